"""TEST INFRASTRUCTURE -- a SECOND, independent restatement of the reference's three vote kernels.

The reference's `ppf_voting`, `backvote` and `rot_voting` exist only as CUDA text inside cupy.RawKernel strings
(models/voting.py:4-148) whose vendored header hard-codes a CUDA toolkit path: they cannot be built or run in this
environment, so parity with them is unpinned by executed reference code.  What can be done is to
restate them twice, independently, and demand that the two restatements agree bit for bit: oracle/cppf_oracle.c (scalar C,
one pair at a time, the checker of the HIP kernels) and this file (numpy, vectorised over the pairs, written from the CUDA
text line by line, cited below).  A transcription slip -- a bound, a sign, an operand order, a float/double promotion --
would have to be made identically in both to go unnoticed.  tests/test_oracle_golden.py compares them.

Conventions taken from the CUDA text and its header (models/include/helper_math.cuh):
  * every `float` expression is evaluated in np.float32, one operation at a time (no contraction);
  * the `double` literals 1e-7, 0.01, 1.01, M_PI promote their sub-expression to float64 (Appendix A of SURVEY.md);
  * length(v) = sqrtf(dot(v, v)), dot = x*x + y*y + z*z left to right (helper_math.cuh:1245,1288);
    cross (…:1417), fracf(v) = v - floorf(v) (…:1338), make_int3(float3) truncates (…:157);
  * cos / sin / tan of a float argument are the float overloads.  Their bits are not specified by CUDA; both
    restatements take them from the same fixed polynomial evaluation (`cs(i, n)`, `tan(rot)` callbacks), so the
    comparison is about the logic around them.
Only tests import this module.
"""
import numpy as np

F = np.float32
D = np.float64
M_PI = 3.14159265358979323846264338327950288


def _dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _length(v):
    return np.sqrt(_dot(v, v)).astype(F)


def _cross(a, b):                      # helper_math.cuh:1417
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1).astype(F)


def _frame(points, point_idxs, scale_x=None):
    """models/voting.py:15-29 (ppf_voting), :81-95 (backvote), :125-136 (rot_voting): a, unit ab, x, y and the mask of pairs
    that do not return early at `if (length(ab) < 1e-7) return;`.  scale_x = odist (x is scaled by it) or None."""
    a = points[point_idxs[:, 0]].astype(F)
    b = points[point_idxs[:, 1]].astype(F)
    ab = (a - b).astype(F)
    L = _length(ab)
    live = ~(L.astype(D) < 1e-7)                                       # float promoted to double for the compare
    with np.errstate(divide="ignore", invalid="ignore"):
        ab = (ab / (L.astype(D) + 1e-7).astype(F)[:, None]).astype(F)  # ab /= (length(ab) + 1e-7): double sum, float divisor
        co = np.stack([np.zeros_like(L), -ab[:, 2], ab[:, 1]], -1).astype(F)
        alt = np.stack([-ab[:, 1], ab[:, 0], np.zeros_like(L)], -1).astype(F)
        co = np.where((_length(co).astype(D) < 1e-7)[:, None], alt, co)
        x = (co / (_length(co).astype(D) + 1e-7).astype(F)[:, None]).astype(F)
        if scale_x is not None:
            x = (x * scale_x.astype(F)[:, None]).astype(F)             # co / (...) * odist, left to right
        y = _cross(x, ab)
    return a, ab, x, y, live


def _n_rots(odist, res, n_rots, adaptive):
    """min(int(odist / res * (2 * M_PI)), n_rots): float quotient, double product, truncation toward zero (:31, :97)"""
    if not adaptive:
        return np.full(odist.shape, n_rots, np.int64)
    with np.errstate(invalid="ignore", over="ignore"):
        q = (odist.astype(F) / F(res)).astype(F).astype(D) * (2 * M_PI)
        n = np.where(np.isfinite(q), np.trunc(np.where(np.isfinite(q), q, 0.0)), 0.0)
    return np.minimum(n, n_rots).astype(np.int64)


def ppf_voting(points, outputs, probs, point_idxs, grid, corner, res, n_rots, adaptive, cs):
    """models/voting.py:8-66.  grid f32[gx,gy,gz] is updated in place with the atomicAdds applied in thread order
    (pair, rotation, corner) -- one legal outcome of the CUDA kernel, the one the C restatement also produces.
    cs(i, n) -> (cos, sin) of float(i * 2 * M_PI / n) as float32."""
    gx, gy, gz = grid.shape
    P = point_idxs.shape[0]
    proj_len, odist = outputs[:, 0].astype(F), outputs[:, 1].astype(F)
    a, ab, x, y, live = _frame(points, point_idxs, odist)
    c = (a - (ab * proj_len[:, None]).astype(F)).astype(F)                         # :23
    prob = np.maximum(probs[point_idxs[:, 0]], probs[point_idxs[:, 1]]).astype(F)  # :25
    nr = np.where(live, _n_rots(odist, res, n_rots, adaptive), 0)
    cr = np.asarray(corner, F)
    keys, cells, vals = [], [], []
    for i in range(int(nr.max()) if P else 0):
        act = np.nonzero(nr > i)[0]
        if act.size == 0:
            continue
        csn = np.array([cs(i, int(n)) for n in nr[act]], F).reshape(-1, 2)
        off = ((csn[:, :1] * x[act]).astype(F) + (csn[:, 1:] * y[act]).astype(F)).astype(F)           # :34
        g = ((((c[act] + off).astype(F) - cr).astype(F)) / F(res)).astype(F)                          # :35
        gd = g.astype(D)
        out = (gd[:, 0] < 0.01) | (gd[:, 1] < 0.01) | (gd[:, 2] < 0.01) | (gd[:, 0] >= gx - 1.01) | \
              (gd[:, 1] >= gy - 1.01) | (gd[:, 2] >= gz - 1.01)                                       # :36-39
        act, g = act[~out], g[~out]
        fl = np.trunc(g).astype(np.int64)                                                             # :40 make_int3
        ce = fl + 1                                                                                   # :41
        r = (g - np.floor(g)).astype(F)                                                               # :42 fracf
        w0, w1 = (F(1.0) - r).astype(F), r                                                            # :44-45
        for k, (sx, sy, sz) in enumerate(((0, 0, 0), (0, 0, 1), (0, 1, 0), (0, 1, 1), (1, 0, 0), (1, 0, 1), (1, 1, 0),
                                          (1, 1, 1))):                                                # :47-63, in that order
            wx, wy, wz = (w1 if sx else w0)[:, 0], (w1 if sy else w0)[:, 1], (w1 if sz else w0)[:, 2]
            w = (((wx * wy).astype(F) * wz).astype(F) * prob[act]).astype(F)
            cx, cy, cz = (ce if sx else fl)[:, 0], (ce if sy else fl)[:, 1], (ce if sz else fl)[:, 2]
            keys.append((act * np.int64(n_rots) + i) * 8 + k)
            cells.append(cx * gy * gz + cy * gz + cz)
            vals.append(w)
    if not keys:
        return 0
    keys, cells, vals = np.concatenate(keys), np.concatenate(cells), np.concatenate(vals)
    order = np.argsort(keys, kind="stable")
    flat = grid.reshape(-1)
    np.add.at(flat, cells[order], vals[order])          # unbuffered, applied in the given order: float32 adds one by one
    return int(keys.size)


def backvote(points, outputs, point_idxs, corner, res, n_rots, dims, gt_center, tol, cs):
    """models/voting.py:74-112 -> out_offsets f32[P,3] (zeros where the kernel writes zeros or returns early into the
    caller's zero-initialised buffer, nocs/inference.py:220)."""
    gx, gy, gz = (int(d) for d in dims)
    P = point_idxs.shape[0]
    proj_len, odist = outputs[:, 0].astype(F), outputs[:, 1].astype(F)
    a, ab, x, y, live = _frame(points, point_idxs, odist)
    c = (a - (ab * proj_len[:, None]).astype(F)).astype(F)
    nr = np.where(live, _n_rots(odist, res, n_rots, True), 0)           # always adaptive (:97)
    cr, gt = np.asarray(corner, F), np.asarray(gt_center, F)
    out = np.zeros((P, 3), F)
    todo = np.ones(P, bool)
    for i in range(int(nr.max()) if P else 0):
        act = np.nonzero((nr > i) & todo)[0]
        if act.size == 0:
            continue
        csn = np.array([cs(i, int(n)) for n in nr[act]], F).reshape(-1, 2)
        off = ((csn[:, :1] * x[act]).astype(F) + (csn[:, 1:] * y[act]).astype(F)).astype(F)
        pc = (c[act] + off).astype(F)                                                         # :100
        far = _length((pc - gt).astype(F)) > F(tol)                                           # :101
        g = ((pc - cr).astype(F) / F(res)).astype(F)                                          # :102
        outside = (g[:, 0] < 0) | (g[:, 1] < 0) | (g[:, 2] < 0) | (g[:, 0] >= F(gx - 1)) | (g[:, 1] >= F(gy - 1)) | \
                  (g[:, 2] >= F(gz - 1))                                                      # :103-104 (int - 1 -> float)
        hit = ~far & ~outside
        out[act[hit]] = -off[hit]                                                             # :108
        todo[act[hit]] = False                                                                # break
    return out


def rot_voting(points, preds_rot, point_idxs, n_rots, cs, tan):
    """models/voting.py:119-147 -> outputs_up f32[P,n_rots,3] (rows of pairs that return early stay zero)."""
    P = point_idxs.shape[0]
    a, ab, x, y, live = _frame(points, point_idxs, None)
    out = np.zeros((P, n_rots, 3), F)
    t = np.array([tan(float(r)) for r in preds_rot], F)
    base = np.where((t > 0)[:, None], ab, -ab).astype(F)                                      # :142
    for i in range(n_rots):
        c_, s_ = cs(i, n_rots)
        off = ((F(c_) * x).astype(F) + (F(s_) * y).astype(F)).astype(F)                       # :141
        up = ((t[:, None] * off).astype(F) + base).astype(F)                                  # :142
        with np.errstate(divide="ignore", invalid="ignore"):
            up = (up / (_length(up).astype(D) + 1e-7).astype(F)[:, None]).astype(F)           # :143
        out[live, i] = up[live]
    return out
