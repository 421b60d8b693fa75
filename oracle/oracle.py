"""ctypes/numpy front end of the CPU oracle (oracle/cppf_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of cppf_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by cppf_amd/.

Every function cites the reference lines it restates (paths relative to qq456cvb/CPPF).
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f = np.float32
_pf = C.POINTER(C.c_float)
_pd = C.POINTER(C.c_double)
_pi32 = C.POINTER(C.c_int32)
_pi64 = C.POINTER(C.c_int64)
_pu8 = C.POINTER(C.c_uint8)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("cppf_oracle.c", "sprin_oracle.c", "backward_oracle.c", "preproc_oracle.c",
                                             "sprin_bwd_oracle.c", "voting_variants.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_expf.restype = C.c_float
        _LIB.orc_expf.argtypes = [C.c_float]
        _LIB.orc_tan.restype = C.c_float
        _LIB.orc_tan.argtypes = [C.c_float]
        _LIB.orc_grid_argmax.restype = C.c_int64
        _LIB.orc_sample_bin.restype = C.c_int
        _LIB.orc_sprin_conv_params.restype = C.c_int64
        _LIB.orc_voxel_dedupe.restype = C.c_int64
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# --------------------------------------------------------------------------- math
def expf(x):
    return float(lib().orc_expf(C.c_float(float(x))))


def sincos(x):
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos(C.c_double(float(x)), C.byref(s), C.byref(c))
    return s.value, c.value


def rot_cs(i, n):
    cs, sn = C.c_float(), C.c_float()
    lib().orc_rot_cs(C.c_int(i), C.c_int(n), C.byref(cs), C.byref(sn))
    return cs.value, sn.value


def tanf(x):
    return float(lib().orc_tan(C.c_float(float(x))))


def set_threads(n):
    lib().orc_set_threads(C.c_int(int(n)))


def num_threads():
    return int(lib().orc_num_threads())


# --------------------------------------------------------------------------- MLP
def pack_params(sd, ppffcs):
    """Flatten a PPFEncoder state_dict (models/model.py:80-87 key names) into one fp32 buffer +
    the offset table orc_pair_mlp() expects."""
    chunks, offs, pos = [], [], 0

    def put(name):
        nonlocal pos
        if name not in sd:
            offs.append(-1)
            return
        a = np.asarray(sd[name], dtype=_f).reshape(-1)
        chunks.append(a)
        offs.append(pos)
        pos += a.size

    for i in range(len(ppffcs) - 1):
        for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc0.weight", "fc0.bias"):
            put(f"res_layers.{i}.{k}")
    put("final.weight")
    put("final.bias")
    return np.concatenate(chunks).astype(_f), np.asarray(offs, dtype=np.int64)


def ppf_features(pc, nrm, idxs):
    """models/model.py:118-129 -> [P,4]"""
    pc, nrm, idxs = _c(pc, _f), _c(nrm, _f), _c(idxs, np.int64)
    out = np.empty((idxs.shape[0], 4), _f)
    lib().orc_ppf_features(_p(pc, _pf), _p(nrm, _pf), _p(idxs, _pi64), C.c_int64(idxs.shape[0]), _p(out, _pf))
    return out


def pair_mlp(pc, nrm, feat, idxs, sd, ppffcs, out_dim, order=0):
    """PPFEncoder.forward_with_idx (models/model.py:117-137) -> logits [P,out_dim]."""
    pc, nrm, feat, idxs = _c(pc, _f), _c(nrm, _f), _c(feat, _f), _c(idxs, np.int64)
    params, offs = pack_params(sd, ppffcs)
    dims = np.asarray(ppffcs, dtype=np.int32)
    P = idxs.shape[0]
    out = np.empty((P, out_dim), _f)
    rc = lib().orc_pair_mlp(_p(pc, _pf), _p(nrm, _pf), _p(feat, _pf), _p(idxs, _pi64), C.c_int64(pc.shape[0]),
                            C.c_int(feat.shape[1]), C.c_int64(P), _p(params, _pf), _p(offs, _pi64),
                            _p(dims, _pi32), C.c_int(len(ppffcs) - 1), C.c_int(out_dim), C.c_int(order),
                            _p(out, _pf))
    if rc != 0:
        raise ValueError(f"orc_pair_mlp failed: {rc}")
    return out


# --------------------------------------------------------------------------- decode
def sample_bin(logits, u, col0=0):
    """inverse-CDF draw over four consecutive segments of the bins (orc_sample_bin); col0 is unused, kept for callers"""
    logits = _c(logits, _f)
    return int(lib().orc_sample_bin(_p(logits, _pf), C.c_int(logits.size), C.c_float(float(u)), C.c_int(col0)))


def decode_center(logits, u, tr_num_bins, vote_range):
    """nocs/inference.py:185-188 with explicit uniforms u[P,2] (u<0: argmax bin)."""
    logits, u = _c(logits, _f), _c(u, _f)
    P = logits.shape[0]
    outputs = np.empty((P, 2), _f)
    bins = np.empty((P, 2), np.int32)
    lib().orc_decode_center(_p(logits, _pf), C.c_int64(P), C.c_int(logits.shape[1]), C.c_int(tr_num_bins),
                            _p(u, _pf), C.c_float(vote_range[0]), C.c_float(vote_range[1]), _p(outputs, _pf),
                            _p(bins, _pi32))
    return outputs, bins


def decode_rot(logits, u, tr_num_bins, rot_num_bins):
    """nocs/inference.py:238-256 -> heads[P,8] = {theta_up, theta_right, aux_up, aux_right, sx,sy,sz,0}"""
    logits, u = _c(logits, _f), _c(u, _f)
    P = logits.shape[0]
    heads = np.empty((P, 8), _f)
    bins = np.empty((P, 2), np.int32)
    lib().orc_decode_rot(_p(logits, _pf), C.c_int64(P), C.c_int(logits.shape[1]), C.c_int(logits.shape[1]),
                         C.c_int(tr_num_bins), C.c_int(rot_num_bins), _p(u, _pf), _p(heads, _pf),
                         _p(bins, _pi32))
    return heads, bins


# --------------------------------------------------------------------------- vote
def grid_setup(pc, res):
    """nocs/inference.py:194-195 -> (corner f32[3], dims i32[3])"""
    pc = _c(pc, _f)
    corner = np.empty(3, _f)
    dims = np.empty(3, np.int32)
    lib().orc_grid_setup(_p(pc, _pf), C.c_int64(pc.shape[0]), C.c_float(res), _p(corner, _pf), _p(dims, _pi32))
    return corner, dims


def ppf_voting(points, outputs, probs, point_idxs, grid, corner, res, n_rots, adaptive, threads=1):
    """models/voting.py:8-66, in place on `grid` (f32[gx,gy,gz]); returns the number of atomicAdds."""
    points, outputs, probs = _c(points, _f), _c(outputs, _f), _c(probs, _f)
    point_idxs, corner = _c(point_idxs, np.int32), _c(corner, _f)
    assert grid.dtype == _f and grid.flags.c_contiguous
    gx, gy, gz = grid.shape
    na = C.c_int64(0)
    args = [_p(points, _pf), _p(outputs, _pf), _p(probs, _pf), _p(point_idxs, _pi32), _p(grid, _pf),
            _p(corner, _pf), C.c_float(res), C.c_int64(point_idxs.shape[0]), C.c_int(n_rots), C.c_int(gx),
            C.c_int(gy), C.c_int(gz), C.c_int(1 if adaptive else 0)]
    if threads == 1:
        lib().orc_ppf_voting(*args, C.byref(na))
        return na.value
    lib().orc_ppf_voting_mt(*args)
    return None


def ppf_voting_f64(points, outputs, probs, point_idxs, dims, corner, res, n_rots, adaptive):
    """models/voting.py:8-66 with the grid accumulated in fp64 -> (grid f64, deposits-per-cell i32)."""
    points, outputs, probs = _c(points, _f), _c(outputs, _f), _c(probs, _f)
    point_idxs, corner = _c(point_idxs, np.int32), _c(corner, _f)
    gx, gy, gz = (int(d) for d in dims)
    grid = np.zeros((gx, gy, gz), np.float64)
    counts = np.zeros((gx, gy, gz), np.int32)
    lib().orc_ppf_voting_f64(_p(points, _pf), _p(outputs, _pf), _p(probs, _pf), _p(point_idxs, _pi32), _p(grid, _pd),
                             _p(counts, _pi32), _p(corner, _pf), C.c_float(res), C.c_int64(point_idxs.shape[0]),
                             C.c_int(n_rots), C.c_int(gx), C.c_int(gy), C.c_int(gz), C.c_int(1 if adaptive else 0))
    return grid, counts


def ppf_voting_fixed(points, outputs, probs, point_idxs, dims, corner, res, n_rots, adaptive, bits, grid=None):
    """models/voting.py:8-66 with every deposit rounded to p2 * 2^-bits (p2 = max(probs) rounded up to a power of two) and
    summed as int64: the specification of cppf_vote_grid_raw.  Returns (grid i64[dims], quantum)."""
    points, outputs, probs = _c(points, _f), _c(outputs, _f), _c(probs, _f)
    point_idxs, corner = _c(point_idxs, np.int32), _c(corner, _f)
    gx, gy, gz = (int(d) for d in dims)
    if grid is None:
        grid = np.zeros((gx, gy, gz), np.int64)
    pmax = float(np.max(probs)) if probs.size else 1.0
    p2 = float(2.0 ** np.ceil(np.log2(pmax))) if pmax > 0 else 1.0
    lib().orc_ppf_voting_fixed(_p(points, _pf), _p(outputs, _pf), _p(probs, _pf), _p(point_idxs, _pi32),
                               grid.ctypes.data_as(C.POINTER(C.c_int64)), _p(corner, _pf), C.c_float(res),
                               C.c_int64(point_idxs.shape[0]), C.c_int(n_rots), C.c_int(gx), C.c_int(gy), C.c_int(gz),
                               C.c_int(1 if adaptive else 0), C.c_int(int(bits)), C.c_double(p2))
    return grid, p2 * 2.0 ** -int(bits)


def grid_argmax(grid):
    """np.argmax(grid) (first maximum, C order), nocs/inference.py:208"""
    g = _c(grid, _f).reshape(-1)
    v = C.c_float()
    i = lib().orc_grid_argmax(_p(g, _pf), C.c_int64(g.size), C.byref(v))
    return int(i), float(v.value)


def center_from_argmax(flat, dims, corner, res):
    """nocs/inference.py:209-210 -> T f64[3]"""
    corner = _c(corner, _f)
    T = np.empty(3, np.float64)
    lib().orc_center_from_argmax(C.c_int64(flat), C.c_int(int(dims[1])), C.c_int(int(dims[2])), _p(corner, _pf),
                                 C.c_double(float(res)), _p(T, _pd))
    return T


def backvote(points, outputs, point_idxs, corner, res, n_rots, dims, gt_center, tol):
    """models/voting.py:74-112 -> (out_offsets f32[P,3], mask bool[P])"""
    points, outputs = _c(points, _f), _c(outputs, _f)
    point_idxs, corner, gt = _c(point_idxs, np.int32), _c(corner, _f), _c(gt_center, _f)
    P = point_idxs.shape[0]
    oo = np.zeros((P, 3), _f)
    mask = np.zeros(P, np.uint8)
    lib().orc_backvote(_p(points, _pf), _p(outputs, _pf), _p(oo, _pf), _p(point_idxs, _pi32), _p(corner, _pf),
                       C.c_float(res), C.c_int64(P), C.c_int(n_rots), C.c_int(int(dims[0])), C.c_int(int(dims[1])),
                       C.c_int(int(dims[2])), _p(gt, _pf), C.c_float(tol), _p(mask, _pu8))
    return oo, mask.astype(bool)


def rot_voting(points, preds_rot, point_idxs, n_rots):
    """models/voting.py:119-147 -> candidates f32[P,n_rots,3]"""
    points, preds_rot, point_idxs = _c(points, _f), _c(preds_rot, _f), _c(point_idxs, np.int32)
    P = point_idxs.shape[0]
    out = np.zeros((P, n_rots, 3), _f)
    lib().orc_rot_voting(_p(points, _pf), _p(preds_rot, _pf), _p(out, _pf), _p(point_idxs, _pi32), C.c_int64(P),
                         C.c_int(n_rots))
    return out


# --------------------------------------------------------------------------- arithmetic variants (voting_variants.c)
# The reference's kernels run under NVRTC defaults (--fmad=true, CUDA's device libm): which products are fused and the last
# bits of cos / sin / tan are the toolchain's choice.  variant 0 = this oracle's own member of that family.
FMAD_LEFT, FMAD_RIGHT, LIBM, ULP_UP, ULP_DOWN, ULP_HASH = 1, 2, 4, 8, 16, 32
VARIANTS = {
    "fmad_left": FMAD_LEFT, "fmad_right": FMAD_RIGHT, "libm": LIBM, "fmad_left+libm": FMAD_LEFT | LIBM,
    "fmad_right+libm": FMAD_RIGHT | LIBM, "ulp_up": ULP_UP, "ulp_down": ULP_DOWN, "ulp_hash": ULP_HASH,
    "fmad_left+ulp_hash": FMAD_LEFT | ULP_HASH, "fmad_right+ulp_hash2": FMAD_RIGHT | ULP_HASH | (2 << 8),
}


def ppf_voting_variant(points, outputs, probs, point_idxs, dims, corner, res, n_rots, adaptive, variant, threads=None):
    """models/voting.py:8-66 under an arithmetic variant -> the exact (fp64) vote grid."""
    points, outputs, probs = _c(points, _f), _c(outputs, _f), _c(probs, _f)
    point_idxs, corner = _c(point_idxs, np.int32), _c(corner, _f)
    gx, gy, gz = (int(d) for d in dims)
    grid = np.zeros((gx, gy, gz), np.float64)
    if threads is None:
        threads = max(1, min(num_threads(), 16, int(2 ** 31 // max(grid.nbytes, 1))))
    lib().orv_ppf_voting_f64(_p(points, _pf), _p(outputs, _pf), _p(probs, _pf), _p(point_idxs, _pi32), _p(grid, _pd),
                             _p(corner, _pf), C.c_float(res), C.c_int64(point_idxs.shape[0]), C.c_int(n_rots), C.c_int(gx),
                             C.c_int(gy), C.c_int(gz), C.c_int(1 if adaptive else 0), C.c_int(int(variant)), C.c_int(threads))
    return grid


def vote_flips(points, outputs, point_idxs, dims, corner, res, n_rots, adaptive, variant):
    """per-sample discrete outcomes of ppf_voting under `variant` against variant 0 -> dict of counts (voting_variants.c)"""
    points, outputs = _c(points, _f), _c(outputs, _f)
    point_idxs, corner = _c(point_idxs, np.int32), _c(corner, _f)
    out = np.zeros(8, np.int64)
    dmax = np.zeros(1, np.float64)
    lib().orv_vote_flips(_p(points, _pf), _p(outputs, _pf), _p(point_idxs, _pi32), _p(corner, _pf), C.c_float(res),
                         C.c_int64(point_idxs.shape[0]), C.c_int(n_rots), C.c_int(int(dims[0])), C.c_int(int(dims[1])),
                         C.c_int(int(dims[2])), C.c_int(1 if adaptive else 0), C.c_int(int(variant)), _p(out, _pi64),
                         _p(dmax, _pd))
    return dict(degenerate_flips=int(out[0]), trip_count_flips=int(out[1]), samples=int(out[2]), in_grid=int(out[3]),
                in_grid_flips=int(out[4]), floor_cell_flips=int(out[5]), max_coord_diff_cells=float(dmax[0]))


def backvote_variant(points, outputs, point_idxs, corner, res, n_rots, dims, gt_center, tol, variant):
    points, outputs = _c(points, _f), _c(outputs, _f)
    point_idxs, corner, gt = _c(point_idxs, np.int32), _c(corner, _f), _c(gt_center, _f)
    P = point_idxs.shape[0]
    oo = np.zeros((P, 3), _f)
    mask = np.zeros(P, np.uint8)
    lib().orv_backvote(_p(points, _pf), _p(outputs, _pf), _p(oo, _pf), _p(point_idxs, _pi32), _p(corner, _pf),
                       C.c_float(res), C.c_int64(P), C.c_int(n_rots), C.c_int(int(dims[0])), C.c_int(int(dims[1])),
                       C.c_int(int(dims[2])), _p(gt, _pf), C.c_float(tol), _p(mask, _pu8), C.c_int(int(variant)))
    return oo, mask.astype(bool)


def rot_voting_variant(points, preds_rot, point_idxs, n_rots, variant):
    points, preds_rot, point_idxs = _c(points, _f), _c(preds_rot, _f), _c(point_idxs, np.int32)
    P = point_idxs.shape[0]
    out = np.zeros((P, n_rots, 3), _f)
    lib().orv_rot_voting(_p(points, _pf), _p(preds_rot, _pf), _p(out, _pf), _p(point_idxs, _pi32), C.c_int64(P),
                         C.c_int(n_rots), C.c_int(int(variant)))
    return out


def pose_tail_variant(pc, nrm, point_idxs, outputs, heads, cfg, sphere_pts, variant, num_rots=72, adaptive=True,
                      angle_tol=1.5, max_rot_pairs=10000):
    """nocs/inference.py:191-303,335 from given (mu, nu) and heads, every vote kernel under `variant` and the centre grid
    accumulated exactly (fp64): what the arithmetic freedom of the reference's toolchain does to the POSE."""
    res = float(cfg["res"])
    corner, dims = grid_setup(pc, res)
    idx32 = point_idxs.astype(np.int32)
    grid = ppf_voting_variant(pc, outputs, np.ones(pc.shape[0], _f), idx32, dims, corner, res, num_rots, adaptive, variant)
    flat = int(np.argmax(grid))
    top2 = np.partition(grid.reshape(-1), -2)[-2:]
    T = center_from_argmax(flat, dims, corner, res)
    _, mask = backvote_variant(pc, outputs, idx32, corner, res, num_rots, dims, T.astype(_f), np.float32(3 * res), variant)
    surv = np.nonzero(mask)[0]
    sel = surv[:max_rot_pairs]
    dirs, bests, counts_all = [], [], []
    for j in range(2):
        if j == 1 and not cfg["regress_right"]:
            continue
        cands = rot_voting_variant(pc, heads[sel, j], idx32[sel], num_rots, variant)
        counts = sphere_count(cands, sphere_pts, angle_tol)
        bi = int(np.argmax(counts))
        best = np.asarray(sphere_pts[bi], np.float64)
        flip, _ = axis_sign(pc, nrm, idx32[surv], heads[surv, 2 + j], best)
        dirs.append(-best if flip else best)
        bests.append(bi)
        counts_all.append(counts)
    sc = scale(heads[surv, 4:7], cfg["scale_mean"]) if surv.size else np.zeros(3)
    return dict(T=T, argmax=flat, peak=float(top2[1]), margin=float(top2[1] - top2[0]), grid=grid, mask=mask,
                up=dirs[0] if dirs else None, right=dirs[1] if len(dirs) > 1 else None, scale=sc, sphere_argmax=bests,
                counts=counts_all, corner=corner, dims=dims)


def sphere_count(cands, sphere_pts, angle_tol_deg):
    """nocs/inference.py:281-282 -> counts i64[S]; thr = cos(angle_tol) rounded to fp32."""
    cands, sph = _c(cands, _f).reshape(-1, 3), _c(sphere_pts, _f)
    thr = np.float32(np.cos(angle_tol_deg / 180 * np.pi))
    counts = np.empty(sph.shape[0], np.int64)
    lib().orc_sphere_count(_p(cands, _pf), C.c_int64(cands.shape[0]), _p(sph, _pf), C.c_int(sph.shape[0]),
                           C.c_float(thr), _p(counts, _pi64))
    return counts


def axis_sign(pc, nrm, point_idxs, aux, best_dir):
    """nocs/inference.py:287-301 -> (flip: bool, (up_loss, down_loss))"""
    pc, nrm, point_idxs, aux = _c(pc, _f), _c(nrm, _f), _c(point_idxs, np.int32), _c(aux, _f)
    bd = _c(best_dir, np.float64)
    losses = np.empty(2, np.float64)
    flip = lib().orc_axis_sign(_p(pc, _pf), _p(nrm, _pf), _p(point_idxs, _pi32), C.c_int64(point_idxs.shape[0]),
                               _p(aux, _pf), C.c_int(1), _p(bd, _pd), _p(losses, _pd))
    return bool(flip), (float(losses[0]), float(losses[1]))


def scale(scale_logits, scale_mean):
    """nocs/inference.py:335 -> f64[3]"""
    sl = _c(scale_logits, _f)
    sm = _c(scale_mean, np.float64)
    out = np.empty(3, np.float64)
    lib().orc_scale(_p(sl, _pf), C.c_int64(sl.shape[0]), C.c_int(sl.shape[1]), _p(sm, _pd), _p(out, _pd))
    return out


# --------------------------------------------------------------------------- host-side pieces
def fibonacci_sphere(samples):
    """utils/util.py:102-118: golden-angle spiral, y from 1 to -1; fp64 -> [samples,3]"""
    phi = math.pi * (3.0 - math.sqrt(5.0))
    pts = np.empty((samples, 3), np.float64)
    for i in range(samples):
        y = 1 - (i / float(samples - 1)) * 2
        r = math.sqrt(1 - y * y)
        t = phi * i
        pts[i] = (math.cos(t) * r, y, math.sin(t) * r)
    return pts


def closed_form_targets(pc, point_idxs):
    """utils/dataset.py:27-36: (mu, nu) of each pair for an object centred at the origin."""
    a = pc[point_idxs[:, 0]]
    b = pc[point_idxs[:, 1]]
    pd = a - b
    pu = pd / (np.linalg.norm(pd, axis=-1, keepdims=True) + 1e-7)
    proj = np.sum(a * pu, -1)
    oc = a - proj[..., None] * pu
    return np.stack([proj, np.linalg.norm(oc, axis=-1)], -1).astype(_f)


def estimate_pose(pc, nrm, feat, point_idxs, sd, cfg, u_tr, u_rot, sphere_pts, num_rots=72, adaptive=True,
                  angle_tol=1.5, max_rot_pairs=10000, order=1, rot_order=None):
    """The glue of nocs/inference.py:177-335 chained from the pieces above, with the stochastic
    draws supplied: u_tr[P,2] (centre bins), u_rot[P,2] (up/right bins, indexed by ORIGINAL pair so
    the second MLP pass of :236 is a gather of first-pass rows -- the MLP is deterministic in eval
    mode), and the 10 000-pair subsample of :278-280 taken as the first survivors in pair order
    (pairs are i.i.d. uniform, so this is the same distribution as the reference's shuffle) -- or, with `rot_order`, exactly
    the positions of the survivor list the caller names (the reference's `idxs = arange(P'); shuffle(idxs); idxs[:10000]`).
    cfg: dict(res, tr_num_bins, rot_num_bins, vote_range, scale_mean, regress_right, ppffcs, out_dim)."""
    res = float(cfg["res"])
    tb, rb = cfg["tr_num_bins"], cfg["rot_num_bins"]
    logits = pair_mlp(pc, nrm, feat, point_idxs, sd, cfg["ppffcs"], cfg["out_dim"], order=order)
    outputs, _ = decode_center(logits, u_tr, tb, cfg["vote_range"])
    corner, dims = grid_setup(pc, res)
    grid = np.zeros(tuple(int(d) for d in dims), _f)
    idx32 = point_idxs.astype(np.int32)
    ppf_voting(pc, outputs, np.ones(pc.shape[0], _f), idx32, grid, corner, res, num_rots, adaptive)
    flat, peak = grid_argmax(grid)
    T = center_from_argmax(flat, dims, corner, res)
    _, mask = backvote(pc, outputs, idx32, corner, res, num_rots, dims, T.astype(_f), np.float32(3 * res))
    surv = np.nonzero(mask)[0]
    heads, _ = decode_rot(logits, u_rot, tb, rb)
    sidx = idx32[surv]
    if rot_order is None:
        sel = surv[:max_rot_pairs]
    else:        # the caller's shuffled subsample (:277-280): positions in the survivor list; out-of-range entries are skipped
        ro = np.asarray(rot_order)[:max_rot_pairs]
        sel = surv[ro[(ro >= 0) & (ro < surv.size)]]
    dirs = []
    for j in range(2):
        if j == 1 and not cfg["regress_right"]:
            continue
        cands = rot_voting(pc, heads[sel, j], idx32[sel], num_rots)
        counts = sphere_count(cands, sphere_pts, angle_tol)
        best = np.asarray(sphere_pts[int(np.argmax(counts))], np.float64)
        flip, _ = axis_sign(pc, nrm, sidx, heads[surv, 2 + j], best)
        dirs.append(-best if flip else best)
    sc = scale(heads[surv, 4:7], cfg["scale_mean"]) if surv.size else np.zeros(3)
    return dict(T=T, argmax=flat, peak=peak, grid=grid, outputs=outputs, logits=logits, mask=mask,
                up=dirs[0] if dirs else None, right=dirs[1] if len(dirs) > 1 else None, scale=sc,
                corner=corner, dims=dims, heads=heads)


# --------------------------------------------------------------------------- SPRIN point encoder (row f1)
def pack_point_encoder(sd, num_layers):
    """Flatten a reference PointEncoder state_dict (models/model.py:36-44, models/sprin.py:64-95) into the
    packed layout documented in sprin_oracle.c.  Returns (packed f32, desc dict)."""
    g = lambda k: np.asarray(sd[k], dtype=np.float32)
    parts = []
    hidden = None
    for l in range(num_layers):
        pre = f"spconvs.{l}.kernel."
        lin = sorted({int(k[len(pre):].split(".")[0]) for k in sd if k.startswith(pre)})
        # Sequential indices: Linear at 0,3,6,..., LayerNorm at 1,4,...; the last Linear has no LayerNorm after it
        lins = [i for i in lin if g(f"{pre}{i}.weight").ndim == 2]
        hid = []
        for i in lins[:-1]:
            W = g(f"{pre}{i}.weight")
            parts += [W.ravel(), g(f"{pre}{i}.bias"), g(f"{pre}{i + 1}.weight"), g(f"{pre}{i + 1}.bias")]
            hid.append(W.shape[0])
        Wk = g(f"{pre}{lins[-1]}.weight")
        parts += [Wk.ravel(), g(f"{pre}{lins[-1]}.bias")]
        rank = Wk.shape[0]
        Wo = g(f"spconvs.{l}.outnet.weight")
        n_out = Wo.shape[0]
        n_in = Wo.shape[1] // rank
        parts += [np.ascontiguousarray(Wo.T).ravel(), g(f"spconvs.{l}.outnet.bias"),
                  g(f"spconvs.{l}.layer_norm.weight"), g(f"spconvs.{l}.layer_norm.bias")]
        Wa = g(f"aggrs.{l}.linear.weight")
        parts += [Wa.ravel(), g(f"aggrs.{l}.linear.bias")]
        if l == 0:
            hidden, desc = hid, dict(hidden=hid, rank=rank, n_nbr_feats=n_in, n_out=n_out, n_glob=Wa.shape[0],
                                     num_layers=num_layers)
        else:
            assert hid == hidden
    return np.concatenate(parts).astype(np.float32), desc


def knn(pc=None, k=60, dist=None):
    """torch.topk(dist, k, largest=False) neighbour sets (models/model.py:47), ascending index order."""
    N = (pc if dist is None else dist).shape[0]
    out = np.empty((N, k), np.int32)
    pcc = _c(pc, _f) if pc is not None else None
    dc = _c(dist, _f) if dist is not None else None
    rc = lib().orc_knn(_p(pcc, _pf) if pcc is not None else None, _p(dc, _pf) if dc is not None else None,
                       C.c_int(N), C.c_int(k), _p(out, _pi32))
    assert rc == 0
    return out


def point_encoder(pc, nrm, nbrs, packed, desc, order=0):
    """PointEncoder.forward_nbrs (models/model.py:63-78) -> f32[N, n_out + n_glob].
    order 0: natural summation order; 1: the order of the device MFMA kernel (bit-exact with it)."""
    pc, nrm = _c(pc, _f), _c(nrm, _f)
    nbrs = _c(nbrs, np.int32)
    N, k = nbrs.shape
    hid = np.asarray(desc["hidden"], np.int32)
    out = np.empty((N, desc["n_out"] + desc["n_glob"]), np.float32)
    packed = _c(packed, _f)
    rc = lib().orc_point_encoder(_p(pc, _pf), _p(nrm, _pf), _p(nbrs, _pi32), C.c_int(N), C.c_int(k), _p(packed, _pf),
                                 _p(hid, _pi32), C.c_int(len(hid)), C.c_int(desc["rank"]), C.c_int(desc["n_nbr_feats"]),
                                 C.c_int(desc["n_out"]), C.c_int(desc["n_glob"]), C.c_int(desc["num_layers"]),
                                 C.c_int(order), _p(out, _pf))
    assert rc == 0
    return out


# --------------------------------------------------------------------------- backward of the point encoder
SPRIN_BWD_MAX_PARTS = 1024


def sprin_bwd_parts(N):
    """number of partial accumulators (= wavefronts) the device kernel uses for N points"""
    return min(SPRIN_BWD_MAX_PARTS, N)


def point_encoder_grad_names():
    """(parameter name, packed offset, shape, transposed) in the packed layout of cppf_point_encoder_pack for the
    standard one-layer encoder (train.py:34)"""
    out, pos = [], 0
    dims = [(32, 6), (64, 32), (32, 64), (32, 32)]
    for i, (h, w) in enumerate(dims):
        out.append((f"spconvs.0.kernel.{3 * i}.weight", pos, (h, w), False)); pos += h * w
        out.append((f"spconvs.0.kernel.{3 * i}.bias", pos, (h,), False)); pos += h
        out.append((f"spconvs.0.kernel.{3 * i + 1}.weight", pos, (h,), False)); pos += h
        out.append((f"spconvs.0.kernel.{3 * i + 1}.bias", pos, (h,), False)); pos += h
    out.append(("spconvs.0.kernel.12.weight", pos, (32, 32), False)); pos += 1024
    out.append(("spconvs.0.kernel.12.bias", pos, (32,), False)); pos += 32
    out.append(("spconvs.0.outnet.weight", pos, (64, 32), True)); pos += 2048      # packed transposed [C][n_out]
    out.append(("spconvs.0.outnet.bias", pos, (32,), False)); pos += 32
    out.append(("spconvs.0.layer_norm.weight", pos, (32,), False)); pos += 32
    out.append(("spconvs.0.layer_norm.bias", pos, (32,), False)); pos += 32
    out.append(("aggrs.0.linear.weight", pos, (8, 32), False)); pos += 256
    out.append(("aggrs.0.linear.bias", pos, (8,), False)); pos += 8
    return out, pos


def point_encoder_backward(pc, nrm, nbrs, packed, grad_out, n_parts=None):
    """autograd of models/model.py:46-61 + models/sprin.py:40-107 w.r.t. the parameters (train.py:91):
    returns ({parameter name: grad}, flat packed gradient)."""
    pc, nrm, nbrs = _c(pc, _f), _c(nrm, _f), _c(nbrs, np.int32)
    packed, grad_out = _c(packed, _f), _c(grad_out, _f)
    N, k = nbrs.shape
    names, total = point_encoder_grad_names()
    L = lib()
    L.orc_point_encoder_backward_params.restype = C.c_int64
    assert L.orc_point_encoder_backward_params() == total
    n_parts = sprin_bwd_parts(N) if n_parts is None else n_parts
    g = np.zeros(total, np.float32)
    rc = L.orc_point_encoder_backward(_p(pc, _pf), _p(nrm, _pf), _p(nbrs, _pi32), C.c_int(N), C.c_int(k), _p(packed, _pf),
                                      _p(grad_out, _pf), C.c_int(n_parts), _p(g, _pf))
    if rc != 0:
        raise ValueError(f"orc_point_encoder_backward failed: {rc}")
    grads = {}
    for name, off, shape, tr in names:
        a = g[off:off + int(np.prod(shape))].reshape(shape)
        grads[name] = a.T.copy() if tr else a.copy()
    return grads, g


# --------------------------------------------------------------------------- backward of the pair MLP (row f2)
BWD_MAX_PARTS = 512


def bwd_parts(P):
    """number of partial accumulators the device kernel uses for P pairs (cppf.h: CPPF_BWD_MAX_PARTS): every
    workgroup gets the same number of 64-pair tiles (+-1)"""
    t = (P + 63) // 64
    if t <= 1:
        return 1
    per = (t + BWD_MAX_PARTS - 1) // BWD_MAX_PARTS
    return (t + per - 1) // per


def pair_mlp_backward(pc, nrm, feat, idxs, sd, ppffcs, out_dim, grad_out, n_parts=None):
    """autograd of models/model.py:117-137 (train.py:91): returns ({param name: grad}, grad_feat[N,F])."""
    pc, nrm, feat, idxs = _c(pc, _f), _c(nrm, _f), _c(feat, _f), _c(idxs, np.int64)
    grad_out = _c(grad_out, _f)
    params, offs = pack_params(sd, ppffcs)
    dims = np.asarray(ppffcs, dtype=np.int32)
    P = idxs.shape[0]
    n_parts = bwd_parts(P) if n_parts is None else n_parts
    gp = np.zeros_like(params)
    gf = np.zeros_like(feat)
    rc = lib().orc_pair_mlp_backward(_p(pc, _pf), _p(nrm, _pf), _p(feat, _pf), _p(idxs, _pi64), C.c_int64(pc.shape[0]),
                                     C.c_int(feat.shape[1]), C.c_int64(P), _p(params, _pf), _p(offs, _pi64),
                                     _p(dims, _pi32), C.c_int(len(ppffcs) - 1), C.c_int(out_dim), _p(grad_out, _pf),
                                     C.c_int(n_parts), C.c_int64(params.size), _p(gp, _pf), _p(gf, _pf))
    if rc != 0:
        raise ValueError(f"orc_pair_mlp_backward failed: {rc}")
    names = []
    for i in range(len(ppffcs) - 1):
        names += [f"res_layers.{i}.{k}" for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc0.weight",
                                                   "fc0.bias")]
    names += ["final.weight", "final.bias"]
    grads = {}
    for nme, o in zip(names, offs):
        if o >= 0:
            grads[nme] = gp[o:o + sd[nme].size].reshape(np.shape(sd[nme]))
    return grads, gf, gp


# --------------------------------------------------------------------------- pre-processing (row f3)
def voxel_dedupe(pc, res):
    """one representative (lowest index) per occupied voxel floor(p / res), ascending -- the role of
    ME.utils.sparse_quantize(..., return_index=True)[1] at nocs/inference.py:140"""
    pc = _c(pc, _f)
    keep = np.empty(pc.shape[0], np.int32)
    n = lib().orc_voxel_dedupe(_p(pc, _pf), C.c_int64(pc.shape[0]), C.c_double(float(res)), _p(keep, _pi32))
    return keep[:n].copy()


def backproject(depth, intrinsics, instance_mask):
    """utils/util.py:598-631 -> (pts f64[n,3], (rows, cols)) like the reference"""
    d = np.ascontiguousarray(depth, dtype=np.float64)
    m = np.ascontiguousarray(instance_mask).astype(np.uint8)
    H, W = d.shape
    kinv = np.ascontiguousarray(np.linalg.inv(np.asarray(intrinsics, np.float64)))
    pts = np.empty((H * W, 3), np.float64)
    pix = np.empty(H * W, np.int32)
    f = lib().orc_backproject
    f.restype = C.c_int64
    n = f(_p(d, _pd), _p(m, _pu8), C.c_int(H), C.c_int(W), _p(kinv, _pd), _p(pts, _pd), _p(pix, _pi32))
    pix = pix[:n]
    return pts[:n].copy(), (pix // W, pix % W)


def estimate_normals(pc, nbrs):
    """PCA normals of the neighbour sets (utils/util.py:61-65 semantics, sign: largest component positive)"""
    pc, nbrs = _c(pc, _f), _c(nbrs, np.int32)
    out = np.empty((pc.shape[0], 3), _f)
    lib().orc_estimate_normals(_p(pc, _pf), _p(nbrs, _pi32), C.c_int64(pc.shape[0]), C.c_int(nbrs.shape[1]), _p(out, _pf))
    return out
