"""Seeded synthetic inputs for tests and bench.py.  numpy only.

An "object" of a category is a closed surface with the category's mean half-extents
(config/category/*.yaml:scale_mean): a capped cylinder for bottle/can/mug/bowl, a box otherwise.
Points get analytic outward normals, a random rigid offset of the centre (so corner != 0) and the
inference-time jitter clip(res/4 * N(0,1), +-res/2) of nocs/inference.py:134.  Pairs are uniform
with replacement like nocs/inference.py:177 (a == b occurs with probability 1/N)."""
import numpy as np

from .config import CATEGORIES

_CYL = {"bottle", "can", "mug", "bowl"}


def make_object(category="bottle", n_points=4096, seed=0, n_feat=40):
    cfg = CATEGORIES[category]
    rng = np.random.default_rng(seed)
    sx, sy, sz = cfg.scale_mean
    n = n_points
    if category in _CYL:
        r, h = sx, sy
        side_area, cap_area = 2 * np.pi * r * 2 * h, np.pi * r * r
        pick = rng.random(n) < side_area / (side_area + 2 * cap_area)
        th = rng.uniform(0, 2 * np.pi, n)
        rad = r * np.sqrt(rng.random(n))
        top = rng.random(n) < 0.5
        pts = np.where(pick[:, None],
                       np.stack([r * np.cos(th), rng.uniform(-h, h, n), r * np.sin(th)], -1),
                       np.stack([rad * np.cos(th), np.where(top, h, -h), rad * np.sin(th)], -1))
        nrm = np.where(pick[:, None], np.stack([np.cos(th), np.zeros(n), np.sin(th)], -1),
                       np.stack([np.zeros(n), np.where(top, 1.0, -1.0), np.zeros(n)], -1))
    else:
        ext = np.array([sx, sy, sz])
        areas = np.array([ext[1] * ext[2], ext[0] * ext[2], ext[0] * ext[1]])
        axis = rng.choice(3, n, p=areas / areas.sum())
        sign = np.where(rng.random(n) < 0.5, 1.0, -1.0)
        pts = rng.uniform(-1, 1, (n, 3)) * ext
        pts[np.arange(n), axis] = sign * ext[axis]
        nrm = np.zeros((n, 3))
        nrm[np.arange(n), axis] = sign
    center = rng.uniform(-0.2, 0.2, 3) + np.array([0.0, 0.0, 0.8])
    pc = pts + center
    pc = pc + np.clip(cfg.res / 4 * rng.standard_normal(pc.shape), -cfg.res / 2, cfg.res / 2)
    feat = rng.standard_normal((n, n_feat))
    return dict(pc=pc.astype(np.float32), normals=nrm.astype(np.float32), feat=feat.astype(np.float32),
                center=center, cfg=cfg, category=category)


def make_pairs(n_points, pairs_per_point, seed=0):
    rng = np.random.default_rng(seed + 1000003)
    return rng.integers(0, n_points, (n_points * pairs_per_point, 2)).astype(np.int64)


def make_uniforms(n_pairs, seed=0):
    rng = np.random.default_rng(seed + 2000003)
    return rng.random((n_pairs, 2), dtype=np.float32), rng.random((n_pairs, 2), dtype=np.float32)


def closed_form_outputs(pc, center, point_idxs, cfg, quantise=True):
    """(mu, nu) of every pair w.r.t. the true centre: mu = a.u, nu = ||a - mu u|| with a taken
    relative to the centre and u the unit vector of a-b (reference utils/dataset.py:27-36), optionally
    snapped to the tr_num_bins bin values by inverting nocs/inference.py:187-188."""
    a = pc[point_idxs[:, 0]].astype(np.float64) - center
    b = pc[point_idxs[:, 1]].astype(np.float64) - center
    d = a - b
    u = d / (np.linalg.norm(d, axis=-1, keepdims=True) + 1e-7)
    mu = np.sum(a * u, -1)
    nu = np.linalg.norm(a - mu[:, None] * u, axis=-1)
    if quantise:
        nb, (v0, v1) = cfg.tr_num_bins, cfg.vote_range
        k0 = np.clip(np.rint((mu + v0) / (2 * v0) * (nb - 1)), 0, nb - 1)
        k1 = np.clip(np.rint(nu / v1 * (nb - 1)), 0, nb - 1)
        mu = k0 / (nb - 1) * 2 * v0 - v0
        nu = k1 / (nb - 1) * v1
    return np.stack([mu, nu], -1).astype(np.float32)


def closed_form_heads(pc, normals, point_idxs, cfg, quantise=True, seed=0, aux_logit=4.0, scale_noise=0.05):
    """Known-answer orientation / scale heads of every pair for the synthetic objects above, whose axes are the world axes (up =
    +y, right = +x; +z with z_right): theta = arccos(u . axis) with u the unit vector of a-b, folded by the category's up
    symmetry, the sign targets from the flipped normal of point a (reference utils/dataset.py:38-60), optionally snapped to the
    rot_num_bins values by inverting nocs/inference.py:251,255.  Returns f32[P,8] rows {theta_up, theta_right, aux_up, aux_right,
    sx, sy, sz, 0}: aux = +-aux_logit, scale logits = small noise around 0 (exp(0) * scale_mean * 2 = the object's extent,
    nocs/inference.py:335).  With closed_form_outputs this is what a perfectly trained network would emit."""
    a = pc[point_idxs[:, 0]].astype(np.float64)
    b = pc[point_idxs[:, 1]].astype(np.float64)
    d = a - b
    u = d / (np.linalg.norm(d, axis=-1, keepdims=True) + 1e-7)
    up = np.array([0.0, 1.0, 0.0])
    right = np.array([0.0, 0.0, 1.0]) if cfg.z_right else np.array([1.0, 0.0, 0.0])
    th_up = np.arccos(np.clip(u @ up, -1, 1))
    if cfg.up_sym:
        th_up = np.minimum(th_up, np.arccos(np.clip(-(u @ up), -1, 1)))
    th_right = np.arccos(np.clip(u @ right, -1, 1))
    if quantise:
        nb = cfg.rot_num_bins
        th_up = np.rint(th_up / np.pi * (nb - 1)) / (nb - 1) * np.pi
        th_right = np.rint(th_right / np.pi * (nb - 1)) / (nb - 1) * np.pi
    n = normals[point_idxs[:, 0]].astype(np.float64).copy()
    n[np.sum(n * u, -1) < 0] *= -1
    rng = np.random.default_rng(seed + 3000003)
    heads = np.zeros((point_idxs.shape[0], 8), np.float32)
    heads[:, 0], heads[:, 1] = th_up, th_right
    heads[:, 2] = np.where(n @ up > 0, aux_logit, -aux_logit)
    heads[:, 3] = np.where(n @ right > 0, aux_logit, -aux_logit)
    heads[:, 4:7] = scale_noise * rng.standard_normal((point_idxs.shape[0], 3))
    return heads


# ---------------------------------------------------------------------------------------------------------------------
# Posed objects with a known answer for the WHOLE pose (training targets and held-out checks, cppf_amd/training.py): the
# shapes above in a random rigid pose and size.  Cylinder categories get a NECK (upper 30 % of the height at 45 % of the
# radius, with a shoulder), so that up and down can be told apart from the geometry; boxes stay boxes (their axes are
# defined up to sign).  The reference trains on ShapeNet renders brought back to the canonical frame
# (utils/dataset.py:202-212); its targets are functions of (cloud, centre, axes, extents) only, which is what is returned.
def random_rotation(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _box_surface(rng, n, ext):
    """n points on the surface of the box [-ext, ext] with outward normals, faces picked by area"""
    ext = np.asarray(ext, np.float64)
    areas = np.array([ext[1] * ext[2], ext[0] * ext[2], ext[0] * ext[1]])
    axis = rng.choice(3, n, p=areas / areas.sum())
    sign = np.where(rng.random(n) < 0.5, 1.0, -1.0)
    pts = rng.uniform(-1, 1, (n, 3)) * ext
    pts[np.arange(n), axis] = sign * ext[axis]
    nrm = np.zeros((n, 3))
    nrm[np.arange(n), axis] = sign
    return pts, nrm


def _cyl_surface(rng, n, r, h, r2=None, y_sh=None):
    """n points on a capped cylinder of radius r, y in [-h, h]; with r2 / y_sh: a NECK of radius r2 above the shoulder y_sh"""
    neck = r2 is not None
    if not neck:
        r2, y_sh = r, h
    parts = [2 * np.pi * r * (y_sh + h), np.pi * r * r]   # body side, bottom cap
    if neck:
        parts += [np.pi * (r * r - r2 * r2), 2 * np.pi * r2 * (h - y_sh), np.pi * r2 * r2]   # shoulder ring, neck side, top cap
    else:
        parts += [np.pi * r * r]
    which = rng.choice(len(parts), n, p=np.array(parts) / np.sum(parts))
    th = rng.uniform(0, 2 * np.pi, n)
    u = rng.random(n)
    c, s_ = np.cos(th), np.sin(th)
    pts, nrm = np.zeros((n, 3)), np.zeros((n, 3))
    m = which == 0
    pts[m] = np.stack([r * c[m], -h + u[m] * (y_sh + h), r * s_[m]], -1)
    nrm[m] = np.stack([c[m], 0 * c[m], s_[m]], -1)
    m = which == 1
    rad = r * np.sqrt(u[m])
    pts[m] = np.stack([rad * c[m], np.full(m.sum(), -h), rad * s_[m]], -1)
    nrm[m] = [0, -1, 0]
    if neck:
        m = which == 2
        rad = np.sqrt(r2 * r2 + u[m] * (r * r - r2 * r2))
        pts[m] = np.stack([rad * c[m], np.full(m.sum(), y_sh), rad * s_[m]], -1)
        nrm[m] = [0, 1, 0]
        m = which == 3
        pts[m] = np.stack([r2 * c[m], y_sh + u[m] * (h - y_sh), r2 * s_[m]], -1)
        nrm[m] = np.stack([c[m], 0 * c[m], s_[m]], -1)
        m = which == 4
        rad = r2 * np.sqrt(u[m])
    else:
        m = which == 2
        rad = r * np.sqrt(u[m])
    pts[m] = np.stack([rad * c[m], np.full(m.sum(), h), rad * s_[m]], -1)
    nrm[m] = [0, 1, 0]
    return pts, nrm, float(np.sum(parts))


def make_posed_object(category="bottle", n_points=2048, seed=0, size_range=(0.8, 1.2), rotate=True, neck=True):
    """-> dict(pc f32[N,3], normals f32[N,3], center f64[3] (centre of the bounding box in the object's frame, like
    utils/dataset.py:162-164), R f64[3,3] (columns = the object's x, y (up), z axes in the world), half_extents f64[3] (what
    exp(scale head) * scale_mean should give, nocs/inference.py:335 / 2), cfg, category).
    bottle / can / bowl: a cylinder with a neck (up and down differ); mug: a cylinder with a HANDLE plate towards +x (the
    `right` axis of the category is defined by it); everything else: a box with the category's extents."""
    cfg = CATEGORIES[category]
    rng = np.random.default_rng(seed + 7000003)
    size = rng.uniform(*size_range)
    sx, sy, sz = (np.array(cfg.scale_mean) * size).tolist()
    n = n_points
    if category == "mug":
        r, h = sz, sy                                            # the body's radius is the z half extent; x also holds the handle
        hx, hy, hz = (sx - r), 0.6 * h, 0.13 * r               # handle plate: half extents
        body_area = 2 * np.pi * r * 2 * h + 2 * np.pi * r * r
        plate_area = 8 * (hx * hy + hx * hz + hy * hz)
        nb = int(rng.binomial(n, body_area / (body_area + plate_area)))
        pb, nb_ = _cyl_surface(rng, nb, r, h)[:2]
        pb[:, 0] -= (sx - r)                                     # bounding box centred: body axis at x = -(sx - r)
        ph, nh = _box_surface(rng, n - nb, [hx, hy, hz])
        ph[:, 0] += sx - hx                                      # the plate reaches x = +sx
        pts, nrm = np.concatenate([pb, ph]), np.concatenate([nb_, nh])
        perm = rng.permutation(n)
        pts, nrm = pts[perm], nrm[perm]
    elif category in _CYL:
        if neck:
            pts, nrm, _ = _cyl_surface(rng, n, sx, sy, 0.45 * sx, 0.4 * sy)     # the neck: upper 30 % of the height at 45 % radius
        else:
            pts, nrm, _ = _cyl_surface(rng, n, sx, sy)
    else:
        pts, nrm = _box_surface(rng, n, [sx, sy, sz])
    R = random_rotation(rng) if rotate else np.eye(3)
    center = rng.uniform(-0.2, 0.2, 3) + np.array([0.0, 0.0, 0.8])
    pc = pts @ R.T + center
    pc = pc + np.clip(cfg.res / 4 * rng.standard_normal(pc.shape), -cfg.res / 2, cfg.res / 2)     # nocs/inference.py:134
    return dict(pc=pc.astype(np.float32), normals=(nrm @ R.T).astype(np.float32), center=center, R=R,
                half_extents=np.array([sx, sy, sz]), cfg=cfg, category=category)


def philox_pairs(seed, n_pairs, n_points):
    """Host twin of the device sampler (cppf_sample_pairs / cppf_stage_batch, csrc/preproc.hip): Philox-4x32-10 keyed by the 64-bit
    `seed`, counter = {pair index, 0 | 1} -> (idx i64[P,2] uniform over [0, n_points), u_tr f32[P,2], u_rot f32[P,2] in [0, 1)), the
    same numbers bit for bit (tests/test_gpu_resident.py), so that a checker on the host can reproduce the pairs a captured chain
    drew on the device (the reference draws them with np.random.randint / torch.multinomial, nocs/inference.py:177,186,250)."""
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
    mask, s32 = np.uint64(0xFFFFFFFF), np.uint64(32)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    p = np.arange(int(n_pairs), dtype=np.uint64)

    def block(c2):
        c = [p & mask, p >> s32, np.full_like(p, c2), np.zeros_like(p)]
        k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
        for _ in range(10):
            p0, p1 = M0 * c[0], M1 * c[2]                       # 32 x 32 -> 64 bit products
            c = [(p1 >> s32) ^ c[1] ^ k0, p1 & mask, (p0 >> s32) ^ c[3] ^ k1, p0 & mask]
            k0, k1 = (k0 + W0) & mask, (k1 + W1) & mask
        return c
    a, b = block(0), block(1)
    N = np.uint64(int(n_points))
    idx = np.stack([(a[0] * N) >> s32, (a[1] * N) >> s32], -1).astype(np.int64)
    u = lambda w: ((w >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24))
    return idx, np.stack([u(a[2]), u(a[3])], -1), np.stack([u(b[0]), u(b[1])], -1)
