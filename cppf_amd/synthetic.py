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
