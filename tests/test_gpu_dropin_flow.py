"""Drop-in use of the three vote callables and the two encoders the way a reference script drives them: launch tuples
`kernel((blocks,1,1), (threads,1,1), (args...))`, host round trips between the stages, torch ops for everything the
kernels do not cover (nocs/inference.py:177-303,335 is the call sequence being exercised; the code below is this
repository's own).  Bins are chosen by arg-max instead of torch.multinomial so the flow is deterministic; the fused
device pipeline in arg-max mode (u < 0) must then produce the same pose."""
import numpy as np
import pytest
import torch

import cppf_amd.synthetic as syn
from cppf_amd.models.model import PPFEncoder
from cppf_amd.models.voting import backvote_kernel, ppf_kernel, rot_voting_kernel
from cppf_amd.utils.util import fibonacci_sphere, num_sphere_bins

pytestmark = pytest.mark.gpu
ROTS, THREADS, TOL_DEG = 72, 512, 1.5


def _gpu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def _argmax_bins(logits, n_bins):
    """[P, n_bins] logits -> f32[P] most likely bin (softmax is monotone, so its arg-max is the logits')"""
    return torch.softmax(logits, -1).argmax(-1).float()


def _blocks(n):
    return ((n + THREADS - 1) // THREADS, 1, 1)


@pytest.mark.parametrize("cat", ["bottle", "mug"])
def test_script_style_instance_equals_fused_pipeline(dev, cat):
    from cppf_amd.inference import estimate_pose
    obj = syn.make_object(cat, 1024, 77)
    cfg, cloud, normals = obj["cfg"], obj["pc"], obj["normals"]
    tb, rb = cfg.tr_num_bins, cfg.rot_num_bins
    torch.manual_seed(0)
    net = PPFEncoder(ppffcs=[84, 32, 32, 16], out_dim=2 * tb + 2 * rb + 2 + 3)
    with torch.no_grad():
        net.final.weight *= 6                       # sharper heads than a random init gives
    net = net.cuda().eval()
    feats = _gpu(obj["feat"][None])                 # stands in for the point encoder's output
    pts_b, nrm_b = _gpu(cloud[None]), _gpu(normals[None])
    sphere = np.array(fibonacci_sphere(num_sphere_bins(TOL_DEG)))
    pairs = syn.make_pairs(1024, 32, 77)            # int64 [P,2], what np.random.randint gives the script
    n_pairs = pairs.shape[0]

    # ---- first pass: centre bins -> (mu, nu)
    with torch.no_grad():
        logits = net(pts_b, nrm_b, feats, idxs=pairs)[0]
    mu = _argmax_bins(logits[:, :tb], tb) / (tb - 1) * 2 * cfg.vote_range[0] - cfg.vote_range[0]
    nu = _argmax_bins(logits[:, tb:2 * tb], tb) / (tb - 1) * cfg.vote_range[1]
    mu_nu = torch.stack([mu, nu], -1).contiguous()

    # ---- centre vote, read back, arg-max on the host like the script
    lo, hi = cloud.min(0), cloud.max(0)
    dims = ((hi - lo) / cfg.res).astype(np.int32) + 1
    grid = torch.zeros(tuple(int(v) for v in dims), dtype=torch.float32, device="cuda")
    pairs32 = _gpu(pairs, torch.int32)
    ppf_kernel(_blocks(n_pairs), (THREADS, 1, 1),
               (_gpu(cloud), mu_nu, torch.ones(cloud.shape[0], device="cuda"), pairs32, grid, _gpu(lo), np.float32(cfg.res),
                n_pairs, ROTS, int(dims[0]), int(dims[1]), int(dims[2]), True))
    votes = grid.cpu().numpy()
    cell = np.array(np.unravel_index(int(np.argmax(votes)), votes.shape))
    centre = lo + cell * cfg.res                    # float64, like corners[0] + cand * res

    # ---- back-vote filter, mask on the host, second pass on the survivors
    offsets = torch.zeros((n_pairs, 3), dtype=torch.float32, device="cuda")
    backvote_kernel(_blocks(n_pairs), (THREADS, 1, 1),
                    (_gpu(cloud), mu_nu, offsets, pairs32, _gpu(lo), np.float32(cfg.res), n_pairs, ROTS, int(dims[0]),
                     int(dims[1]), int(dims[2]), _gpu(centre, torch.float32), np.float32(3 * cfg.res)))
    alive = np.any(offsets.cpu().numpy() != 0, -1)
    kept = pairs[alive]
    with torch.no_grad():
        logits2 = net(pts_b, nrm_b, feats, idxs=kept)[0]
    angle = {"up": _argmax_bins(logits2[:, 2 * tb:2 * tb + rb], rb) / (rb - 1) * np.pi,
             "right": _argmax_bins(logits2[:, 2 * tb + rb:2 * tb + 2 * rb], rb) / (rb - 1) * np.pi}
    sign_logit = {"up": logits2[:, -5], "right": logits2[:, -4]}
    log_scale = logits2[:, -3:]

    # ---- orientation vote per axis: candidates, sphere histogram with a torch mm, sign by the auxiliary head
    kept32 = _gpu(kept, torch.int32)
    mu_nu_kept = mu_nu[torch.from_numpy(np.nonzero(alive)[0]).cuda()].contiguous()
    sph_t = torch.tensor(sphere.T, dtype=torch.float32, device="cuda")
    a, b = cloud[kept[:, 0]], cloud[kept[:, 1]]
    ab = a - b
    ab_unit = ab / (np.sqrt(np.sum(ab ** 2, -1)) + 1e-7)[:, None]
    n_a = normals[kept[:, 0]].copy()
    n_a[np.sum(n_a * ab_unit, -1) < 0] *= -1
    bce = torch.nn.BCEWithLogitsLoss()
    axes = {}
    for name in (("up", "right") if cfg.regress_right else ("up",)):
        cands = torch.zeros((kept.shape[0], ROTS, 3), dtype=torch.float32, device="cuda")
        rot_voting_kernel(_blocks(kept.shape[0]), (THREADS, 1, 1),
                          (_gpu(cloud), mu_nu_kept, angle[name].contiguous(), cands, kept32, _gpu(lo), np.float32(cfg.res),
                           kept.shape[0], ROTS, int(dims[0]), int(dims[1]), int(dims[2])))
        used = cands[:10000].reshape(-1, 3)          # the first 10 000 survivors (the script shuffles; pairs are i.i.d.)
        hits = (used.mm(sph_t) > np.cos(TOL_DEG / 180 * np.pi)).sum(0).cpu().numpy()
        axis = np.array(sphere[int(np.argmax(hits))])
        with torch.no_grad():
            side = torch.from_numpy((np.sum(n_a * axis, -1) > 0).astype(np.float32)).cuda()
            flip = bce(sign_logit[name], 1.0 - side).item() < bce(sign_logit[name], side).item()
        axes[name] = -axis if flip else axis
    scale = np.exp(log_scale.mean(0).cpu().numpy()) * cfg.scale_mean * 2

    # ---- the fused device pipeline on the same inputs, arg-max bins (u < 0)
    neg = torch.full((n_pairs, 2), -1.0, device="cuda")
    with torch.no_grad():
        fused = estimate_pose(net, pts_b[0], nrm_b[0], feats[0], _gpu(pairs), neg, neg, cfg, sphere, pc_host=cloud)
    assert alive.sum() > 20 and fused["n_surv"] == int(alive.sum())
    np.testing.assert_allclose(fused["T"], centre, atol=1e-9)
    np.testing.assert_allclose(fused["up"], axes["up"], atol=1e-9)
    np.testing.assert_allclose(fused["scale"], scale, rtol=2e-5)              # fp32 mean in torch vs fp64 sums here
    if cfg.regress_right:
        np.testing.assert_allclose(fused["best_right"] * np.sign(fused["best_right"] @ axes["right"]), axes["right"], atol=1e-9)


def test_reference_call_sequence_recovers_a_trained_pose(dev):
    """cppf_amd/dropin.py -- the per-instance body of nocs/inference.py:177-339 with its own stage order, launch tuples, host round
    trips, torch.multinomial and np.random.shuffle, only the two imports switched -- on a held-out posed object with the trained
    networks: the pose it finds is the true one, like the fused estimate_pose (same networks, its own random draws)."""
    import os
    from conftest import GOLDEN
    from cppf_amd import training
    from cppf_amd.dropin import reference_style_instance
    sph = np.array(fibonacci_sphere(num_sphere_bins(TOL_DEG)))
    for cat, seed in (("bottle", 900011), ("mug", 900012)):
        cfg = syn.CATEGORIES[cat]
        penc, enc = training.load_weights(os.path.join(GOLDEN, f"trained_{cat}.npz"), cfg, dev)
        ob = syn.make_posed_object(cat, 1400, seed)
        torch.manual_seed(1)
        pose = reference_style_instance(penc, enc, ob["pc"], ob["normals"], cfg, sph, n_pairs=100000, rng=np.random.RandomState(3))
        e1 = training.pose_errors(pose, ob)
        e2 = training.pose_errors(training.infer(penc, enc, ob, dev, seed=5, sphere=sph), ob)
        for e in (e1, e2):
            assert e["t_cells"] <= 3.0 and e["up_deg_mod_sign"] <= 10.0 and e["scale_rel"] <= 0.2, (cat, e1, e2)
        assert pose["n_surv"] > 5000


def test_kernels_accept_any_cuda_array_interface_object(dev):
    """"accepting torch tensors (any object with data_ptr() / __cuda_array_interface__)": a foreign device array
    (here a minimal stand-in for a CuPy array) goes through the three callables zero-copy, outputs updated in place"""
    class Foreign:
        def __init__(self, t):
            self._t = t          # keeps the memory alive
            kinds = {torch.float32: "<f4", torch.int32: "<i4"}
            self.__cuda_array_interface__ = {"shape": tuple(t.shape), "typestr": kinds[t.dtype], "data": (t.data_ptr(), False),
                                             "version": 2, "strides": None}
    obj = syn.make_object("bottle", 600, 5)
    cfg = obj["cfg"]
    pairs = syn.make_pairs(600, 20, 5).astype(np.int32)
    mu_nu = syn.closed_form_outputs(obj["pc"], obj["center"], pairs, cfg)
    lo = obj["pc"].min(0)
    dims = ((obj["pc"].max(0) - lo) / cfg.res).astype(np.int32) + 1
    P = pairs.shape[0]
    args_t = (_gpu(obj["pc"]), _gpu(mu_nu), torch.ones(600, device="cuda"), _gpu(pairs), None, _gpu(lo), np.float32(cfg.res), P, ROTS,
              int(dims[0]), int(dims[1]), int(dims[2]), True)
    g_t = torch.zeros(tuple(int(v) for v in dims), device="cuda")
    g_f = torch.zeros_like(g_t)
    ppf_kernel(_blocks(P), (THREADS, 1, 1), args_t[:4] + (g_t,) + args_t[5:])
    ppf_kernel(_blocks(P), (THREADS, 1, 1), tuple(Foreign(a) if isinstance(a, torch.Tensor) else a for a in args_t[:4]) + (Foreign(g_f),) +
               (Foreign(args_t[5]),) + args_t[6:])
    assert torch.equal(g_t, g_f) and float(g_f.sum()) > 1000
    centre = _gpu((lo + np.array(np.unravel_index(int(g_t.argmax()), g_t.shape)) * cfg.res).astype(np.float32))
    o_t, o_f = torch.zeros((P, 3), device="cuda"), torch.zeros((P, 3), device="cuda")
    for out, wrap in ((o_t, lambda a: a), (o_f, Foreign)):
        backvote_kernel(_blocks(P), (THREADS, 1, 1), (wrap(args_t[0]), wrap(args_t[1]), wrap(out), wrap(args_t[3]), wrap(args_t[5]),
                                                       np.float32(cfg.res), P, ROTS, int(dims[0]), int(dims[1]), int(dims[2]),
                                                       wrap(centre), np.float32(3 * cfg.res)))
    assert torch.equal(o_t, o_f) and bool((o_f != 0).any())
    rot = torch.rand(P, device="cuda") * 3.14
    c_t, c_f = torch.zeros((P, ROTS, 3), device="cuda"), torch.zeros((P, ROTS, 3), device="cuda")
    for out, wrap in ((c_t, lambda a: a), (c_f, Foreign)):
        rot_voting_kernel(_blocks(P), (THREADS, 1, 1), (wrap(args_t[0]), wrap(args_t[1]), wrap(rot), wrap(out), wrap(args_t[3]),
                                                         wrap(args_t[5]), np.float32(cfg.res), P, ROTS, int(dims[0]), int(dims[1]), int(dims[2])))
    assert torch.equal(c_t, c_f)
