"""BASELINE.json `configs[1..4]` at their stated sizes, whole chains against the oracle (through the C ABI like every GPU test).

  C2  single object N=4096 K=128  (P =   524 288)  centre chain + full pose             nocs/inference.py:177-339
  C3  single object N=4096 K=256  (P = 1 048 576)  fused chain (CenterPipeline) + pose
  C4  one GPU's share of the 64-object batch: 8 mixed-category C2-size objects through BatchPoseRunner
  C5  per-instance N=8192 K=256   (P = 2 097 152), fine grid (res 2e-3), full PosePipeline

Every discrete outcome (arg-max index, survivor count, survivor mask, sampled bins via (mu, nu)) is compared bit for bit, T / up
to 1e-12 and the scale to 1e-6 -- far inside the 1e-4 of the north star.  The oracle's vote grid is the serial fp32 sum (one of
the orders the reference's atomicAdd may take); the device's is the exact fixed-point sum, so should the two arg-max ever differ
the test demands that the device agrees with the exact fp64 sum and that the oracle's top two cells are closer than fp32
summation error (a genuine tie of the reference itself)."""
import numpy as np
import pytest
import torch

import cppf_amd.synthetic as syn
from cppf_amd.config import CATEGORIES, NOCS_CATEGORIES
from cppf_amd.models.model import PPFEncoder

pytestmark = pytest.mark.gpu


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def seeded_sd(seed=0, gain=1.0):
    torch.manual_seed(seed)
    enc = PPFEncoder([84, 32, 32, 16], 141)
    sd = {k: v.detach().numpy().copy() for k, v in enc.state_dict().items()}
    for k in ("final.weight", "final.bias"):            # gain > 1: peakier bin distributions, more pairs survive the back-vote
        sd[k] = sd[k] * np.float32(gain)
    return sd


def make_encoder(sd, dev):
    enc = PPFEncoder([84, 32, 32, 16], 141)
    enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return enc.to(dev).eval()


def ocfg_of(cfg, res=None):
    return dict(res=cfg.res if res is None else res, tr_num_bins=32, rot_num_bins=36, vote_range=cfg.vote_range,
                scale_mean=cfg.scale_mean, regress_right=cfg.regress_right, ppffcs=[84, 32, 32, 16], out_dim=141)


def with_res(cfg, res):
    import dataclasses
    return dataclasses.replace(cfg, res=res)


def check_argmax(oracle, flat_dev, o, ob, idx, cfg_res, num_rots=72):
    """bit-exact arg-max; a mismatch is tolerated only as a tie of the reference's own fp32 summation (see module docstring)"""
    if flat_dev == o["argmax"]:
        return
    g64, cnt = oracle.ppf_voting_f64(ob["pc"], o["outputs"], np.ones(ob["pc"].shape[0], np.float32), idx.astype(np.int32),
                                     o["dims"], o["corner"], cfg_res, num_rots, True)
    assert flat_dev == int(np.argmax(g64)), (flat_dev, o["argmax"], int(np.argmax(g64)))
    top = np.sort(o["grid"].reshape(-1))[-2:]
    assert top[1] - top[0] <= 2.0 ** -22 * top[1] * max(int(cnt.reshape(-1)[flat_dev]), 1) ** 0.5 + 1e-6, top


def check_pose(r, o, cfg):
    np.testing.assert_allclose(r["T"], o["T"], rtol=0, atol=1e-12)
    assert r["n_surv"] == int(o["mask"].sum())
    np.testing.assert_allclose(r["up"], o["up"], atol=1e-12)
    if cfg.regress_right:
        # right before the orthogonalisation of nocs/inference.py:305-312 is the oracle's `right`; compare after it
        right = o["right"] - np.dot(o["up"], o["right"]) * o["up"]
        right = right / (np.linalg.norm(right) + 1e-9)
        np.testing.assert_allclose(r["right"], right, atol=1e-12)
    np.testing.assert_allclose(r["scale"], o["scale"], rtol=1e-6)
    np.testing.assert_allclose(r["peak"], o["peak"], rtol=2e-5)


def single_object_chain(oracle, golden, dev, cat, n_points, k, seed, res=None, gain=4.0):
    """CenterPipeline (the benchmarked chain) and PosePipeline (both captured forms) on one object vs oracle.estimate_pose"""
    from cppf_amd.inference import CenterPipeline, PosePipeline, grid_shape
    ob = syn.make_object(cat, n_points, seed)
    cfg = ob["cfg"] if res is None else with_res(ob["cfg"], res)
    idx = syn.make_pairs(n_points, k, seed)
    P = idx.shape[0]
    u_tr, u_rot = syn.make_uniforms(P, seed)
    sd = seeded_sd(0, gain)
    enc = make_encoder(sd, dev)
    sph = golden("sphere.npz")["pts"]
    corners, dims = grid_shape(ob["pc"], cfg.res)
    o = oracle.estimate_pose(ob["pc"], ob["normals"], ob["feat"], idx, sd, ocfg_of(cfg), u_tr, u_rot, sph)
    assert tuple(int(d) for d in o["dims"]) == tuple(dims)

    cp = CenterPipeline(enc, cfg, n_points, P, dims, dev)
    cp.load(ob["pc"], ob["normals"], ob["feat"], idx, u_tr, u_rot, corners[0].copy())
    for _ in range(2):                                   # capture, then replay
        oi, ov = cp.run()
    np.testing.assert_array_equal(cp.outputs.cpu().numpy(), o["outputs"])          # (mu, nu) of every pair: sampled bins
    np.testing.assert_array_equal(cp.heads.cpu().numpy(), o["heads"])              # every head of every pair
    check_argmax(oracle, int(oi.item()), o, ob, idx, cfg.res)
    np.testing.assert_allclose(float(ov.item()), o["peak"], rtol=2e-5)
    del cp

    pp = PosePipeline(enc, cfg, n_points, P, dims, dev, sph)
    pp.load(ob["pc"], ob["normals"], ob["feat"], idx, u_tr, u_rot, corners[0].copy())
    for form in (False, True):                           # split (second pass on the survivors) and full-first
        pp.adapt(P if form else 0)
        r = pp.run()
        check_argmax(oracle, r["argmax"], o, ob, idx, cfg.res)
        if r["argmax"] == o["argmax"]:
            check_pose(r, o, cfg)
            np.testing.assert_array_equal(r["ws"].mask.cpu().numpy().astype(bool), o["mask"])
            np.testing.assert_array_equal(r["heads"].cpu().numpy()[o["mask"]], o["heads"][o["mask"]])
    return o, P


def test_c2_full_chain_matches_oracle(oracle, golden, dev):
    """BASELINE.json configs[1]: N=4096, K=128, bottle -- the benchmark's own object (seed 0) and weights"""
    o, P = single_object_chain(oracle, golden, dev, "bottle", 4096, 128, 0, gain=1.0)
    assert P == 524288


def test_c2_full_chain_peaky_weights(oracle, golden, dev):
    """same size, final layer x4: thousands of survivors reach the second pass, orientation vote and scale"""
    o, P = single_object_chain(oracle, golden, dev, "camera", 4096, 128, 7, gain=4.0)
    assert int(o["mask"].sum()) > 1000


def test_c3_fused_chain_matches_oracle(oracle, golden, dev):
    """BASELINE.json configs[2]: N=4096, K=256 (P = 1 048 576) through the fused PPF -> MLP(MFMA) -> decode -> vote chain"""
    o, P = single_object_chain(oracle, golden, dev, "bottle", 4096, 256, 1, gain=4.0)
    assert P == 1048576


def test_c5_full_chain_fine_grid_matches_oracle(oracle, golden, dev):
    """BASELINE.json configs[4]: per-instance N=8192, K=256 (P = 2 097 152), fine grid res 2e-3 (52x152x52 class, many LDS
    tiles): MLP -> decode -> vote -> arg-max -> back-vote -> second pass -> orientation -> scale, one chain"""
    o, P = single_object_chain(oracle, golden, dev, "bottle", 8192, 256, 3, res=2e-3, gain=4.0)
    assert P == 2097152 and int(np.prod(o["dims"])) > 300000


def test_c4_one_gpu_share_mixed_categories(oracle, golden, dev):
    """BASELINE.json configs[3]: 64 mixed-category objects, 8 per GPU -- one GPU's share at the stated size (8 objects of
    N=4096, K=128 cycling the six NOCS categories) through BatchPoseRunner, every record against the oracle chain"""
    from cppf_amd.batch import BatchPoseRunner
    sd = seeded_sd(0, 4.0)
    encoders = {c: make_encoder(sd, dev) for c in NOCS_CATEGORIES}
    objects = []
    for j in range(8):
        cat = NOCS_CATEGORIES[j % 6]
        ob = syn.make_object(cat, 4096, 200 + j)
        idx = syn.make_pairs(4096, 128, 200 + j)
        u_tr, u_rot = syn.make_uniforms(idx.shape[0], 200 + j)
        objects.append(dict(pc=ob["pc"], normals=ob["normals"], feat=ob["feat"], point_idxs=idx, u_tr=u_tr, u_rot=u_rot,
                            cfg=ob["cfg"]))
    runner = BatchPoseRunner(encoders, dev)
    recs = runner.run(objects).cpu().numpy()
    recs2 = runner.run(objects).cpu().numpy()            # replays the cached graphs (forms may have adapted: same records)
    np.testing.assert_array_equal(recs, recs2)
    assert recs.shape == (8, 20) and recs[:, 15].tolist() == list(range(8))
    sph = golden("sphere.npz")["pts"]
    for j, obj in enumerate(objects):
        cfg = obj["cfg"]
        o = oracle.estimate_pose(obj["pc"], obj["normals"], obj["feat"], obj["point_idxs"], sd, ocfg_of(cfg), obj["u_tr"],
                                 obj["u_rot"], sph)
        check_argmax(oracle, int(recs[j, 12]), o, obj, obj["point_idxs"], cfg.res)
        if int(recs[j, 12]) != o["argmax"]:
            continue
        assert int(recs[j, 14]) == int(o["mask"].sum()), (j, cfg.category)
        np.testing.assert_allclose(recs[j, 0:3], o["T"], atol=1e-12)
        np.testing.assert_allclose(recs[j, 3:6], o["up"], atol=1e-12)
        np.testing.assert_allclose(recs[j, 9:12], o["scale"], rtol=1e-6)
