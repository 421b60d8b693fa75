"""The kernel configuration the HEADLINE is timed on, under the oracle directly.

bench.py's timed regions run `cppf_pair_mlp_decode_batch` on FOUR equal-length C2 pair lists (N=4096, K=128, P=524 288 each):
the XCD-pinned workgroup mapping (`per_xcd = 8 / n_lists`, csrc/pair_mlp.hip pair_mlp_batch_kernel), then every object's vote +
arg-max (CenterBatchPipeline), replayed from a hipGraph.  Here that branch -- with 2, 4 and 8 lists, distinct clouds, int64 and
int32 pair lists, with and without the orientation / scale heads -- is compared per pair with oracle.pair_mlp(order=1) +
decode_center / decode_rot (models/model.py:117-137, nocs/inference.py:185-188,245-256), NOT with another launch of the same
kernel; the launch geometry is read back through cppf_pair_mlp_batch_plan so that the test fails if the lists stop taking the
XCD-pinned branch.  Then one CenterBatchPipeline of four C2 objects: (mu, nu) of every pair, every grid cell against the exact
fp64 vote sum, the four arg-maxes (nocs/inference.py:207-211)."""
import numpy as np
import pytest
import torch

import cppf_amd.synthetic as syn
from cppf_amd.models.model import PPFEncoder, batch_plan, forward_decode_batch
from test_gpu_configs import check_argmax, make_encoder, ocfg_of, seeded_sd
from test_gpu_parity import check_grid

pytestmark = pytest.mark.gpu

N_POINTS, K = 4096, 128            # BASELINE.json configs[1]
P = N_POINTS * K


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


_cache = {}


def c2_object(oracle, seed, sd):
    """C2-size object `seed` (the benchmark's generator and seeds: bench.make_center_set uses 100 * rank + i) with the oracle's
    decoded (mu, nu) and heads of every pair; computed once per session"""
    hit = _cache.get(seed)
    if hit is None:
        ob = syn.make_object("bottle", N_POINTS, seed=seed)
        idx = syn.make_pairs(N_POINTS, K, seed=seed)
        u_tr, u_rot = syn.make_uniforms(P, seed=seed)
        cfg = ob["cfg"]
        lo = oracle.pair_mlp(ob["pc"], ob["normals"], ob["feat"], idx, sd, cfg.ppffcs, cfg.out_dim, order=1)
        outputs, _ = oracle.decode_center(lo, u_tr, cfg.tr_num_bins, cfg.vote_range)
        heads, _ = oracle.decode_rot(lo, u_rot, cfg.tr_num_bins, cfg.rot_num_bins)
        del lo
        hit = _cache[seed] = dict(ob=ob, cfg=cfg, idx=idx, u_tr=u_tr, u_rot=u_rot, outputs=outputs, heads=heads)
    return hit


@pytest.mark.parametrize("n_lists,idx_dtype,with_heads", [
    (4, np.int64, False),            # the headline's launch: bench.py default (--mlp-batch 4, centre heads)
    (4, np.int32, True), (4, np.int64, True), (4, np.int32, False),
    (2, np.int64, False), (2, np.int32, True),
    (8, np.int64, True), (8, np.int32, False)])
def test_xcd_pinned_batch_matches_oracle_per_pair(oracle, dev, n_lists, idx_dtype, with_heads):
    sd = seeded_sd(0)
    enc = make_encoder(sd, dev)
    plan = batch_plan([P] * n_lists)
    assert plan["per_xcd"] == 8 // n_lists and plan["grid"] % 8 == 0 and plan["grid"] >= 8 * n_lists, plan
    objs = [c2_object(oracle, j, sd) for j in range(n_lists)]
    items = []
    for o in objs:
        it = dict(encoder=enc, pc=t(o["ob"]["pc"], dev), pc_normal=t(o["ob"]["normals"], dev), feat=t(o["ob"]["feat"], dev),
                  idxs=t(o["idx"].astype(idx_dtype), dev), u_tr=t(o["u_tr"], dev), vote_range=o["cfg"].vote_range)
        if with_heads:
            it["u_rot"] = t(o["u_rot"], dev)
        items.append(it)
    with torch.no_grad():
        for rep in range(2):
            got = forward_decode_batch(items)
    torch.cuda.synchronize()
    for j, ((outputs, heads), o) in enumerate(zip(got, objs)):
        np.testing.assert_array_equal(outputs.cpu().numpy(), o["outputs"], err_msg=f"list {j}: (mu, nu)")
        assert (heads is not None) == with_heads
        if with_heads:
            np.testing.assert_array_equal(heads.cpu().numpy(), o["heads"], err_msg=f"list {j}: heads")
    # the lists are distinct problems (a launch that mixed them up would not pass by accident)
    assert not np.array_equal(objs[0]["outputs"], objs[1]["outputs"])


def test_nearly_equal_lists_take_the_pinned_branch_too(oracle, dev):
    """lengths within 10 % of each other (the plan's rule), not multiples of the tile: still XCD-pinned, still the oracle's bits"""
    sd = seeded_sd(0)
    enc = make_encoder(sd, dev)
    lens = [P, P - 37, P - 40000, P - 16]
    assert batch_plan(lens)["per_xcd"] == 2
    objs = [c2_object(oracle, j, sd) for j in range(4)]
    items = [dict(encoder=enc, pc=t(o["ob"]["pc"], dev), pc_normal=t(o["ob"]["normals"], dev), feat=t(o["ob"]["feat"], dev),
                  idxs=t(o["idx"][:n], dev), u_tr=t(o["u_tr"][:n], dev), u_rot=t(o["u_rot"][:n], dev), vote_range=o["cfg"].vote_range)
             for o, n in zip(objs, lens)]
    with torch.no_grad():
        got = forward_decode_batch(items)
    for (outputs, heads), o, n in zip(got, objs, lens):
        np.testing.assert_array_equal(outputs.cpu().numpy(), o["outputs"][:n])
        np.testing.assert_array_equal(heads.cpu().numpy(), o["heads"][:n])
    # beyond 10 %: contiguous workgroup ranges
    assert batch_plan([P, P // 2, P, P])["per_xcd"] == 0 and batch_plan([P] * 3)["per_xcd"] == 0


@pytest.mark.parametrize("vote_batch,vote_workgroups", [(True, 0), (True, 128), (False, 128), (False, 0)])
def test_center_batch_pipeline_of_four_c2_objects(oracle, dev, vote_batch, vote_workgroups):
    """the headline's captured chain: 4 C2 objects, one pair-kernel launch, then the four votes -- in ONE vote + ONE reduce launch
    (64 workgroups per object: the timed regions' default; 128), or a launch per object (round 4's chain: 128 wide, full width) --
    replayed: every pair's (mu, nu), every grid cell, the four arg-maxes"""
    from cppf_amd.inference import CenterBatchPipeline, CenterPipeline, grid_shape
    sd = seeded_sd(0)
    enc = make_encoder(sd, dev)
    objs = [c2_object(oracle, j, sd) for j in range(4)]
    pipes = []
    for o in objs:
        corners, dims = grid_shape(o["ob"]["pc"], o["cfg"].res)
        p = CenterPipeline(enc, o["cfg"], N_POINTS, P, dims, dev, 72, adaptive=True, with_heads=False, vote_workgroups=vote_workgroups)
        p.load(o["ob"]["pc"], o["ob"]["normals"], o["ob"]["feat"], o["idx"], o["u_tr"], o["u_rot"], corners[0].copy())
        pipes.append((p, corners, dims))
    bp = CenterBatchPipeline([p for p, _, _ in pipes], vote_batch=vote_batch, vote_workgroups=vote_workgroups)
    for rep in range(3):                 # capture, replay, replay
        res = bp.run()
    torch.cuda.synchronize()
    for (p, corners, dims), o, (oi, ov) in zip(pipes, objs, res):
        np.testing.assert_array_equal(p.outputs.cpu().numpy(), o["outputs"])
        idx32 = o["idx"].astype(np.int32)
        g64, _ = check_grid(oracle, p.grid.cpu().numpy(), o["ob"]["pc"], o["outputs"], idx32, corners[0], dims, o["cfg"].res, 72, True,
                            bits_slack=2 if (vote_workgroups or vote_batch) else 0)
        go = np.zeros(dims, np.float32)
        oracle.ppf_voting(o["ob"]["pc"], o["outputs"], np.ones(N_POINTS, np.float32), idx32, go, corners[0], o["cfg"].res, 72, True)
        oflat, opeak = oracle.grid_argmax(go)
        ref = dict(argmax=int(oflat), outputs=o["outputs"], dims=dims, corner=corners[0], grid=go)
        check_argmax(oracle, int(oi.item()), ref, o["ob"], o["idx"], o["cfg"].res)
        assert int(oi.item()) == int(np.argmax(g64))            # and the exact sum's, always
        np.testing.assert_allclose(float(ov.item()), opeak, rtol=2e-5)
