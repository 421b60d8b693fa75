"""PoseChain / BatchPoseRunner chains: the instances of a frame (the reference loops over them, nocs/inference.py:120-339) share
their launches -- one pair-kernel launch (cppf_pair_mlp_decode_batch), one vote + one reduce launch (cppf_vote_argmax_batch) and the
six launches of cppf_pose_tail_batch for the whole chain.  Per instance the 21-double pose record must be the record of the
instance's own captured pipeline BIT FOR BIT (both captured forms), and the oracle's pose."""
import numpy as np
import pytest
import torch

import cppf_amd.synthetic as syn
from cppf_amd.config import NOCS_CATEGORIES
from cppf_amd.inference import PoseChain, PosePipeline, grid_shape
from test_gpu_configs import check_argmax, check_pose, make_encoder, ocfg_of, seeded_sd

pytestmark = pytest.mark.gpu


def make_members(dev, sph, enc, specs, dynamic, point_encoder=None, **kw):
    pipes, data = [], []
    for j, (cat, n, k, seed) in enumerate(specs):
        ob = syn.make_object(cat, n, seed)
        idx = syn.make_pairs(n, k, seed)
        u_tr, u_rot = syn.make_uniforms(idx.shape[0], seed)
        corners, dims = grid_shape(ob["pc"], ob["cfg"].res)
        if dynamic:
            p = PosePipeline(enc, ob["cfg"], 2048, idx.shape[0], False, dev, sph, dynamic=True, point_encoder=point_encoder, **kw)
        else:
            p = PosePipeline(enc, ob["cfg"], n, idx.shape[0], dims, dev, sph, point_encoder=point_encoder, **kw)
        p.load(ob["pc"], ob["normals"], None if point_encoder is not None else ob["feat"], idx, u_tr, u_rot, corners[0].copy(), dims=dims if dynamic else None)
        pipes.append(p)
        data.append(dict(ob=ob, idx=idx, u_tr=u_tr, u_rot=u_rot, corners=corners, dims=dims))
    return pipes, data


@pytest.mark.parametrize("dynamic", [False, True])
def test_chain_records_equal_member_pipelines_and_oracle(oracle, golden, dev, dynamic):
    sph = golden("sphere.npz")["pts"]
    sd = seeded_sd(0, 4.0)
    enc = make_encoder(sd, dev)
    specs = [("bottle", 1500, 40, 1), ("mug", 1200, 50, 2), ("laptop", 1800, 32, 3), ("camera", 900, 64, 4), ("can", 1024, 48, 5)]
    pipes, data = make_members(dev, sph, enc, specs, dynamic)
    chain = PoseChain(pipes)
    for form in (False, True):                    # split (second pass on the survivors) and full-first
        want = []
        for p in pipes:
            p.adapt(p.idx.shape[0] if form else 0)
            r = p.run()
            want.append(p.ws.rec.cpu().numpy().copy())
        chain.full_first = form
        for rep in range(3):                      # capture, replay, replay
            recs = torch.zeros((len(pipes), 21), dtype=torch.float64, device=dev)
            chain.run_async(list(recs))
            torch.cuda.synchronize()
            got = recs.cpu().numpy()
            for j, w in enumerate(want):
                assert np.array_equal(got[j], w), (form, rep, j, got[j], w)
    poses = chain.run()
    for pose, d, (cat, n, k, seed) in zip(poses, data, specs):
        cfg = d["ob"]["cfg"]
        o = oracle.estimate_pose(d["ob"]["pc"], d["ob"]["normals"], d["ob"]["feat"], d["idx"], sd, ocfg_of(cfg), d["u_tr"], d["u_rot"], sph)
        check_argmax(oracle, pose["argmax"], o, d["ob"], d["idx"], cfg.res)
        if pose["argmax"] == o["argmax"]:
            check_pose(pose, o, cfg)
            np.testing.assert_array_equal(pose["ws"].mask.cpu().numpy().astype(bool), o["mask"])
    # new inputs for one member: the captured chain reads the members' static buffers
    d = data[1]
    u2, v2 = syn.make_uniforms(d["idx"].shape[0], 99)
    pipes[1].load(None, None, None, None, u2, v2, None)
    poses2 = chain.run()
    solo = pipes[1].run()
    assert poses2[1]["argmax"] == solo["argmax"] and np.array_equal(poses2[1]["T"], solo["T"]) and poses2[1]["n_surv"] == solo["n_surv"]
    assert poses2[0]["argmax"] == poses[0]["argmax"] and poses2[1]["n_surv"] != poses[1]["n_surv"]
    with pytest.raises(ValueError):
        PoseChain([pipes[0], pipes[0]])


@pytest.mark.parametrize("n_bins,gain", [(3600, 12.0), (37, 1.0), (480, 12.0)])
def test_chain_sphere_count_equals_member_pipelines_on_other_spheres(dev, n_bins, gain):
    """the chain's orientation count (rot_sphere_band_even_body: a block owns a contiguous share of the survivors) against the
    members' own launches (groups of 2 or 8 pairs) on a fine sphere (0.2 degrees: 3 600 bins, above the default 64 KB of LDS), a
    coarse one and the default, with few survivors and with the cap on the voting pairs reached (nocs/inference.py:277; set to 1 000
    here): records bit for bit, both forms"""
    from cppf_amd.utils.util import fibonacci_sphere
    sph = np.array(fibonacci_sphere(n_bins))
    enc = make_encoder(seeded_sd(3, gain), dev)
    specs = [("bottle", 1500, 40, 11), ("mug", 1200, 10, 12), ("laptop", 1800, 64, 13), ("bowl", 700, 3, 14)]
    pipes, _ = make_members(dev, sph, enc, specs, True, max_rot_pairs=1000)
    chain = PoseChain(pipes)
    survivors = []
    for form in (False, True):
        want = []
        for p in pipes:
            p.adapt(p.idx.shape[0] if form else 0)
            p.run()
            want.append(p.ws.rec.cpu().numpy().copy())
        chain.full_first = form
        for rep in range(2):
            recs = torch.zeros((len(pipes), 21), dtype=torch.float64, device=dev)
            chain.run_async(list(recs))
            torch.cuda.synchronize()
            got = recs.cpu().numpy()
            for j, w in enumerate(want):
                assert np.array_equal(got[j], w), (form, rep, j, got[j], w)
        survivors = [int(w[18]) for w in want]
    assert min(survivors) < 1000 and (gain < 2 or max(survivors) > 1000)      # (both sides of the cap are exercised)


def test_chain_with_point_encoders_and_many_tile_member(oracle, golden, dev):
    """kNN + SPRIN at the head of the chain (nocs/inference.py:180-181); a member whose grid needs >= 4 tiles (fine resolution)
    takes its own vote launches inside the chain"""
    import dataclasses
    from cppf_amd.models.model import PointEncoder
    sph = golden("sphere.npz")["pts"]
    enc = make_encoder(seeded_sd(0, 4.0), dev)
    torch.manual_seed(5)
    penc = PointEncoder(k=60, spfcs=[32, 64, 32, 32], num_layers=1, out_dim=32).eval().to(dev)
    specs = [("bottle", 1100, 40, 11), ("bowl", 1300, 36, 12), ("mug", 1000, 44, 13)]
    pipes, data = make_members(dev, sph, enc, specs, True, point_encoder=penc)
    # a fourth member on a fine grid: many tiles
    ob = syn.make_object("bottle", 1500, 14)
    cfg = dataclasses.replace(ob["cfg"], res=2e-3)
    idx = syn.make_pairs(1500, 40, 14)
    u_tr, u_rot = syn.make_uniforms(idx.shape[0], 14)
    corners, dims = grid_shape(ob["pc"], cfg.res)
    pm = PosePipeline(enc, cfg, 2048, idx.shape[0], True, dev, sph, dynamic=True, point_encoder=penc)
    pm.load(ob["pc"], ob["normals"], None, idx, u_tr, u_rot, corners[0].copy(), dims=dims)
    assert pm.many_tiles
    pipes.append(pm)
    want = []
    for p in pipes:
        p.run()
        want.append(p.ws.rec.cpu().numpy().copy())
    chain = PoseChain(pipes)
    for rep in range(2):
        recs = torch.zeros((len(pipes), 21), dtype=torch.float64, device=dev)
        chain.run_async(list(recs))
        torch.cuda.synchronize()
        for j, w in enumerate(want):
            assert np.array_equal(recs[j].cpu().numpy(), w), (rep, j)


def test_runner_chains_equal_per_instance_runs(oracle, golden, dev):
    """BatchPoseRunner: the same mixed-category batch through chains (captured on the second sighting of a combination) and with
    chain_len = 1 -- identical records; the chains were really used"""
    from cppf_amd.batch import BatchPoseRunner
    sd = seeded_sd(0, 4.0)
    encs = {c: make_encoder(sd, dev) for c in NOCS_CATEGORIES}
    objects = []
    for j in range(8):
        ob = syn.make_object(NOCS_CATEGORIES[j % 6], 1024 + 100 * (j % 4), 400 + j)
        idx = syn.make_pairs(ob["pc"].shape[0], 40, 400 + j)
        u_tr, u_rot = syn.make_uniforms(idx.shape[0], 400 + j)
        objects.append(dict(pc=ob["pc"], normals=ob["normals"], feat=ob["feat"], point_idxs=idx, u_tr=u_tr, u_rot=u_rot, cfg=ob["cfg"]))
    solo = BatchPoseRunner(encs, dev, chain_len=1)
    want = solo.run(objects).cpu().numpy()
    runner = BatchPoseRunner(encs, dev, chain_len=3)  # (the default shares launches only from 2 x n_lanes instances per lane on)
    got1 = runner.run(objects).cpu().numpy()          # first sighting: the members' own graphs
    assert not runner._chains
    got2 = runner.run(objects).cpu().numpy()          # second: chains captured
    got3 = runner.run(objects).cpu().numpy()          # third: replayed
    assert len(runner._chains) == 3 and sorted(len(c.pipes) for c in runner._chains.values()) == [2, 3, 3]
    for g in (got1, got2, got3):
        np.testing.assert_array_equal(g, want)
    # a different batch composition: new combinations run solo first, then as chains -- same records either way
    objs2 = objects[3:] + objects[:2]
    w2 = solo.run(objs2).cpu().numpy()
    for rep in range(3):
        np.testing.assert_array_equal(runner.run(objs2).cpu().numpy(), w2)
    sph = golden("sphere.npz")["pts"]
    auto = BatchPoseRunner(encs, dev)                 # 8 instances on 3 lanes: no chains by default, same records
    np.testing.assert_array_equal(auto.run(objects).cpu().numpy(), want)
    np.testing.assert_array_equal(auto.run(objects).cpu().numpy(), want)
    assert not auto._chains
    for j in (0, 5):
        obj = objects[j]
        o = oracle.estimate_pose(obj["pc"], obj["normals"], obj["feat"], obj["point_idxs"], sd, ocfg_of(obj["cfg"]), obj["u_tr"], obj["u_rot"], sph)
        assert int(want[j, 12]) == o["argmax"] and int(want[j, 14]) == int(o["mask"].sum())
