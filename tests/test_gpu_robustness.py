"""Host-logic hazards around the HIP path: configurations outside the fused kernels, weight updates while several streams
share one encoder, the rotation-table cache inside the vote's workspace, and the two captured forms of PosePipeline."""
import dataclasses

import numpy as np
import pytest
import torch

import cppf_amd.synthetic as syn
from cppf_amd import _lib
from cppf_amd._torch_util import workspace
from cppf_amd.models import voting
from cppf_amd.models.model import PPFEncoder

pytestmark = pytest.mark.gpu


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def sd_of(enc):
    return {k: v.detach().cpu().numpy().copy() for k, v in enc.state_dict().items()}


@pytest.mark.parametrize("tr_bins,rot_bins", [(16, 24), (32, 20), (40, 36)])
def test_pose_with_non_default_bin_counts(oracle, golden, dev, tr_bins, rot_bins):
    """config/config.yaml's tr_num_bins / rot_num_bins are knobs: the whole-pose entry points must serve other values through
    the logits + decode kernels (the fused kernels are specialised for 32 / 36) -- estimate_pose, PosePipeline (captured) and
    the second-pass entry point on its own"""
    from cppf_amd.inference import PosePipeline, estimate_pose, grid_shape
    ob = syn.make_object("camera", 1024, 11)
    cfg = dataclasses.replace(ob["cfg"], tr_num_bins=tr_bins, rot_num_bins=rot_bins)
    assert cfg.out_dim == 2 * tr_bins + 2 * rot_bins + 5
    idx = syn.make_pairs(1024, 24, 11)
    P = idx.shape[0]
    u_tr, u_rot = syn.make_uniforms(P, 11)
    torch.manual_seed(4)
    enc = PPFEncoder(cfg.ppffcs, cfg.out_dim)
    with torch.no_grad():
        enc.final.weight.mul_(4)
        enc.final.bias.mul_(4)
    sd = sd_of(enc)
    enc = enc.to(dev).eval()
    assert not enc.fused_decode_supported(tr_bins, rot_bins)
    sph = golden("sphere.npz")["pts"]
    ocfg = dict(res=cfg.res, tr_num_bins=tr_bins, rot_num_bins=rot_bins, vote_range=cfg.vote_range, scale_mean=cfg.scale_mean,
                regress_right=cfg.regress_right, ppffcs=cfg.ppffcs, out_dim=cfg.out_dim)
    o = oracle.estimate_pose(ob["pc"], ob["normals"], ob["feat"], idx, sd, ocfg, u_tr, u_rot, sph, order=1 if cfg.out_dim <= 144 else 0)
    with torch.no_grad():
        r = estimate_pose(enc, t(ob["pc"], dev), t(ob["normals"], dev), t(ob["feat"], dev), t(idx, dev), t(u_tr, dev),
                          t(u_rot, dev), cfg, sph, pc_host=ob["pc"])
    np.testing.assert_array_equal(r["outputs"].cpu().numpy(), o["outputs"])
    assert r["argmax"] == o["argmax"] and r["n_surv"] == int(o["mask"].sum()) > 0
    np.testing.assert_array_equal(r["heads"].cpu().numpy()[o["mask"]], o["heads"][o["mask"]])
    np.testing.assert_allclose(r["T"], o["T"], atol=1e-12)
    np.testing.assert_allclose(r["up"], o["up"], atol=1e-12)
    np.testing.assert_allclose(r["scale"], o["scale"], rtol=1e-6)
    # captured pipeline: full-first form only, and adapt() must not leave it
    corners, dims = grid_shape(ob["pc"], cfg.res)
    pp = PosePipeline(enc, cfg, 1024, P, dims, dev, sph)
    pp.load(ob["pc"], ob["normals"], ob["feat"], idx, u_tr, u_rot, corners[0].copy())
    for _ in range(3):
        q = pp.run()
        assert pp.full_first
        assert q["argmax"] == o["argmax"] and q["n_surv"] == r["n_surv"]
        np.testing.assert_array_equal(q["up"], r["up"])
        np.testing.assert_array_equal(q["scale"], r["scale"])
    pp.adapt(0)
    assert pp.full_first
    # forward_decode_sel on its own (host-synchronising fallback)
    surv = torch.nonzero(r["ws"].mask.bool())[:, 0].to(torch.int32)
    n = torch.tensor([surv.numel()], dtype=torch.int32, device=dev)
    heads = torch.zeros((P, 8), dtype=torch.float32, device=dev)
    with torch.no_grad():
        enc.forward_decode_sel(t(ob["pc"], dev), t(ob["normals"], dev), t(ob["feat"], dev), t(idx, dev), t(u_rot, dev), surv, n,
                               heads, tr_num_bins=tr_bins, rot_num_bins=rot_bins)
    np.testing.assert_array_equal(heads.cpu().numpy()[o["mask"]], o["heads"][o["mask"]])
    assert float(heads.cpu().numpy()[~o["mask"]].__abs__().max()) == 0.0


def test_weight_update_between_batches_with_lanes(golden, dev):
    """one encoder shared by three lanes (streams): after an in-place parameter update between two batches every lane must
    replay with the NEW weights -- records equal those of a fresh runner built after the update, on every repetition"""
    from cppf_amd.batch import BatchPoseRunner
    cfg = syn.make_object("bottle", 64, 0)["cfg"]
    torch.manual_seed(0)
    enc = PPFEncoder(cfg.ppffcs, cfg.out_dim).eval().to(dev)
    objs = []
    for j in range(9):
        ob = syn.make_object("bottle", 2048, 70 + j)
        idx = syn.make_pairs(2048, 64, 70 + j)
        u1, u2 = syn.make_uniforms(idx.shape[0], 70 + j)
        objs.append(dict(pc=ob["pc"], normals=ob["normals"], feat=ob["feat"], point_idxs=idx, u_tr=u1, u_rot=u2, cfg=ob["cfg"]))
    runner = BatchPoseRunner({"bottle": enc}, dev, n_lanes=3)
    before = runner.run(objs).cpu().numpy()
    for step in range(3):
        with torch.no_grad():
            for p in enc.parameters():
                p.add_(torch.randn_like(p) * 0.05)              # bumps _version like an optimizer step
        got = runner.run(objs).cpu().numpy()
        fresh_enc = PPFEncoder(cfg.ppffcs, cfg.out_dim).eval().to(dev)
        fresh_enc.load_state_dict(enc.state_dict())
        want = BatchPoseRunner({"bottle": fresh_enc}, dev, n_lanes=1).run(objs).cpu().numpy()
        np.testing.assert_array_equal(got, want)
        assert not np.array_equal(got, before)
        before = got


def test_center_pipelines_sharing_an_encoder_on_two_streams(dev):
    """two CenterPipelines, one encoder, two streams: a weight update noticed by the first must be waited for by the second"""
    from cppf_amd.inference import CenterPipeline, grid_shape
    ob = syn.make_object("mug", 2048, 3)
    cfg = ob["cfg"]
    idx = syn.make_pairs(2048, 64, 3)
    P = idx.shape[0]
    u_tr, u_rot = syn.make_uniforms(P, 3)
    torch.manual_seed(1)
    enc = PPFEncoder(cfg.ppffcs, cfg.out_dim).eval().to(dev)
    corners, dims = grid_shape(ob["pc"], cfg.res)
    pipes, streams = [], [torch.cuda.Stream(device=dev) for _ in range(2)]
    for _ in range(2):
        p = CenterPipeline(enc, cfg, 2048, P, dims, dev)
        p.load(ob["pc"], ob["normals"], ob["feat"], idx, u_tr, u_rot, corners[0].copy())
        pipes.append(p)
    for rep in range(6):
        with torch.no_grad():
            for q in enc.parameters():
                q.add_(torch.randn_like(q) * 0.05)
        torch.cuda.synchronize()
        for p, s in zip(pipes, streams):
            with torch.cuda.stream(s):
                p.run()
        torch.cuda.synchronize()
        assert torch.equal(pipes[0].outputs, pipes[1].outputs)
        assert torch.equal(pipes[0].heads, pipes[1].heads)
        assert int(pipes[0].out_idx) == int(pipes[1].out_idx)
        with torch.no_grad():
            ref, _ = enc.forward_decode(pipes[0].pc, pipes[0].nrm, pipes[0].feat, pipes[0].idx, pipes[0].u_tr, cfg.vote_range)
        assert torch.equal(ref, pipes[0].outputs)


def test_vote_workspace_cache_survives_clobbering(dev):
    """the vote caches its rotation table in the caller's workspace (stamp at byte 248, table from byte 256): a caller that
    overwrites table bytes but keeps the stamp -- an arena shared between entry points, a recycled allocator block -- must still
    get the right grid (the kernels re-validate the cache and rebuild it)"""
    ob = syn.make_object("bottle", 2048, 9)
    cfg = ob["cfg"]
    idx = syn.make_pairs(2048, 32, 9)
    outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
    from cppf_amd.inference import grid_shape
    corners, dims = grid_shape(ob["pc"], cfg.res)
    pc, out_d, idx_d, corner = t(ob["pc"], dev), t(outputs, dev), t(idx.astype(np.int32), dev), t(corners[0], dev)
    grid = torch.zeros(dims, dtype=torch.float32, device=dev)
    oi, ov = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.float32, device=dev)

    def vote():
        voting.vote_argmax(pc, out_d, None, idx_d, grid, corner, cfg.res, 72, True, oi, ov, accumulate=False)
        torch.cuda.synchronize()
        return grid.clone(), int(oi.item())

    g0, f0 = vote()
    g1, f1 = vote()                                   # second launch: table from the cache
    assert torch.equal(g0, g1) and f0 == f1
    ws = workspace(256, dev, "vote")
    assert ws.numel() >= 32768
    stamp = ws[248:256].clone()
    for lo, hi in ((256, 256 + 4096), (256 + 8192, 256 + 21024), (256, 256 + 21024)):
        vote()
        assert torch.equal(ws[248:256], stamp)       # a valid stamp is in place ...
        ws[lo:hi] = 0x5a                              # ... and the table under it is destroyed
        g2, f2 = vote()
        assert torch.equal(g0, g2) and f0 == f2
        g3, f3 = vote()
        assert torch.equal(g0, g3) and f0 == f3


def test_pose_pipeline_forms_keep_their_own_outputs(golden, dev):
    """split -> full-first -> split: run() must return the outputs / heads tensors of the graph that just ran, not tensors
    the other form's capture allocated (stale data)"""
    from cppf_amd.inference import PosePipeline, grid_shape
    sph = golden("sphere.npz")["pts"]
    torch.manual_seed(0)
    cfg = syn.make_object("camera", 64, 0)["cfg"]
    enc = PPFEncoder(cfg.ppffcs, cfg.out_dim)
    with torch.no_grad():
        enc.final.weight.mul_(4)
        enc.final.bias.mul_(4)
    enc = enc.eval().to(dev)
    pp = None
    seen = []
    for step, (seed, form) in enumerate([(1, False), (2, True), (3, False), (4, True), (5, False)]):
        ob = syn.make_object("camera", 1024, seed)
        idx = syn.make_pairs(1024, 24, seed)
        P = idx.shape[0]
        u_tr, u_rot = syn.make_uniforms(P, seed)
        corners, dims = grid_shape(ob["pc"], cfg.res)
        if pp is None:
            pp = PosePipeline(enc, cfg, 1024, P, (40, 40, 40), dev, sph, dynamic=True)
        pp.load(ob["pc"], ob["normals"], ob["feat"], idx, u_tr, u_rot, corners[0].copy(), dims=dims)
        pp.adapt(P if form else 0)
        assert pp.full_first == form
        r = pp.run()
        with torch.no_grad():
            want_out, want_heads = enc.forward_decode(pp.pc[:1024], pp.nrm[:1024], pp.feat[:1024], pp.idx, pp.u_tr,
                                                      cfg.vote_range, pp.u_rot)
        assert torch.equal(r["outputs"], want_out), (step, form)
        mask = r["ws"].mask.bool()
        assert int(mask.sum()) == r["n_surv"] > 0
        assert torch.equal(r["heads"][mask], want_heads[mask]), (step, form)
        seen.append(r["argmax"])
    assert len(set(seen)) > 1                         # different instances really went through


def test_rot_vote_on_a_shuffled_subsample_like_the_reference(oracle, golden, dev):
    """nocs/inference.py:277-280 draws the orientation vote's pairs by shuffling the survivors: `rot_order` reproduces a given
    shuffle exactly (estimate_pose, PosePipeline's static buffer, entries beyond the survivor count skipped)"""
    from cppf_amd.inference import PosePipeline, estimate_pose, grid_shape
    ob = syn.make_object("camera", 1024, 21)
    cfg = ob["cfg"]
    idx = syn.make_pairs(1024, 32, 21)
    P = idx.shape[0]
    u_tr, u_rot = syn.make_uniforms(P, 21)
    torch.manual_seed(0)
    enc = PPFEncoder(cfg.ppffcs, cfg.out_dim)
    with torch.no_grad():
        enc.final.weight.mul_(4)
        enc.final.bias.mul_(4)
    sd = sd_of(enc)
    enc = enc.to(dev).eval()
    sph = golden("sphere.npz")["pts"]
    ocfg = dict(res=cfg.res, tr_num_bins=32, rot_num_bins=36, vote_range=cfg.vote_range, scale_mean=cfg.scale_mean,
                regress_right=cfg.regress_right, ppffcs=cfg.ppffcs, out_dim=cfg.out_dim)
    o0 = oracle.estimate_pose(ob["pc"], ob["normals"], ob["feat"], idx, sd, ocfg, u_tr, u_rot, sph, max_rot_pairs=150)
    n_surv = int(o0["mask"].sum())
    assert n_surv > 300
    rng = np.random.RandomState(5)
    order = np.arange(n_surv)
    rng.shuffle(order)                                        # the reference's own three lines
    order = order[:150].astype(np.int32)
    args = (enc, t(ob["pc"], dev), t(ob["normals"], dev), t(ob["feat"], dev), t(idx, dev), t(u_tr, dev), t(u_rot, dev), cfg, sph)
    for ro in (order, np.concatenate([order[:100], [n_surv + 5, -1, 2 ** 30], order[100:]]).astype(np.int32)):
        o = oracle.estimate_pose(ob["pc"], ob["normals"], ob["feat"], idx, sd, ocfg, u_tr, u_rot, sph, max_rot_pairs=153,
                                 rot_order=ro)
        with torch.no_grad():
            r = estimate_pose(*args, pc_host=ob["pc"], max_rot_pairs=153, rot_order=ro)
        cands = oracle.rot_voting(ob["pc"], o["heads"][np.nonzero(o["mask"])[0][ro[(ro >= 0) & (ro < n_surv)]], 0],
                                  idx.astype(np.int32)[np.nonzero(o["mask"])[0][ro[(ro >= 0) & (ro < n_surv)]]], 72)
        counts = oracle.sphere_count(cands, sph, 1.5)
        np.testing.assert_array_equal(r["ws"].counts[0].cpu().numpy(), counts)          # the subsample's counts, bin by bin
        np.testing.assert_allclose(r["up"], o["up"], atol=1e-12)
        assert r["n_surv"] == n_surv
    # a different subsample gives different counts (the order is really honoured)
    with torch.no_grad():
        r_first = estimate_pose(*args, pc_host=ob["pc"], max_rot_pairs=150)
    assert not np.array_equal(r_first["ws"].counts[0].cpu().numpy(), counts)
    np.testing.assert_allclose(r_first["up"], o0["up"], atol=1e-12)
    # captured pipeline with a static order buffer
    corners, dims = grid_shape(ob["pc"], cfg.res)
    pp = PosePipeline(enc, cfg, 1024, P, dims, dev, sph, max_rot_pairs=150, rot_order_len=150)
    pp.load(ob["pc"], ob["normals"], ob["feat"], idx, u_tr, u_rot, corners[0].copy())
    q0 = pp.run()
    np.testing.assert_allclose(q0["up"], o0["up"], atol=1e-12)                            # default order = first survivors
    pp.rot_order.copy_(t(order, dev))
    q = pp.run()
    o = oracle.estimate_pose(ob["pc"], ob["normals"], ob["feat"], idx, sd, ocfg, u_tr, u_rot, sph, max_rot_pairs=150,
                             rot_order=order)
    np.testing.assert_allclose(q["up"], o["up"], atol=1e-12)
    np.testing.assert_array_equal(pp.ws.counts[0].cpu().numpy(),
                                  oracle.sphere_count(oracle.rot_voting(ob["pc"], o["heads"][np.nonzero(o["mask"])[0][order], 0],
                                                                         idx.astype(np.int32)[np.nonzero(o["mask"])[0][order]], 72),
                                                      sph, 1.5))
