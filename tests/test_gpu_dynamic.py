"""Shape-polymorphic (`_dyn`) entry points and pipelines: one captured chain for ragged instance shapes.

The reference re-derives N and the grid dims per instance on the host (nocs/inference.py:140-142,194-195).  The `_dyn`
kernels read them from a device record instead; everything here is checked BIT FOR BIT against the by-value entry points on
the real shape (which test_gpu_parity.py pins on the oracle), plus against the oracle chain directly for the batch runner."""
import numpy as np
import pytest
import torch

import cppf_amd.synthetic as syn
from cppf_amd import _lib
from cppf_amd.config import CATEGORIES
from cppf_amd.inference import CenterPipeline, PosePipeline, estimate_pose, grid_class, grid_shape
from cppf_amd.models import voting
from cppf_amd.models.model import PPFEncoder, PointEncoder

pytestmark = pytest.mark.gpu
I32, F32 = torch.int32, torch.float32


def t(x, dev, dtype=None):
    a = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        a = a.to(dtype)
    return a.to(dev)


def _encoder(dev, seed=0, gain=4.0):
    torch.manual_seed(seed)
    enc = PPFEncoder([84, 32, 32, 16], 141)
    with torch.no_grad():
        enc.final.weight.mul_(gain)
        enc.final.bias.mul_(gain)
    return enc.to(dev).eval()


@pytest.mark.parametrize("cat,n,res_scale,n_cap", [("bottle", 700, 1.0, 1024), ("mug", 1024, 1.0, 1024), ("camera", 333, 1.0, 2048),
                                                   ("bottle", 900, 0.5, 1024), ("laptop", 512, 0.4, 1024)])
def test_dyn_vote_equals_by_value_vote(dev, cat, n, res_scale, n_cap):
    """cppf_vote_argmax_dyn with the shape in a device record == cppf_vote_argmax on the real shape: grid bit for bit,
    arg-max index and value; capacity-sized point / prob buffers with garbage behind the cloud."""
    ob = syn.make_object(cat, n, 7)
    cfg = ob["cfg"]
    res = cfg.res * res_scale
    idx = syn.make_pairs(n, 48, 7)
    out = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
    corners, dims = grid_shape(ob["pc"], res)
    T, many, cap = grid_class(dims)
    assert T > 0
    pc, outputs, idx_d, corner = t(ob["pc"], dev), t(out, dev), t(idx, dev), t(corners[0].copy(), dev)
    probs = torch.ones(n, dtype=F32, device=dev)
    grid = torch.empty(dims, dtype=F32, device=dev)
    i0, v0 = voting.vote_argmax(pc, outputs, probs, idx_d, grid, corner, res, 72, True, accumulate=False)
    # capacity-sized inputs: rows behind the cloud hold NaN / huge values and must never be read
    pc_cap = torch.full((n_cap, 3), float("nan"), dtype=F32, device=dev)
    pc_cap[:n] = pc
    probs_cap = torch.full((n_cap,), 1e30, dtype=F32, device=dev)
    probs_cap[:n] = 1.0
    flat = torch.full((cap,), -7.0, dtype=F32, device=dev)
    shape = torch.tensor([n, *dims], dtype=I32, device=dev)
    oi, ov = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=F32, device=dev)
    voting.vote_argmax_dyn(pc_cap, outputs, probs_cap, idx_d, flat, shape, corner, res, 72, True, oi, ov, many_tiles=many)
    G = dims[0] * dims[1] * dims[2]
    assert int(oi) == int(i0) and float(ov) == float(v0)
    assert torch.equal(flat[:G].view(dims), grid)
    assert bool((flat[G:] == -7.0).all())                   # nothing written past the real grid
    if not many:                                            # the 2 048-workgroup geometry serves small grids too
        flat2 = torch.empty(64 * _lib.lib().cppf_vote_tile_cells(), dtype=F32, device=dev)
        voting.vote_argmax_dyn(pc_cap, outputs, probs_cap, idx_d, flat2, shape, corner, res, 72, True, oi, ov, many_tiles=True)
        assert int(oi) == int(i0) and torch.equal(flat2[:G].view(dims), grid)


def test_dyn_vote_reports_a_record_beyond_its_capacities(dev):
    ob = syn.make_object("bottle", 512, 1)
    cfg = ob["cfg"]
    idx = syn.make_pairs(512, 16, 1)
    out = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg)
    corners, dims = grid_shape(ob["pc"], cfg.res)
    pc, outputs, idx_d, corner = t(ob["pc"], dev), t(out, dev), t(idx, dev), t(corners[0].copy(), dev)
    probs = torch.ones(512, dtype=F32, device=dev)
    oi, ov = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=F32, device=dev)
    flat = torch.zeros(3 * _lib.lib().cppf_vote_tile_cells(), dtype=F32, device=dev)
    for bad in ([513, *dims], [512, 200, 200, 200], [512, 0, 5, 5], [0, *dims]):
        shape = torch.tensor(bad, dtype=I32, device=dev)
        flat.fill_(3.0)
        voting.vote_argmax_dyn(pc, outputs, probs, idx_d, flat, shape, corner, cfg.res, 72, True, oi, ov)
        assert int(oi) == -1 and np.isnan(float(ov)) and bool((flat == 3.0).all()), bad
    # a fine grid (>= 4 tiles) on the few-tiles geometry is refused as well
    fine = grid_shape(ob["pc"], cfg.res / 2)[1]
    assert grid_class(fine)[1]
    voting.vote_argmax_dyn(pc, outputs, probs, idx_d, flat, torch.tensor([512, *fine], dtype=I32, device=dev), corner,
                           cfg.res / 2, 72, True, oi, ov, many_tiles=False)
    assert int(oi) == -1


@pytest.mark.parametrize("n,n_cap", [(60, 64), (700, 1024), (1024, 1024), (1500, 4096)])
def test_dyn_knn_and_point_encoder_equal_by_value(dev, n, n_cap):
    ob = syn.make_object("mug", n, 3)
    torch.manual_seed(5)
    penc = PointEncoder(k=60, spfcs=[32, 64, 32, 32], num_layers=1, out_dim=32).eval().to(dev)
    pc, nrm = t(ob["pc"], dev), t(ob["normals"], dev)
    with torch.no_grad():
        ref = penc(pc[None], nrm[None])[0]
        ref_nbrs = penc.neighbours(pc)
    pc_cap = torch.full((n_cap, 3), float("nan"), dtype=F32, device=dev)
    nrm_cap = torch.full((n_cap, 3), float("nan"), dtype=F32, device=dev)
    pc_cap[:n], nrm_cap[:n] = pc, nrm
    n_dev = torch.tensor([n, 0, 0, 0], dtype=I32, device=dev)
    nbrs = torch.full((n_cap, 60), -1, dtype=I32, device=dev)
    out = torch.full((n_cap, 40), -5.0, dtype=F32, device=dev)
    with torch.no_grad():
        penc.forward_dyn(pc_cap, nrm_cap, n_dev, out=out, nbrs=nbrs)
    assert torch.equal(nbrs[:n], ref_nbrs) and torch.equal(out[:n], ref)
    assert bool((nbrs[n:] == -1).all()) and bool((out[n:] == -5.0).all())


def test_dynamic_pipeline_serves_ragged_shapes_with_one_graph(golden, dev):
    """PosePipeline(dynamic=True): instances of different N and grid dims replay ONE captured graph and give the records of
    the exact-shape eager chain (estimate_pose), features from the device point encoder included."""
    sph = golden("sphere.npz")["pts"]
    enc = _encoder(dev)
    torch.manual_seed(9)
    penc = PointEncoder(k=60, spfcs=[32, 64, 32, 32], num_layers=1, out_dim=32).eval().to(dev)
    cfg = CATEGORIES["bottle"]
    P = 1024 * 24
    pipe = PosePipeline(enc, cfg, 1024, P, False, dev, sph, point_encoder=penc, dynamic=True)
    graph = None
    for seed, n, stretch in ((1, 1024, 1.0), (2, 640, 0.8), (3, 901, 1.15), (4, 64, 0.5), (5, 1000, 1.0)):
        ob = syn.make_object("bottle", n, seed)
        pc = ((ob["pc"] - ob["center"]) * np.float32(stretch) + ob["center"]).astype(np.float32)
        idx = np.random.default_rng(seed).integers(0, n, (P, 2)).astype(np.int64)
        u_tr, u_rot = syn.make_uniforms(P, seed)
        corners, dims = grid_shape(pc, cfg.res)
        with torch.no_grad():
            feat = penc(t(pc, dev)[None], t(ob["normals"], dev)[None])[0]
            ref = estimate_pose(enc, t(pc, dev), t(ob["normals"], dev), feat, t(idx, dev), t(u_tr, dev), t(u_rot, dev), cfg,
                                sph, pc_host=pc)
        pipe.load(pc, ob["normals"], None, idx, u_tr, u_rot, corners[0].copy(), dims=dims)
        r = pipe.run()
        graph = graph or pipe._graph
        assert pipe._graph is graph                       # no re-capture
        assert r["argmax"] == ref["argmax"] and r["n_surv"] == ref["n_surv"] and r["peak"] == ref["peak"]
        for k in ("T", "up", "scale", "R"):
            np.testing.assert_array_equal(r[k], ref[k])
        sv = ref["ws"].mask.bool()
        assert torch.equal(r["outputs"], ref["outputs"]) and torch.equal(r["heads"][sv], ref["heads"][sv])
        assert torch.equal(pipe.grid_view, ref["ws"].grid)
    with pytest.raises(_lib.CppfError):
        pipe.set_shape(1025, (10, 10, 10))
    with pytest.raises(_lib.CppfError):
        pipe.set_shape(59, (10, 10, 10))                   # fewer points than the encoder's k


def test_captured_pipeline_follows_weight_updates(golden, dev):
    """a captured graph bakes the address of the packed weight image in.  In-place parameter updates (optimizer
    step, load_state_dict) must reach the replay; `.data` edits do after invalidate()."""
    enc = _encoder(dev, 1)
    cfg = CATEGORIES["mug"]
    ob = syn.make_object("mug", 512, 11)
    idx = syn.make_pairs(512, 32, 11)
    P = idx.shape[0]
    u_tr, u_rot = syn.make_uniforms(P, 11)
    corners, dims = grid_shape(ob["pc"], cfg.res)
    pipe = CenterPipeline(enc, cfg, 512, P, dims, dev)
    pipe.load(ob["pc"], ob["normals"], ob["feat"], idx, u_tr, u_rot, corners[0].copy())

    def fresh():
        e2 = PPFEncoder([84, 32, 32, 16], 141).to(dev).eval()
        e2.load_state_dict(enc.state_dict())
        with torch.no_grad():
            o, _ = e2.forward_decode(pipe.pc, pipe.nrm, pipe.feat, pipe.idx, pipe.u_tr, cfg.vote_range, pipe.u_rot)
        return o.clone()

    pipe.run()
    pipe.run()
    g = pipe._graph
    addr = enc._packed.data_ptr()
    assert torch.equal(pipe.outputs, fresh())
    with torch.no_grad():                                   # what an optimizer step does
        for p in enc.parameters():
            p.add_(0.01 * torch.randn_like(p))
    pipe.run()
    assert pipe._graph is g and enc._packed.data_ptr() == addr      # same graph, image rebuilt in place
    assert torch.equal(pipe.outputs, fresh())
    other = _encoder(dev, 2)
    enc.load_state_dict(other.state_dict())
    pipe.run()
    assert pipe._graph is g and torch.equal(pipe.outputs, fresh())
    before = pipe.outputs.clone()
    enc.final.bias.data[:16].add_(3.0)                      # bypasses the version counter ...
    pipe.run()
    assert torch.equal(pipe.outputs, before)
    enc.invalidate()                                        # ... until the caller says so
    pipe.run()
    assert pipe._graph is g and torch.equal(pipe.outputs, fresh()) and not torch.equal(pipe.outputs, before)


def test_batch_runner_ragged_batch_bounded_cache(oracle, golden, dev):
    """BASELINE config 4 with realistic raggedness: every instance its own N and grid, mixed categories, more shape buckets
    than the cache holds.  Records equal the oracle chain; the pipeline cache never exceeds its bound; the scratch cache holds
    only live pipelines' buffers."""
    from cppf_amd import _torch_util
    from cppf_amd.batch import BatchPoseRunner
    from cppf_amd.config import NOCS_CATEGORIES
    sph = golden("sphere.npz")["pts"]
    torch.manual_seed(0)
    ref_enc = PPFEncoder([84, 32, 32, 16], 141)
    sd = {k: v.detach().numpy().copy() for k, v in ref_enc.state_dict().items()}
    for k in ("final.weight", "final.bias"):
        sd[k] = sd[k] * 4
    encoders = {}
    for c in NOCS_CATEGORIES:
        e = PPFEncoder([84, 32, 32, 16], 141)
        e.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        encoders[c] = e.to(dev).eval()
    rng = np.random.default_rng(0)
    objects = []
    for j in range(10):
        cat = NOCS_CATEGORIES[j % 6]
        n = int(rng.integers(200, 1400))
        ob = syn.make_object(cat, n, 300 + j)
        P = 16384
        idx = rng.integers(0, n, (P, 2)).astype(np.int64)
        u_tr, u_rot = syn.make_uniforms(P, 300 + j)
        objects.append(dict(pc=ob["pc"], normals=ob["normals"], feat=ob["feat"], point_idxs=idx, u_tr=u_tr, u_rot=u_rot,
                            cfg=ob["cfg"]))
    scoped_before = {k[1] for k in _torch_util._ws_cache if isinstance(k[1], tuple)}
    runner = BatchPoseRunner(encoders, dev, n_bucket=512, max_pipelines=4)
    recs = runner.run(objects).cpu().numpy()
    assert len(runner._pipes) <= 4
    recs2 = runner.run(objects).cpu().numpy()
    np.testing.assert_array_equal(recs, recs2)
    assert len(runner._pipes) <= 4
    live = {("scope", id(p)) for p in runner._pipes.values()}
    scoped = {k[1] for k in _torch_util._ws_cache if isinstance(k[1], tuple)}
    assert scoped - scoped_before <= live                   # evicted pipelines took their scratch with them
    for j, obj in enumerate(objects):
        cfg = obj["cfg"]
        ocfg = dict(res=cfg.res, tr_num_bins=32, rot_num_bins=36, vote_range=cfg.vote_range, scale_mean=cfg.scale_mean,
                    regress_right=cfg.regress_right, ppffcs=[84, 32, 32, 16], out_dim=141)
        o = oracle.estimate_pose(obj["pc"], obj["normals"], obj["feat"], obj["point_idxs"], sd, ocfg, obj["u_tr"],
                                 obj["u_rot"], sph)
        assert int(recs[j, 12]) == o["argmax"] and int(recs[j, 14]) == int(o["mask"].sum()), j
        np.testing.assert_allclose(recs[j, 0:3], o["T"], atol=1e-12)
        np.testing.assert_allclose(recs[j, 3:6], o["up"], atol=1e-12)
        np.testing.assert_allclose(recs[j, 9:12], o["scale"], rtol=1e-6)
    objects[3].pop("feat")
    with pytest.raises(ValueError):
        runner.run(objects)


def test_tile_capacity_classes(dev):
    """`many_tiles` is a tile capacity class (ABI 4: 0 = 3 tiles, 1 = 64, 4..64 = that many): a 16-tile grid through class 16 (a quarter
    of class 1's queues) and through class 1 gives the by-value launch's grid and arg-max bit for bit; a grid that needs more tiles
    than the class holds is refused (arg-max -1 / NaN, nothing voted); BatchPoseRunner's pipelines take the 16 class for such grids"""
    ob = syn.make_object("bottle", 2048, 21)
    cfg = ob["cfg"]
    idx = syn.make_pairs(2048, 40, 21)
    out = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
    pc, outputs, idx_d = t(ob["pc"], dev), t(out, dev), t(idx, dev)
    cells = _lib.lib().cppf_vote_tile_cells()
    oi, ov = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=F32, device=dev)
    seen = set()
    for res in (2e-3, 1.6e-3):
        corners, dims = grid_shape(ob["pc"], res)
        T, many, cap = grid_class(dims)
        seen.add(many)
        corner = t(corners[0].copy(), dev)
        grid = torch.empty(dims, dtype=F32, device=dev)
        i0, v0 = voting.vote_argmax(pc, outputs, None, idx_d, grid, corner, res, 72, True, accumulate=False)
        shape = torch.tensor([2048, *dims], dtype=I32, device=dev)
        G = dims[0] * dims[1] * dims[2]
        for cls in {many, 1, 64, max(T, 4)}:
            flat = torch.full((_lib.tiles_cap(cls) * cells,), -7.0, dtype=F32, device=dev)
            voting.vote_argmax_dyn(pc, outputs, None, idx_d, flat, shape, corner, res, 72, True, oi, ov, many_tiles=cls)
            assert int(oi) == int(i0) and float(ov) == float(v0), (res, cls)
            assert torch.equal(flat[:G].view(dims), grid) and bool((flat[G:] == -7.0).all()), (res, cls)
        if T > 4:        # one tile short of what the grid needs: refused, nothing written
            flat = torch.full(((T - 1) * cells,), 3.0, dtype=F32, device=dev)
            voting.vote_argmax_dyn(pc, outputs, None, idx_d, flat, shape, corner, res, 72, True, oi, ov, many_tiles=T - 1)
            assert int(oi) == -1 and np.isnan(float(ov)) and bool((flat == 3.0).all()), (res, T)
    assert seen == {16, 1}, seen          # 16 tiles at res 2e-3, more at 1.6e-3
    # a shape-polymorphic pipeline of class 16 holds a quarter of the queues and grid of class 1
    from cppf_amd.batch import BatchPoseRunner
    small, big = BatchPoseRunner.footprint_bytes(2048, 2 ** 19, 16, None), BatchPoseRunner.footprint_bytes(2048, 2 ** 19, 1, None)
    assert big - small == 48 * 2 ** 19 * 12 + 48 * cells * 4
