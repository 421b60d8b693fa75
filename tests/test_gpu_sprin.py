"""SPRIN point encoder on the MI355X through the C ABI: HIP vs the oracle
(bit-exact: same operation order) and vs the reference's own outputs (tests/golden/sprin_*.npz, 2e-5)."""
import numpy as np
import pytest
import torch

from cppf_amd import _lib
from cppf_amd.models.model import PointEncoder

pytestmark = pytest.mark.gpu


def _cloud(n, seed, dup=0):
    rng = np.random.default_rng(seed)
    th, h = rng.uniform(0, 2 * np.pi, n), rng.uniform(-0.15, 0.15, n)
    pc = (np.stack([0.05 * np.cos(th), h, 0.05 * np.sin(th)], -1) + rng.normal(0, 1e-3, (n, 3))).astype(np.float32)
    nrm = np.stack([np.cos(th), np.zeros(n), np.sin(th)], -1) + rng.normal(0, 0.05, (n, 3))
    nrm = (nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)).astype(np.float32)
    if dup:                                   # exact duplicates: equal keys at the selection boundary
        pc[-dup:] = pc[:dup]
    return pc, nrm


def _encoder(dev, num_layers=1, seed=3, k=60):
    torch.manual_seed(seed)
    enc = PointEncoder(k=k, spfcs=[32, 64, 32, 32], num_layers=num_layers, out_dim=32).eval()
    with torch.no_grad():
        for name, p in enc.named_parameters():
            if "layer_norm" in name or (".kernel." in name and p.ndim == 1 and name.endswith("weight")):
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
    return enc.to(dev)


def _oracle_out(oracle, enc, pc, nrm, nbrs):
    sd = {k: v.detach().cpu().numpy() for k, v in enc.state_dict().items()}
    packed, desc = oracle.pack_point_encoder(sd, enc.num_layers)
    return oracle.point_encoder(pc, nrm, nbrs, packed, desc, order=1)     # the MFMA kernel's summation order


@pytest.mark.parametrize("n,k,dup", [(1000, 60, 0), (257, 64, 0), (64, 64, 0), (515, 7, 40), (2000, 33, 0), (4096, 60, 3), (8000, 60, 0), (9001, 16, 0), (10050, 9, 0)])
def test_knn_points_matches_oracle(dev, oracle, n, k, dup):
    pc, _ = _cloud(n, n + k, dup)
    enc = PointEncoder(k=k, spfcs=[32, 64, 32, 32], num_layers=1, out_dim=32)
    got = enc.neighbours(torch.from_numpy(pc).to(dev)).cpu().numpy()
    assert np.array_equal(got, oracle.knn(pc, k))


def test_knn_duplicate_heavy_cloud_takes_the_full_bisection(dev, oracle):
    """3 distinct locations x 400 copies: > 256 keys tie below the pruning bound, so the kernel's fallback
    (bisection over all keys) runs; ties resolve to the lowest indices like the oracle's."""
    rng = np.random.default_rng(7)
    pc = np.repeat(rng.normal(0, 0.1, (3, 3)).astype(np.float32), 400, 0)[rng.permutation(1200)]
    for k in (60, 5):
        enc = PointEncoder(k=k, spfcs=[32, 64, 32, 32], num_layers=1, out_dim=32)
        got = enc.neighbours(torch.from_numpy(pc).to(dev)).cpu().numpy()
        assert np.array_equal(got, oracle.knn(pc, k))


def test_knn_from_dist_matches_torch_topk(dev, oracle, golden):
    z = golden("sprin_l2.npz")                       # holds the reference's own torch.cdist matrix
    k = int(z["k"])
    enc = PointEncoder(k=k, spfcs=[32, 64, 32, 32], num_layers=1, out_dim=32)
    pc = torch.from_numpy(z["pc"]).to(dev)
    got = enc.neighbours(pc, torch.from_numpy(z["dist"]).to(dev)).cpu().numpy()
    assert np.array_equal(got, z["nbrs_topk"].astype(np.int32))
    # a matrix with negative and repeated entries: ordering and tie rule
    rng = np.random.default_rng(0)
    d = rng.integers(-3, 4, (96, 96)).astype(np.float32)
    got = enc.neighbours(pc, torch.from_numpy(d).to(dev)).cpu().numpy()
    assert np.array_equal(got, oracle.knn(None, k, dist=d))


def test_point_encoder_matches_reference_and_oracle(dev, oracle, golden):
    z = golden("sprin_l1.npz")
    enc = PointEncoder(k=int(z["k"]), spfcs=list(z["spfcs"]), num_layers=1, out_dim=32).eval()
    enc.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")})
    enc = enc.to(dev)
    pc, nrm = torch.from_numpy(z["pc"][None]).to(dev), torch.from_numpy(z["nrm"][None]).to(dev)
    nbrs = torch.from_numpy(z["nbrs_topk"].astype(np.int64)[None]).to(dev)
    with torch.no_grad():
        out_nbrs = enc.forward_nbrs(pc, nrm, nbrs)
        out_dist = enc(pc, nrm, torch.cdist(pc, pc))             # the call of nocs/inference.py:180-181
        out_none = enc(pc, nrm)                                   # extension: no N x N matrix
    assert out_nbrs.shape == (1, 256, 40)
    want = _oracle_out(oracle, enc, z["pc"], z["nrm"], z["nbrs_topk"].astype(np.int32))
    assert np.array_equal(out_nbrs[0].cpu().numpy(), want)        # same operation order: bit-exact
    for o in (out_nbrs, out_dist, out_none):
        np.testing.assert_allclose(o[0].cpu().numpy(), z["out"], atol=2e-5, rtol=0)   # the reference's own output


@pytest.mark.parametrize("n,k,layers", [(1024, 60, 1), (333, 17, 1), (300, 60, 2), (130, 64, 3)])
def test_point_encoder_bit_exact_vs_oracle(dev, oracle, n, k, layers):
    pc, nrm = _cloud(n, 5 * n + k)
    enc = _encoder(dev, layers, seed=n, k=k)
    nbrs = oracle.knn(pc, k)
    with torch.no_grad():
        out = enc.forward_nbrs(torch.from_numpy(pc[None]).to(dev), torch.from_numpy(nrm[None]).to(dev),
                               torch.from_numpy(nbrs.astype(np.int64)[None]).to(dev))
    assert np.array_equal(out[0].cpu().numpy(), _oracle_out(oracle, enc, pc, nrm, nbrs))


def test_unsupported_shapes_raise_and_autograd_uses_composite(dev, golden):
    z = golden("sprin_l2.npz")
    enc = PointEncoder(k=int(z["k"]), spfcs=list(z["spfcs"]), num_layers=2, out_dim=32).eval()
    enc.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")})
    enc = enc.to(dev)
    pc, nrm = torch.from_numpy(z["pc"][None]).to(dev), torch.from_numpy(z["nrm"][None]).to(dev)
    with torch.no_grad(), pytest.raises(_lib.CppfError, match="no device kernel"):
        enc(pc, nrm, torch.cdist(pc, pc))
    out = enc(pc, nrm, torch.cdist(pc, pc))                        # grad enabled -> torch composite (train.py:64)
    assert out.requires_grad
    np.testing.assert_allclose(out[0].detach().cpu().numpy(), z["out"], atol=5e-5, rtol=0)
    with torch.no_grad(), pytest.raises(_lib.CppfError, match="HIP device only"):
        enc.cpu()(pc.cpu(), nrm.cpu(), None)


def test_pipeline_with_point_encoder_equals_precomputed_features(dev, oracle):
    """PosePipeline(point_encoder=...) -- kNN + SPRIN + pair path + pose tail in one captured graph -- gives the
    record of the same pipeline fed with the oracle's features; replaying the graph is stable."""
    import cppf_amd.synthetic as syn
    from cppf_amd.inference import PosePipeline, grid_shape
    from cppf_amd.models.model import PPFEncoder
    from cppf_amd.utils.util import fibonacci_sphere
    ob = syn.make_object("mug", 1024, 11)
    cfg, pc, nrm = ob["cfg"], ob["pc"], ob["normals"]
    torch.manual_seed(4)
    ppf = PPFEncoder([84, 32, 32, 16], 2 * cfg.tr_num_bins + 2 * cfg.rot_num_bins + 5).eval().to(dev)
    penc = _encoder(dev, 1, seed=9, k=60)
    idx = syn.make_pairs(1024, 32, 11)
    u_tr, u_rot = syn.make_uniforms(idx.shape[0], 11)
    corners, dims = grid_shape(pc, cfg.res)
    corner = corners[0]
    sphere = np.array(fibonacci_sphere(480))
    feat = _oracle_out(oracle, penc, pc, nrm, oracle.knn(pc, 60))
    a = PosePipeline(ppf, cfg, 1024, idx.shape[0], dims, dev, sphere, point_encoder=penc)
    a.load(pc, nrm, None, idx, u_tr, u_rot, corner)
    ra = a.run()
    ra2 = a.run()
    b = PosePipeline(ppf, cfg, 1024, idx.shape[0], dims, dev, sphere)
    b.load(pc, nrm, feat, idx, u_tr, u_rot, corner)
    rb = b.run()
    assert np.array_equal(a.feat.cpu().numpy(), feat)
    for key in ("argmax", "n_surv"):
        assert ra[key] == rb[key] == ra2[key]
    for key in ("T", "up", "scale"):
        assert np.array_equal(ra[key], rb[key]) and np.array_equal(ra[key], ra2[key])


def test_batch_runner_with_point_encoders(dev, oracle):
    """BatchPoseRunner(point_encoders=...): objects arrive without `feat`; every record equals the pipeline fed with
    the oracle's features."""
    import cppf_amd.synthetic as syn
    from cppf_amd.batch import BatchPoseRunner
    from cppf_amd.models.model import PPFEncoder
    cats = ["bottle", "mug", "bowl"]
    encs, pencs, objs, objs_feat = {}, {}, [], []
    for i, c in enumerate(cats):
        cfg = syn.make_object(c, 64, 0)["cfg"]
        torch.manual_seed(20 + i)
        encs[c] = PPFEncoder([84, 32, 32, 16], 2 * cfg.tr_num_bins + 2 * cfg.rot_num_bins + 5).eval().to(dev)
        pencs[c] = _encoder(dev, 1, seed=30 + i, k=60)
    for j in range(5):
        c = cats[j % 3]
        ob = syn.make_object(c, 768, 40 + j)
        idx = syn.make_pairs(768, 24, 40 + j)
        u_tr, u_rot = syn.make_uniforms(idx.shape[0], 40 + j)
        base = dict(pc=ob["pc"], normals=ob["normals"], point_idxs=idx, u_tr=u_tr, u_rot=u_rot, cfg=ob["cfg"])
        objs.append(base)
        objs_feat.append(dict(base, feat=_oracle_out(oracle, pencs[c], ob["pc"], ob["normals"], oracle.knn(ob["pc"], 60))))
    a = BatchPoseRunner(encs, dev, point_encoders=pencs).run(objs).cpu().numpy()
    b = BatchPoseRunner(encs, dev).run(objs_feat).cpu().numpy()
    assert a.shape == (5, 20) and np.array_equal(a, b)


@pytest.mark.parametrize("sizes,caps,ready", [
    ([717, 1203, 1890, 960], [1024, 2048, 2048, 1024], [0, 0, 0, 0]),      # the reference-default batch's clouds in their buckets
    ([64, 2011, 300, 1544, 1333, 1777, 90, 1024], [64, 3072, 512, 2048, 2048, 2048, 128, 1024], [0, 1, 0, 0, 1, 0, 0, 1]),
    ([515], [1024], [0]),
])
def test_point_encoder_batch_equals_the_per_cloud_entry_points_and_the_oracle(dev, oracle, sizes, caps, ready):
    """cppf_point_encoder_forward_batch (three launches for up to 8 clouds; members of different categories carry different weights;
    capacity-sized buffers with the point count in device memory; members whose neighbour sets are already there): every member's
    neighbour sets and features equal cppf_knn_dyn + cppf_point_encoder_forward_dyn's and the oracle's, bit for bit; rows beyond a
    member's point count are left untouched."""
    from cppf_amd.models.model import point_encoder_forward_batch
    k = 60
    encs = [_encoder(dev, 1, seed=40 + (i % 3), k=k) for i in range(len(sizes))]
    members, singles, clouds = [], [], []
    for i, (n, cap, rdy) in enumerate(zip(sizes, caps, ready)):
        pc, nrm = _cloud(n, 900 + i)
        clouds.append((pc, nrm))
        pcd, nrmd = torch.full((cap, 3), 7.0, device=dev), torch.full((cap, 3), 7.0, device=dev)
        pcd[:n], nrmd[:n] = torch.from_numpy(pc).to(dev), torch.from_numpy(nrm).to(dev)
        n_dev = torch.tensor([n, 0, 0, 0], dtype=torch.int32, device=dev)
        nbrs = torch.full((cap, k), -5, dtype=torch.int32, device=dev)
        if rdy:
            nbrs[:n] = torch.from_numpy(oracle.knn(pc, k).astype(np.int32)).to(dev)
        out = torch.full((cap, 40), -3.0, device=dev)
        members.append(dict(encoder=encs[i], pc=pcd, nrm=nrmd, n_dev=n_dev, out=out, nbrs=nbrs, nbrs_ready=bool(rdy)))
        singles.append(encs[i].forward_dyn(pcd, nrmd, n_dev, out=torch.full((cap, 40), -3.0, device=dev),
                                           nbrs=torch.full((cap, k), -5, dtype=torch.int32, device=dev)).cpu().numpy())
    outs = point_encoder_forward_batch(members)
    assert outs is not None and len(outs) == len(sizes)
    for i, (n, cap) in enumerate(zip(sizes, caps)):
        pc, nrm = clouds[i]
        got, nb = outs[i].cpu().numpy(), members[i]["nbrs"].cpu().numpy()
        want_nb = oracle.knn(pc, k)
        assert np.array_equal(nb[:n], want_nb) and np.all(nb[n:] == -5)
        assert np.array_equal(got, singles[i])
        assert np.array_equal(got[:n], _oracle_out(oracle, encs[i], pc, nrm, want_nb)) and np.all(got[n:] == -3.0)
    # members that do not qualify: the caller loops over forward_dyn
    assert point_encoder_forward_batch([dict(members[0], encoder=_encoder(dev, 2, seed=1, k=k))]) is None
    assert point_encoder_forward_batch([members[0], dict(members[0], encoder=_encoder(dev, 1, seed=1, k=17))]) is None
    # the C ABI's argument checks
    L = _lib.lib()
    arr = (_lib.PointEncItem * 1)()
    import ctypes as C
    hid = (C.c_int * 4)(32, 64, 32, 32)
    assert L.cppf_point_encoder_forward_batch(0, C.cast(arr, C.c_void_p), k, hid, 4, 32, 2, 32, 8, 1, None) == -1
    assert L.cppf_point_encoder_forward_batch(9, C.cast(arr, C.c_void_p), k, hid, 4, 32, 2, 32, 8, 1, None) == -1
    assert L.cppf_point_encoder_forward_batch(1, C.cast(arr, C.c_void_p), k, hid, 4, 32, 2, 32, 8, 2, None) == -3
    assert L.cppf_point_encoder_forward_batch(1, C.cast(arr, C.c_void_p), k, hid, 4, 32, 2, 32, 8, 1, None) == -1      # (null pointers)
