"""Pre-processing on the device: voxel de-duplication and PCA normals, against the oracle
(bit-exact) and against numpy definitions.  Parity with MinkowskiEngine / open3d themselves is unpinned (absent here,
and their results are not fully specified: representative point per voxel, sign of a normal)."""
import numpy as np
import pytest
import torch

from cppf_amd.utils.util import estimate_normals, sparse_quantize

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,res", [(5000, 0.02), (70000, 0.004), (1, 0.01), (300, 5.0)])
def test_voxel_dedupe_matches_oracle_and_numpy_unique(dev, oracle, n, res):
    rng = np.random.default_rng(n)
    pc = rng.uniform(-0.2, 0.3, (n, 3)).astype(np.float32)
    pc[n // 2:] = pc[:n - n // 2]                                  # exact duplicates: one survivor each, the lower index
    coords, idx = sparse_quantize(pc, return_index=True, quantization_size=res)
    assert np.array_equal(idx, oracle.voxel_dedupe(pc, res).astype(np.int64))
    keys = np.floor(pc.astype(np.float64) / res).astype(np.int64)
    _, first = np.unique(keys, axis=0, return_index=True)
    assert np.array_equal(idx, np.sort(first))
    assert np.array_equal(coords, keys[idx].astype(np.int32))
    t_coords, t_idx = sparse_quantize(torch.from_numpy(pc).to(dev), quantization_size=res)   # torch in -> torch out
    assert t_idx.is_cuda and np.array_equal(t_idx.cpu().numpy(), idx)


def test_normals_match_oracle_numpy_and_known_answers(dev, oracle):
    rng = np.random.default_rng(3)
    # a noisy plane: the normal is known
    n0 = np.array([0.3, -0.5, 0.8]); n0 /= np.linalg.norm(n0)
    a = np.cross(n0, [1, 0, 0]); a /= np.linalg.norm(a)
    b = np.cross(n0, a)
    P = (rng.uniform(-1, 1, (3000, 1)) * a + rng.uniform(-1, 1, (3000, 1)) * b + rng.normal(0, 1e-3, (3000, 1)) * n0).astype(np.float32)
    nr = estimate_normals(P, 30)
    assert nr.shape == (3000, 3) and nr.dtype == np.float32
    assert np.array_equal(nr, oracle.estimate_normals(P, oracle.knn(P, 30)))            # bit-exact vs the oracle
    assert np.degrees(np.arccos(np.clip(np.abs(nr @ n0), 0, 1))).max() < 2.0
    np.testing.assert_allclose(np.linalg.norm(nr.astype(np.float64), axis=1), 1.0, atol=1e-6)
    assert (nr[np.arange(3000), np.abs(nr).argmax(1)] > 0).all()                       # the sign convention
    # a sphere: normals are radial; against numpy's eigh on the same neighbour sets, up to sign
    S = rng.normal(size=(4096, 3)); S = (S / np.linalg.norm(S, axis=1, keepdims=True) * 0.5).astype(np.float32)
    ns = estimate_normals(torch.from_numpy(S).to(dev), 60)
    assert ns.is_cuda
    ns = ns.cpu().numpy()
    radial = S / np.linalg.norm(S, axis=1, keepdims=True)
    assert np.degrees(np.arccos(np.clip(np.abs((ns * radial).sum(1)), 0, 1))).max() < 3.0
    nb = oracle.knn(S, 60)
    for i in range(0, 4096, 257):
        q = S[nb[i]].astype(np.float64)
        w, v = np.linalg.eigh(np.cov(q.T, bias=True))
        e = v[:, 0] * np.sign(v[np.abs(v[:, 0]).argmax(), 0])
        np.testing.assert_allclose(ns[i], e, atol=2e-6)


def test_raw_points_to_pose_entirely_on_device(dev, oracle):
    """The instance loop of nocs/inference.py:131-339 from raw points: voxel de-duplication -> normals -> kNN + SPRIN
    features -> pair path -> pose, every stage a device kernel of this package; each intermediate equals the oracle's."""
    import cppf_amd.synthetic as syn
    from cppf_amd.inference import estimate_pose
    from cppf_amd.models.model import PPFEncoder, PointEncoder
    from cppf_amd.utils.util import fibonacci_sphere
    ob = syn.make_object("mug", 3000, 21)
    cfg = ob["cfg"]
    raw = np.concatenate([ob["pc"], ob["pc"][:500] + np.float32(1e-5)])          # near-duplicates inside the same voxels
    _, keep = sparse_quantize(raw, return_index=True, quantization_size=cfg.res)
    assert np.array_equal(keep, oracle.voxel_dedupe(raw, cfg.res))
    pc = raw[keep].astype(np.float32)
    nrm = estimate_normals(pc, 60)
    assert np.array_equal(nrm, oracle.estimate_normals(pc, oracle.knn(pc, 60)))
    torch.manual_seed(5)
    penc = PointEncoder(k=60, spfcs=[32, 64, 32, 32], num_layers=1, out_dim=32).eval()
    psd = {k: v.detach().numpy().copy() for k, v in penc.state_dict().items()}
    enc = PPFEncoder(cfg.ppffcs, cfg.out_dim).eval()
    sd = {k: v.detach().numpy().copy() for k, v in enc.state_dict().items()}
    penc, enc = penc.to(dev), enc.to(dev)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    with torch.no_grad():
        feat = penc(d(pc)[None], d(nrm)[None])[0]
    packed, desc = oracle.pack_point_encoder(psd, 1)
    feat_o = oracle.point_encoder(pc, nrm, oracle.knn(pc, 60), packed, desc, order=1)
    assert np.array_equal(feat.cpu().numpy(), feat_o)
    n = pc.shape[0]
    idx = syn.make_pairs(n, 24, 21)
    u_tr, u_rot = syn.make_uniforms(idx.shape[0], 21)
    sph = np.array(fibonacci_sphere(480))
    with torch.no_grad():
        r = estimate_pose(enc, d(pc), d(nrm), feat, d(idx), d(u_tr), d(u_rot), cfg, sph, pc_host=pc)
    ocfg = dict(res=cfg.res, tr_num_bins=32, rot_num_bins=36, vote_range=cfg.vote_range, scale_mean=cfg.scale_mean,
                regress_right=cfg.regress_right, ppffcs=cfg.ppffcs, out_dim=cfg.out_dim)
    o = oracle.estimate_pose(pc, nrm, feat_o, idx, sd, ocfg, u_tr, u_rot, sph)
    assert r["argmax"] == o["argmax"] and r["n_surv"] == int(o["mask"].sum())
    assert np.allclose(r["T"], o["T"], atol=1e-9) and np.allclose(r["up"], o["up"], atol=1e-9)
    assert np.allclose(r["scale"], o["scale"], rtol=1e-5)


def test_backproject_on_device_equals_oracle(oracle, golden, dev):
    """cppf_backproject (utils/util.py:598-631 on the device): pixel list and fp64 points bit for bit the oracle's, which
    tests/test_oracle_golden.py pins on the reference's own function; uint16 and float32 depth, numpy and device inputs,
    an empty mask; and the cloud goes on to voxel de-duplication without leaving the device."""
    from cppf_amd.utils.util import backproject, sparse_quantize
    g = golden("backproject.npz")
    K = g["intrinsics"]
    for k in range(2):
        for d in (g["depth"], g["depth"].astype(np.float32) * np.float32(0.37)):
            po, (ro, co) = oracle.backproject(d, K, g["masks"][k])
            pts, (rows, cols) = backproject(d, K, g["masks"][k])
            assert np.array_equal(rows, ro) and np.array_equal(cols, co)
            assert np.array_equal(pts, po)
    pts, (rows, cols) = backproject(g["depth"], K, np.zeros_like(g["masks"][0]))
    assert pts.shape == (0, 3) and rows.size == 0
    # device in, device out: one depth upload per frame, the points never visit the host
    d_dev = torch.from_numpy(g["depth"].view(np.int16)).to(dev)
    pd, pix = backproject(d_dev, K, torch.from_numpy(g["masks"][0]).to(dev), return_device=True)
    po, (ro, co) = oracle.backproject(g["depth"], K, g["masks"][0])
    assert pd.is_cuda and np.array_equal(pd.cpu().numpy(), po) and np.array_equal(pix.cpu().numpy(), ro * g["depth"].shape[1] + co)
    pc = (pd / 1000.0)                                                      # nocs/inference.py:132
    pc = torch.stack([-pc[:, 0], -pc[:, 1], pc[:, 2]], -1).float()          # :136-137
    _, keep = sparse_quantize(pc, return_index=True, quantization_size=0.004)
    ko = oracle.voxel_dedupe(pc.cpu().numpy(), 0.004)
    assert np.array_equal(keep.cpu().numpy() if hasattr(keep, "cpu") else keep, ko)


def _philox4x32_10(c, k):
    """numpy restatement of Philox-4x32-10 (Salmon et al., SC'11; the Random123 reference constants) on arrays of counters"""
    c = [np.asarray(x, np.uint64) for x in c]
    k0, k1 = np.uint64(k[0]), np.uint64(k[1])
    M0, M1, W0, W1, lo = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & lo, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & lo]
        k0, k1 = (k0 + W0) & lo, (k1 + W1) & lo
    return c


def test_device_pair_sampler_is_philox_and_uniform(dev):
    """cppf_sample_pairs (the pair list of nocs/inference.py:177 and the stand-ins for torch.multinomial's draws, drawn on the device):
    equal to a numpy Philox-4x32-10 keyed by the seed with counter = pair index -- so a pair's draw is a function of (seed, index)
    alone --, indices uniform over [0, N), uniforms in [0, 1); N by value and from a device record"""
    import torch
    from cppf_amd import _lib
    from cppf_amd._torch_util import stream_ptr
    L = _lib.lib()
    P, N, seed = 200003, 3001, 0x1234_5678_9ABC_DEF1
    idx = torch.empty((P, 2), dtype=torch.int64, device=dev)
    u = torch.empty((2, P, 2), dtype=torch.float32, device=dev)
    _lib.check(L.cppf_sample_pairs(idx.data_ptr(), u[0].data_ptr(), u[1].data_ptr(), P, N, None, seed, None, stream_ptr(dev)), "sample")
    p = np.arange(P, dtype=np.uint64)
    key = (seed & 0xFFFFFFFF, seed >> 32)
    a = _philox4x32_10([p, np.zeros_like(p), np.zeros_like(p), np.zeros_like(p)], key)
    b = _philox4x32_10([p, np.zeros_like(p), np.ones_like(p), np.zeros_like(p)], key)
    want_idx = np.stack([(a[0] * np.uint64(N)) >> np.uint64(32), (a[1] * np.uint64(N)) >> np.uint64(32)], -1).astype(np.int64)
    f = lambda r: ((r >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24))
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    assert np.array_equal(u[0].cpu().numpy(), np.stack([f(a[2]), f(a[3])], -1)) and np.array_equal(u[1].cpu().numpy(), np.stack([f(b[0]), f(b[1])], -1))
    import cppf_amd.synthetic as syn             # the host twin bench_cpu.py re-draws a device-resident batch's pairs with
    ti, tu, tv = syn.philox_pairs(seed, P, N)
    assert np.array_equal(ti, want_idx) and np.array_equal(tu, u[0].cpu().numpy()) and np.array_equal(tv, u[1].cpu().numpy())
    i, uu = idx.cpu().numpy(), u.cpu().numpy()
    assert i.min() == 0 and i.max() == N - 1 and uu.min() >= 0.0 and uu.max() < 1.0
    counts = np.bincount(i.reshape(-1), minlength=N)
    chi2 = float(((counts - 2 * P / N) ** 2 / (2 * P / N)).sum())
    assert abs(chi2 - N) < 6 * np.sqrt(2 * N), chi2                      # chi-square of a uniform draw: N +- sqrt(2N)
    assert abs(uu.mean() - 0.5) < 2e-3 and abs(np.corrcoef(i[:, 0], i[:, 1])[0, 1]) < 0.01
    # a shorter list is a prefix; another seed is another list; N from a device record
    idx2 = torch.empty((1000, 2), dtype=torch.int64, device=dev)
    nd = torch.tensor([N, 1, 1, 1], dtype=torch.int32, device=dev)
    _lib.check(L.cppf_sample_pairs(idx2.data_ptr(), None, None, 1000, 1, nd.data_ptr(), seed, None, stream_ptr(dev)), "sample")
    assert torch.equal(idx2, idx[:1000])
    _lib.check(L.cppf_sample_pairs(idx2.data_ptr(), None, None, 1000, N, None, seed + 1, None, stream_ptr(dev)), "sample")
    assert not torch.equal(idx2, idx[:1000])
