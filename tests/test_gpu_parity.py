"""Parity of the HIP path (through the C ABI) with the oracle on identical seeded inputs.

Bar: bit-exact for every integer / index / discrete outcome (sampled bins, masks, survivor lists,
sphere counts, arg-max) and -- because the kernels and the oracle follow the same arithmetic
conventions -- also bit-exact for the per-pair float outputs (logits, (mu,nu), offsets,
candidates).  Only the vote grid is compared with a tolerance: fp32 atomic accumulation order is
unspecified, in the reference as well (atol = 1e-5 * number of contributions).
"""
import numpy as np
import pytest
import torch

import cppf_amd.synthetic as syn
from cppf_amd import _lib
from cppf_amd.config import CATEGORIES
from cppf_amd.models import voting
from cppf_amd.models.model import PPFEncoder
from cppf_amd._torch_util import stream_ptr, workspace
from conftest import sd_from_npz

pytestmark = pytest.mark.gpu


def t(x, dev, dtype=None):
    a = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        a = a.to(dtype)
    return a.to(dev)


def make_encoder(sd, ppffcs, out_dim, dev):
    enc = PPFEncoder(ppffcs, out_dim)
    enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return enc.to(dev).eval()


def seeded_sd(seed=0, ppffcs=(84, 32, 32, 16), out_dim=141):
    torch.manual_seed(seed)
    enc = PPFEncoder(list(ppffcs), out_dim)
    return {k: v.detach().numpy().copy() for k, v in enc.state_dict().items()}


# ------------------------------------------------------------------------------------ pair MLP
@pytest.mark.parametrize("tag,out_dim", [("141", 141), ("9", 9)])
def test_mlp_mfma_bit_exact_vs_oracle_and_close_to_reference(oracle, golden, dev, tag, out_dim):
    g = golden(f"mlp_{tag}.npz")
    sd = sd_from_npz(g)
    enc = make_encoder(sd, [84, 32, 32, 16], out_dim, dev)
    with torch.no_grad():
        y = enc(t(g["pc"], dev)[None], t(g["nrm"], dev)[None], t(g["feat"], dev)[None], idxs=g["idxs"])
        y2 = enc.forward_with_idx(t(g["pc"], dev), t(g["nrm"], dev), t(g["feat"], dev), t(g["idxs"], dev))
        y3 = enc.forward_with_idx(t(g["pc"], dev), t(g["nrm"], dev), t(g["feat"], dev), t(g["idxs"], dev, torch.int32))
    assert y.shape == (1, g["idxs"].shape[0], out_dim) and y.dtype == torch.float32 and y.device == dev
    yo = oracle.pair_mlp(g["pc"], g["nrm"], g["feat"], g["idxs"], sd, [84, 32, 32, 16], out_dim, order=1)
    np.testing.assert_array_equal(y[0].cpu().numpy(), yo)               # MFMA chain == fmaf chain, bit for bit
    assert torch.equal(y[0], y2) and torch.equal(y2, y3)
    np.testing.assert_allclose(y[0].cpu().numpy(), g["logits"], rtol=0, atol=2e-6)   # reference forward


def test_mlp_generic_architecture(oracle, golden, dev):
    g = golden("mlp_generic.npz")
    sd = sd_from_npz(g)
    enc = make_encoder(sd, [44, 24, 24], 10, dev)
    with torch.no_grad():
        y = enc.forward_with_idx(t(g["pc"], dev), t(g["nrm"], dev), t(g["feat"], dev), t(g["idxs"], dev))
    yo = oracle.pair_mlp(g["pc"], g["nrm"], g["feat"], g["idxs"], sd, [44, 24, 24], 10, order=0)
    np.testing.assert_array_equal(y.cpu().numpy(), yo)
    np.testing.assert_allclose(y.cpu().numpy(), g["logits"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("P", [1, 15, 33, 1000, 4097])
def test_mlp_ragged_sizes(oracle, dev, P):
    ob = syn.make_object("bottle", 300, 5)
    idx = np.random.default_rng(P).integers(0, 300, (P, 2)).astype(np.int64)
    sd = seeded_sd(1)
    enc = make_encoder(sd, [84, 32, 32, 16], 141, dev)
    with torch.no_grad():
        y = enc.forward_with_idx(t(ob["pc"], dev), t(ob["normals"], dev), t(ob["feat"], dev), t(idx, dev))
    yo = oracle.pair_mlp(ob["pc"], ob["normals"], ob["feat"], idx, sd, [84, 32, 32, 16], 141, order=1)
    np.testing.assert_array_equal(y.cpu().numpy(), yo)


def test_mlp_empty_and_dense(oracle, dev):
    ob = syn.make_object("can", 40, 2)
    sd = seeded_sd(2)
    enc = make_encoder(sd, [84, 32, 32, 16], 141, dev)
    pc, n, f = t(ob["pc"], dev), t(ob["normals"], dev), t(ob["feat"], dev)
    with torch.no_grad():
        y0 = enc.forward_with_idx(pc, n, f, torch.zeros((0, 2), dtype=torch.int64, device=dev))
        yd = enc(pc[None], n[None], f[None])                       # dense all-pairs branch
    assert y0.shape == (0, 141)
    assert yd.shape == (1, 40, 40, 141)
    ii, jj = np.meshgrid(np.arange(40), np.arange(40), indexing="ij")
    allp = np.stack([ii.reshape(-1), jj.reshape(-1)], -1).astype(np.int64)
    yo = oracle.pair_mlp(ob["pc"], ob["normals"], ob["feat"], allp, sd, [84, 32, 32, 16], 141, order=1)
    np.testing.assert_array_equal(yd.reshape(-1, 141).cpu().numpy(), yo)


# ------------------------------------------------------------------------------------ decode
@pytest.mark.parametrize("sharpen", [8, 60, 400])
def test_fused_decode_bit_exact(oracle, dev, sharpen):
    """sharpen = 8: non-uniform distributions; 60 / 400: logit spreads of tens to thousands, where the softmax weights run
    through the subnormals to zero (ldexp underflow in the weight function, csrc/cppf_math.h:det_exp2w) -- same bins either way"""
    ob = syn.make_object("bottle", 1024, 7)
    idx = syn.make_pairs(1024, 16, 7)
    P = idx.shape[0]
    u_tr, u_rot = syn.make_uniforms(P, 7)
    u_tr[:50] = -1.0                                               # arg-max mode rows
    u_rot[25:75] = -1.0
    u_tr[100, 0] = 0.0
    u_tr[101, 1] = np.float32(1.0) - np.float32(2 ** -24)
    sd = seeded_sd(0)
    for k in ("final.weight", "final.bias"):                       # sharper, non-uniform distributions
        sd[k] = sd[k] * sharpen
    enc = make_encoder(sd, [84, 32, 32, 16], 141, dev)
    cfg = ob["cfg"]
    with torch.no_grad():
        outputs, heads = enc.forward_decode(t(ob["pc"], dev), t(ob["normals"], dev), t(ob["feat"], dev), t(idx, dev),
                                            t(u_tr, dev), cfg.vote_range, t(u_rot, dev))
        logits = enc.forward_with_idx(t(ob["pc"], dev), t(ob["normals"], dev), t(ob["feat"], dev), t(idx, dev))
    lo = oracle.pair_mlp(ob["pc"], ob["normals"], ob["feat"], idx, sd, [84, 32, 32, 16], 141, order=1)
    oo, bins = oracle.decode_center(lo, u_tr, 32, cfg.vote_range)
    ho, rbins = oracle.decode_rot(lo, u_rot, 32, 36)
    np.testing.assert_array_equal(outputs.cpu().numpy(), oo)
    np.testing.assert_array_equal(heads.cpu().numpy(), ho)
    assert sharpen > 8 or (len(np.unique(bins)) > 20 and len(np.unique(rbins)) > 20)
    if sharpen >= 60:   # the spread really reaches the underflow region
        assert float((lo[:, :32].max(1) - lo[:, :32].min(1)).max()) > (90.0 if sharpen == 60 else 200.0)
    # decode-from-memory kernels agree too
    L = _lib.lib()
    out2 = torch.empty_like(outputs)
    heads2 = torch.empty_like(heads)
    utr_d, urot_d = t(u_tr, dev), t(u_rot, dev)
    _lib.check(L.cppf_decode_center(logits.data_ptr(), P, 141, 32, cfg.vote_range[0], cfg.vote_range[1],
                                    utr_d.data_ptr(), out2.data_ptr(), stream_ptr(dev)), "decode_center")
    _lib.check(L.cppf_decode_rot(logits.data_ptr(), P, 141, 141, 32, 36, urot_d.data_ptr(), heads2.data_ptr(),
                                 stream_ptr(dev)), "decode_rot")
    torch.cuda.synchronize()
    assert torch.equal(out2, outputs) and torch.equal(heads2, heads)


def test_second_pass_on_selected_pairs(oracle, dev):
    """cppf_pair_mlp_decode_sel (nocs/inference.py:236-256 on point_idxs[mask]): the heads rows of the selected pairs equal
    the full decode's (hence the oracle's), rows that are not selected are untouched, the count comes from the device, a
    smaller max_sel caps it, an empty selection writes nothing; int32 and int64 pair lists."""
    ob = syn.make_object("mug", 777, 9)
    idx = syn.make_pairs(777, 20, 9)
    P = idx.shape[0]
    u_tr, u_rot = syn.make_uniforms(P, 9)
    u_rot[::97] = -1.0
    sd = seeded_sd(3)
    for k in ("final.weight", "final.bias"):
        sd[k] = sd[k] * 6
    enc = make_encoder(sd, [84, 32, 32, 16], 141, dev)
    cfg = ob["cfg"]
    pc, nrm, feat, urot_d = t(ob["pc"], dev), t(ob["normals"], dev), t(ob["feat"], dev), t(u_rot, dev)
    rng = np.random.default_rng(1)
    for idx_d in (t(idx, dev), t(idx.astype(np.int32), dev)):
        with torch.no_grad():
            _, full = enc.forward_decode(pc, nrm, feat, idx_d, t(u_tr, dev), cfg.vote_range, urot_d)
            enc.forward_decode(pc, nrm, feat, idx_d, t(u_tr, dev), cfg.vote_range, None)      # the first pass of a pose chain
        for n_sel, max_sel in ((0, P), (1, P), (15, P), (16, P), (17, P), (1000, P), (1000, 333), (P, P)):
            sel = np.sort(rng.choice(P, n_sel, replace=False)).astype(np.int32)
            sel_d = torch.full((P,), -1, dtype=torch.int32, device=dev)                        # garbage behind the count
            sel_d[:n_sel] = t(sel, dev)
            cnt = torch.tensor([n_sel], dtype=torch.int32, device=dev)
            heads = torch.full((P, 8), -3.0, dtype=torch.float32, device=dev)
            with torch.no_grad():
                enc.forward_decode_sel(pc, nrm, feat, idx_d, urot_d, sel_d, cnt, heads, max_sel=max_sel)
            used = sel[:min(n_sel, max_sel)]
            got = heads.cpu().numpy()
            np.testing.assert_array_equal(got[used], full.cpu().numpy()[used])
            rest = np.ones(P, bool)
            rest[used] = False
            assert np.all(got[rest] == -3.0)
    lo = oracle.pair_mlp(ob["pc"], ob["normals"], ob["feat"], idx, sd, [84, 32, 32, 16], 141, order=1)
    ho, _ = oracle.decode_rot(lo, u_rot, 32, 36)
    np.testing.assert_array_equal(full.cpu().numpy(), ho)


# ------------------------------------------------------------------------------------ centre vote
def run_vote(dev, pc, outputs, idx32, corner, dims, res, n_rots, adaptive, probs=None, grid0=None, workgroups=0):
    N = pc.shape[0]
    probs = np.ones(N, np.float32) if probs is None else probs
    if grid0 is not None:       # += into a pre-filled grid (the reference's semantics)
        grid = t(grid0, dev).clone()
        oi, ov = voting.vote_argmax(t(pc, dev), t(outputs, dev), t(probs, dev), t(idx32, dev), grid, t(corner, dev),
                                    res, n_rots, adaptive, workgroups=workgroups)
    else:                       # overwrite mode on a poisoned grid must equal += on a zeroed one
        grid = torch.full(tuple(int(d) for d in dims), float("nan"), dtype=torch.float32, device=dev)
        oi, ov = voting.vote_argmax(t(pc, dev), t(outputs, dev), t(probs, dev), t(idx32, dev), grid, t(corner, dev),
                                    res, n_rots, adaptive, accumulate=False, workgroups=workgroups)
    torch.cuda.synchronize()
    return grid.cpu().numpy(), int(oi.item()), float(ov.item())


def oracle_vote(oracle, pc, outputs, idx32, corner, dims, res, n_rots, adaptive, probs=None, grid0=None):
    probs = np.ones(pc.shape[0], np.float32) if probs is None else probs
    grid = np.zeros(tuple(int(d) for d in dims), np.float32) if grid0 is None else grid0.copy()
    na = oracle.ppf_voting(pc, outputs, probs, idx32, grid, corner, res, n_rots, adaptive)
    return grid, na


def check_grid(oracle, gg, pc, outputs, idx32, corner, dims, res, n_rots, adaptive, probs=None, grid0=None, bits_slack=0):
    """GPU grid vs the exact (fp64) vote sum.  Tiled path: each deposit is rounded to the fixed-point
    quantum p2*2^-bits (csrc/vote.hip), so |gpu - exact| <= deposits/2 quanta + fp32 rounding of the
    chunk partial sums.  Global-atomics path: fp32 atomic order noise, like the reference itself."""
    probs = np.ones(pc.shape[0], np.float32) if probs is None else probs
    g64, cnt = oracle.ppf_voting_f64(pc, outputs, probs, idx32, dims, corner, res, n_rots, adaptive)
    if grid0 is not None:
        g64 = g64 + grid0
    bits = _lib.lib().cppf_vote_fixed_point_bits(idx32.shape[0], n_rots, int(dims[0]), int(dims[1]), int(dims[2]))
    bits = bits - bits_slack if bits > 0 else bits       # (a narrower launch, CPPF_VOTE_WORKGROUPS: longer chunks, up to two bits coarser)
    pmax = float(np.max(probs))
    if bits > 0 and np.all(np.isfinite(probs)) and np.all(probs >= 0):
        p2 = 2.0 ** np.ceil(np.log2(pmax)) if pmax > 0 else 1.0
        tol = 2e-6 * np.abs(g64) + cnt * (0.5 * p2 * 2.0 ** -bits) + 1e-30
    else:   # fp32 atomics in whatever order they land: the rounding of n additions into one cell grows like sqrt(n) ulps of the sum
        tol = (2e-5 + 1.2e-7 * np.sqrt(cnt)) * np.abs(g64) + 1e-6 * max(pmax, 1e-30)
    err = np.abs(gg.astype(np.float64) - g64)
    worst = np.unravel_index(np.argmax(err - tol), err.shape)
    assert np.all(err <= tol), f"cell {worst}: gpu {gg[worst]} exact {g64[worst]} tol {tol[worst]} deposits {cnt[worst]}"
    return g64, cnt


def vote_case(oracle, dev, cat, n, k, seed, adaptive, quantise=True, n_rots=72, res_scale=1.0, corner_shift=0.0):
    ob = syn.make_object(cat, n, seed)
    cfg = ob["cfg"]
    res = cfg.res * res_scale
    idx = syn.make_pairs(n, k, seed)
    outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=quantise)
    corner, dims = oracle.grid_setup(ob["pc"], res)
    corner = (corner + np.float32(corner_shift)).astype(np.float32)
    idx32 = idx.astype(np.int32)
    go, na = oracle_vote(oracle, ob["pc"], outputs, idx32, corner, dims, res, n_rots, adaptive)
    gg, flat, peak = run_vote(dev, ob["pc"], outputs, idx32, corner, dims, res, n_rots, adaptive)
    check_grid(oracle, gg, ob["pc"], outputs, idx32, corner, dims, res, n_rots, adaptive)
    return ob, go, na, gg, flat, peak, dims


@pytest.mark.parametrize("cat,adaptive,quantise", [("bottle", True, True), ("bottle", False, True), ("bottle", True, False),
                                                    ("laptop", True, True), ("mug", False, False), ("bed", True, True)])
def test_vote_grid_and_argmax_match_oracle(oracle, dev, cat, adaptive, quantise):
    ob, go, na, gg, flat, peak, dims = vote_case(oracle, dev, cat, 1024, 32, 11, adaptive, quantise)
    assert na > 0
    # (vote_case already checked every cell against the exact fp64 sum)
    np.testing.assert_allclose(gg, go, rtol=2e-5, atol=1e-5 * max(1.0, go.max()))  # and the fp32 serial restatement
    np.testing.assert_allclose(gg.sum(dtype=np.float64), na / 8, rtol=1e-6)     # checksum: total weight = #votes
    oflat, opeak = oracle.grid_argmax(go)
    srt = np.sort(go.reshape(-1))
    assert srt[-1] - srt[-2] > 1e-4 * srt[-1], "fixture must have a dominant peak"
    assert flat == oflat                                                        # bit-exact arg-max index
    assert abs(peak - opeak) <= 2e-5 * opeak


def test_vote_null_probs_equals_ones(dev):
    """probs = NULL is the all-ones tensor every caller of the reference passes (nocs/inference.py:201): same grid, bit for bit,
    on the tiled path and on the global-atomics fallback (> 64 tiles)"""
    for res_scale, n in ((1.0, 1024), (0.12, 256)):
        ob = syn.make_object("bottle", n, 3)
        cfg = ob["cfg"]
        res = float(np.float32(cfg.res * res_scale))
        idx = syn.make_pairs(n, 16, 3)
        outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg)
        from cppf_amd.inference import grid_shape
        corners, dims = grid_shape(ob["pc"], res)
        args = (t(ob["pc"], dev), t(outputs, dev))
        g1 = torch.zeros(tuple(int(d) for d in dims), dtype=torch.float32, device=dev)
        g0 = torch.zeros_like(g1)
        i1, v1 = voting.vote_argmax(*args, torch.ones(n, device=dev), t(idx, dev), g1, t(corners[0], dev), res, 72, True)
        i0, v0 = voting.vote_argmax(*args, None, t(idx, dev), g0, t(corners[0], dev), res, 72, True)
        torch.cuda.synchronize()
        if res_scale == 1.0:     # (the fallback's fp32 atomics are order-dependent: compared with a tolerance there)
            assert torch.equal(g0, g1) and int(i0.item()) == int(i1.item()) and float(v0.item()) == float(v1.item())
        else:
            assert _lib.lib().cppf_vote_tiles(int(dims[0]), int(dims[1]), int(dims[2])) == 0      # no LDS plan: global atomics
            torch.testing.assert_close(g0, g1, rtol=2e-5, atol=1e-6)
        assert float(g1.sum().item()) > 100


def test_vote_tiling_paths(oracle, dev):
    # fine grid -> x and y tiles; very fine grid -> global-atomic fallback; coarse -> single tile
    for res_scale, expect in ((1.0, "tiles"), (0.5, "tiles"), (0.2, "global"), (4.0, "one")):
        ob, go, na, gg, flat, peak, dims = vote_case(oracle, dev, "bottle", 512, 16, 3, True, True,
                                                     res_scale=res_scale)
        G = int(np.prod(dims))
        if expect == "global":
            assert G > 16 * 32768
        if expect == "one":
            assert G <= 32768
        assert flat == oracle.grid_argmax(go)[0]
    # wide-and-flat grid: forces y cuts (gy*gz > tile budget)
    pc = np.array([[0, 0, 0], [0.02, 2.0, 0.5], [0.01, 1.0, 0.2], [0.0, 0.5, 0.4]], np.float32)
    rng = np.random.default_rng(0)
    idx32 = rng.integers(0, 4, (4096, 2)).astype(np.int32)
    outputs = np.stack([rng.uniform(-0.2, 0.2, 4096), rng.uniform(0, 0.3, 4096)], -1).astype(np.float32)
    corner = np.array([-0.02, -0.1, -0.1], np.float32)
    dims = (6, 300, 140)                     # plane of 42 000 cells > tile budget: 2 y cuts x 6 x slabs
    go, na = oracle_vote(oracle, pc, outputs, idx32, corner, dims, 0.01, 72, True)
    gg, flat, peak = run_vote(dev, pc, outputs, idx32, corner, dims, 0.01, 72, True)
    assert na > 0
    check_grid(oracle, gg, pc, outputs, idx32, corner, dims, 0.01, 72, True)
    assert flat == oracle.grid_argmax(go)[0]


def test_vote_rotation_table_cached_in_workspace(oracle, dev):
    """The (cos, sin) table is built by the first launch on a workspace and read back by later ones (csrc/vote.hip,
    VOTE_TAB_STAMP): a fresh workspace, repeated launches and a change of n_rots all give the same, correct grid.  (The
    launch that builds the table leaves a PENDING stamp that only the following reduce kernel validates: late workgroups
    of a 2 000-workgroup launch once read the table their own launch was still writing -- an intermittent failure of
    test_randomised_soak in round 2; this case has that geometry.)"""
    from cppf_amd import _torch_util
    ob = syn.make_object("bottle", 2048, 31)
    cfg = ob["cfg"]
    res = cfg.res * 0.45                                     # ~20 tiles -> 2 000+ workgroups
    idx = syn.make_pairs(2048, 48, 31).astype(np.int32)
    out = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
    from cppf_amd.inference import grid_shape
    corners, dims = grid_shape(ob["pc"], res)
    assert _lib.lib().cppf_vote_tiles(*dims) >= 8
    rng = np.random.default_rng(3)
    probs = rng.uniform(0.4, 2.0, 2048).astype(np.float32)
    grids = {}
    for rep, n_rots in enumerate((72, 72, 72, 40, 72, 72)):
        if rep in (0, 3):
            _torch_util._ws_cache.clear()                    # a fresh (uninitialised) workspace: no stamp
        gg, flat, peak = run_vote(dev, ob["pc"], out, idx, corners[0], dims, res, n_rots, True, probs)
        check_grid(oracle, gg, ob["pc"], out, idx, corners[0], dims, res, n_rots, True, probs)
        if n_rots in grids:
            assert np.array_equal(gg, grids[n_rots]), rep
        grids[n_rots] = gg


def test_vote_edge_cases(oracle, dev):
    pc = np.array([[0, 0, 0], [0.1, 0, 0], [0, 0.1, 0], [0, 0, 0.1]], np.float32)
    corner = np.array([-0.2, -0.2, -0.2], np.float32)
    idx = np.array([[0, 0], [1, 0], [2, 0], [3, 0], [1, 2], [2, 2]], np.int32)
    out = np.array([[0.1, 0.05], [0.1, 0.003], [0.0, 0.01], [0.05, 0.25], [-0.1, 0.1], [0.1, 0.1]], np.float32)
    probs = np.array([1.0, 0.5, 2.0, 0.25], np.float32)                        # prob = max(probs[a], probs[b])
    for adaptive in (True, False):
        for n_rots in (72, 7, 1, 100):
            go, na = oracle_vote(oracle, pc, out, idx, corner, (20, 20, 20), 0.02, n_rots, adaptive, probs)
            gg, flat, peak = run_vote(dev, pc, out, idx, corner, (20, 20, 20), 0.02, n_rots, adaptive, probs)
            check_grid(oracle, gg, pc, out, idx, corner, (20, 20, 20), 0.02, n_rots, adaptive, probs)
            np.testing.assert_allclose(gg, go, rtol=1e-5, atol=2e-6)
            assert flat == oracle.grid_argmax(go)[0]
    # everything out of the grid: grid untouched, arg-max = first cell (np.argmax of zeros)
    gg, flat, peak = run_vote(dev, pc, out, idx, corner + 5, (20, 20, 20), 0.02, 72, True)
    assert gg.sum() == 0 and flat == 0 and peak == 0
    # accumulation into a pre-filled grid (+=, like the reference's atomicAdd) and n_ppfs = 0
    g0 = np.random.default_rng(1).random((20, 20, 20)).astype(np.float32)
    go, _ = oracle_vote(oracle, pc, out, idx, corner, (20, 20, 20), 0.02, 72, True, grid0=g0)
    gg, flat, _ = run_vote(dev, pc, out, idx, corner, (20, 20, 20), 0.02, 72, True, grid0=g0)
    check_grid(oracle, gg, pc, out, idx, corner, (20, 20, 20), 0.02, 72, True, grid0=g0)
    assert flat == oracle.grid_argmax(go)[0]
    gg, flat, _ = run_vote(dev, pc, out[:0], idx[:0], corner, (20, 20, 20), 0.02, 72, True, grid0=g0)
    np.testing.assert_array_equal(gg, g0)
    assert flat == int(np.argmax(g0))


def test_vote_fixed_point_carry_and_prob_ranges(oracle, dev):
    """LDS accumulation is 32-bit fixed point with a carry log (csrc/vote.hip): force wrap-arounds
    (thousands of unit weights into a handful of cells), scale probs across 60 binades, and use
    negative / non-finite probs (fp32-atomic fallback)."""
    rng = np.random.default_rng(5)
    pc = np.array([[0, 0, 0], [0.1, 0, 0], [0, 0.1, 0], [0.03, 0.02, 0.1]], np.float32)
    corner = np.array([-0.2, -0.2, -0.2], np.float32)
    P = 6000
    idx = np.tile(np.array([[1, 0], [2, 0], [3, 1]], np.int32), (P // 3, 1))
    # nu ~ 0: all 72 rotations of a pair land on (nearly) the same point -> ~72 per pair into <= 8 cells
    out = np.stack([np.full(P, 0.05), np.full(P, 1e-6)], -1).astype(np.float32)
    for probs in (np.ones(4, np.float32), np.array([3.0, 0.5, 7.25, 1.0], np.float32),
                  np.full(4, 2.0 ** -30, np.float32), np.full(4, 2.0 ** 30, np.float32)):
        go, na = oracle_vote(oracle, pc, out, idx, corner, (20, 20, 20), 0.02, 72, False, probs)
        gg, flat, peak = run_vote(dev, pc, out, idx, corner, (20, 20, 20), 0.02, 72, False, probs)
        assert na == P * 72 * 8 and go.max() > 20000 * probs.min()      # far beyond one 2^32 wrap at 24 bits
        g64, cnt = check_grid(oracle, gg, pc, out, idx, corner, (20, 20, 20), 0.02, 72, False, probs)
        assert cnt.max() > 100000
        # the serial fp32 restatement (like any ordering of fp32 atomicAdd) has drifted ~1e-3 by now
        np.testing.assert_allclose(go, g64, rtol=5e-3, atol=1e-4 * probs.max())
        assert flat == int(np.argmax(g64))
    # exactness: the fixed-point sum of identical weights is exact, the fp32 serial sum is not
    gg, _, _ = run_vote(dev, pc, out[:3000], idx[:3000], corner, (20, 20, 20), 0.02, 72, False)
    assert abs(gg.sum(dtype=np.float64) - 3000 * 72) < 0.05
    # negative / NaN / inf probs: fp32 atomics path, still the reference's arithmetic
    out2 = np.stack([rng.uniform(-0.1, 0.1, 900), rng.uniform(0, 0.15, 900)], -1).astype(np.float32)
    idx2 = rng.integers(0, 4, (900, 2)).astype(np.int32)
    for probs in (np.array([1.0, -2.0, 0.5, -0.25], np.float32), np.array([1.0, np.inf, 0.5, 1.0], np.float32)):
        go, na = oracle_vote(oracle, pc, out2, idx2, corner, (20, 20, 20), 0.02, 72, True, probs)
        gg, flat, peak = run_vote(dev, pc, out2, idx2, corner, (20, 20, 20), 0.02, 72, True, probs)
        fin = np.isfinite(go)
        assert np.array_equal(fin, np.isfinite(gg))
        np.testing.assert_allclose(gg[fin], go[fin], rtol=1e-4, atol=1e-5)
    gg, _, _ = run_vote(dev, pc, out2, idx2, corner, (20, 20, 20), 0.02, 72, True, np.zeros(4, np.float32))
    assert not gg.any()                                                   # all-zero probs: nothing deposited


def test_ppf_kernel_drop_in_call_convention(oracle, dev):
    """Called exactly like the reference's launch at nocs/inference.py:197-205."""
    ob = syn.make_object("bottle", 256, 4)
    cfg = ob["cfg"]
    idx = syn.make_pairs(256, 8, 4)
    outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg)
    corner, dims = oracle.grid_setup(ob["pc"], cfg.res)
    pc = ob["pc"]
    block_size = (pc.shape[0] ** 2 + 512 - 1) // 512
    grid_obj = torch.zeros(tuple(int(d) for d in dims), dtype=torch.float32, device=dev)
    ret = voting.ppf_kernel(
        (block_size, 1, 1), (512, 1, 1),
        (t(pc, dev), t(outputs, dev), torch.ones(256, device=dev), t(idx, dev, torch.int32), grid_obj, t(corner, dev),
         np.float32(cfg.res), idx.shape[0], 72, grid_obj.shape[0], grid_obj.shape[1], grid_obj.shape[2], True))
    assert ret is None
    go, _ = oracle_vote(oracle, pc, outputs, idx.astype(np.int32), corner, dims, cfg.res, 72, True)
    np.testing.assert_allclose(grid_obj.cpu().numpy(), go, rtol=2e-5, atol=1e-5 * go.max())
    with pytest.raises(TypeError):
        voting.ppf_kernel((1, 1, 1), (512, 1, 1), (t(pc, dev),))
    with pytest.raises(TypeError):
        voting.ppf_kernel((1, 1, 1), (512, 1, 1),
                          (t(pc, dev), t(outputs, dev), torch.ones(256, device=dev), t(idx, dev), grid_obj,
                           t(corner, dev), cfg.res, idx.shape[0], 72, *grid_obj.shape, True))   # int64 idxs
    with pytest.raises(ValueError):
        voting.ppf_kernel((1, 1, 1), (512, 1, 1),
                          (t(pc, dev), t(outputs, dev), torch.ones(256, device=dev), t(idx, dev, torch.int32),
                           grid_obj, torch.from_numpy(corner), cfg.res, idx.shape[0], 72, *grid_obj.shape, True))


def test_grid_argmax_tie_rule_and_negatives(dev):
    L = _lib.lib()
    rng = np.random.default_rng(0)
    for n in (1, 63, 64, 65, 1000, 300001):
        g = rng.integers(-5, 5, n).astype(np.float32)            # many ties
        oi, ov = voting.grid_argmax(t(g, dev))
        assert int(oi.item()) == int(np.argmax(g)) and float(ov.item()) == g.max()
    g = -np.abs(rng.normal(size=5000)).astype(np.float32) - 1      # all negative
    oi, ov = voting.grid_argmax(t(g, dev))
    assert int(oi.item()) == int(np.argmax(g)) and float(ov.item()) == g.max()


# ------------------------------------------------------------------------------------ back-vote / rot
def test_backvote_compaction_bit_exact(oracle, dev):
    ob = syn.make_object("bottle", 1024, 9)
    cfg = ob["cfg"]
    idx = syn.make_pairs(1024, 20, 9)
    P = idx.shape[0]
    outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg)
    corner, dims = oracle.grid_setup(ob["pc"], cfg.res)
    idx32 = idx.astype(np.int32)
    center = (ob["center"] + 0.5 * cfg.res).astype(np.float32)
    oo, mask = oracle.backvote(ob["pc"], outputs, idx32, corner, cfg.res, 72, dims, center, np.float32(3 * cfg.res))
    # drop-in call (nocs/inference.py:219-228)
    output_ocs = torch.zeros((P, 3), dtype=torch.float32, device=dev)
    voting.backvote_kernel(((P + 511) // 512, 1, 1), (512, 1, 1),
                           (t(ob["pc"], dev), t(outputs, dev), output_ocs, t(idx32, dev), t(corner, dev),
                            np.float32(cfg.res), P, 72, int(dims[0]), int(dims[1]), int(dims[2]), t(center, dev),
                            np.float32(3 * cfg.res)))
    np.testing.assert_array_equal(output_ocs.cpu().numpy(), oo)
    assert 0.05 < mask.mean() < 1
    # C ABI with mask + compaction
    L = _lib.lib()
    m = torch.empty(P, dtype=torch.uint8, device=dev)
    surv = torch.full((P,), -1, dtype=torch.int32, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    output_ocs.zero_()
    # (device tensors are kept in named variables: a temporary's data_ptr() dangles once it is freed)
    pc_d, out_d, idx_d, cor_d, cen_d = t(ob["pc"], dev), t(outputs, dev), t(idx32, dev), t(corner, dev), t(center, dev)
    _lib.check(L.cppf_backvote(pc_d.data_ptr(), out_d.data_ptr(), output_ocs.data_ptr(),
                               idx_d.data_ptr(), cor_d.data_ptr(), cfg.res, P, 72, int(dims[0]),
                               int(dims[1]), int(dims[2]), cen_d.data_ptr(), float(np.float32(3 * cfg.res)),
                               m.data_ptr(), stream_ptr(dev)), "backvote")
    ws = workspace(L.cppf_compact_workspace_bytes(P), dev, "compact")
    _lib.check(L.cppf_compact_mask(m.data_ptr(), P, surv.data_ptr(), cnt.data_ptr(), ws.data_ptr(), ws.numel(),
                                   stream_ptr(dev)), "compact")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(m.cpu().numpy().astype(bool), mask)
    ref = np.nonzero(mask)[0]
    assert int(cnt.item()) == ref.size
    np.testing.assert_array_equal(surv.cpu().numpy()[:ref.size], ref)


@pytest.mark.parametrize("n", [0, 1, 1023, 1024, 1025, 70000, 1200000])
def test_compaction_sizes(dev, n):
    L = _lib.lib()
    rng = np.random.default_rng(n)
    mask = (rng.random(n) < 0.37).astype(np.uint8)
    m = t(mask, dev) if n else torch.empty(0, dtype=torch.uint8, device=dev)
    surv = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt = torch.full((1,), -7, dtype=torch.int32, device=dev)
    ws = workspace(L.cppf_compact_workspace_bytes(n), dev, "compact")
    _lib.check(L.cppf_compact_mask(m.data_ptr() if n else ws.data_ptr(), n, surv.data_ptr(), cnt.data_ptr(),
                                   ws.data_ptr(), ws.numel(), stream_ptr(dev)), "compact")
    ref = np.nonzero(mask)[0]
    assert int(cnt.item()) == ref.size
    np.testing.assert_array_equal(surv.cpu().numpy()[:ref.size], ref)


def test_rot_voting_and_sphere_count_bit_exact(oracle, golden, dev):
    ob = syn.make_object("camera", 512, 13)
    idx = syn.make_pairs(512, 6, 13)
    P = idx.shape[0]
    idx32 = idx.astype(np.int32)
    rng = np.random.default_rng(13)
    theta = (rng.integers(0, 36, P) / 35 * np.pi).astype(np.float32)     # both sides of pi/2, incl. 0 and pi
    co = oracle.rot_voting(ob["pc"], theta, idx32, 72)
    candidates = torch.zeros((P, 72, 3), dtype=torch.float32, device=dev)
    voting.rot_voting_kernel(((P + 511) // 512, 1, 1), (512, 1, 1),
                             (t(ob["pc"], dev), torch.zeros((P, 2), device=dev), t(theta, dev), candidates,
                              t(idx32, dev), torch.zeros(3, device=dev), np.float32(4e-3), P, 72, 10, 10, 10))
    np.testing.assert_array_equal(candidates.cpu().numpy(), co)
    # fused count on a selection, against the oracle's mm + threshold
    sph = golden("sphere.npz")["pts"]
    sel = np.sort(rng.choice(P, 700, replace=False)).astype(np.int32)
    counts_o = oracle.sphere_count(co[sel[:500]], sph, 1.5)
    L = _lib.lib()
    counts = torch.zeros(480, dtype=torch.int32, device=dev)
    nsel = torch.tensor([700], dtype=torch.int32, device=dev)
    thr = float(np.float32(np.cos(1.5 / 180 * np.pi)))
    pc_d, th_d, idx_d, sel_d, sph_d = t(ob["pc"], dev), t(theta, dev), t(idx32, dev), t(sel, dev), t(sph.astype(np.float32), dev)
    for sorted_y in (0, 1):          # full sweep and the banded search (Fibonacci bins: y descending)
        counts.zero_()
        _lib.check(L.cppf_rot_sphere_count(pc_d.data_ptr(), th_d.data_ptr(), 1,
                                           idx_d.data_ptr(), sel_d.data_ptr(), nsel.data_ptr(), P, 500, 72,
                                           sph_d.data_ptr(), 480, thr, sorted_y, counts.data_ptr(),
                                           stream_ptr(dev)), "rot_sphere_count")
        torch.cuda.synchronize()
        np.testing.assert_array_equal(counts.cpu().numpy().astype(np.int64), counts_o)
    # ascending order and a loose threshold (wide band) give the same counts as the oracle as well
    rev = sph[::-1].copy()
    counts_r = oracle.sphere_count(co[sel[:500]], rev, 25.0)
    rev_d = t(rev.astype(np.float32), dev)
    thr2 = float(np.float32(np.cos(25.0 / 180 * np.pi)))
    counts.zero_()
    _lib.check(L.cppf_rot_sphere_count(pc_d.data_ptr(), th_d.data_ptr(), 1, idx_d.data_ptr(), sel_d.data_ptr(),
                                       nsel.data_ptr(), P, 500, 72, rev_d.data_ptr(), 480, thr2, -1, counts.data_ptr(),
                                       stream_ptr(dev)), "rot_sphere_count")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(counts.cpu().numpy().astype(np.int64), counts_r)
    assert counts_o.sum() > 0


def test_axis_sign_and_scale_sums(oracle, dev):
    ob = syn.make_object("mug", 600, 21)
    idx = syn.make_pairs(600, 10, 21)
    P = idx.shape[0]
    idx32 = idx.astype(np.int32)
    rng = np.random.default_rng(21)
    heads = rng.normal(0, 1.5, (P, 8)).astype(np.float32)
    sel = np.sort(rng.choice(P, 2500, replace=False)).astype(np.int32)
    bd = np.array([0.2, 0.95, -0.1])
    bd /= np.linalg.norm(bd)
    flip, (up, down) = oracle.axis_sign(ob["pc"], ob["normals"], idx32[sel], heads[sel, 3], bd)
    L = _lib.lib()
    out = torch.empty(3, dtype=torch.float64, device=dev)
    out4 = torch.empty(4, dtype=torch.float64, device=dev)
    ws = workspace(L.cppf_reduce_workspace_bytes(), dev, "reduce")
    nsel = torch.tensor([2500], dtype=torch.int32, device=dev)
    hd = t(heads, dev)
    pc_d, n_d, idx_d, sel_d, bd_d = t(ob["pc"], dev), t(ob["normals"], dev), t(idx32, dev), t(sel, dev), t(bd, dev)
    _lib.check(L.cppf_axis_sign(pc_d.data_ptr(), n_d.data_ptr(), idx_d.data_ptr(),
                                sel_d.data_ptr(), nsel.data_ptr(), P, hd.data_ptr() + 12, 8, bd_d.data_ptr(),
                                out.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr(dev)), "axis_sign")
    _lib.check(L.cppf_scale_sum(hd.data_ptr() + 16, 8, sel_d.data_ptr(), nsel.data_ptr(), P, out4.data_ptr(),
                                ws.data_ptr(), ws.numel(), stream_ptr(dev)), "scale_sum")
    o, o4 = out.cpu().numpy(), out4.cpu().numpy()
    assert o[2] == 2500 and o4[3] == 2500
    np.testing.assert_allclose([o[0] / 2500, o[1] / 2500], [up, down], rtol=1e-12)
    np.testing.assert_allclose(o4[:3], heads[sel, 4:7].astype(np.float64).sum(0), rtol=1e-12)
    sc = oracle.scale(heads[sel, 4:7], [0.06, 0.05, 0.045])
    mean = (o4[:3] / 2500).astype(np.float32)
    np.testing.assert_allclose(np.exp(mean).astype(np.float64) * np.array([0.06, 0.05, 0.045]) * 2, sc, rtol=1e-6)


# ------------------------------------------------------------------------------------ the pose tail's fused launches
@pytest.mark.parametrize("n_pairs_per_point,known_answer", [(20, True), (3, True), (40, False)])
def test_backvote_count_and_scatter_match_oracle(oracle, dev, n_pairs_per_point, known_answer):
    """cppf_pose_tail_begin + cppf_backvote_count + cppf_compact_scatter = center_from_argmax + backvote + count/scan/scatter"""
    ob = syn.make_object("bottle", 1024, 9)
    cfg = ob["cfg"]
    idx = syn.make_pairs(1024, n_pairs_per_point, 9)
    P = idx.shape[0]
    outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg)
    if not known_answer:                          # few survivors, many chunks without any
        rng = np.random.default_rng(5)
        keep = rng.random(P) < 0.02
        outputs = np.where(keep[:, None], outputs, rng.uniform(0.0, 0.25, (P, 2)).astype(np.float32)).astype(np.float32)
    corner, dims = oracle.grid_setup(ob["pc"], cfg.res)
    idx32 = idx.astype(np.int32)
    # the arg-max cell nearest the true centre: the begin launch turns it into T (fp64) and its float copy
    cand = np.round((ob["center"] - corner.astype(np.float64)) / cfg.res).astype(np.int64)
    flat = int((cand[0] * dims[1] + cand[1]) * dims[2] + cand[2])
    T = corner.astype(np.float64) + cand * float(cfg.res)
    center = T.astype(np.float32)
    oo, mask = oracle.backvote(ob["pc"], outputs, idx32, corner, cfg.res, 72, dims, center, np.float32(3 * cfg.res))
    L = _lib.lib()
    n_chunks = (P + 1023) // 1024
    zero = torch.full((16 + 4 * ((n_chunks + 3) // 4 * 4) + 32,), 0xAB, dtype=torch.uint8, device=dev)   # garbage on entry
    T64 = zero[:32].view(torch.float64)                     # inside the zeroed region, like the pipeline's record
    chunk_counts = zero[32:32 + 4 * n_chunks].view(torch.int32)
    T32 = torch.empty(3, dtype=torch.float32, device=dev)
    oi = torch.tensor([flat], dtype=torch.int64, device=dev)
    pk = torch.tensor([7.5], dtype=torch.float32, device=dev)
    ip = torch.empty(2, dtype=torch.float64, device=dev)
    pc_d, out_d, idx_d, cor_d = t(ob["pc"], dev), t(outputs, dev), t(idx32, dev), t(corner, dev)
    _lib.check(L.cppf_pose_tail_begin(oi.data_ptr(), cor_d.data_ptr(), float(cfg.res), int(dims[1]), int(dims[2]), None,
                                      T64.data_ptr(), T32.data_ptr(), pk.data_ptr(), ip.data_ptr(), zero.data_ptr(),
                                      zero.numel(), stream_ptr(dev)), "begin")
    m = torch.empty(P, dtype=torch.uint8, device=dev)
    surv = torch.full((P,), -1, dtype=torch.int32, device=dev)
    cnt = torch.full((1,), -5, dtype=torch.int32, device=dev)
    _lib.check(L.cppf_backvote_count(pc_d.data_ptr(), out_d.data_ptr(), idx_d.data_ptr(), cor_d.data_ptr(), cfg.res, P, 72,
                                     int(dims[0]), int(dims[1]), int(dims[2]), None, T32.data_ptr(),
                                     float(np.float32(3 * cfg.res)), m.data_ptr(), chunk_counts.data_ptr(), None,
                                     stream_ptr(dev)), "backvote_count")
    _lib.check(L.cppf_compact_scatter(m.data_ptr(), P, chunk_counts.data_ptr(), surv.data_ptr(), cnt.data_ptr(),
                                      stream_ptr(dev)), "compact_scatter")
    torch.cuda.synchronize()
    # the same from the int64 pair list, leaving the int32 copy behind
    m64 = torch.empty(P, dtype=torch.uint8, device=dev)
    cc64 = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
    idx64_d, idx32_out = t(idx.astype(np.int64), dev), torch.full((P, 2), -1, dtype=torch.int32, device=dev)
    _lib.check(L.cppf_backvote_count64(pc_d.data_ptr(), out_d.data_ptr(), idx64_d.data_ptr(), idx32_out.data_ptr(),
                                       cor_d.data_ptr(), cfg.res, P, 72, int(dims[0]), int(dims[1]), int(dims[2]), None,
                                       T32.data_ptr(), float(np.float32(3 * cfg.res)), m64.data_ptr(), cc64.data_ptr(), None,
                                       stream_ptr(dev)), "backvote_count64")
    torch.cuda.synchronize()
    assert torch.equal(m64, m) and torch.equal(cc64, chunk_counts)
    np.testing.assert_array_equal(idx32_out.cpu().numpy(), idx32)
    np.testing.assert_array_equal(T64.cpu().numpy()[:3], T)
    np.testing.assert_array_equal(T32.cpu().numpy(), center)
    np.testing.assert_array_equal(ip.cpu().numpy(), [flat, 7.5])
    assert not zero[24:32].any().item() and not zero[32 + 4 * n_chunks:].any().item()
    np.testing.assert_array_equal(m.cpu().numpy().astype(bool), mask)
    pad = np.zeros(n_chunks * 1024, bool)
    pad[:P] = mask
    np.testing.assert_array_equal(chunk_counts.cpu().numpy(), pad.reshape(n_chunks, 1024).sum(1))
    ref = np.nonzero(mask)[0]
    assert int(cnt.item()) == ref.size and (ref.size > 0)
    np.testing.assert_array_equal(surv.cpu().numpy()[:ref.size], ref)


@pytest.mark.parametrize("n", [1, 1023, 1024, 1025, 70000, 1200000])
def test_compact_scatter_sizes(dev, n):
    L = _lib.lib()
    rng = np.random.default_rng(n)
    mask = (rng.random(n) < (0.37 if n != 70000 else 0.001)).astype(np.uint8)
    n_chunks = (n + 1023) // 1024
    pad = np.zeros(n_chunks * 1024, np.int32)
    pad[:n] = mask
    m, cc = t(mask, dev), t(pad.reshape(n_chunks, 1024).sum(1).astype(np.int32), dev)
    surv = torch.empty(n, dtype=torch.int32, device=dev)
    cnt = torch.full((1,), -7, dtype=torch.int32, device=dev)
    _lib.check(L.cppf_compact_scatter(m.data_ptr(), n, cc.data_ptr(), surv.data_ptr(), cnt.data_ptr(), stream_ptr(dev)), "scatter")
    ref = np.nonzero(mask)[0]
    assert int(cnt.item()) == ref.size
    np.testing.assert_array_equal(surv.cpu().numpy()[:ref.size], ref)
    assert L.cppf_compact_scatter(m.data_ptr(), 8192 * 1024 + 1, cc.data_ptr(), surv.data_ptr(), cnt.data_ptr(),
                                  stream_ptr(dev)) == -3      # CPPF_EUNSUPPORTED: the three-launch form serves it


@pytest.mark.parametrize("n_dirs", [1, 2])
def test_sphere_count_dirs_and_pose_sums_match_oracle(oracle, golden, dev, n_dirs):
    ob = syn.make_object("mug", 600, 21)
    idx = syn.make_pairs(600, 10, 21)
    P = idx.shape[0]
    idx32 = idx.astype(np.int32)
    rng = np.random.default_rng(21)
    heads = rng.normal(0, 1.5, (P, 8)).astype(np.float32)
    heads[:, :2] = (rng.integers(0, 36, (P, 2)) / 35 * np.pi).astype(np.float32)
    sel = np.sort(rng.choice(P, 2500, replace=False)).astype(np.int32)
    sph = golden("sphere.npz")["pts"]
    L = _lib.lib()
    thr = float(np.float32(np.cos(1.5 / 180 * np.pi)))
    counts = torch.zeros((2, 480), dtype=torch.int32, device=dev)
    nsel = torch.tensor([2500], dtype=torch.int32, device=dev)
    hd = t(heads, dev)
    pc_d, n_d, idx_d, sel_d = t(ob["pc"], dev), t(ob["normals"], dev), t(idx32, dev), t(sel, dev)
    s32_d, s64_d = t(sph.astype(np.float32), dev), t(sph, dev)
    _lib.check(L.cppf_rot_sphere_count_dirs(pc_d.data_ptr(), hd.data_ptr(), 8, 1, n_dirs, idx_d.data_ptr(), sel_d.data_ptr(),
                                            nsel.data_ptr(), P, 2000, 72, s32_d.data_ptr(), 480, thr, 1, counts.data_ptr(),
                                            480, stream_ptr(dev)), "rot_sphere_count_dirs")
    best_idx = torch.full((2,), -1, dtype=torch.int64, device=dev)
    best_dir = torch.full((2, 3), 9.0, dtype=torch.float64, device=dev)
    sign = torch.full((2, 3), 9.0, dtype=torch.float64, device=dev)
    scale = torch.full((4,), 9.0, dtype=torch.float64, device=dev)
    ws = workspace(L.cppf_pose_sums_workspace_bytes(), dev, "pose_sums")
    ticket = torch.zeros(4, dtype=torch.int32, device=dev)
    for _ in range(2):                                                     # the ticket is left ready for the next call
        _lib.check(L.cppf_pose_sums(pc_d.data_ptr(), n_d.data_ptr(), idx_d.data_ptr(), sel_d.data_ptr(), nsel.data_ptr(), P,
                                    hd.data_ptr() + 8, 8, n_dirs, counts.data_ptr(), 480, 480, s64_d.data_ptr(),
                                    hd.data_ptr() + 16, 8, best_idx.data_ptr(), best_dir.data_ptr(), sign.data_ptr(),
                                    scale.data_ptr(), ws.data_ptr(), ws.numel(), ticket.data_ptr(), stream_ptr(dev)),
                   "pose_sums")
    torch.cuda.synchronize()
    assert int(ticket[0].item()) == 0
    cn, bi, bd, sg, sc = (x.cpu().numpy() for x in (counts, best_idx, best_dir, sign, scale))
    for j in range(n_dirs):
        co = oracle.rot_voting(ob["pc"], heads[sel[:2000], j], idx32[sel[:2000]], 72)
        counts_o = oracle.sphere_count(co, sph, 1.5)
        np.testing.assert_array_equal(cn[j].astype(np.int64), counts_o)
        assert bi[j] == int(np.argmax(counts_o))                           # first maximum
        np.testing.assert_array_equal(bd[j], sph[bi[j]])
        _, (up, down) = oracle.axis_sign(ob["pc"], ob["normals"], idx32[sel], heads[sel, 2 + j], sph[bi[j]])
        assert sg[j, 2] == 2500
        np.testing.assert_allclose([sg[j, 0] / 2500, sg[j, 1] / 2500], [up, down], rtol=1e-12)
    if n_dirs == 1:                                                        # the second direction's outputs are not touched
        assert not cn[1].any() and bi[1] == -1 and (bd[1] == 9.0).all() and (sg[1] == 9.0).all()
    assert sc[3] == 2500
    np.testing.assert_allclose(sc[:3], heads[sel, 4:7].astype(np.float64).sum(0), rtol=1e-12)


# ------------------------------------------------------------------------------------ end to end
@pytest.mark.parametrize("cat", ["bottle", "camera"])
def test_estimate_pose_matches_oracle_chain(oracle, golden, dev, cat):
    from cppf_amd.inference import estimate_pose
    ob = syn.make_object(cat, 1024, 31)
    cfg = ob["cfg"]
    idx = syn.make_pairs(1024, 32, 31)
    P = idx.shape[0]
    u_tr, u_rot = syn.make_uniforms(P, 31)
    sd = seeded_sd(0)
    for k in ("final.weight", "final.bias"):
        sd[k] = sd[k] * 4
    enc = make_encoder(sd, [84, 32, 32, 16], 141, dev)
    sph = golden("sphere.npz")["pts"]
    with torch.no_grad():
        r = estimate_pose(enc, t(ob["pc"], dev), t(ob["normals"], dev), t(ob["feat"], dev), t(idx, dev), t(u_tr, dev),
                          t(u_rot, dev), cfg, sph, pc_host=ob["pc"])
    ocfg = dict(res=cfg.res, tr_num_bins=32, rot_num_bins=36, vote_range=cfg.vote_range, scale_mean=cfg.scale_mean,
                regress_right=cfg.regress_right, ppffcs=[84, 32, 32, 16], out_dim=141)
    o = oracle.estimate_pose(ob["pc"], ob["normals"], ob["feat"], idx, sd, ocfg, u_tr, u_rot, sph)
    np.testing.assert_array_equal(r["outputs"].cpu().numpy(), o["outputs"])
    np.testing.assert_array_equal(r["heads"].cpu().numpy()[o["mask"]], o["heads"][o["mask"]])   # second pass: survivors' rows
    assert r["argmax"] == o["argmax"]                                   # bit-exact vote-grid arg-max
    np.testing.assert_allclose(r["T"], o["T"], rtol=0, atol=1e-12)
    assert r["n_surv"] == int(o["mask"].sum())
    np.testing.assert_array_equal(r["ws"].mask.cpu().numpy().astype(bool), o["mask"])
    np.testing.assert_allclose(r["up"], o["up"], atol=1e-12)
    if cfg.regress_right:
        assert r["best_right"] is not None
    np.testing.assert_allclose(r["scale"], o["scale"], rtol=1e-6)       # <= 1e-4 bar of the north star
    np.testing.assert_allclose(r["peak"], o["peak"], rtol=2e-5)


def test_pose_pipeline_split_and_full_first_forms_agree(oracle, golden, dev):
    """PosePipeline's two captured forms (centre heads + second pass on the survivors | every head in the first pass) give the
    same pose record and the same heads rows for the survivors; the form follows the survivors' share of the last instance"""
    from cppf_amd.inference import PosePipeline, grid_shape
    ob = syn.make_object("camera", 1024, 5)
    cfg = ob["cfg"]
    idx = syn.make_pairs(1024, 24, 5)
    P = idx.shape[0]
    u_tr, u_rot = syn.make_uniforms(P, 5)
    sd = seeded_sd(0)
    for k in ("final.weight", "final.bias"):
        sd[k] = sd[k] * 4
    enc = make_encoder(sd, [84, 32, 32, 16], 141, dev)
    sph = golden("sphere.npz")["pts"]
    corners, dims = grid_shape(ob["pc"], cfg.res)
    pp = PosePipeline(enc, cfg, 1024, P, dims, dev, sph, 72)
    pp.load(ob["pc"], ob["normals"], ob["feat"], idx, u_tr, u_rot, corners[0].copy())
    recs, heads = [], []
    for form in (False, True, False, True):
        pp.adapt(P if form else 0)                       # force the form
        assert pp.full_first == form
        r = pp.run()
        mask = r["ws"].mask.cpu().numpy().astype(bool)
        recs.append(r["ws"].rec.cpu().numpy().copy())
        heads.append(r["heads"].cpu().numpy()[mask].copy())
        assert r["n_surv"] == mask.sum() > 0
    for k in (1, 2, 3):
        np.testing.assert_array_equal(recs[k], recs[0])
        np.testing.assert_array_equal(heads[k], heads[0])
    # the form chosen after a run follows the share of survivors (hysteresis 0.10 / 0.25)
    share = r["n_surv"] / P
    assert pp.full_first == (share >= 0.10)                                # (the last forced form was full-first)
    ocfg = dict(res=cfg.res, tr_num_bins=32, rot_num_bins=36, vote_range=cfg.vote_range, scale_mean=cfg.scale_mean,
                regress_right=cfg.regress_right, ppffcs=[84, 32, 32, 16], out_dim=141)
    o = oracle.estimate_pose(ob["pc"], ob["normals"], ob["feat"], idx, sd, ocfg, u_tr, u_rot, sph)
    assert r["argmax"] == o["argmax"] and r["n_surv"] == int(o["mask"].sum())
    np.testing.assert_array_equal(heads[0], o["heads"][o["mask"]])
    np.testing.assert_allclose(r["up"], o["up"], atol=1e-12)


# ------------------------------------------------------------------------------------ full-size properties
def test_full_size_properties_c2(dev):
    """N=4096, K=128 (BASELINE.json config 2): size-independent properties, no oracle."""
    ob = syn.make_object("bottle", 4096, 0)
    cfg = ob["cfg"]
    idx = syn.make_pairs(4096, 128, 0)
    P = idx.shape[0]
    outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
    from cppf_amd.inference import grid_shape
    corners, dims = grid_shape(ob["pc"], cfg.res)
    corner = corners[0]
    idx32 = idx.astype(np.int32)
    g1, flat1, peak1 = run_vote(dev, ob["pc"], outputs, idx32, corner, dims, cfg.res, 72, True)
    # known answer: the arg-max is the cell holding the true centre (utils/dataset.py:27-36)
    cell = np.array(np.unravel_index(flat1, dims))
    assert np.all(np.abs(cell - (ob["center"] - corner) / cfg.res) <= 1.0)
    # permutation invariance of the pairs (same multiset of votes)
    perm = np.random.default_rng(0).permutation(P)
    g2, flat2, _ = run_vote(dev, ob["pc"], outputs[perm], idx32[perm], corner, dims, cfg.res, 72, True)
    assert flat2 == flat1
    np.testing.assert_allclose(g2, g1, rtol=1e-4, atol=1e-4 * g1.max())
    # linearity: voting twice into the same grid doubles it
    g3, flat3, _ = run_vote(dev, ob["pc"], outputs, idx32, corner, dims, cfg.res, 72, True, grid0=g1)
    np.testing.assert_allclose(g3, 2 * g1, rtol=1e-4, atol=1e-4 * g1.max())
    assert flat3 == flat1
    # probs scale the grid: probs = 0.5 halves every vote exactly
    gh, flath, _ = run_vote(dev, ob["pc"], outputs, idx32, corner, dims, cfg.res, 72, True,
                            probs=np.full(4096, 0.5, np.float32))
    np.testing.assert_allclose(gh, 0.5 * g1, rtol=1e-4, atol=1e-4 * g1.max())
    # checksum: total mass is an integer count of in-grid votes, <= 72 per pair
    mass = g1.sum(dtype=np.float64)
    assert abs(mass - round(mass)) < 1e-3 * mass ** 0.5 + 0.5 and mass <= 72.0 * P
    # non-adaptive deposits at least as much mass
    g4, _, _ = run_vote(dev, ob["pc"], outputs, idx32, corner, dims, cfg.res, 72, False)
    assert g4.sum(dtype=np.float64) >= mass * 0.999


def test_large_config_c5_properties(oracle, dev):
    """BASELINE.json config 5 sizes: N=8192, K=256 (P = 2 097 152 pairs), fine grid res 2e-3
    (~400 k cells -> 14 LDS tiles x 146 pair chunks scheduled over the CUs, 24-bit fixed point)."""
    ob = syn.make_object("bottle", 8192, 3)
    cfg = ob["cfg"]
    res = 2e-3
    idx = syn.make_pairs(8192, 256, 3)
    P = idx.shape[0]
    outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
    from cppf_amd.inference import grid_shape
    corners, dims = grid_shape(ob["pc"], res)
    assert int(np.prod(dims)) > 300000
    bits = _lib.lib().cppf_vote_fixed_point_bits(P, 72, *dims)
    assert 16 <= bits <= 24
    idx32 = idx.astype(np.int32)
    gg, flat, peak = run_vote(dev, ob["pc"], outputs, idx32, corners[0], dims, res, 72, True)
    cell = np.array(np.unravel_index(flat, dims))
    assert np.all(np.abs(cell - (ob["center"] - corners[0]) / res) <= 1.5)
    assert flat == int(np.argmax(gg)) and peak == gg.max()
    # exact-sum check on a 131 072-pair prefix (keeps the CPU side to a couple of seconds)
    n = 131072
    g2, f2, _ = run_vote(dev, ob["pc"], outputs[:n], idx32[:n], corners[0], dims, res, 72, True)
    check_grid(oracle, g2, ob["pc"], outputs[:n], idx32[:n], corners[0], dims, res, 72, True)
    mass = gg.sum(dtype=np.float64)
    assert abs(mass - round(mass)) < 1.0 and mass <= 72.0 * P


def test_pipelines_graph_replay_equals_eager(golden, dev):
    """CenterPipeline / PosePipeline (static buffers + hipGraph replay) give the eager results, also after
    new data is loaded into the same graph."""
    from cppf_amd.inference import CenterPipeline, PosePipeline, estimate_pose, grid_shape
    sph = golden("sphere.npz")["pts"]
    sd = seeded_sd(0)
    for k in ("final.weight", "final.bias"):
        sd[k] = sd[k] * 4
    enc = make_encoder(sd, [84, 32, 32, 16], 141, dev)
    cfg = CATEGORIES["camera"]
    pipe = None
    for seed in (41, 42):
        ob = syn.make_object("camera", 1024, seed)
        idx = syn.make_pairs(1024, 32, seed)
        P = idx.shape[0]
        u_tr, u_rot = syn.make_uniforms(P, seed)
        corners, dims = grid_shape(ob["pc"], cfg.res)
        with torch.no_grad():
            ref = estimate_pose(enc, t(ob["pc"], dev), t(ob["normals"], dev), t(ob["feat"], dev), t(idx, dev),
                                t(u_tr, dev), t(u_rot, dev), cfg, sph, pc_host=ob["pc"])
        if pipe is None or pipe.dims != tuple(dims):
            pipe = PosePipeline(enc, cfg, 1024, P, dims, dev, sph)
        pipe.load(ob["pc"], ob["normals"], ob["feat"], idx, u_tr, u_rot, corners[0].copy())
        for rep in range(2):                      # first call captures, second replays
            r = pipe.run()
            assert r["argmax"] == ref["argmax"] and r["n_surv"] == ref["n_surv"]
            np.testing.assert_array_equal(r["T"], ref["T"])
            np.testing.assert_array_equal(r["up"], ref["up"])
            np.testing.assert_allclose(r["scale"], ref["scale"], rtol=1e-12)
            sv = ref["ws"].mask.bool()
            assert torch.equal(r["outputs"], ref["outputs"]) and torch.equal(r["heads"][sv], ref["heads"][sv])
        cp = CenterPipeline(enc, cfg, 1024, P, dims, dev, use_graph=False)
        cp.load(ob["pc"], ob["normals"], ob["feat"], idx, u_tr, u_rot, corners[0].copy())
        oi, ov = cp.run()
        assert int(oi.item()) == ref["argmax"]


def test_batch_of_mixed_categories_matches_per_object_oracle(oracle, golden, dev):
    """BASELINE.json config 4 in miniature: mixed NOCS categories through BatchPoseRunner (one cached
    hipGraph pipeline per shape, records in object order); every object is checked against the oracle chain."""
    from cppf_amd.batch import BatchPoseRunner
    from cppf_amd.config import NOCS_CATEGORIES
    sd = seeded_sd(0)
    for k in ("final.weight", "final.bias"):
        sd[k] = sd[k] * 4
    encoders = {c: make_encoder(sd, [84, 32, 32, 16], 141, dev) for c in NOCS_CATEGORIES}
    objects = []
    for j in range(8):
        cat = NOCS_CATEGORIES[j % 6]
        ob = syn.make_object(cat, 768, 100 + j)
        idx = syn.make_pairs(768, 24, 100 + j)
        u_tr, u_rot = syn.make_uniforms(idx.shape[0], 100 + j)
        objects.append(dict(pc=ob["pc"], normals=ob["normals"], feat=ob["feat"], point_idxs=idx, u_tr=u_tr, u_rot=u_rot,
                            cfg=ob["cfg"]))
    runner = BatchPoseRunner(encoders, dev)
    recs = runner.run(objects).cpu().numpy()
    recs2 = runner.run(objects).cpu().numpy()               # second pass replays the cached graphs
    np.testing.assert_array_equal(recs, recs2)
    assert recs.shape == (8, 20) and recs[:, 15].tolist() == list(range(8))
    sph = golden("sphere.npz")["pts"]
    for j, obj in enumerate(objects):
        cfg = obj["cfg"]
        ocfg = dict(res=cfg.res, tr_num_bins=32, rot_num_bins=36, vote_range=cfg.vote_range, scale_mean=cfg.scale_mean,
                    regress_right=cfg.regress_right, ppffcs=[84, 32, 32, 16], out_dim=141)
        o = oracle.estimate_pose(obj["pc"], obj["normals"], obj["feat"], obj["point_idxs"], sd, ocfg, obj["u_tr"],
                                 obj["u_rot"], sph)
        assert int(recs[j, 12]) == o["argmax"] and int(recs[j, 14]) == int(o["mask"].sum())
        np.testing.assert_allclose(recs[j, 0:3], o["T"], atol=1e-12)
        np.testing.assert_allclose(recs[j, 3:6], o["up"], atol=1e-12)
        np.testing.assert_allclose(recs[j, 9:12], o["scale"], rtol=1e-6)


def test_pair_sharded_center_equals_unsharded_bit_for_bit(oracle, dev):
    """sharding.estimate_center_sharded (pairs split across ranks, integer vote images all-reduced): with a single rank it is the
    plain chain, and the slices of two emulated ranks (integer images summed on one device) give the SAME grid and arg-max"""
    from cppf_amd import sharding
    from cppf_amd.inference import estimate_center, grid_shape
    from cppf_amd.models import voting
    ob = syn.make_object("mug", 1024, 2)
    cfg = ob["cfg"]
    torch.manual_seed(2)
    enc = PPFEncoder(cfg.ppffcs, cfg.out_dim).eval().to(dev)
    idx = torch.from_numpy(syn.make_pairs(1024, 32, 2)).to(dev)
    u_tr = torch.from_numpy(syn.make_uniforms(idx.shape[0], 2)[0]).to(dev)
    corners, dims = grid_shape(ob["pc"], cfg.res)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    pc, nrm, feat, corner = d(ob["pc"]), d(ob["normals"]), d(ob["feat"]), d(corners[0])
    with torch.no_grad():
        i0, v0, outputs, _, g0 = estimate_center(enc, pc, nrm, feat, idx, u_tr, cfg, corner, dims)
        g0 = g0.clone()
        i1, v1, g1 = sharding.estimate_center_sharded(enc, pc, nrm, feat, idx, u_tr, cfg, corner, dims, 0, 1)
        assert torch.equal(g0, g1) and int(i0) == int(i1)
        bits = voting.vote_fixed_point_bits(idx.shape[0], 72, dims)
        total = torch.zeros(dims, dtype=torch.int64, device=dev)
        q = torch.zeros(1, dtype=torch.float32, device=dev)
        for r in range(2):
            lo, hi = sharding.shard_pairs(idx.shape[0], r, 2)
            voting.vote_grid_raw(pc, outputs[lo:hi].contiguous(), None, idx[lo:hi].contiguous(), total, q, corner, cfg.res, 72, True,
                                 fixed_bits=bits, accumulate=True)
        gs, gi, gv = voting.grid_from_raw(total, q)
        assert torch.equal(gs, g0) and int(gi) == int(i0) and float(gv) == float(v0)


def test_randomised_soak(dev):
    """a few seconds of tests/soak_gpu.py: random categories, sizes, grid resolutions, output regimes, rotation
    counts, weights -- vote grids, arg-max, back-vote offsets and kNN sets against the oracle"""
    import soak_gpu
    n_v, n_b, n_k, n_i = soak_gpu.run(10.0, 12345, dev)
    assert n_v >= 15 and n_b >= 15 and n_k >= 15 and n_i >= 10


def test_batch_runner_device_sampled_inputs(golden, dev):
    """objects without a pair list: pairs and bin uniforms are drawn on the device from (seed, object index) -- the same
    seed reproduces the batch bit for bit, another seed does not, and every instance still produces a finite pose"""
    from cppf_amd.batch import BatchPoseRunner
    from cppf_amd.models.model import PPFEncoder
    cats = ["bottle", "mug", "bowl"]
    encs = {}
    for i, c in enumerate(cats):
        cfg = syn.make_object(c, 64, 0)["cfg"]
        torch.manual_seed(i)
        encs[c] = PPFEncoder(cfg.ppffcs, cfg.out_dim).eval().to(dev)
    objs = []
    for j in range(5):
        ob = syn.make_object(cats[j % 3], 512, j)
        objs.append(dict(pc=ob["pc"], normals=ob["normals"], feat=ob["feat"], cfg=ob["cfg"], n_pairs=512 * 24))
    runner = BatchPoseRunner(encs, dev)
    a = runner.run(objs, seed=5).cpu().numpy()
    b = runner.run(objs, seed=5).cpu().numpy()
    c = runner.run(objs, seed=6).cpu().numpy()
    assert a.shape[0] == 5 and np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.all(np.isfinite(a)) and np.array_equal(a[:, 15], np.arange(5))


def test_batch_runner_lanes_do_not_share_scratch(golden, dev):
    """instances of ONE shape but different data, two in flight on two streams: every record equals the one the same
    instance produces alone (captured pipelines own their scratch: the per-point table, the vote's partial tiles and the
    reduction buffers must not be shared between the two lanes)"""
    from cppf_amd.batch import BatchPoseRunner
    from cppf_amd.models.model import PPFEncoder
    cfg = syn.make_object("bottle", 64, 0)["cfg"]
    torch.manual_seed(0)
    encs = {"bottle": PPFEncoder(cfg.ppffcs, cfg.out_dim).eval().to(dev)}
    objs = []
    for j in range(8):
        ob = syn.make_object("bottle", 2048, 50 + j)
        ob["pc"] = (ob["pc"] - ob["pc"].min(0) + np.float32(0.01 * j)).astype(np.float32)      # same extents -> same grid dims
        idx = syn.make_pairs(2048, 64, 50 + j)
        u1, u2 = syn.make_uniforms(idx.shape[0], 50 + j)
        objs.append(dict(pc=ob["pc"], normals=ob["normals"], feat=ob["feat"], point_idxs=idx, u_tr=u1, u_rot=u2, cfg=ob["cfg"]))
    runner = BatchPoseRunner(encs, dev)
    alone = []
    for o in objs:
        r = runner.run_object(o)
        alone.append(np.concatenate([r["T"], r["up"], r["scale"], [r["argmax"], r["n_surv"]]]))
    for rep in range(3):
        recs = runner.run(objs).cpu().numpy()
        for j, a in enumerate(alone):
            got = np.concatenate([recs[j, 0:3], recs[j, 3:6], recs[j, 9:12], recs[j, 12:13], recs[j, 14:15]])
            assert np.array_equal(got, a), (rep, j, got, a)


@pytest.mark.parametrize("with_heads", [False, True])
def test_batched_decode_equals_per_list_calls(dev, with_heads):
    """cppf_pair_mlp_decode_batch: several pair lists -- own clouds, sizes, index widths, two different networks -- in ONE launch give
    the bits of one cppf_pair_mlp_decode call each; argument errors are reported"""
    from cppf_amd.models.model import forward_decode_batch
    encs = [make_encoder(seeded_sd(s), [84, 32, 32, 16], 141, dev) for s in (0, 5)]
    items, want = [], []
    for j, (n, k, seed) in enumerate([(700, 9, 1), (2048, 40, 2), (64, 1, 3), (1500, 33, 4), (300, 5, 5)]):
        ob = syn.make_object(["bottle", "mug", "can", "laptop", "bowl"][j], n, seed)
        idx = syn.make_pairs(n, k, seed)
        if j % 2:
            idx = idx.astype(np.int32)
        u_tr, u_rot = syn.make_uniforms(idx.shape[0], seed)
        u_tr[:3] = -1.0
        enc = encs[j % 2]
        it = dict(encoder=enc, pc=t(ob["pc"], dev), pc_normal=t(ob["normals"], dev), feat=t(ob["feat"], dev), idxs=t(idx, dev),
                  u_tr=t(u_tr, dev), vote_range=ob["cfg"].vote_range)
        if with_heads:
            it["u_rot"] = t(u_rot, dev)
        items.append(it)
        with torch.no_grad():
            want.append(enc.forward_decode(it["pc"], it["pc_normal"], it["feat"], it["idxs"], it["u_tr"], it["vote_range"],
                                           it.get("u_rot")))
    for lo, hi in ((0, 5), (0, 1), (1, 3)):
        with torch.no_grad():
            got = forward_decode_batch(items[lo:hi])
        for j, (o, h) in enumerate(got):
            wo, wh = want[lo + j]
            assert torch.equal(o, wo) and (h is None) == (wh is None) and (h is None or torch.equal(h, wh))
    mixed = [dict(items[0]), dict(items[1])]
    mixed[1]["u_rot"] = None if with_heads else t(syn.make_uniforms(items[1]["idxs"].shape[0], 9)[1], dev)
    with pytest.raises(_lib.CppfError):
        forward_decode_batch(mixed)
    with pytest.raises(ValueError):
        forward_decode_batch(items + items)


def test_center_batch_pipeline_equals_member_pipelines(golden, dev):
    """CenterBatchPipeline: the members' pair lists in one launch of the pair kernel, then each member's vote -- the same (mu, nu),
    grid and arg-max as every member run on its own; replay after new data; argument checks"""
    from cppf_amd.inference import CenterBatchPipeline, CenterPipeline, grid_shape
    enc = make_encoder(seeded_sd(3), [84, 32, 32, 16], 141, dev)
    pipes, data = [], []
    for j, (cat, n, k) in enumerate([("bottle", 1200, 24), ("mug", 800, 40), ("bowl", 1500, 16)]):
        ob = syn.make_object(cat, n, 30 + j)
        idx = syn.make_pairs(n, k, 30 + j)
        u_tr, u_rot = syn.make_uniforms(idx.shape[0], 30 + j)
        corners, dims = grid_shape(ob["pc"], ob["cfg"].res)
        p = CenterPipeline(enc, ob["cfg"], n, idx.shape[0], dims, dev, 72, adaptive=True, with_heads=False,
                           vote_workgroups=128 if j == 1 else 0)
        p.load(ob["pc"], ob["normals"], ob["feat"], idx, u_tr, u_rot, corners[0].copy())
        pipes.append(p)
        data.append((ob, idx, u_tr, u_rot, corners))
    want = []
    for p in pipes:
        p.run()
        want.append((p.outputs.clone(), p.grid.clone(), int(p.out_idx), float(p.out_val)))
    bp = CenterBatchPipeline(pipes)
    for rep in range(3):
        res = bp.run()
        torch.cuda.synchronize()
        for p, (o, g, i, v), (oi, ov) in zip(pipes, want, res):
            assert torch.equal(p.outputs, o) and torch.equal(p.grid, g) and int(oi) == i and float(ov) == v
    # new uniforms for one member: the captured chain reads the static buffers (member.outputs is the tensor of whichever chain
    # was captured last -- here the batch's)
    ob, idx, u_tr, u_rot, corners = data[0]
    u2, _ = syn.make_uniforms(idx.shape[0], 777)
    pipes[0].load(None, None, None, None, u2, None, None)
    bp.run()
    torch.cuda.synchronize()
    fresh = CenterPipeline(enc, ob["cfg"], ob["pc"].shape[0], idx.shape[0], pipes[0].dims, dev, 72, adaptive=True, with_heads=False)
    fresh.load(ob["pc"], ob["normals"], ob["feat"], idx, u2, u_rot, corners[0].copy())
    fresh.run()
    torch.cuda.synchronize()
    assert torch.equal(pipes[0].outputs, fresh.outputs) and torch.equal(pipes[0].grid, fresh.grid)
    assert int(pipes[0].out_idx) == int(fresh.out_idx) and not torch.equal(fresh.outputs, want[0][0])
    with pytest.raises(ValueError):
        CenterBatchPipeline([])


def test_plain_c_host_votes_on_the_device_and_matches_the_oracle(c_host, dev):
    """the drop-in boundary from a host that is not Python: tests/c_host/vote_host.c allocates with hipMalloc, zeroes the vote's
    workspace once, calls cppf_vote_argmax three times on one stream and compares grid, arg-max, peak and centre with
    orc_ppf_voting / orc_grid_argmax / orc_center_from_argmax (models/voting.py:8-66, nocs/inference.py:207-210)"""
    import subprocess
    p = subprocess.run([c_host, "gpu"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "device ok" in p.stdout, p.stdout


def test_plain_c_host_runs_the_whole_hot_path_and_matches_the_oracle_bit_for_bit(c_host, dev):
    """the same host, the whole path: cppf_pair_mlp_pack + cppf_pair_mlp_decode (PPF, MLP on the reference's architecture with
    seeded random weights, centre decode; int64 pair list) -> cppf_vote_argmax; every (mu, nu) equal to orc_pair_mlp(order 1) +
    orc_decode_center bit for bit, the arg-max the oracle's (models/model.py:117-137, nocs/inference.py:185-211)"""
    import subprocess
    p = subprocess.run([c_host, "chain"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "chain ok" in p.stdout and " 0 of " in p.stdout, p.stdout
    print(p.stdout)
