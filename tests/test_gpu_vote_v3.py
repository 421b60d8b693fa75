"""The round-3 tiled vote (csrc/vote.hip: v3_bin_kernel / v3_vote_kernel<FUSED> / v3_reduce_kernel): halo tiles cut along x, along y
and along both, the fused (< 4 tiles) and the binned (>= 4 tiles) forms on the same inputs, run-to-run bit reproducibility although
the tile queues fill in a racy order, state kept in the workspace between calls, and a workspace that was never initialised."""
import ctypes as C

import numpy as np
import pytest
import torch

import cppf_amd.synthetic as syn
from cppf_amd import _lib
from cppf_amd._torch_util import stream_ptr
from cppf_amd.models import voting
from test_gpu_parity import check_grid, run_vote, t

pytestmark = pytest.mark.gpu


def case(n=1500, k=40, seed=3, cat="bottle", quantise=True):
    ob = syn.make_object(cat, n, seed)
    idx = syn.make_pairs(n, k, seed)
    out = syn.closed_form_outputs(ob["pc"], ob["center"], idx, ob["cfg"], quantise=quantise)
    return ob, idx.astype(np.int32), out


@pytest.mark.parametrize("dims,res,expect_tiles", [
    ((76, 26, 26), 4e-3, 2),        # cut along x only (fused)
    ((26, 76, 26), 4e-3, 2),        # cut along y only (fused)
    ((60, 60, 52), 2.5e-3, None),   # cut along x and y: many tiles (binned)
    ((150, 30, 40), 3e-3, None),    # long in x
    ((20, 20, 20), 8e-3, 1),        # one tile, no halo
])
def test_halo_tiles_every_cut_direction(oracle, dev, dims, res, expect_tiles):
    """every cell -- also those next to a cut, which the owner tile's halo and the extra plane hand to the neighbour -- equals the
    exact fp64 vote sum within half a quantum per deposit; arg-max identical to the exact grid's"""
    L = _lib.lib()
    T = L.cppf_vote_tiles(*dims)
    assert T > 0 and (expect_tiles is None or T == expect_tiles)
    ob, idx, out = case()
    # put the object in the middle of the requested grid, and once more straddling the central cuts
    for shift in (0.0, 0.37):
        corner = (ob["center"] - 0.5 * np.array(dims) * res + shift * res * np.array([5, 7, 3])).astype(np.float32)
        gg, flat, peak = run_vote(dev, ob["pc"], out, idx, corner, dims, res, 72, True)
        g64, cnt = check_grid(oracle, gg, ob["pc"], out, idx, corner, dims, res, 72, True)
        assert flat == int(np.argmax(g64)) and g64.max() > 100
        assert peak == gg.max()
        # += into a pre-filled grid (the reference's semantics) on the same workspace
        g0 = np.random.default_rng(1).random(dims).astype(np.float32)
        gg2, flat2, _ = run_vote(dev, ob["pc"], out, idx, corner, dims, res, 72, True, grid0=g0)
        np.testing.assert_allclose(gg2, g0 + gg, rtol=1e-6, atol=1e-6)


def test_fused_and_binned_forms_give_the_same_bits(dev):
    """a 2-tile grid through the fused kernel (by-value launch and the few-tile *_dyn class) and through the binned kernels (the
    many-tile *_dyn class serves small grids too): the grids are the exact sum of the same quantised deposits -> bit-identical"""
    from cppf_amd.inference import grid_class, grid_shape
    ob, idx, out = case(n=3000, k=60, seed=5)
    res = ob["cfg"].res
    corners, dims = grid_shape(ob["pc"], res)
    Tn, many, cap = grid_class(dims)
    assert Tn == 2 and not many
    pc, o, i64, corner = t(ob["pc"], dev), t(out, dev), t(idx.astype(np.int64), dev), t(corners[0], dev)
    grid = torch.empty(dims, dtype=torch.float32, device=dev)
    i0, v0 = voting.vote_argmax(pc, o, None, i64, grid, corner, res, 72, True, accumulate=False)
    shape = torch.tensor([ob["pc"].shape[0], *dims], dtype=torch.int32, device=dev)
    G = int(np.prod(dims))
    for many_tiles in (False, True):
        flat = torch.zeros(64 * _lib.lib().cppf_vote_tile_cells() if many_tiles else cap, dtype=torch.float32, device=dev)
        oi, ov = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.float32, device=dev)
        voting.vote_argmax_dyn(pc, o, None, i64, flat, shape, corner, res, 72, True, oi, ov, many_tiles=many_tiles)
        assert torch.equal(flat[:G].view(dims), grid) and int(oi) == int(i0) and float(ov) == float(v0)


def test_many_tile_vote_is_bit_reproducible(dev):
    """the binned path's queues fill in whatever order the workgroups of the bin kernel reach their atomics; the grid must not care"""
    ob, idx, out = case(n=4096, k=64, seed=9)
    res = 2e-3
    from cppf_amd.inference import grid_shape
    corners, dims = grid_shape(ob["pc"], res)
    assert _lib.lib().cppf_vote_tiles(*dims) >= 4
    pc, o, i32, corner = t(ob["pc"], dev), t(out, dev), t(idx, dev), t(corners[0], dev)
    grids = []
    for rep in range(6):
        grid = torch.empty(dims, dtype=torch.float32, device=dev)
        oi, ov = voting.vote_argmax(pc, o, None, i32, grid, corner, res, 72, True, accumulate=False)
        grids.append((grid.clone(), int(oi), float(ov)))
    for g, i, v in grids[1:]:
        assert torch.equal(g, grids[0][0]) and i == grids[0][1] and v == grids[0][2]
    assert grids[0][0].sum().item() > 1e5


def test_uninitialised_workspace_is_reported_until_zeroed(oracle, dev):
    """include/cppf.h: the vote workspace must start zeroed.  One that never was (garbage header) must not be used silently:
    every call on it reports arg-max -1 / peak NaN until the caller zeroes cppf_vote_workspace_init_bytes() bytes"""
    L = _lib.lib()
    ob, idx, out = case(n=1200, k=30, seed=2)
    res = ob["cfg"].res
    from cppf_amd.inference import grid_shape
    corners, dims = grid_shape(ob["pc"], res)
    for dims_, res_ in ((dims, res), (grid_shape(ob["pc"], res / 2.2)[1], res / 2.2)):      # fused and binned
        corner = t(grid_shape(ob["pc"], res_)[0][0], dev)
        P = idx.shape[0]
        need = L.cppf_vote_workspace_bytes(P, 72, *dims_)
        ws = torch.randint(1, 255, (need,), dtype=torch.uint8, device=dev)            # garbage, no zero byte
        pc, o, i32 = t(ob["pc"], dev), t(out, dev), t(idx, dev)
        grid = torch.zeros(dims_, dtype=torch.float32, device=dev)
        oi, ov = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.float32, device=dev)

        def call():
            rc = L.cppf_vote_argmax(pc.data_ptr(), o.data_ptr(), None, i32.data_ptr(), 0, grid.data_ptr(), corner.data_ptr(),
                                    float(res_), pc.shape[0], P, 72, *[int(d) for d in dims_], 1, 0, oi.data_ptr(), ov.data_ptr(),
                                    ws.data_ptr(), ws.numel(), stream_ptr(dev))
            assert rc == 0
            torch.cuda.synchronize()
        for _ in range(3):     # EVERY call on the garbage block says so (round 3 stamped the header valid after the first report and
            oi.zero_()         # the next call silently added the garbage "extra plane" to the grid): the header stays poisoned
            ov.zero_()
            call()
            assert int(oi) == -1 and np.isnan(float(ov))
        # zeroing the header alone is NOT the contract (include/cppf.h, ABI 2): the whole state -- header + plane -- is
        # the caller follows the header's rule now (zero once) and everything works, repeatedly
        ws[:min(int(L.cppf_vote_workspace_init_bytes()), need)].zero_()
        for _ in range(2):
            call()
            gg = grid.cpu().numpy()
            g64, cnt = check_grid(oracle, gg, ob["pc"], out, idx, grid_shape(ob["pc"], res_)[0][0], dims_, res_, 72, True)
            assert int(oi) == int(np.argmax(g64))


@pytest.mark.parametrize("dims,res", [((26, 76, 26), 4e-3), ((60, 60, 52), 2.5e-3)])
def test_negative_probs_on_cut_grids(oracle, dev, dims, res):
    """negative / non-finite probs switch the launch to fp32 LDS atomics and fp32 partial tiles; the halo words then reach the
    neighbour's cells through the extra plane as floats -- on a fused (2-tile) and a binned (many-tile) grid"""
    from test_gpu_parity import oracle_vote
    ob, idx, out = case(n=900, k=20, seed=4)
    corner = (ob["center"] - 0.5 * np.array(dims) * res).astype(np.float32)
    rng = np.random.default_rng(3)
    for bad in (-0.75, np.inf):
        probs = rng.uniform(0.25, 2.0, ob["pc"].shape[0]).astype(np.float32)
        probs[rng.integers(0, probs.size, 40)] = bad
        go, na = oracle_vote(oracle, ob["pc"], out, idx, corner, dims, res, 72, True, probs)
        gg, flat, peak = run_vote(dev, ob["pc"], out, idx, corner, dims, res, 72, True, probs)
        fin = np.isfinite(go)
        assert np.array_equal(fin, np.isfinite(gg))
        np.testing.assert_allclose(gg[fin], go[fin], rtol=2e-4, atol=2e-5 * float(np.abs(go[fin]).max()))
    # and back to the fixed-point path on the same workspace: the plane must be clean again
    gg, flat, peak = run_vote(dev, ob["pc"], out, idx, corner, dims, res, 72, True)
    g64, cnt = check_grid(oracle, gg, ob["pc"], out, idx, corner, dims, res, 72, True)
    assert flat == int(np.argmax(g64))


@pytest.mark.parametrize("n_pairs", [700, 262144 + 77, 3 * 262144 + 1000, 5 * 262144 - 3])
def test_binned_vote_super_round_lengths(oracle, dev, n_pairs):
    """v3_bin_kernel culls and queues its pairs in super-rounds of bin_sr x 512 per workgroup, bin_sr = 1 .. 8 by the length of the pair
    list (ragged tails included): many-tile grid, a trained network's regime (every block dense: queued unculled) and a random-weight
    network's (half of the circles miss the grid: culled, compacted), every cell against the exact sum"""
    n, k = 4096, 320
    ob = syn.make_object("bottle", n, 11)
    idx = syn.make_pairs(n, k, 11)[:n_pairs].astype(np.int32)
    cfg = ob["cfg"]
    res = 2.2e-3
    dims = (48, 140, 48)
    assert _lib.lib().cppf_vote_tiles(*dims) >= 4
    corner = (ob["center"] - 0.5 * np.array(dims) * res).astype(np.float32)
    ka = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
    kb = np.random.default_rng(5).integers(0, 32, (n_pairs, 2))
    un = np.stack([kb[:, 0] / 31 * 2 * cfg.vote_range[0] - cfg.vote_range[0], kb[:, 1] / 31 * cfg.vote_range[1]], -1).astype(np.float32)
    for out in (ka, un):
        gg, flat, peak = run_vote(dev, ob["pc"], out, idx, corner, dims, res, 72, True)
        g64, cnt = check_grid(oracle, gg, ob["pc"], out, idx, corner, dims, res, 72, True)
        assert flat == int(np.argmax(g64)) and peak == gg.max()


def test_bin_kernel_staging_overflow_path(oracle, dev):
    """horizontal circles of 30-95 cells radius on a 35-tile grid (tiles of 29 x 40 cells): a pair visits ~10 tiles, so one batch round
    of a workgroup (512 pairs) produces more records than the LDS staging area holds -- waves find it full, ask for a flush and retry
    their batch (`stuck` / `again` in v3_bin_kernel)"""
    dims, res = (200, 200, 24), 4e-3
    out10 = (C.c_int32 * 10)()
    half, k = 1024, 24
    rng = np.random.default_rng(9)
    base = rng.uniform(-0.03, 0.03, (half, 3)).astype(np.float32) * np.array([1, 1, 0.2], np.float32)
    pc = np.concatenate([base, base + np.array([0, 0, 0.02], np.float32)]).astype(np.float32)      # point i + half sits right above point i
    idx = np.stack([np.repeat(np.arange(half), k), np.repeat(np.arange(half), k) + half], -1).astype(np.int32)
    P = idx.shape[0]
    assert _lib.lib().cppf_vote_plan_query(P, 72, *dims, out10) == 0 and out10[0] == 3 and out10[1] >= 16
    tx, ty, nty = out10[2], out10[3], out10[5]
    out = np.stack([rng.uniform(-0.02, 0.02, P), rng.uniform(0.12, 0.38, P)], -1).astype(np.float32)
    corner = (-0.5 * np.array(dims) * res).astype(np.float32)
    gg, flat, peak = run_vote(dev, pc, out, idx, corner, dims, res, 72, True)
    g64, cnt = check_grid(oracle, gg, pc, out, idx, corner, dims, res, 72, True)
    # tiles a pair's samples fall into (numpy, the reference's formula in fp64): enough for a round's records to overflow the staging area
    a, b = pc[idx[:, 0]].astype(np.float64), pc[idx[:, 1]].astype(np.float64)
    u = (a - b) / (np.linalg.norm(a - b, axis=1, keepdims=True) + 1e-7)
    co = np.stack([np.zeros(P), -u[:, 2], u[:, 1]], -1)
    co /= np.linalg.norm(co, axis=1, keepdims=True) + 1e-7
    x = co * out[:, 1:2]
    y = np.cross(x, u)
    th = np.arange(72) * 2 * np.pi / 72
    g = ((a - u * out[:, :1])[:, None, :] + x[:, None, :] * np.cos(th)[None, :, None] + y[:, None, :] * np.sin(th)[None, :, None] - corner) / res
    ok = np.all((g >= 0.01) & (g < np.array(dims) - 1.01), axis=2)
    tile = np.where(ok, (np.floor(g[..., 0]) // tx) * nty + np.floor(g[..., 1]) // ty, -1)
    visits = np.mean([len(set(t[t >= 0])) for t in tile[:2000]])
    assert visits * 512 > 3072 * 1.2, visits
    assert peak == gg.max()
    gg2, flat2, _ = run_vote(dev, pc, out, idx, corner, dims, res, 72, True)
    np.testing.assert_array_equal(gg, gg2)
    assert flat == flat2


# --------------------------------------------------------------------------- the vote as exact integers (cppf_vote_grid_raw)
@pytest.mark.parametrize("n,k,res,bits,unit", [(1500, 40, 4e-3, 0, True), (1500, 40, 4e-3, 20, True), (3000, 50, 2e-3, 0, True),
                                              (3000, 50, 2e-3, 17, False), (1200, 30, 4e-3, 24, False)])
def test_integer_grid_equals_the_oracles_fixed_point_vote(oracle, dev, n, k, res, bits, unit):
    """cppf_vote_grid_raw: every cell's sum of quanta, fused (2 tiles) and binned (many tiles), chosen and forced bits, unit and
    non-unit probs -- EQUAL to oracle.ppf_voting_fixed (floor(w 2^bits / p2 + 1/2) per deposit, summed as int64), and
    grid_from_raw of it equal to what cppf_vote_argmax writes when it chooses the same bits"""
    from cppf_amd.inference import grid_shape
    ob, idx, out = case(n=n, k=k, seed=6)
    corners, dims = grid_shape(ob["pc"], res)
    rng = np.random.default_rng(8)
    probs = None if unit else rng.uniform(0.3, 1.7, n).astype(np.float32)
    pc, o, i32, corner = t(ob["pc"], dev), t(out, dev), t(idx, dev), t(corners[0], dev)
    pr = None if unit else t(probs, dev)
    raw = torch.full(dims, -7, dtype=torch.int64, device=dev)
    q = torch.zeros(1, dtype=torch.float32, device=dev)
    voting.vote_grid_raw(pc, o, pr, i32, raw, q, corner, res, 72, True, fixed_bits=bits)
    qv = float(q)
    p2 = 1.0 if unit else 2.0
    used = int(round(np.log2(p2 / qv)))
    assert qv > 0 and (bits == 0 or used == bits) and 8 <= used <= 24
    assert bits != 0 or used >= voting.vote_fixed_point_bits(idx.shape[0], 72, dims)   # the query: a lower bound of what a launch chooses
    want, qo = oracle.ppf_voting_fixed(ob["pc"], out, np.ones(n, np.float32) if unit else probs, idx, dims, corners[0], res, 72, True, used)
    assert qo == qv
    got = raw.cpu().numpy()
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} of {want.size} cells differ"
    assert want.sum() > 1e6
    # += (accumulate) adds the integers
    voting.vote_grid_raw(pc, o, pr, i32, raw, q, corner, res, 72, True, fixed_bits=used, accumulate=True)
    assert np.array_equal(raw.cpu().numpy(), 2 * want)
    # one conversion, the arg-max of the converted grid
    grid, gi, gv = voting.grid_from_raw(torch.from_numpy(want).to(dev), q)
    assert np.array_equal(grid.cpu().numpy(), (want.astype(np.float64) * qv).astype(np.float32))
    assert int(gi) == int(np.argmax(grid.cpu().numpy())) and float(gv) == float(grid.max())
    if bits == 0:
        g2 = torch.empty(dims, dtype=torch.float32, device=dev)
        i2, v2 = voting.vote_argmax(pc, o, pr, i32, g2, corner, res, 72, True, accumulate=False)
        assert torch.equal(g2, grid) and int(i2) == int(gi) and float(v2) == float(gv)


def test_integer_grids_of_pair_slices_add_up_to_the_whole_lists_grid(dev):
    """pair-sharded vote, ranks emulated on one device: with the bits of the WHOLE list forced, the slices' integer grids sum to
    the whole list's integer grid exactly, for any number of slices (C5-like fine grid: the binned path, racy queues)"""
    from cppf_amd import sharding
    from cppf_amd.inference import grid_shape
    ob, idx, out = case(n=4096, k=96, seed=12)
    res = 2e-3
    corners, dims = grid_shape(ob["pc"], res)
    pc, o, i32, corner = t(ob["pc"], dev), t(out, dev), t(idx, dev), t(corners[0], dev)
    P = idx.shape[0]
    i0, v0, g0, q0 = sharding.vote_sharded(pc, o, i32, corner, dims, res, P, 1)
    assert float(q0) > 0 and float(v0) > 50
    bits = voting.vote_fixed_point_bits(P, 72, dims)
    for world in (2, 3, 8):
        total = torch.zeros(dims, dtype=torch.int64, device=dev)
        q = torch.zeros(1, dtype=torch.float32, device=dev)
        for r in range(world):
            lo, hi = sharding.shard_pairs(P, r, world)
            voting.vote_grid_raw(pc, o[lo:hi].contiguous(), None, i32[lo:hi].contiguous(), total, q, corner, res, 72, True,
                                 fixed_bits=bits, accumulate=True)
        g, gi, gv = voting.grid_from_raw(total, q)
        assert float(q) == float(q0) and torch.equal(g, g0) and int(gi) == int(i0) and float(gv) == float(v0), world


def test_integer_vote_rejects_what_it_cannot_serve(dev):
    L = _lib.lib()
    ob, idx, out = case(n=600, k=10, seed=1)
    pc, o, i32 = t(ob["pc"], dev), t(out, dev), t(idx, dev)
    corner = t(np.zeros(3, np.float32), dev)
    raw = torch.zeros((8, 8, 8), dtype=torch.int64, device=dev)
    q = torch.zeros(1, dtype=torch.float32, device=dev)
    with pytest.raises(_lib.CppfError):
        voting.vote_grid_raw(pc, o, None, i32, raw, q, corner, 4e-3, 72, True, fixed_bits=25)      # bits beyond fp32 deposits
    big = torch.zeros((300, 300, 300), dtype=torch.int64, device=dev)
    with pytest.raises(_lib.CppfError):
        voting.vote_grid_raw(pc, o, None, i32, big, q, corner, 4e-3, 72, True)                     # > 64 tiles: no integer path (global fp32 atomics)
    # an empty slice (more ranks than pairs): zero image, quantum +inf -> ignored by the MIN over ranks, converts to a zero grid
    raw.fill_(5)
    voting.vote_grid_raw(pc, o[:0].contiguous(), None, i32[:0].contiguous(), raw, q, corner, 4e-3, 72, True)
    assert int(raw.abs().sum()) == 0 and float(q) == float("inf")
    g, gi, gv = voting.grid_from_raw(raw, q)
    assert float(g.abs().sum()) == 0.0
    from cppf_amd import sharding
    i_, v_, g_, q_ = sharding.vote_sharded(pc, o[:0].contiguous(), i32[:0].contiguous(), corner, (8, 8, 8), 4e-3, idx.shape[0], 1)
    assert float(g_.abs().sum()) == 0.0
    # negative probs: the launch accumulates in fp32 -> quantum 0 flags the image as not valid, the converted grid is NaN
    probs = torch.full((600,), -1.0, dtype=torch.float32, device=dev)
    voting.vote_grid_raw(pc, o, probs, i32, raw, q, corner, 4e-3, 72, True)
    assert float(q) == 0.0
    g, gi, gv = voting.grid_from_raw(raw, q)
    assert bool(torch.isnan(g).all())


# --------------------------------------------------------------------------- more than 72 rotations (nocs/inference.py:39 --num_rots)
@pytest.mark.parametrize("n,k,res,n_rots,adaptive", [(4096, 128, 4e-3, 144, True), (4096, 64, 4e-3, 360, False), (8192, 64, 2e-3, 144, True),
                                                     (2048, 40, 2e-3, 100, False), (1500, 40, 4e-3, 73, True)])
def test_more_than_72_rotations_run_as_passes_of_the_same_kernels(oracle, dev, n, k, res, n_rots, adaptive):
    """n_rots > 72 = ceil(n_rots / 72) passes of the fused / binned kernels over windows of 72 rotation indices (round 2's kernels, kept
    for this until round 3, are gone): config-size clouds, fused and binned grids, adaptive on and off.  Every cell against the exact
    fp64 vote sum (half a quantum per deposit), the arg-max identical, and the integer image EQUAL to the oracle's fixed-point vote."""
    from cppf_amd.inference import grid_shape
    ob, idx, out = case(n=n, k=k, seed=21)
    corners, dims = grid_shape(ob["pc"], res)
    P = idx.shape[0]
    L = _lib.lib()
    plan = (C.c_int32 * 10)()
    assert L.cppf_vote_plan_query(P, n_rots, *[int(d) for d in dims], plan) == 0 and plan[0] in (2, 3)
    gg, flat, peak = run_vote(dev, ob["pc"], out, idx, corners[0], dims, res, n_rots, adaptive)
    g64, cnt = oracle.ppf_voting_f64(ob["pc"], out, np.ones(n, np.float32), idx, dims, corners[0], res, n_rots, adaptive)
    passes = -(-n_rots // 72)
    q = 2.0 ** -8                                            # (the coarsest quantum any pass may have chosen)
    bits = voting.vote_fixed_point_bits(P, n_rots, dims)
    tol = 2e-6 * np.abs(g64) * passes + cnt * 0.5 * 2.0 ** -bits + 1e-30
    err = np.abs(gg.astype(np.float64) - g64)
    assert np.all(err <= tol), f"max excess {np.max(err - tol)} at {np.unravel_index(np.argmax(err - tol), err.shape)}"
    assert flat == int(np.argmax(g64)) and peak == gg.max() and g64.max() > 100
    if not adaptive:
        assert cnt.sum() > 72 * 8 * P * 0.05                 # far more deposits than 72 rotations could make
    # the integer image: one scale for all passes, equal to the oracle's statement
    pc, o, i32, corner = t(ob["pc"], dev), t(out, dev), t(idx, dev), t(corners[0], dev)
    raw = torch.empty(dims, dtype=torch.int64, device=dev)
    qd = torch.zeros(1, dtype=torch.float32, device=dev)
    voting.vote_grid_raw(pc, o, None, i32, raw, qd, corner, res, n_rots, adaptive)
    used = int(round(-np.log2(float(qd))))
    want, _ = oracle.ppf_voting_fixed(ob["pc"], out, np.ones(n, np.float32), idx, dims, corners[0], res, n_rots, adaptive, used)
    assert np.array_equal(raw.cpu().numpy(), want)


@pytest.mark.parametrize("n,k", [(1500, 40), (4096, 128)])
def test_vote_workgroup_hint(oracle, dev, n, k):
    """CPPF_VOTE_WORKGROUPS (cppf.h): fewer, longer-lived vote workgroups for callers with several instances in flight.  Chunking
    and the fixed-point scale follow the width (up to two bits coarser than the default launch describes); the grid stays the sum
    of the quantised deposits -- every cell against the exact fp64 sum, arg-max identical -- by value and through the *_dyn form;
    undefined flag bits and a hint on the integer-image entry point are refused."""
    from cppf_amd.inference import grid_class, grid_shape
    ob, idx, out = case(n=n, k=k, seed=21)
    res = ob["cfg"].res
    corners, dims = grid_shape(ob["pc"], res)
    pc, o, i64, corner = t(ob["pc"], dev), t(out, dev), t(idx.astype(np.int64), dev), t(corners[0], dev)
    g64, cnt = oracle.ppf_voting_f64(ob["pc"], out, np.ones(n, np.float32), idx, dims, corners[0], res, 72, True)
    bits = _lib.lib().cppf_vote_fixed_point_bits(idx.shape[0], 72, *[int(d) for d in dims])
    grid0 = torch.empty(dims, dtype=torch.float32, device=dev)
    i0, _ = voting.vote_argmax(pc, o, None, i64, grid0, corner, res, 72, True, accumulate=False)
    assert int(i0) == int(np.argmax(g64))
    Tn, many, cap = grid_class(dims)
    shape = torch.tensor([n, *dims], dtype=torch.int32, device=dev)
    G = int(np.prod(dims))
    for w in (64, 128, 200, 256):
        tol = 2e-6 * np.abs(g64) + cnt * (0.5 * 2.0 ** -(bits - 2)) + 1e-30
        grid = torch.full(dims, float("nan"), dtype=torch.float32, device=dev)
        oi, ov = voting.vote_argmax(pc, o, None, i64, grid, corner, res, 72, True, accumulate=False, workgroups=w)
        gg = grid.cpu().numpy()
        assert np.all(np.abs(gg.astype(np.float64) - g64) <= tol) and int(oi) == int(i0) and float(ov) == gg.max()
        if w == 256:
            assert torch.equal(grid, grid0)            # the default width, spelled out
        flat = torch.zeros(cap, dtype=torch.float32, device=dev)
        oi2, ov2 = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.float32, device=dev)
        voting.vote_argmax_dyn(pc, o, None, i64, flat, shape, corner, res, 72, True, oi2, ov2, many_tiles=many, workgroups=w)
        gd = flat[:G].view(dims).cpu().numpy()
        assert np.all(np.abs(gd.astype(np.float64) - g64) <= tol) and int(oi2) == int(i0)
        # += semantics with the hint
        g1 = torch.ones(dims, dtype=torch.float32, device=dev)
        voting.vote_argmax(pc, o, None, i64, g1, corner, res, 72, True, accumulate=True, workgroups=w)
        np.testing.assert_allclose(g1.cpu().numpy(), gg + 1.0, rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError):
        voting.vote_argmax(pc, o, None, i64, grid0, corner, res, 72, True, workgroups=32)
    L = _lib.lib()
    need = L.cppf_vote_workspace_bytes(idx.shape[0], 72, *[int(d) for d in dims])
    ws = torch.zeros(need, dtype=torch.uint8, device=dev)
    oi, ov = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.float32, device=dev)
    args = [pc.data_ptr(), o.data_ptr(), None, i64.data_ptr(), 1, grid0.data_ptr(), corner.data_ptr(), float(res), n, idx.shape[0], 72,
            int(dims[0]), int(dims[1]), int(dims[2]), 1]
    tail = [oi.data_ptr(), ov.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr(dev)]
    assert L.cppf_vote_argmax(*args, 2, *tail) == -1           # bit 1 is not defined
    assert L.cppf_vote_argmax(*args, 1 << 20, *tail) == -1
    raw = torch.zeros(dims, dtype=torch.int64, device=dev)
    q = torch.zeros(1, dtype=torch.float32, device=dev)
    rc = L.cppf_vote_grid_raw(pc.data_ptr(), o.data_ptr(), None, i64.data_ptr(), 1, raw.data_ptr(), q.data_ptr(), corner.data_ptr(),
                              float(res), n, idx.shape[0], 72, int(dims[0]), int(dims[1]), int(dims[2]), 1, (128 << 8), 0,
                              ws.data_ptr(), ws.numel(), stream_ptr(dev))
    assert rc == -1                                            # integer images keep the default plan's bits
