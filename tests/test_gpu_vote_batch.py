"""cppf_vote_argmax_batch: the votes of several objects in one vote launch + one reduce launch (csrc/vote.hip v3_vote_batch_kernel).
Reference: n launches of ppf_kernel (models/voting.py:8-66, nocs/inference.py:192-205) + np.argmax (:207-208), one per instance of the
loop at nocs/inference.py:120.

Per object the batched call must give the grid, arg-max and peak of its own cppf_vote_argmax call BIT FOR BIT -- at ANY width: the
grid is the exact integer sum of the quantised deposits, whichever workgroup took whichever pair, and the fixed-point scale does
not follow the launch width (csrc/vote.hip v3_fused_bits_pairs; v3_split for grids of >= 4 tiles) -- and the oracle's vote within
the fixed-point tolerance (tests/test_gpu_parity.py check_grid)."""
import numpy as np
import pytest
import torch

import cppf_amd.synthetic as syn
from cppf_amd import _lib
from cppf_amd.inference import grid_shape
from cppf_amd.models import voting
from test_gpu_parity import check_grid

pytestmark = pytest.mark.gpu


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def make_case(cat, n, k, seed, res_scale=1.0, idx_dtype=np.int64, quantise=True):
    ob = syn.make_object(cat, n, seed)
    cfg = ob["cfg"]
    res = float(np.float32(cfg.res * res_scale))
    idx = syn.make_pairs(n, k, seed).astype(idx_dtype)
    outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=quantise)
    corners, dims = grid_shape(ob["pc"], res)
    return dict(ob=ob, res=res, idx=idx, outputs=outputs, corner=corners[0].copy(), dims=dims)


def item_of(c, dev, poison=True):
    grid = torch.full(tuple(c["dims"]), float("nan") if poison else 0.0, dtype=torch.float32, device=dev)
    return dict(points=t(c["ob"]["pc"], dev), outputs=t(c["outputs"], dev), point_idxs=t(c["idx"], dev), grid=grid,
                corner=t(c["corner"], dev), res=c["res"], out_idx=torch.full((1,), -7, dtype=torch.int64, device=dev),
                out_val=torch.zeros(1, dtype=torch.float32, device=dev))


def single(c, dev, width):
    it = item_of(c, dev)
    voting.vote_argmax(it["points"], it["outputs"], None, it["point_idxs"], it["grid"], it["corner"], it["res"], 72, True,
                       it["out_idx"], it["out_val"], accumulate=False, workgroups=width)
    torch.cuda.synchronize()
    return it


@pytest.mark.parametrize("n_items,width", [(4, 0), (2, 0), (3, 64), (4, 96), (1, 0)])
def test_batched_votes_equal_single_votes_bit_for_bit(oracle, dev, n_items, width):
    """few-tile objects of different categories, sizes and index widths: one launch for all == one launch each at the same width"""
    specs = [("bottle", 2048, 40, 1, np.int64), ("mug", 1500, 64, 2, np.int32), ("laptop", 3000, 24, 3, np.int64),
             ("bowl", 1024, 96, 4, np.int32)][:n_items]
    cases = [make_case(cat, n, k, seed, idx_dtype=dt) for cat, n, k, seed, dt in specs]
    w = voting.vote_batch_workgroups(n_items, width)
    assert w == (width or max(64, 256 // n_items))
    items = [item_of(c, dev) for c in cases]
    for rep in range(2):          # the second call finds the workspaces (headers, extra planes, rotation tables) as the first left them
        voting.vote_argmax_batch(items, 72, True, accumulate=False, workgroups=width)
    torch.cuda.synchronize()
    for c, it in zip(cases, items):
        for w1 in (w, 0, 128):        # the object alone at the batch's width, at full width, at the runner's 128: the same bits
            one = single(c, dev, w1)
            assert torch.equal(it["grid"], one["grid"]) and int(it["out_idx"]) == int(one["out_idx"]) and float(it["out_val"]) == float(one["out_val"])
        bits = _lib.lib().cppf_vote_fixed_point_bits(c["idx"].shape[0], 72, *c["dims"])
        g64, _ = check_grid(oracle, it["grid"].cpu().numpy(), c["ob"]["pc"], c["outputs"], c["idx"].astype(np.int32), c["corner"], c["dims"],
                            c["res"], 72, True, bits_slack=3)
        assert bits > 0 and int(it["out_idx"]) == int(np.argmax(g64))


def test_eight_objects_in_one_launch(oracle, dev):
    """8 objects, 64 workgroups each (512 workgroups: two rounds of the chip): every cell against the exact fp64 vote sum"""
    cases = [make_case(["bottle", "can", "camera", "mug"][j % 4], 1024 + 128 * j, 48, 10 + j) for j in range(8)]
    assert voting.vote_batch_workgroups(8) == 64
    items = [item_of(c, dev) for c in cases]
    voting.vote_argmax_batch(items, 72, True)
    torch.cuda.synchronize()
    for c, it in zip(cases, items):
        g64, _ = check_grid(oracle, it["grid"].cpu().numpy(), c["ob"]["pc"], c["outputs"], c["idx"].astype(np.int32), c["corner"], c["dims"],
                            c["res"], 72, True, bits_slack=4)
        assert int(it["out_idx"]) == int(np.argmax(g64))
        np.testing.assert_allclose(float(it["out_val"]), g64.max(), rtol=2e-5)


def test_mixed_batch_many_tile_and_accumulate(oracle, dev):
    """an object whose grid needs >= 4 tiles (fine resolution) gets its own bin launch and shares the vote + reduce launches;
    accumulate adds to pre-filled grids (the reference's += semantics, models/voting.py:56-63)"""
    cases = [make_case("bottle", 1500, 32, 21), make_case("bottle", 2048, 32, 22, res_scale=0.5), make_case("mug", 1200, 40, 23)]
    assert _lib.lib().cppf_vote_tiles(*cases[1]["dims"]) >= 4 and _lib.lib().cppf_vote_tiles(*cases[0]["dims"]) < 4
    items = [item_of(c, dev, poison=False) for c in cases]
    for it in items:
        it["grid"].fill_(0.25)
    voting.vote_argmax_batch(items, 72, True, accumulate=True)
    torch.cuda.synchronize()
    for j, (c, it) in enumerate(zip(cases, items)):
        one = item_of(c, dev, poison=False)
        one["grid"].fill_(0.25)
        voting.vote_argmax(one["points"], one["outputs"], None, one["point_idxs"], one["grid"], one["corner"], one["res"], 72, True,
                           one["out_idx"], one["out_val"], accumulate=True, workgroups=0 if j == 1 else voting.vote_batch_workgroups(3))
        torch.cuda.synchronize()
        assert torch.equal(it["grid"], one["grid"]) and int(it["out_idx"]) == int(one["out_idx"])
        check_grid(oracle, it["grid"].cpu().numpy(), c["ob"]["pc"], c["outputs"], c["idx"].astype(np.int32), c["corner"], c["dims"], c["res"],
                   72, True, grid0=np.full(c["dims"], 0.25, np.float32), bits_slack=3)


@pytest.mark.parametrize("width", [0, 64, 128])
def test_many_tile_members_share_the_launches_at_any_width(oracle, dev, width):
    """grids of >= 4 tiles (a posed object's bounding box; a fine grid) beside few-tile ones, by value and shape-polymorphic (tile
    class 16): one bin launch each, ONE vote launch and ONE reduce launch for all -- every grid, arg-max and peak equal to the
    object's own call at full width, at 64 and at the runner's 128 (the binned vote's scale does not follow the width either)"""
    posed = syn.make_posed_object("bottle", 1800, 41)
    c_posed = dict(ob=posed, res=float(np.float32(posed["cfg"].res)), idx=syn.make_pairs(1800, 48, 41))
    c_posed["outputs"] = syn.closed_form_outputs(posed["pc"], posed["pc"].mean(0).astype(np.float64), c_posed["idx"], posed["cfg"], quantise=True)
    corners, c_posed["dims"] = grid_shape(posed["pc"], c_posed["res"])
    c_posed["corner"] = corners[0].copy()
    cases = [make_case("mug", 1500, 40, 42), c_posed, make_case("bottle", 2048, 32, 43, res_scale=0.5), make_case("can", 1024, 64, 44)]
    tiles = [_lib.lib().cppf_vote_tiles(*c["dims"]) for c in cases]
    assert tiles[0] < 4 and 4 <= tiles[1] <= 16 and 4 <= tiles[2] <= 16 and tiles[3] < 4
    items = [item_of(c, dev) for c in cases]
    # member 1 shape-polymorphic: a flat capacity buffer of the 16-tile class, dims in a device record
    cap_cells = 16 * int(_lib.lib().cppf_vote_tile_cells())
    n1 = cases[1]["ob"]["pc"].shape[0]
    pts = torch.zeros((2048, 3), dtype=torch.float32, device=dev)
    pts[:n1] = t(cases[1]["ob"]["pc"], dev)
    items[1].update(points=pts, grid=torch.full((cap_cells,), float("nan"), dtype=torch.float32, device=dev),
                    shape=torch.tensor([n1, *cases[1]["dims"]], dtype=torch.int32, device=dev), many_tiles=16)
    for rep in range(2):
        voting.vote_argmax_batch(items, 72, True, accumulate=False, workgroups=width)
    torch.cuda.synchronize()
    for j, (c, it) in enumerate(zip(cases, items)):
        G = int(np.prod(c["dims"]))
        got = it["grid"][:G].view(*c["dims"]) if j == 1 else it["grid"]
        for w1 in (0, 64, 128):
            one = single(c, dev, w1)
            assert torch.equal(got, one["grid"]), (j, w1)
            assert int(it["out_idx"]) == int(one["out_idx"]) and float(it["out_val"]) == float(one["out_val"]), (j, w1)
        g64, _ = check_grid(oracle, got.cpu().numpy(), c["ob"]["pc"], c["outputs"], c["idx"].astype(np.int32), c["corner"], c["dims"],
                            c["res"], 72, True)
        assert int(it["out_idx"]) == int(np.argmax(g64))


def test_batched_dyn_items_equal_by_value(oracle, dev):
    """shape-polymorphic items (dims in a device record, capacity-sized buffers: BatchPoseRunner's pipelines) in one launch"""
    cases = [make_case("bottle", 1800, 40, 31), make_case("camera", 1300, 56, 32), make_case("can", 900, 64, 33)]
    cap_n, cap_cells = 2048, 3 * int(_lib.lib().cppf_vote_tile_cells())
    items, w = [], voting.vote_batch_workgroups(3)
    for c in cases:
        n = c["ob"]["pc"].shape[0]
        pts = torch.zeros((cap_n, 3), dtype=torch.float32, device=dev)
        pts[:n] = t(c["ob"]["pc"], dev)
        shape = torch.tensor([n, *c["dims"]], dtype=torch.int32, device=dev)
        items.append(dict(points=pts, outputs=t(c["outputs"], dev), point_idxs=t(c["idx"], dev),
                          grid=torch.full((cap_cells,), float("nan"), dtype=torch.float32, device=dev), corner=t(c["corner"], dev), res=c["res"],
                          out_idx=torch.zeros(1, dtype=torch.int64, device=dev), out_val=torch.zeros(1, dtype=torch.float32, device=dev),
                          shape=shape, many_tiles=False))
    voting.vote_argmax_batch(items, 72, True)
    torch.cuda.synchronize()
    for c, it in zip(cases, items):
        one = single(c, dev, w)
        G = int(np.prod(c["dims"]))
        assert torch.equal(it["grid"][:G].view(*c["dims"]), one["grid"]) and int(it["out_idx"]) == int(one["out_idx"])
        assert float(it["out_val"]) == float(one["out_val"])
    # a record beyond the capacities: that item reports -1 / NaN, its neighbours are untouched
    items[1]["shape"][0] = cap_n + 1
    voting.vote_argmax_batch(items, 72, True)
    torch.cuda.synchronize()
    assert int(items[1]["out_idx"]) == -1 and np.isnan(float(items[1]["out_val"]))
    assert int(items[0]["out_idx"]) == int(single(cases[0], dev, w)["out_idx"]) and int(items[2]["out_idx"]) == int(single(cases[2], dev, w)["out_idx"])
    items[1]["shape"][0] = cases[1]["ob"]["pc"].shape[0]
    voting.vote_argmax_batch(items, 72, True)
    torch.cuda.synchronize()
    assert int(items[1]["out_idx"]) == int(single(cases[1], dev, w)["out_idx"])


def test_argument_errors(dev):
    c = make_case("bottle", 512, 8, 1)
    with pytest.raises(ValueError):
        voting.vote_argmax_batch([], 72, True)
    with pytest.raises(ValueError):
        voting.vote_argmax_batch([item_of(c, dev)] * 9, 72, True)
    with pytest.raises(ValueError):
        voting.vote_argmax_batch([item_of(c, dev)], 72, True, workgroups=32)
    bad = item_of(c, dev)
    bad["outputs"] = bad["outputs"].double()
    with pytest.raises(TypeError):
        voting.vote_argmax_batch([bad], 72, True)


def test_sticky_failure_across_rotation_windows(oracle, dev):
    """n_rots = 144 runs two passes (windows of 72 rotations).  A pass that gives up must make the WHOLE call report "no valid
    image" (quantum 0), even when a later pass succeeds (the second pass used to overwrite the report with a
    valid-looking quantum while the grid lacked the first window).

    Construction: collinear points on the y axis, every pair's circle centred on the origin in the plane y = 0; 99 % of the 4 M
    pairs have radius 11.4 cells (adaptive n = 71: all rotations in window 0, all in the grid), 1 % radius 14 cells (n = 87: 15
    rotations in window 1).  One tile, 256 workgroups: a workgroup deposits ~1.16 M unit weights onto ~500 cells in window 0 --
    with the caller forcing 24 fixed-point bits that is ~4 500 wrap-arounds for a 2 048-entry carry log, the plan's own 22 bits
    give ~1 100 -- while window 1 holds a few thousand deposits and succeeds whatever the bits."""
    rng = np.random.default_rng(3)
    N, P, res = 64, 1 << 22, 0.01
    pc = np.zeros((N, 3), np.float32)
    pc[:, 1] = np.linspace(-0.1, 0.1, N, dtype=np.float32) + np.float32(0.0007)
    idx = rng.integers(0, N, (P, 2)).astype(np.int64)
    a, b = pc[idx[:, 0], 1].astype(np.float64), pc[idx[:, 1], 1].astype(np.float64)
    u = np.sign(a - b)                                     # (a == b: degenerate, skipped by the kernel like the reference)
    outputs = np.zeros((P, 2), np.float32)
    outputs[:, 0] = (u * a).astype(np.float32)             # mu: centre = a - u mu = the origin's plane
    outputs[:, 1] = np.where(rng.random(P) < 0.01, 14.0 * res, 11.4 * res).astype(np.float32)
    dims, corner = (30, 30, 30), np.full(3, -0.15, np.float32)
    assert _lib.lib().cppf_vote_tiles(*dims) == 1
    plan_bits = voting.vote_fixed_point_bits(P, 144, dims)
    assert plan_bits <= 22
    raw = torch.zeros(dims, dtype=torch.int64, device=dev)
    q = torch.full((1,), -1.0, dtype=torch.float32, device=dev)
    args = (t(pc, dev), t(outputs, dev), None, t(idx, dev), raw, q, t(corner, dev), res)
    voting.vote_grid_raw(*args, 144, True, fixed_bits=24)
    torch.cuda.synchronize()
    assert float(q) == 0.0, "window 0 overflowed its carry log: the call must not report a valid image"
    # the workspace recovers: the next call with the plan's own bits is valid and EQUAL to the oracle's fixed-point vote
    voting.vote_grid_raw(*args, 144, True, fixed_bits=0)
    torch.cuda.synchronize()
    used = int(round(-np.log2(float(q))))
    assert float(q) > 0.0 and used >= plan_bits
    want, _ = oracle.ppf_voting_fixed(pc, outputs, np.ones(N, np.float32), idx, dims, corner, res, 144, True, used)
    assert np.array_equal(raw.cpu().numpy(), want) and want.sum() > 1e12
    # ... and the f32 arg-max entry on the same inputs reports -1 / NaN when forced into the same overflow is not possible (it
    # chooses its own bits): it must simply work
    grid = torch.zeros(dims, dtype=torch.float32, device=dev)
    oi, ov = voting.vote_argmax(args[0], args[1], None, args[3], grid, args[6], res, 144, True, accumulate=False)
    torch.cuda.synchronize()
    assert int(oi) == int(np.argmax(want)) or want.reshape(-1)[int(oi)] == want.max()
