"""One real frame.  tests/golden/demo_0000_depth.png is the reference's demo depth image (data/demo/0000_depth.png, a NOCS
REAL275-style Kinect frame: holes, depth noise, no analytic normals), read without OpenCV, cut into a few rectangular
"instances" (a detector's masks are upstream of the path), and taken through nocs/inference.py:131-142 on the device --
back-projection, voxel de-duplication, PCA normals, each against the oracle on this real data -- then kNN + SPRIN + the whole
pose with the networks trained on synthetic objects (tests/golden/trained_*.npz)."""
import os
import time

import numpy as np
import pytest

from conftest import GOLDEN

DEPTH = os.path.join(GOLDEN, "demo_0000_depth.png")
# (category, rows, cols, depth window in mm around the rectangle's median depth): the mug in front of the laptop, the two bowls,
# the mug with the red handle, the can, and the laptop itself
RECTS = [("mug", (262, 356), (124, 206), 90), ("bowl", (184, 246), (288, 366), 90), ("bowl", (194, 250), (370, 442), 90),
         ("mug", (186, 250), (436, 504), 90), ("can", (112, 184), (376, 408), 60), ("laptop", (118, 322), (92, 302), 260)]


def instances(depth):
    out = []
    for cat, (r0, r1), (c0, c1), win in RECTS:
        m = np.zeros(depth.shape, bool)
        patch = depth[r0:r1, c0:c1]
        med = np.median(patch[patch > 0])
        m[r0:r1, c0:c1] = np.abs(patch.astype(np.int64) - med) <= win
        out.append((cat, m))
    return out


def test_depth_png_reader_without_opencv():
    from cppf_amd.utils.util import read_depth_png
    d = read_depth_png(DEPTH)
    assert d.dtype == np.uint16 and d.shape == (480, 640)
    assert int(d.max()) == 1409 and int(d.astype(np.int64).sum()) == 244066561 and abs(float((d == 0).mean()) - 0.2979) < 1e-3
    try:
        from PIL import Image
    except ImportError:
        return
    assert np.array_equal(d, np.array(Image.open(DEPTH)))
    # every PNG filter type, 8 and 16 bit, against PIL's encoder
    import io
    import tempfile
    rng = np.random.default_rng(0)
    for arr in (rng.integers(0, 65535, (37, 53)).astype(np.uint16), (np.arange(40 * 64).reshape(40, 64) * 13 % 251).astype(np.uint8),
                np.cumsum(rng.integers(0, 9, (64, 48)), 1).astype(np.uint16)):
        with tempfile.NamedTemporaryFile(suffix=".png") as f:
            Image.fromarray(arr).save(f.name, optimize=True)
            assert np.array_equal(read_depth_png(f.name), arr)
    with tempfile.NamedTemporaryFile(suffix=".png") as f:
        Image.fromarray(rng.integers(0, 255, (8, 8, 3)).astype(np.uint8)).save(f.name)
        with pytest.raises(ValueError):
            read_depth_png(f.name)


def test_oracle_preprocessing_on_the_real_frame(oracle):
    """CPU: the oracle's back-projection / voxel de-duplication / normals on real depth (what the device path is held to below)"""
    from cppf_amd.frames import NOCS_INTRINSICS
    from cppf_amd.utils.util import read_depth_png
    depth = read_depth_png(DEPTH)
    sizes = []
    for cat, m in instances(depth):
        pts, (rows, cols) = oracle.backproject(depth, NOCS_INTRINSICS, m)
        assert pts.shape[0] == int((m & (depth > 0)).sum()) and np.all(depth[rows, cols] > 0)
        pc = pts / 1000
        pc[:, 0] = -pc[:, 0]
        pc[:, 1] = -pc[:, 1]
        keep = oracle.voxel_dedupe(pc.astype(np.float32), 4e-3 if cat != "laptop" else 1e-2)
        sizes.append(len(keep))
        assert 0.3 < pc[:, 2].mean() < 1.5 and len(keep) <= pts.shape[0]
    assert min(sizes) > 150 and max(sizes) < 20000, sizes


@pytest.mark.gpu
def test_real_frame_preprocessing_equals_oracle_and_poses_are_sane(oracle, dev):
    import torch
    from cppf_amd import training
    from cppf_amd.config import CATEGORIES
    from cppf_amd.frames import NOCS_INTRINSICS, frame_poses, instance_cloud
    from cppf_amd.utils.util import read_depth_png
    depth = read_depth_png(DEPTH)
    inst = instances(depth)
    d_dev = torch.from_numpy(depth.view(np.int16)).to(dev)
    # ---- device pre-processing == oracle, instance by instance, on the real data
    for cat, m in inst:
        cfg = CATEGORIES[cat]
        pc, nrm = instance_cloud(d_dev, NOCS_INTRINSICS, m, cfg)
        pts, _ = oracle.backproject(depth, NOCS_INTRINSICS, m)
        p = pts / 1000.0
        p = np.stack([-p[:, 0], -p[:, 1], p[:, 2]], -1)
        keep = oracle.voxel_dedupe(p.astype(np.float32), cfg.res)
        want = p[keep].astype(np.float32)
        assert np.array_equal(pc.cpu().numpy(), want), cat
        assert np.array_equal(nrm.cpu().numpy(), oracle.estimate_normals(want, oracle.knn(want, cfg.knn))), cat
    # ---- the whole loop with trained networks (synthetic training: bottle weights stand in for can / bowl)
    nets = {}
    for cat, src in (("mug", "mug"), ("laptop", "laptop"), ("bowl", "bottle"), ("can", "bottle")):
        penc, enc = training.load_weights(os.path.join(GOLDEN, f"trained_{src}.npz"), CATEGORIES[src], dev)
        nets[cat] = (enc, penc)
    encs = {c: v[0] for c, v in nets.items()}
    pencs = {c: v[1] for c, v in nets.items()}
    poses = frame_poses(depth, inst, encs, pencs, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    poses = frame_poses(depth, inst, encs, pencs, device=dev)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / len(inst) * 1e3
    print(f"real frame: {len(inst)} instances, {ms:.2f} ms per instance incl. pre-processing (eager, one at a time)")
    for (cat, m), pose in zip(inst, poses):
        assert pose is not None and pose["n_surv"] > 0 and np.all(np.isfinite(pose["T"])) and np.all(np.isfinite(pose["scale"]))
        cfg = CATEGORIES[cat]
        pc, _ = instance_cloud(d_dev, NOCS_INTRINSICS, m, cfg)
        lo, hi = pc.min(0).values.cpu().numpy(), pc.max(0).values.cpu().numpy()
        assert np.all(pose["T"] >= lo - cfg.res) and np.all(pose["T"] <= hi + cfg.res)          # the centre is a grid cell of the cloud's box
        assert abs(np.linalg.norm(pose["up"]) - 1) < 1e-6 and 0.02 < pose["scale_norm"] < 1.5
    # the front mug (a real mug seen by a real sensor, network trained on synthetic cylinders with a handle): the voted centre
    # falls inside the mug -- within its radius of the cloud's centroid axis -- and its size comes out as a mug's
    pc, _ = instance_cloud(d_dev, NOCS_INTRINSICS, inst[0][1], CATEGORIES["mug"])
    c = pc.mean(0).cpu().numpy()
    assert np.linalg.norm(poses[0]["T"] - c) < 0.08, (poses[0]["T"], c)
    assert 0.05 < poses[0]["scale_norm"] < 0.4


@pytest.mark.gpu
@pytest.mark.parametrize("batch_prestage", [True, False])
def test_frame_runner_equals_the_eager_loop(oracle, dev, batch_prestage):
    """(batch_prestage: the members' frame stages in eight shared launches -- cppf_frame_cloud_dyn_batch, two chains of four for the
    frame's eight instances -- or every member's own sixteen, three chains on three lanes.)
    FrameRunner (depth + one label image uploaded per frame, every instance's pre-processing count-driven on the device at the
    head of a captured, shape-polymorphic chain, the instances of a lane sharing their launches, ONE read-back per frame) gives
    frame_poses' poses -- the eager per-instance loop, nocs/inference.py:108-142,177-339 -- bit for bit: first sighting (members'
    own graphs), captured chains, replays; overlapping masks, an instance too small for the kNN (skipped like :121-123), an empty
    mask; the device stage's cloud / normals / grid record equal the oracle's pre-processing."""
    import torch
    from cppf_amd import training
    from cppf_amd.config import CATEGORIES
    from cppf_amd.frames import NOCS_INTRINSICS, FrameRunner, frame_poses
    from cppf_amd.inference import grid_shape
    from cppf_amd.utils.util import read_depth_png
    depth = read_depth_png(DEPTH)
    inst = instances(depth)
    tiny = np.zeros(depth.shape, bool)
    tiny[300:304, 150:156] = True                          # 24 pixels: fewer points than the 60-neighbour kNN needs
    inst = inst + [("mug", tiny), ("can", np.zeros(depth.shape, bool))]
    inst[1] = ("bowl", inst[1][1] | inst[2][1])            # overlapping masks: the label image carries both bits
    nets = {}
    for cat, src in (("mug", "mug"), ("laptop", "laptop"), ("bowl", "bottle"), ("can", "bottle")):
        penc, enc = training.load_weights(os.path.join(GOLDEN, f"trained_{src}.npz"), CATEGORIES[src], dev)
        nets[cat] = (enc, penc)
    encs = {c: v[0] for c, v in nets.items()}
    pencs = {c: v[1] for c, v in nets.items()}
    want = frame_poses(depth, inst, encs, pencs, device=dev, seed=3)
    assert want[6] is None and want[7] is None and all(w is not None for w in want[:6])
    runner = FrameRunner(encs, pencs, dev, batch_prestage=batch_prestage)
    served = []
    for rep in range(6):     # solo graphs (two instances need many-tile pipelines: eager this once), new members solo, chains captured, replays
        got = runner.run(depth, inst, seed=3)
        served.append(dict(runner.last))
        for i, (w, g) in enumerate(zip(want, got)):
            assert (w is None) == (g is None), (rep, i)
            if w is None:
                continue
            assert g["n_points"] == w["n_points"] and g["argmax"] == w["argmax"] and g["n_surv"] == w["n_surv"], (rep, i)
            for k in ("T", "up", "right", "scale", "R"):
                assert np.array_equal(g[k], w[k]), (rep, i, k)
            assert g["peak"] == w["peak"]
    # the union of two bowls and the laptop (res 1e-2 on a 0.5 m object) need >= 4 vote tiles: eager once, many-tile pipelines after
    assert served[0] == {"captured": 4, "eager": 2, "skipped": 2} and runner._many_tile_cats == {"bowl", "laptop"}, served
    assert all(s_ == {"captured": 6, "eager": 0, "skipped": 2} for s_ in served[1:]), served
    assert len(runner._chains) == (2 if batch_prestage else 3)
    # the device stage against the oracle's pre-processing (member 0 of lane 0 still holds instance 0's cloud)
    cat, m = inst[0]
    cfg = CATEGORIES[cat]
    pipe = next(m_["pipe"] for m_ in runner._members.values() if int(m_["slot"][0].item()) == 0 and m_["key"][0] == cat)
    n = int(pipe.shape[0])
    pts, _ = oracle.backproject(depth, NOCS_INTRINSICS, m)
    p = pts / 1000.0
    p = np.stack([-p[:, 0], -p[:, 1], p[:, 2]], -1)
    keep = oracle.voxel_dedupe(p.astype(np.float32), cfg.res)
    pc_want = p[keep].astype(np.float32)
    assert n == len(keep) and np.array_equal(pipe.pc[:n].cpu().numpy(), pc_want)
    assert np.array_equal(pipe.nrm[:n].cpu().numpy(), oracle.estimate_normals(pc_want, oracle.knn(pc_want, cfg.knn)))
    corners, dims = grid_shape(pc_want, cfg.res)
    assert tuple(int(v) for v in pipe.shape[1:4].cpu()) == tuple(dims) and np.array_equal(pipe.corner.cpu().numpy(), corners[0])
    # another frame (different seed = different pairs): the captured chains read the new draws
    got2 = runner.run(depth, inst, seed=4)
    want2 = frame_poses(depth, inst, encs, pencs, device=dev, seed=4)
    for w, g in zip(want2, got2):
        assert (w is None) == (g is None) and (w is None or (g["argmax"] == w["argmax"] and np.array_equal(g["up"], w["up"])))
    assert any(a is not None and b is not None and a["n_surv"] != b["n_surv"] for a, b in zip(got, got2))


@pytest.mark.gpu
def test_frame_runner_keeps_its_graphs_when_the_instances_change(dev):
    """A video: the instances of a frame change order, number and mask (members were keyed by lane, chain position and
    label bit, so every such change minted new pipelines and captures).  Members are keyed by what their launches are sized for --
    (category, capacity, grid class) -- and told per frame which instance they serve: permuted / dropped / re-added instances reuse
    the members that exist, poses stay those of the eager loop, and a chain cache smaller than a frame's groups still serves it"""
    from cppf_amd import training
    from cppf_amd.config import CATEGORIES
    from cppf_amd.frames import FrameRunner, frame_poses
    from cppf_amd.utils.util import read_depth_png
    depth = read_depth_png(DEPTH)
    inst = instances(depth)
    nets = {}
    for cat, src in (("mug", "mug"), ("laptop", "laptop"), ("bowl", "bottle"), ("can", "bottle")):
        penc, enc = training.load_weights(os.path.join(GOLDEN, f"trained_{src}.npz"), CATEGORIES[src], dev)
        nets[cat] = (enc, penc)
    encs = {c: v[0] for c, v in nets.items()}
    pencs = {c: v[1] for c, v in nets.items()}

    def check(runner, frame, seed):
        want = frame_poses(depth, frame, encs, pencs, device=dev, seed=seed)
        got = runner.run(depth, frame, seed=seed)
        for w, g in zip(want, got):
            assert (w is None) == (g is None)
            if w is not None:
                assert g["argmax"] == w["argmax"] and g["n_surv"] == w["n_surv"] and np.array_equal(g["T"], w["T"]) and np.array_equal(g["up"], w["up"])

    runner = FrameRunner(encs, pencs, dev)
    for rep in range(3):
        check(runner, inst, rep)
    n_members = len(runner._members)
    rng = np.random.default_rng(0)
    for rep in range(6):                       # permutations, subsets, the full frame again
        order = rng.permutation(len(inst))[:rng.integers(2, len(inst) + 1)]
        check(runner, [inst[i] for i in order], 10 + rep)
        assert len(runner._members) == n_members, (rep, len(runner._members), n_members)
    check(runner, inst, 99)
    assert len(runner._members) == n_members
    # a chain cache that cannot hold one frame's groups: chains of the running frame are never evicted, the frame is served all the same
    small = FrameRunner(encs, pencs, dev, chain_len=1, n_lanes=2, max_chains=2)
    for rep in range(4):
        check(small, inst, rep)
    assert len(small._chains) <= 2


@pytest.mark.gpu
def test_pipelined_frames_equal_the_synchronous_ones(dev):
    """FrameRunner.submit(): a video loop that enqueues frame k + 1 before it collects frame k (the host's share of a frame under the
    device's).  Different frames (subsets, permutations, seeds) back to back, three submitted before the first result is asked for
    (the third collects the oldest: two pinned sets), results out of order: every frame's poses equal run()'s on a fresh look, the
    first sightings (members' own graphs), the captures and the replays alike."""
    from cppf_amd import training
    from cppf_amd.config import CATEGORIES
    from cppf_amd.frames import FrameRunner
    from cppf_amd.utils.util import read_depth_png
    depth = read_depth_png(DEPTH)
    inst = instances(depth)
    nets = {}
    for cat, src in (("mug", "mug"), ("laptop", "laptop"), ("bowl", "bottle"), ("can", "bottle")):
        penc, enc = training.load_weights(os.path.join(GOLDEN, f"trained_{src}.npz"), CATEGORIES[src], dev)
        nets[cat] = (enc, penc)
    encs = {c: v[0] for c, v in nets.items()}
    pencs = {c: v[1] for c, v in nets.items()}
    rng = np.random.default_rng(5)
    frames = []
    for k in range(9):
        order = rng.permutation(len(inst))[:rng.integers(2, len(inst) + 1)] if k % 3 else np.arange(len(inst))
        d = depth if k % 2 == 0 else np.ascontiguousarray(np.where(depth > 0, depth + np.uint16(1 + k), depth).astype(depth.dtype))
        frames.append((d, [inst[i] for i in order], 40 + k))
    sync = FrameRunner(encs, pencs, dev)
    want = []
    for rep in range(2):                        # (twice: the second pass replays captured chains)
        want = [sync.run(d, f, seed=sd) for d, f, sd in frames]
    piped = FrameRunner(encs, pencs, dev)
    for rep in range(2):
        pend = []
        for k, (d, f, sd) in enumerate(frames):
            pend.append(piped.submit(d, f, seed=sd))
            if k % 4 == 3:                       # collect some in the middle, newest first
                for p in reversed(pend[-2:]):
                    p.result()
        got = [p.result() for p in pend]
        for k, (w_f, g_f) in enumerate(zip(want, got)):
            assert len(w_f) == len(g_f)
            for w, g in zip(w_f, g_f):
                assert (w is None) == (g is None), (rep, k)
                if w is not None:
                    assert g["argmax"] == w["argmax"] and g["n_surv"] == w["n_surv"] and g["n_points"] == w["n_points"], (rep, k)
                    for key in ("T", "up", "right", "scale"):
                        assert np.array_equal(g[key], w[key]), (rep, k, key)
    assert pend[0].result() is got[0]           # idempotent


@pytest.mark.gpu
@pytest.mark.parametrize("depth_float,label_bytes,idx64", [(False, 4, True), (True, 1, False), (False, 2, False)])
def test_frame_cloud_batch_equals_the_single_entry_points(dev, depth_float, label_bytes, idx64):
    """cppf_frame_cloud_dyn_batch (eight launches for all members) against cppf_frame_cloud_dyn_bit + cppf_sample_pairs per member
    through the C ABI: members of different capacity, resolution and k, u16 and f32 depth, 8 / 16 / 32-bit label images, int32 and
    int64 pair lists, a member whose mask is empty, one with fewer points than k_min: clouds, normals, neighbour sets, corners,
    shape records, pairs and uniforms bit for bit"""
    import ctypes as C
    import torch
    from cppf_amd import _lib
    from cppf_amd._torch_util import stream_ptr
    from cppf_amd.frames import NOCS_INTRINSICS
    from cppf_amd.utils.util import read_depth_png
    L = _lib.lib()
    depth = read_depth_png(DEPTH)
    H, W = depth.shape
    inst = instances(depth)
    masks = [inst[0][1], inst[5][1], inst[1][1] | inst[2][1], np.zeros_like(inst[0][1]), inst[4][1] & (np.arange(W)[None, :] < 377)]
    spec = [(4096, 0.004, 30), (65536, 0.01, 60), (8192, 0.003, 16), (4096, 0.004, 30), (4096, 0.004, 60)]      # (capacity, res, k)
    ldt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[label_bytes]
    labels = np.zeros((H, W), ldt)
    for i, m in enumerate(masks):
        labels |= (m.astype(ldt) << ldt(i))
    dd = torch.from_numpy(depth.astype(np.float32) if depth_float else depth.view(np.int16)).to(dev)
    ld = torch.from_numpy(labels.view({1: np.uint8, 2: np.int16, 4: np.int32}[label_bytes])).to(dev)
    kinv = np.ascontiguousarray(np.linalg.inv(NOCS_INTRINSICS))
    n_pairs = 5000
    idt = torch.int64 if idx64 else torch.int32

    def buffers(cap, k):
        z = lambda *s_, dt=torch.float32: torch.full(s_, -7, dtype=dt, device=dev)
        return dict(pc=z(cap, 3), nrm=z(cap, 3), corner=z(3), shape=z(4, dt=torch.int32), nbrs=z(cap, k, dt=torch.int32),
                    idx=z(n_pairs, 2, dt=idt), u_tr=z(n_pairs, 2), u_rot=z(n_pairs, 2),
                    ws=torch.zeros(int(L.cppf_frame_cloud_workspace_bytes(H, W, cap, k)), dtype=torch.uint8, device=dev),
                    slot=torch.zeros(2, dtype=torch.int64, device=dev))

    one, many = [buffers(c, k) for c, _, k in spec], [buffers(c, k) for c, _, k in spec]
    for i, b in enumerate(one + many):
        j = i % len(spec)
        b["slot"].copy_(torch.tensor([j, 1000003 * 7 + j], dtype=torch.int64))
    with torch.cuda.device(dev):
        for b, (cap, res, k) in zip(one, spec):
            _lib.check(L.cppf_frame_cloud_dyn_bit(dd.data_ptr(), 0 if depth_float else 1, ld.data_ptr(), label_bytes, b["slot"].data_ptr(), H, W,
                                                  kinv.ctypes.data, 1000.0, res, k, k + 1, cap, b["pc"].data_ptr(), b["nrm"].data_ptr(),
                                                  b["corner"].data_ptr(), b["shape"].data_ptr(), b["nbrs"].data_ptr(), b["ws"].data_ptr(),
                                                  b["ws"].numel(), stream_ptr(dev)), "cppf_frame_cloud_dyn_bit")
            if idx64:
                _lib.check(L.cppf_sample_pairs(b["idx"].data_ptr(), b["u_tr"].data_ptr(), b["u_rot"].data_ptr(), n_pairs, 1,
                                               b["shape"].data_ptr(), 0, b["slot"].data_ptr() + 8, stream_ptr(dev)), "cppf_sample_pairs")
        arr = (_lib.FrameCloudItem * len(spec))()
        for a, b, (cap, res, k) in zip(arr, many, spec):
            a.label_bit_dev, a.seed_dev = b["slot"].data_ptr(), b["slot"].data_ptr() + 8
            a.pc_out, a.nrm_out, a.corner_out, a.shape_out, a.nbrs_out = (b[n].data_ptr() for n in ("pc", "nrm", "corner", "shape", "nbrs"))
            a.idx, a.u_tr, a.u_rot, a.workspace, a.workspace_bytes = b["idx"].data_ptr(), b["u_tr"].data_ptr(), b["u_rot"].data_ptr(), b["ws"].data_ptr(), b["ws"].numel()
            a.res, a.n_pairs, a.knn_k, a.k_min, a.n_cap, a.idx_is_i64 = res, n_pairs, k, k + 1, cap, 1 if idx64 else 0
        _lib.check(L.cppf_frame_cloud_dyn_batch(len(spec), C.cast(arr, C.c_void_p), dd.data_ptr(), 0 if depth_float else 1, ld.data_ptr(),
                                                label_bytes, H, W, kinv.ctypes.data, 1000.0, stream_ptr(dev)), "cppf_frame_cloud_dyn_batch")
    torch.cuda.synchronize()
    counts = []
    for j, (a, b) in enumerate(zip(one, many)):
        n = int(a["shape"][0])
        counts.append(n)
        assert torch.equal(a["shape"], b["shape"]) and torch.equal(a["corner"], b["corner"]), j
        # (whole buffers, pre-filled alike: what either form leaves untouched beyond the cloud is part of the comparison)
        assert torch.equal(a["pc"], b["pc"]) and torch.equal(a["nrm"], b["nrm"]) and torch.equal(a["nbrs"], b["nbrs"]), j
        assert torch.all(b["nrm"][n:] == -7) and torch.all(b["nbrs"][n:] == -7)
        if idx64:
            assert torch.equal(a["idx"], b["idx"]) and torch.equal(a["u_tr"], b["u_tr"]) and torch.equal(a["u_rot"], b["u_rot"]), j
        else:        # the single sampler draws int64 lists: the same numbers, from the Philox twin on the host
            import cppf_amd.synthetic as syn
            idx_w, utr_w, urot_w = syn.philox_pairs(1000003 * 7 + j, n_pairs, n)
            assert np.array_equal(b["idx"].cpu().numpy(), idx_w.astype(np.int32)), j
            assert np.array_equal(b["u_tr"].cpu().numpy(), utr_w) and np.array_equal(b["u_rot"].cpu().numpy(), urot_w), j
    assert counts[3] == 0 and counts[4] == 0 and counts[1] > 2000 and counts[0] > 100, counts      # (member 4: fewer points than k_min)
    # argument checks
    assert L.cppf_frame_cloud_dyn_batch(0, C.cast(arr, C.c_void_p), dd.data_ptr(), 1, ld.data_ptr(), 4, H, W, kinv.ctypes.data, 1000.0, None) == -1
    assert L.cppf_frame_cloud_dyn_batch(9, C.cast(arr, C.c_void_p), dd.data_ptr(), 1, ld.data_ptr(), 4, H, W, kinv.ctypes.data, 1000.0, None) == -1
    assert L.cppf_frame_cloud_dyn_batch(1, C.cast(arr, C.c_void_p), dd.data_ptr(), 1, ld.data_ptr(), 3, H, W, kinv.ctypes.data, 1000.0, None) == -1
    arr[0].workspace_bytes = 16
    assert L.cppf_frame_cloud_dyn_batch(1, C.cast(arr, C.c_void_p), dd.data_ptr(), 1, ld.data_ptr(), 4, H, W, kinv.ctypes.data, 1000.0, None) == -2
