"""Objects that are already on the device ("inputs already resident on device"): BatchPoseRunner.put() + run().

  cppf_stage_batch        one launch at the head of a captured chain: clouds copied from where the caller keeps them, the grid of
                          nocs/inference.py:194-195 set up, pairs and bin uniforms drawn (:177,186,250) -- against the single calls
  record assembly         the host end of nocs/inference.py:299-339 in the last launch of the batched tail
                          (CppfPoseTailItem.record_out) -- against cppf_amd.inference.assemble_batch and against the oracle chain
"""
import ctypes as C

import numpy as np
import pytest
import torch

import cppf_amd.synthetic as syn
from cppf_amd import _lib, sharding
from cppf_amd._torch_util import stream_ptr
from cppf_amd.config import NOCS_CATEGORIES
from cppf_amd.inference import grid_shape
from test_gpu_configs import check_argmax, make_encoder, ocfg_of, seeded_sd

pytestmark = pytest.mark.gpu


def draw(dev, seed, n_pairs, n_points):
    """cppf_sample_pairs on its own: (idx i64[P,2], u_tr f32[P,2], u_rot f32[P,2]) as host arrays"""
    idx = torch.empty((n_pairs, 2), dtype=torch.int64, device=dev)
    u = torch.empty((2, n_pairs, 2), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().cppf_sample_pairs(idx.data_ptr(), u[0].data_ptr(), u[1].data_ptr(), n_pairs, n_points, None,
                                            int(seed) & 0xFFFFFFFFFFFFFFFF, None, stream_ptr(dev)), "cppf_sample_pairs")
    return idx.cpu().numpy(), u[0].cpu().numpy(), u[1].cpu().numpy()


def test_stage_batch_equals_the_single_calls(dev):
    """ragged clouds (one of a single point, one filling its buffer), with and without features, a static-shape member (no shape
    record), a member that draws nothing, int64 and int32 pair lists (the same numbers)"""
    L = _lib.lib()
    rng = np.random.default_rng(5)
    specs = [(1000, 1024, 4099, 40, True), (1, 64, 77, 40, True), (2048, 2048, 100000, 8, True), (333, 1024, 0, 40, True),
             (700, 1024, 5000, 40, False)]                       # (n, capacity, pairs, F, dynamic)
    n_items = len(specs)
    desc_host = np.zeros((n_items, _lib.STAGE_DESC_WORDS), np.uint64)
    arr = (_lib.StageItem * n_items)()
    keep, want = [], []
    for i, (n, cap, P, F, dyn) in enumerate(specs):
        pc = (rng.standard_normal((n, 3)) * 0.1 + rng.uniform(-1, 1, 3)).astype(np.float32)
        nrm = rng.standard_normal((n, 3)).astype(np.float32)
        feat = rng.standard_normal((n, F)).astype(np.float32) if i != 2 else None
        res = np.float32(0.004 if i % 2 == 0 else 0.01)
        src = [torch.from_numpy(a).to(dev) if a is not None else None for a in (pc, nrm, feat)]
        dst = dict(pc=torch.full((cap, 3), 7.0, device=dev), nrm=torch.full((cap, 3), 7.0, device=dev),
                   feat=torch.full((cap, F), 7.0, device=dev), corner=torch.zeros(3, device=dev),
                   shape=torch.full((4,), -5, dtype=torch.int32, device=dev),
                   idx=torch.full((max(P, 1), 2), -1, dtype=torch.int32 if i == 4 else torch.int64, device=dev),
                   u=torch.full((2, max(P, 1), 2), -1.0, device=dev))
        seed = 0xFEDCBA9876543210 + i
        desc_host[i] = (src[0].data_ptr(), src[1].data_ptr(), 0 if feat is None else src[2].data_ptr(), n, seed, 40 + i)
        a = arr[i]
        a.pc, a.nrm, a.feat, a.corner = dst["pc"].data_ptr(), dst["nrm"].data_ptr(), dst["feat"].data_ptr(), dst["corner"].data_ptr()
        a.shape = dst["shape"].data_ptr() if dyn else None
        a.idx = dst["idx"].data_ptr() if P else None
        a.u_tr, a.u_rot = dst["u"][0].data_ptr(), (dst["u"][1].data_ptr() if i != 1 else None)
        a.n_pairs, a.n_cap, a.F, a.res, a.idx_is_i64 = P, cap, F, float(res), 0 if i == 4 else 1
        keep.append((src, dst))
        want.append((pc, nrm, feat, res, seed))
    desc = torch.from_numpy(desc_host.view(np.int64)).to(dev)
    for i in range(n_items):
        arr[i].desc = desc[i].data_ptr()
    _lib.check(L.cppf_stage_batch(n_items, arr, stream_ptr(dev)), "cppf_stage_batch")
    torch.cuda.synchronize()
    for i, ((n, cap, P, F, dyn), (pc, nrm, feat, res, seed), (_, dst)) in enumerate(zip(specs, want, keep)):
        np.testing.assert_array_equal(dst["pc"].cpu().numpy()[:n], pc)
        np.testing.assert_array_equal(dst["nrm"].cpu().numpy()[:n], nrm)
        assert (dst["pc"].cpu().numpy()[n:] == 7.0).all() and (dst["nrm"].cpu().numpy()[n:] == 7.0).all()      # rows behind the cloud untouched
        if feat is not None:
            np.testing.assert_array_equal(dst["feat"].cpu().numpy()[:n], feat)
        else:
            assert (dst["feat"].cpu().numpy() == 7.0).all()
        corners, dims = grid_shape(pc, res)
        np.testing.assert_array_equal(dst["corner"].cpu().numpy(), corners[0])
        assert dst["shape"].cpu().tolist() == ([n, *dims] if dyn else [-5] * 4)
        if P:
            idx, u_tr, u_rot = draw(dev, seed, P, n)
            assert dst["idx"].dtype == (torch.int32 if i == 4 else torch.int64)
            np.testing.assert_array_equal(dst["idx"].cpu().numpy().astype(np.int64), idx)
            np.testing.assert_array_equal(dst["u"][0].cpu().numpy(), u_tr)
            if i != 1:
                np.testing.assert_array_equal(dst["u"][1].cpu().numpy(), u_rot)
            else:
                assert (dst["u"][1].cpu().numpy() == -1.0).all()
            assert idx.min() >= 0 and idx.max() < n
    # argument errors are reported, not launched
    assert L.cppf_stage_batch(0, arr, None) == -1 and L.cppf_stage_batch(9, arr, None) == -1 and L.cppf_stage_batch(1, None, None) == -1
    bad = (_lib.StageItem * 1)()
    C.memmove(bad, arr, C.sizeof(_lib.StageItem))
    bad[0].n_cap = 0
    assert L.cppf_stage_batch(1, bad, None) == -1


def mixed_batch(n_objects, sizes, n_pairs, seed0):
    objs = []
    for j in range(n_objects):
        ob = syn.make_object(NOCS_CATEGORIES[j % 6], sizes[j % len(sizes)], seed0 + j)
        objs.append(dict(pc=ob["pc"], normals=ob["normals"], feat=ob["feat"], cfg=ob["cfg"], n_pairs=n_pairs))
    return objs


@pytest.mark.parametrize("chain_len,n_lanes", [(None, 3), (1, 2), (8, 1)])
def test_resident_batch_equals_the_host_path_and_the_oracle(oracle, golden, dev, chain_len, n_lanes):
    """8 mixed-category objects of ragged sizes: records of run(put(objects)) -- assembled on the device, never read back by the
    runner -- against run(objects) with the same pairs handed over as host arrays (records assembled on the host), and against the
    oracle's pose of every object; first batch (eager chains), second (captured), third (forms adapted)"""
    from cppf_amd.batch import BatchPoseRunner
    sd = seeded_sd(0, 4.0)
    encoders = {c: make_encoder(sd, dev) for c in NOCS_CATEGORIES}
    P, seed = 40000, 11
    objects = mixed_batch(8, (1024, 700, 1500, 2048), P, 300)
    runner = BatchPoseRunner(encoders, dev, chain_len=chain_len, n_lanes=n_lanes)
    resident = runner.put(objects)
    assert all(o["pc"].is_cuda and len(o["dims"]) == 3 for o in resident)
    recs = [runner.run(resident, seed=seed) for _ in range(3)]
    assert all(r.is_cuda and r.shape == (8, sharding.RECORD) for r in recs)
    recs = [r.cpu().numpy() for r in recs]
    np.testing.assert_array_equal(recs[0], recs[1])
    np.testing.assert_array_equal(recs[0], recs[2])
    assert recs[0][:, 15].tolist() == list(range(8)) and (recs[0][:, 16:] == 0).all()
    # the host path on the same draws
    host_objs = []
    for j, o in enumerate(objects):
        idx, u_tr, u_rot = draw(dev, seed * 1000003 + j, P, o["pc"].shape[0])
        host_objs.append(dict(o, point_idxs=idx, u_tr=u_tr, u_rot=u_rot))
    want = BatchPoseRunner(encoders, dev).run(host_objs)
    want = want.cpu().numpy() if torch.is_tensor(want) else np.asarray(want)
    for cols in ((0, 3), (12, 16)):                                      # T, arg-max, peak, survivors, id: the same bits
        np.testing.assert_array_equal(recs[0][:, cols[0]:cols[1]], want[:, cols[0]:cols[1]])
    np.testing.assert_allclose(recs[0][:, 3:9], want[:, 3:9], atol=1e-12)          # up, right
    np.testing.assert_allclose(recs[0][:, 9:12], want[:, 9:12], rtol=1e-6)         # scale (fp32 exp on either side)
    assert (want[:, 14] > 0).all()
    sph = golden("sphere.npz")["pts"]
    for j, obj in enumerate(host_objs):
        cfg = obj["cfg"]
        o = oracle.estimate_pose(obj["pc"], obj["normals"], obj["feat"], obj["point_idxs"], sd, ocfg_of(cfg), obj["u_tr"], obj["u_rot"], sph)
        check_argmax(oracle, int(recs[0][j, 12]), o, obj, obj["point_idxs"], cfg.res)
        if int(recs[0][j, 12]) != o["argmax"]:
            continue
        assert int(recs[0][j, 14]) == int(o["mask"].sum()), (j, cfg.category)
        np.testing.assert_allclose(recs[0][j, 0:3], o["T"], atol=1e-12)
        np.testing.assert_allclose(recs[0][j, 3:6], o["up"], atol=1e-12)
        if cfg.regress_right:            # the oracle's `right` is the sign-corrected best bin; the record holds it orthogonalised (:305-312)
            r = o["right"] - np.dot(o["up"], o["right"]) * o["up"]
            np.testing.assert_allclose(recs[0][j, 6:9], r / (np.linalg.norm(r) + 1e-9), atol=1e-9)
        else:
            r = np.array([0, -o["up"][2], o["up"][1]])
            np.testing.assert_allclose(recs[0][j, 6:9], r / (np.linalg.norm(r) + 1e-9), atol=1e-9)
        np.testing.assert_allclose(recs[0][j, 9:12], o["scale"], rtol=1e-6)


@pytest.mark.parametrize("overlap", [False, True])
def test_resident_objects_given_as_device_tensors_and_a_new_batch_per_run(dev, overlap):
    """(overlap: BatchPoseRunner(overlap_batches=True) -- a batch's chains wait for their inputs' events only, so lanes run ahead into
    the next batch while the caller's stream still reads this one's records)  put() of device tensors (dims computed on the device); a runner fed a DIFFERENT batch every run (new addresses, new sizes in the
    same buckets, new seeds) returns what a fresh runner returns for that batch -- nothing of a replay is baked into the graphs"""
    from cppf_amd.batch import BatchPoseRunner
    sd = seeded_sd(1, 4.0)
    encoders = {c: make_encoder(sd, dev) for c in NOCS_CATEGORIES}
    runner = BatchPoseRunner(encoders, dev, overlap_batches=overlap)
    outs = []
    batches = [mixed_batch(5, (900, 1000, 800), 30000, 400 + 10 * b) for b in range(3)]
    for b, objs in enumerate(batches):
        on_dev = [dict(o, pc=torch.from_numpy(o["pc"]).to(dev), normals=torch.from_numpy(o["normals"]).to(dev),
                       feat=torch.from_numpy(o["feat"]).to(dev)) for o in objs]
        res = runner.put(on_dev)
        assert [o["dims"] for o in res] == [grid_shape(o["pc"], o["cfg"].res)[1] for o in objs]
        outs.append(runner.run(res, seed=b).cpu().numpy())
    for b, objs in enumerate(batches):
        fresh = BatchPoseRunner(encoders, dev)
        np.testing.assert_array_equal(fresh.run(fresh.put(objs), seed=b).cpu().numpy(), outs[b])
    with pytest.raises(ValueError):
        runner.run([batches[0][0], runner.put(batches[0])[1]])            # host and resident objects mixed
    with pytest.raises(ValueError):
        runner.put([dict(batches[0][0], point_idxs=np.zeros((4, 2), np.int64))])


def test_overlapping_batches_return_each_batch_its_own_records(dev):
    """overlap_batches=True with nothing between the runs: many batches of DIFFERENT objects enqueued back to back without a
    synchronisation -- resident and host-array batches alternating, more batches in flight than the two record buffers -- every
    returned tensor must hold its own batch's records (what a fresh, non-overlapping runner returns for it)"""
    from cppf_amd.batch import BatchPoseRunner
    sd = seeded_sd(2, 4.0)
    encoders = {c: make_encoder(sd, dev) for c in NOCS_CATEGORIES}
    runner = BatchPoseRunner(encoders, dev, overlap_batches=True, n_lanes=3)
    batches = [mixed_batch(int(n), (600, 1000, 800, 1200), 20000, 700 + 20 * b) for b, n in enumerate((8, 3, 8, 13, 1, 8, 8, 5))]
    for warm in range(2):                                   # chains captured on their second sighting
        for b, objs in enumerate(batches):
            runner.run(runner.put(objs) if b % 2 == 0 else objs, seed=b)
    torch.cuda.synchronize()
    outs = [runner.run(runner.put(objs) if b % 2 == 0 else objs, seed=b) for b, objs in enumerate(batches)]      # no sync in between
    outs = [o.cpu().numpy() for o in outs]
    fresh = BatchPoseRunner(encoders, dev)
    for b, objs in enumerate(batches):
        want = fresh.run(fresh.put(objs), seed=b).cpu().numpy()
        np.testing.assert_array_equal(outs[b], want, err_msg=f"batch {b}")


def test_gather_words_rows_from_separate_buffers(dev):
    """cppf_gather_words: the 8-byte words of up to 32 separately allocated buffers into the rows of one array with one launch
    (FrameRunner's records and shape words); the Python helper cuts longer lists into launches of 32"""
    import ctypes as C
    from cppf_amd import _lib
    from cppf_amd._torch_util import gather_words
    rng = np.random.default_rng(0)
    for n_rows, words in ((1, 21), (8, 21), (32, 2), (45, 5)):
        srcs = [torch.from_numpy(rng.standard_normal(words + 3)).to(dev) for _ in range(n_rows)]       # (longer than a row: only `words` are taken)
        dst = torch.full((n_rows, words), -1.0, dtype=torch.float64, device=dev)
        gather_words(dst, srcs, dev)
        torch.cuda.synchronize()
        assert torch.equal(dst, torch.stack([s_[:words] for s_ in srcs]))
    shapes = [torch.tensor([5 + i, 7, 8, 9], dtype=torch.int32, device=dev) for i in range(6)]        # i32[4] = two words
    out = torch.zeros((6, 4), dtype=torch.int32, device=dev)
    gather_words(out, shapes, dev)
    assert out.cpu().tolist() == [[5 + i, 7, 8, 9] for i in range(6)]
    L = _lib.lib()
    ptrs = (C.c_void_p * 1)(srcs[0].data_ptr())
    assert L.cppf_gather_words(33, ptrs, 4, dst.data_ptr(), None) == -1
    assert L.cppf_gather_words(1, ptrs, 4, dst.data_ptr() + 4, None) == -1            # (destination not 8-byte aligned)
    assert L.cppf_gather_words(1, None, 4, dst.data_ptr(), None) == -1
    assert L.cppf_gather_words(0, None, 4, None, None) == 0
