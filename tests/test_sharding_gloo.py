"""Object sharding + the single end-of-batch collective on 2 CPU ranks (gloo).  The per-object
compute is replaced by a deterministic function of the object id; what is under test is the N>1 path:
round-robin assignment, ragged shard sizes, record packing, gather order, identical result on every
rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cppf_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_pose(obj):
    rng = np.random.default_rng(obj)
    return dict(T=rng.normal(size=3), up=rng.normal(size=3), right=rng.normal(size=3), scale=rng.random(3),
                argmax=int(rng.integers(0, 50000)), peak=float(rng.random() * 100), n_surv=int(rng.integers(0, 1000)))


def _worker(rank, world, port, n_objects, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = sharding.init_distributed("gloo")
    assert (r, w) == (rank, world)
    mine = sharding.shard_objects(n_objects, rank, world)
    recs = torch.stack([sharding.pack_record(j, _fake_pose(j)) for j in mine]) if mine else \
        torch.zeros((0, sharding.RECORD), dtype=torch.float64)
    allrec = sharding.gather_records(recs, n_objects, rank, world, validate=True)
    torch.save(allrec, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_objects", [7, 2, 64])
def test_two_rank_gather_matches_single_rank(tmp_path, n_objects):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_objects, str(tmp_path)), nprocs=world, join=True)
    ref = torch.stack([sharding.pack_record(j, _fake_pose(j)) for j in range(n_objects)])
    for rank in range(world):
        got = torch.load(os.path.join(tmp_path, f"rank{rank}.pt"))
        assert got.shape == (n_objects, sharding.RECORD)
        assert torch.equal(got, ref)                      # object order, every rank, bit-identical


def test_shard_objects_round_robin():
    assert sharding.shard_objects(64, 3, 8) == list(range(3, 64, 8))
    assert sharding.shard_objects(5, 7, 8) == []
    allj = sorted(j for r in range(8) for j in sharding.shard_objects(13, r, 8))
    assert allj == list(range(13))
    # the layout is fixed: row s of a rank's records is object rank + s * world; a caller that breaks it is told (validate)
    one = sharding.gather_records(torch.stack([sharding.pack_record(j, _fake_pose(j)) for j in (0, 1, 2)]), 3, 0, 1, validate=True)
    assert one[:, 15].tolist() == [0.0, 1.0, 2.0]
    with pytest.raises(AssertionError):
        sharding.gather_records(torch.stack([sharding.pack_record(j, _fake_pose(j)) for j in (2, 0, 1)]), 3, 0, 1, validate=True)
    # validation is opt-in at every world size (the same call must not behave differently with one rank); all-zero rows --
    # objects a caller skipped -- pass it
    mis = sharding.gather_records(torch.stack([sharding.pack_record(j, _fake_pose(j)) for j in (2, 0, 1)]), 3, 0, 1)
    assert mis[:, 15].tolist() == [2.0, 0.0, 1.0]
    skipped = torch.stack([sharding.pack_record(j, _fake_pose(j)) for j in (0, 1, 2)])
    skipped[1] = 0
    assert sharding.gather_records(skipped, 3, 0, 1, validate=True)[:, 15].tolist() == [0.0, 0.0, 2.0]
    with pytest.raises(ValueError):
        sharding.gather_records(torch.zeros((1, sharding.RECORD), dtype=torch.float64), 3, 0, 1)
    perm = sharding._object_order(7, 3, torch.device("cpu")).tolist()      # rank-major rows -> object order
    assert perm == [0, 3, 6, 1, 4, 7, 2] and sharding._object_order(7, 3, torch.device("cpu")) is sharding._object_order(7, 3, "cpu")


def _forced_worker(rank, world, port, out_dir):
    """ONE rank with the group forced on (CPPF_FORCE_DIST=1): the collective branches run -- what the -m gpu test does with
    backend nccl on one MI355X (tests/test_gpu_rccl.py), here over gloo"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      CPPF_FORCE_DIST="1")
    r, w, _ = sharding.init_distributed("gloo")
    assert (r, w) == (0, 1) and dist.is_initialized() and dist.get_world_size() == 1
    recs = torch.stack([sharding.pack_record(j, _fake_pose(j)) for j in range(5)])
    calls = []
    orig = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    out = sharding.gather_records(recs, 5, 0, 1, validate=True)
    dist.all_gather_into_tensor = orig
    assert calls == [1] and torch.equal(out, recs)                     # the collective ran, the records came back unchanged
    g = torch.arange(24, dtype=torch.int64).reshape(2, 3, 4)
    assert torch.equal(sharding.allreduce_grid(g.clone(), 1), g)       # (a one-rank sum)
    torch.save(out, os.path.join(out_dir, "forced.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_single_rank_with_the_group_forced_on_takes_the_collective_branch(tmp_path):
    mp.spawn(_forced_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert torch.load(os.path.join(tmp_path, "forced.pt")).shape == (5, sharding.RECORD)


# --------------------------------------------------------------------------- intra-object pair sharding
def _vote_case():
    import cppf_amd.synthetic as syn
    from oracle import oracle as O
    ob = syn.make_object("bottle", 256, 5)
    cfg = ob["cfg"]
    idx = syn.make_pairs(256, 16, 5).astype(np.int32)
    outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
    corner, dims = O.grid_setup(ob["pc"], cfg.res)
    return ob, cfg, idx, outputs, corner, dims


def _vote_worker(rank, world, port, out_dir):
    """each rank votes its slice of the pairs into a private INTEGER grid (the oracle's statement of cppf_vote_grid_raw:
    every deposit rounded to 2^-bits, summed as int64); one all-reduce sums them"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from oracle import oracle as O
    sharding.init_distributed("gloo")
    ob, cfg, idx, outputs, corner, dims = _vote_case()
    lo, hi = sharding.shard_pairs(idx.shape[0], rank, world)
    raw, q = O.ppf_voting_fixed(ob["pc"], outputs[lo:hi], np.ones(256, np.float32), idx[lo:hi], dims, corner, cfg.res, 72, True, 24)
    g = torch.from_numpy(raw)
    sharding.allreduce_grid(g, world)
    torch.save(g, os.path.join(out_dir, f"grid{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pair_sharded_integer_vote_allreduce_equals_single_rank_bit_for_bit(tmp_path, world):
    """north_star: "bit-exact vote-grid argmax indices" for a pair list split over GPUs.  The exchanged image is integer, so the
    all-reduced grid EQUALS the single-rank grid (rounds 1-3 all-reduced fp32 grids: equal to rounding only)."""
    from oracle import oracle as O
    mp.spawn(_vote_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ob, cfg, idx, outputs, corner, dims = _vote_case()
    full, q = O.ppf_voting_fixed(ob["pc"], outputs, np.ones(256, np.float32), idx, dims, corner, cfg.res, 72, True, 24)
    assert q == 2.0 ** -24 and full.sum() > 0
    grids = [torch.load(os.path.join(tmp_path, f"grid{r}.pt")).numpy() for r in range(world)]
    for g in grids:
        assert np.array_equal(g, full)                                # every rank: the single-rank integers
    # ... and the integer image is the reference's vote to half a quantum per deposit
    g64, cnt = O.ppf_voting_f64(ob["pc"], outputs, np.ones(256, np.float32), idx, dims, corner, cfg.res, 72, True)
    assert np.all(np.abs(full * q - g64) <= cnt * 0.5 * q + 1e-12)
    f32 = np.zeros(tuple(dims), np.float32)
    O.ppf_voting(ob["pc"], outputs, np.ones(256, np.float32), idx, f32, corner, cfg.res, 72, True)
    assert int(np.argmax(full)) == int(np.argmax(f32))


def test_shard_pairs_is_a_balanced_partition():
    for n, w in ((10, 3), (524288, 8), (5, 8), (0, 4)):
        cuts = [sharding.shard_pairs(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in cuts]
        assert max(sizes) - min(sizes) <= 1
