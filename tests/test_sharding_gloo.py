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
    allrec = sharding.gather_records(recs, n_objects, rank, world)
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
    one = sharding.gather_records(torch.stack([sharding.pack_record(j, _fake_pose(j)) for j in (2, 0, 1)]), 3, 0, 1)
    assert one[:, 15].tolist() == [0.0, 1.0, 2.0]


# --------------------------------------------------------------------------- intra-object pair sharding
def _vote_worker(rank, world, port, out_dir):
    """each rank votes its slice of the pairs with the oracle into a private grid; one all-reduce sums them"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import cppf_amd.synthetic as syn
    from oracle import oracle as O
    sharding.init_distributed("gloo")
    ob = syn.make_object("bottle", 256, 5)
    cfg = ob["cfg"]
    idx = syn.make_pairs(256, 16, 5).astype(np.int32)
    outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
    corner, dims = O.grid_setup(ob["pc"], cfg.res)
    lo, hi = sharding.shard_pairs(idx.shape[0], rank, world)
    grid = np.zeros(tuple(dims), np.float32)
    O.ppf_voting(ob["pc"], outputs[lo:hi], np.ones(256, np.float32), idx[lo:hi], grid, corner, cfg.res, 72, True)
    g = torch.from_numpy(grid)
    sharding.allreduce_grid(g, world)
    torch.save(g, os.path.join(out_dir, f"grid{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_pair_sharded_vote_allreduce_matches_single_rank(tmp_path):
    import cppf_amd.synthetic as syn
    from oracle import oracle as O
    world = 2
    mp.spawn(_vote_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ob = syn.make_object("bottle", 256, 5)
    cfg = ob["cfg"]
    idx = syn.make_pairs(256, 16, 5).astype(np.int32)
    outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
    corner, dims = O.grid_setup(ob["pc"], cfg.res)
    full = np.zeros(tuple(dims), np.float32)
    O.ppf_voting(ob["pc"], outputs, np.ones(256, np.float32), idx, full, corner, cfg.res, 72, True)
    g0, g1 = (torch.load(os.path.join(tmp_path, f"grid{r}.pt")).numpy() for r in range(world))
    assert np.array_equal(g0, g1)                                     # every rank holds the same summed grid
    np.testing.assert_allclose(g0, full, rtol=0, atol=1e-5 * full.max())   # fp32 sums in a different order
    assert int(np.argmax(g0)) == int(np.argmax(full))


def test_shard_pairs_is_a_balanced_partition():
    for n, w in ((10, 3), (524288, 8), (5, 8), (0, 4)):
        cuts = [sharding.shard_pairs(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in cuts]
        assert max(sizes) - min(sizes) <= 1
