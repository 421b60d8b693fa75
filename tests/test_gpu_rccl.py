"""The RCCL branch of the multi-GPU path executed on the one MI355X there is: a process group with backend "nccl" (= RCCL on
ROCm) and ONE rank, CPPF_FORCE_DIST=1 making sharding.* and bench.py take their collective branches -- communicator set-up with
device_id, the f64 all_gather_into_tensor of the result records on the device, the i64 all_reduce of the vote's integer image,
barrier, the device-side max-over-ranks of bench.py.  (World > 1 on this box needs gloo: tests/test_gpu_multirank.py.)
The reference's analogue is the per-instance loop of nocs/inference.py:120; north_star: "a single RCCL gather over xGMI"."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", CPPF_FORCE_DIST="1")
    os.environ.pop("CPPF_DIST_BACKEND", None)
    import torch.distributed as dist
    import cppf_amd.synthetic as syn
    from cppf_amd import sharding
    from cppf_amd.batch import BatchPoseRunner
    from cppf_amd.inference import grid_shape
    from test_gpu_multirank import _batch, _encoders
    r, w, local = sharding.init_distributed()
    assert (r, w, local) == (0, 1, 0) and dist.is_initialized() and dist.get_backend() == "nccl"
    dev = torch.device("cuda", 0)

    # every collective of the path is counted: the test fails if a shortcut skips RCCL
    seen = {"all_gather_into_tensor": 0, "all_reduce": 0}
    for name in seen:
        orig = getattr(dist, name)

        def counted(*a, _o=orig, _n=name, **k):
            seen[_n] += 1
            assert a[0].is_cuda, f"{_n} was handed a host tensor under nccl"
            return _o(*a, **k)
        setattr(dist, name, counted)

    # 1. the end-of-batch gather: f64 records on the device through all_gather_into_tensor
    recs = torch.rand((5, sharding.RECORD), dtype=torch.float64, device=dev)
    recs[:, 15] = torch.arange(5, device=dev).double()
    out = sharding.gather_records(recs, 5, 0, 1, dev, validate=True)
    assert out.is_cuda and torch.equal(out, recs) and seen["all_gather_into_tensor"] == 1

    # 2. the batch driver end to end (its gather goes through the group) == the same batch without a group's help
    objects = _batch(n_objects=4)
    runner = BatchPoseRunner(_encoders(dev), dev)
    got = runner.run(objects, 0, 1)
    assert seen["all_gather_into_tensor"] == 2
    torch.save(got.cpu(), os.path.join(out_dir, "batch.pt"))

    # 3. the pair-sharded vote: integer image all-reduced over RCCL (i64 SUM), quantum MIN-reduced
    ob = syn.make_object("bottle", 2048, 3)
    idx = syn.make_pairs(2048, 48, 3)
    o = syn.closed_form_outputs(ob["pc"], ob["center"], idx, ob["cfg"], quantise=True)
    corners, dims = grid_shape(ob["pc"], 2e-3)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    n0 = seen["all_reduce"]
    i1, v1, g1, q1 = sharding.vote_sharded(d(ob["pc"]), d(o), d(idx), d(corners[0]), dims, 2e-3, idx.shape[0], 1)
    assert seen["all_reduce"] == n0 + 2 and float(q1) > 0
    i0, v0, g0, q0 = sharding.vote_sharded(d(ob["pc"]), d(o), d(idx), d(corners[0]), dims, 2e-3, idx.shape[0], 1, force_collective=False)
    assert torch.equal(g0, g1) and int(i0) == int(i1) and float(v0) == float(v1)
    g32 = g1.clone()
    sharding.allreduce_grid(g32, 1)                           # f32 SUM too (a one-rank sum: unchanged)
    assert torch.equal(g32, g1)

    # 4. what bench.py does around its timed region
    dist.barrier()
    tmax = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    assert float(tmax) == 1.25
    dist.destroy_process_group()
    with open(os.path.join(out_dir, "ok"), "w") as f:
        json.dump(seen, f)


def test_rccl_group_of_one_runs_every_collective_of_the_path(dev, tmp_path):
    from cppf_amd.batch import BatchPoseRunner
    from test_gpu_multirank import _batch, _encoders
    mp.spawn(_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    seen = json.load(open(os.path.join(tmp_path, "ok")))
    assert seen["all_gather_into_tensor"] >= 2 and seen["all_reduce"] >= 4
    single = BatchPoseRunner(_encoders(dev), dev).run(_batch(n_objects=4)).cpu()      # no group in this process: the shortcut
    assert torch.equal(torch.load(os.path.join(tmp_path, "batch.pt")), single)


@pytest.mark.parametrize("extra", [["--steps", "6", "--objects", "3"],
                                   ["--config", "c4", "--steps", "8", "--n-points", "1024", "--pairs-per-point", "32"]])
def test_bench_with_the_rccl_group_forced_on(extra):
    """`bench.py --gpus 1` with CPPF_FORCE_DIST=1: the line the driver's scaling run prints per N, produced through the RCCL
    branches (gather inside the timed region, barrier, device-side max over ranks) -- same schema, backend reported"""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               CPPF_FORCE_DIST="1")
    env.pop("CPPF_DIST_BACKEND", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--warmup", "1", "--no-secondary",
                        "--no-cpu-baseline", "--min-seconds", "0.2"] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["unit"] == "pairs/s"
    assert d["dist"]["backend"] == "nccl" and d["dist"]["forced_single_rank"] is True
    assert d["dist"]["ranks_seen"] == 1 and d["dist"]["device_per_rank"] == [0] and d["dist"]["shared_gpu"] is False
    assert d["regions"] >= 5 and d["region_ms_min_max"][0] <= d["ms_per_step"] * d["steps"] <= d["region_ms_min_max"][1]
