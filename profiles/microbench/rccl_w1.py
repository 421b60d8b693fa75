"""RCCL at world size 1 on one MI355X: the collectives cppf_amd/sharding.py issues (backend "nccl" = RCCL), executed for real --
communicator set-up, the f64 all_gather_into_tensor of the result records, the i64 all_reduce of a vote grid image, barrier --
with their latencies.  One rank has nobody to talk to, so these are the FIXED costs (launch + RCCL kernel + stream sync) every
world size pays; link time comes on top at W > 1.      python profiles/microbench/rccl_w1.py > profiles/r4_rccl_w1.txt"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("CPPF_FORCE_DIST", "1")
os.environ.setdefault("MASTER_PORT", "29533")
from cppf_amd import sharding  # noqa: E402


def timed(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


def main():
    t0 = time.perf_counter()
    rank, world, local = sharding.init_distributed()
    dev = torch.device("cuda", local)
    x = torch.zeros(1, device=dev)
    dist.all_reduce(x)                      # communicator creation happens on first use
    torch.cuda.synchronize()
    print(f"backend {dist.get_backend()}  world {world}  device {torch.cuda.get_device_name(dev)}")
    print(f"init_process_group + first collective (communicator set-up): {(time.perf_counter() - t0) * 1e3:.1f} ms")
    print("median / min / max in microseconds, host wall clock around call + torch.cuda.synchronize(), 200 calls each")
    for n_obj in (1, 8, 20, 64):
        recs = torch.rand((n_obj, sharding.RECORD), dtype=torch.float64, device=dev)
        recs[:, 15] = torch.arange(n_obj, device=dev).double()
        out = sharding.gather_records(recs, n_obj, 0, 1, dev, force_collective=True, validate=True)
        assert torch.equal(out, recs)
        m = timed(lambda: sharding.gather_records(recs, n_obj, 0, 1, dev, force_collective=True))
        print(f"gather_records (f64 all_gather_into_tensor + index_select), {n_obj:3d} records of 160 B: {m[0]:8.1f} {m[1]:8.1f} {m[2]:8.1f}")
    for dims, name in (((26, 76, 26), "C2 grid 26x76x26"), ((52, 152, 52), "C5 grid 52x152x52")):
        g = torch.randint(0, 1 << 40, dims, dtype=torch.int64, device=dev)
        ref = g.clone()
        sharding.allreduce_grid(g, 1, force_collective=True)
        assert torch.equal(g, ref)
        m = timed(lambda: sharding.allreduce_grid(g, 1, force_collective=True))
        print(f"allreduce_grid i64 ({name}, {g.numel() * 8 / 1e6:.2f} MB): {m[0]:8.1f} {m[1]:8.1f} {m[2]:8.1f}")
    m = timed(lambda: dist.barrier())
    print(f"barrier: {m[0]:8.1f} {m[1]:8.1f} {m[2]:8.1f}")
    t = torch.zeros(1, dtype=torch.float64, device=dev)
    m = timed(lambda: dist.all_reduce(t, op=dist.ReduceOp.MAX))
    print(f"all_reduce MAX of one f64 (bench.py's max-over-ranks time): {m[0]:8.1f} {m[1]:8.1f} {m[2]:8.1f}")
    m = timed(lambda: torch.cuda.synchronize())
    print(f"(torch.cuda.synchronize alone: {m[0]:8.1f} {m[1]:8.1f} {m[2]:8.1f})")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
