"""The multi-rank side of the benchmark: the self-launcher (`python bench.py --gpus N` without a launcher), the CPU binding of a rank
next to its GPU, and what the process group saw (`dist` of the printed line: backend, ranks, per-rank region times, imbalance)."""
import os
import sys

import torch

from cppf_amd import sharding

BENCH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench.py")


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) without a launcher's environment: re-run this command as N ranks under
    torch.distributed.run on this node -- rank r on GPU r over RCCL when the node has N GPUs; with fewer GPUs the ranks share them
    (rank r on GPU r mod n) and rendezvous over gloo, so that the whole multi-rank code path runs on a one-GPU box (reported as
    dist.shared_gpu).  The ranks print through this process's stdout: rank 0's JSON line stays the last line."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH] + list(argv)
    return subprocess.call(cmd, env=env)


def bind_rank_cpus(local, world, dev):
    """Give this rank the CPUs next to its GPU: /sys/bus/pci/devices/<GPU>/local_cpulist (the GPU's NUMA node), split evenly between
    the ranks whose GPUs share that list; torch's intra-op threads are capped to the share (at most 8: the host side of a rank is one
    enqueueing thread).  -> [numa node, CPUs in the share, first, last] (numa -1 / zeros: nothing bound -- no sysfs entry, one rank,
    CPPF_BENCH_NO_BIND=1).  A launcher that pinned the ranks already (the mask is narrower than the machine) is left alone."""
    none = [-1, 0, 0, 0]
    if world <= 1 or os.environ.get("CPPF_BENCH_NO_BIND") or not hasattr(os, "sched_setaffinity"):
        return none
    try:
        def cpulist(i):
            p = torch.cuda.get_device_properties(i)
            base = f"/sys/bus/pci/devices/{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0/"
            with open(base + "local_cpulist") as f:
                text = f.read().strip()
            with open(base + "numa_node") as f:
                numa = int(f.read().strip())
            cpus = []
            for part in text.split(","):
                lo, _, hi = part.partition("-")
                cpus += list(range(int(lo), int(hi or lo) + 1))
            return numa, cpus
        allowed = os.sched_getaffinity(0)
        if len(allowed) < (os.cpu_count() or 1):
            return none
        numa, cpus = cpulist(dev.index)
        cpus = [c_ for c_ in cpus if c_ in allowed]
        n_local = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        n_dev = torch.cuda.device_count()
        sharers = [r for r in range(n_local) if cpulist(r % n_dev)[1] == cpulist(dev.index)[1]]
        share = cpus[sharers.index(local)::len(sharers)] if local in sharers else cpus
        if not share:
            return none
        os.sched_setaffinity(0, share)
        torch.set_num_threads(max(1, min(8, len(share))))
        return [numa, len(share), min(share), max(share)]
    except (OSError, ValueError, AttributeError, RuntimeError):
        return none


def dist_info(world, dev, own_ms=None, binding=None):
    """Which collective library carried the gather / barrier / max-over-ranks of this run, and what the group saw (None: no process
    group).  COLLECTIVE: every rank calls it, outside the timed regions.  ranks_seen = an all-reduced 1 per rank; device_per_rank;
    shared_gpu = ranks outnumber the node's GPUs (gloo rendezvous, ranks time-share the devices: a code-path run, not a scaling
    measurement); rank_region_ms = every rank's median time for its OWN steps of a region (HIP events, before the gather) and
    imbalance = max / min of them -- a straggler shows here, host contention does not; cpu_binding per rank (bind_rank_cpus)."""
    if not torch.distributed.is_initialized():
        return None
    cd = sharding.collective_device(dev)
    one = torch.ones(1, dtype=torch.int64, device=cd)
    torch.distributed.all_reduce(one)
    med = own_ms[len(own_ms) // 2] if own_ms else 0.0
    mine = torch.tensor([float(dev.index), med] + [float(v) for v in (binding or [-1, 0, 0, 0])], dtype=torch.float64, device=cd)
    every = torch.empty(max(world, 1) * mine.numel(), dtype=torch.float64, device=cd)
    torch.distributed.all_gather_into_tensor(every, mine)
    every = every.view(max(world, 1), mine.numel()).cpu().tolist()
    devs, ms = [int(r[0]) for r in every], [r[1] for r in every]
    return {"backend": torch.distributed.get_backend(), "forced_single_rank": world == 1, "ranks_seen": int(one.item()),
            "device_per_rank": devs, "shared_gpu": len(set(devs)) < len(devs),
            "rank_region_ms": [float(f"{v:.6g}") for v in ms], "imbalance": (max(ms) / min(ms)) if min(ms) > 0 else None,
            "cpu_binding": [None if r[2] < 0 else {"numa": int(r[2]), "cpus": int(r[3]), "range": [int(r[4]), int(r[5])]} for r in every]}
