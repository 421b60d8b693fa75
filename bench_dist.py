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
    """Give this rank the CPUs next to its GPU: /sys/bus/pci/devices/<GPU>/local_cpulist (the GPU's NUMA node), whole physical cores
    in contiguous blocks split evenly between the ranks whose GPUs share that list (one rank: the whole list -- a process the
    scheduler parked on the other socket enqueues measurably slower); with several ranks torch's intra-op threads are capped to the
    share (at most 8: the host side of a rank is one enqueueing thread).  -> [numa node, CPUs in the share, first, last] (numa -1 /
    zeros: nothing bound -- no sysfs entry, CPPF_BENCH_NO_BIND=1).  A launcher that pinned the ranks already (the mask is narrower
    than the machine) is left alone."""
    none = [-1, 0, 0, 0]
    if os.environ.get("CPPF_BENCH_NO_BIND") or not hasattr(os, "sched_setaffinity"):
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
        os.environ["CPPF_BENCH_ORIG_CPUS"] = ",".join(str(c_) for c_ in sorted(allowed))    # (the CPU worker gets the whole machine back)
        numa, cpus = cpulist(dev.index)
        cpus = [c_ for c_ in cpus if c_ in allowed]
        n_local = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        n_dev = torch.cuda.device_count()
        sharers = [r for r in range(n_local) if cpulist(r % n_dev)[1] == cpulist(dev.index)[1]] if world > 1 else [local]
        cores = {}                                    # physical core (its lowest hardware thread) -> its hardware threads in the list
        for c_ in cpus:
            try:
                with open(f"/sys/devices/system/cpu/cpu{c_}/topology/thread_siblings_list") as f:
                    key = int(f.read().replace("-", ",").split(",")[0])
            except (OSError, ValueError):
                key = c_
            cores.setdefault(key, []).append(c_)
        keys = sorted(cores)
        if local in sharers and len(sharers) > 1:
            per = max(1, len(keys) // len(sharers))
            i = sharers.index(local)
            keys = keys[i * per:(i + 1) * per] if i < len(sharers) - 1 else keys[i * per:]
        share = sorted(c_ for k_ in keys for c_ in cores[k_])
        if not share:
            return none
        os.sched_setaffinity(0, share)
        if world > 1:
            torch.set_num_threads(max(1, min(8, len(share))))
        return [numa, len(share), min(share), max(share)]
    except (OSError, ValueError, AttributeError, RuntimeError):
        return none


def binding_dict(b):
    """bind_rank_cpus' result as it is printed"""
    return None if not b or b[0] < 0 else {"numa": int(b[0]), "cpus": int(b[1]), "range": [int(b[2]), int(b[3])]}


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
            "cpu_binding": [binding_dict(r[2:6]) for r in every]}
