"""Object-level data parallelism.

Object instances are independent (the reference's per-instance loop body, nocs/inference.py:120-339,
carries no cross-instance state), so a batch shards one object per GPU, round-robin, with no traffic
during compute and ONE collective at the end: a gather of a fixed-size result record per object.
One process per GPU (torchrun); backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU (tests).
The record is ~100 bytes per object, so the collective is latency-bound; nothing here is designed
around ring bandwidth."""
import os

import torch
import torch.distributed as dist

RECORD = 20  # f64 per object: T[3] up[3] right[3] scale[3] argmax peak n_surv object_id + 4 spare


def init_distributed(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:   # RCCL on GPUs; CPPF_DIST_BACKEND=gloo lets two ranks share one GPU when debugging on a small box
            backend = os.environ.get("CPPF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_objects(n_objects, rank, world):
    """Object j runs on rank j mod world."""
    return list(range(rank, n_objects, world))


def pack_record(object_id, pose):
    """dict from cppf_amd.inference.estimate_pose -> f64[RECORD]"""
    r = torch.zeros(RECORD, dtype=torch.float64)
    r[0:3] = torch.as_tensor(pose["T"], dtype=torch.float64)
    r[3:6] = torch.as_tensor(pose["up"], dtype=torch.float64)
    r[6:9] = torch.as_tensor(pose["right"], dtype=torch.float64)
    r[9:12] = torch.as_tensor(pose["scale"], dtype=torch.float64)
    r[12], r[13], r[14], r[15] = float(pose["argmax"]), float(pose["peak"]), float(pose["n_surv"]), float(object_id)
    return r


def gather_records(local_records, n_objects, rank, world, device=None):
    """The single end-of-batch collective.  local_records: f64[n_local, RECORD] (object id in column
    15).  Returns f64[n_objects, RECORD] in object order on every rank (all_gather: the same cost as a
    gather at this size and every rank can continue with the poses)."""
    if world == 1:
        out = local_records
    else:
        n_max = (n_objects + world - 1) // world
        dev = device if device is not None else local_records.device
        out_dev = dev
        if dist.get_backend() == "gloo":      # CPU collective (tests; two ranks sharing one GPU): stage through the host
            dev = torch.device("cpu")
        buf = torch.full((n_max, RECORD), -1.0, dtype=torch.float64, device=dev)
        buf[:local_records.shape[0]] = local_records.to(dev)
        allb = torch.empty((world * n_max, RECORD), dtype=torch.float64, device=dev)   # rank-major concatenation
        dist.all_gather_into_tensor(allb, buf)
        out = allb.to(out_dev)
        out = out[out[:, 15] >= 0]
    order = torch.argsort(out[:, 15])
    out = out[order]
    assert out.shape[0] == n_objects, f"gathered {out.shape[0]} records for {n_objects} objects"
    return out


# ----------------------------------------------------------------------------------------------------------
# Intra-object pair sharding (optional row): for a scene with fewer instances than GPUs
# the pairs of ONE object are split across ranks, every rank votes its slice into a private full grid, and the
# grids are summed with one all-reduce before the arg-max.  This is the only place the path has a real exchange
# step: f32[G] = 0.2-1.6 MB, i.e. tens of microseconds over xGMI; the summation order of the all-reduce is the
# collective's, so the grid matches the single-GPU grid to fp32 rounding (same class as the reference's atomics).
def shard_pairs(n_pairs, rank, world):
    """contiguous, balanced slice [lo, hi) of the pair list for this rank"""
    base, rem = divmod(int(n_pairs), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_grid(grid, world):
    """sum the per-rank vote grids in place (RCCL over xGMI with backend nccl; gloo in the CPU tests)"""
    if world > 1:
        dist.all_reduce(grid, op=dist.ReduceOp.SUM)
    return grid


def estimate_center_sharded(encoder, pc, pc_normal, feat, point_idxs, u_tr, cfg, corner, dims, rank, world,
                            num_rots=72, adaptive=True):
    """estimate_center() with the pair list split across `world` ranks: returns (out_idx, out_val, grid) where
    grid is the all-reduced vote grid, identical on every rank."""
    from .inference import estimate_center
    from .models import voting
    lo, hi = shard_pairs(point_idxs.shape[0], rank, world)
    _, _, _, _, grid = estimate_center(encoder, pc, pc_normal, feat, point_idxs[lo:hi].contiguous(),
                                       u_tr[lo:hi].contiguous(), cfg, corner, dims, num_rots, adaptive)
    allreduce_grid(grid, world)
    idx, val = voting.grid_argmax(grid)
    return idx, val, grid
