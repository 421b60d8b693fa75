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

from . import _lib

RECORD = 20  # f64 per object: T[3] up[3] right[3] scale[3] argmax peak n_surv object_id + 4 spare


def forced():
    """CPPF_FORCE_DIST=1: create the process group and take the collective branches even with ONE rank, so that the RCCL
    path (communicator set-up, f64 all_gather_into_tensor, integer all_reduce, barrier) executes on a single-GPU box."""
    return os.environ.get("CPPF_FORCE_DIST", "0") not in ("", "0")


def init_distributed(backend=None, force=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / LOCAL_WORLD_SIZE / MASTER_* from the environment (torchrun contract).  The group is
    created when WORLD_SIZE > 1, or when `force` (default: CPPF_FORCE_DIST) asks for it with a single rank.  Returns (rank, world,
    device index): LOCAL_RANK -- or, when this node has fewer GPUs than local ranks, LOCAL_RANK mod the GPUs present: the ranks then
    share devices and rendezvous over gloo (RCCL needs a GPU per rank), so that a one-GPU box runs the whole multi-rank path."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # ranks on this node: LOCAL_WORLD_SIZE (torchrun sets it); without it the job is taken to be ONE node (WORLD_SIZE) -- a value every
    # rank agrees on, so that all of them pick the same backend (a per-rank guess made rank 0 choose nccl and rank 1 gloo)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "0") or 0) or world
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    shared = 0 < n_dev < local_world
    if shared:
        local = local % n_dev
    if force is None:
        force = forced()
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:   # RCCL on GPUs; CPPF_DIST_BACKEND=gloo lets two ranks share one GPU when debugging on a small box
            backend = os.environ.get("CPPF_DIST_BACKEND") or ("nccl" if n_dev > 0 and not shared else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def collective_device(device):
    """where a tensor must live to go through the group's collectives: the GPU for RCCL, the host for gloo"""
    return torch.device("cpu") if dist.get_backend() == "gloo" else device


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


_PERM = {}


def _object_order(n_objects, world, device):
    """row of the rank-major all_gather result that holds object j: rank j mod W, slot j // W (cached per shape and device)"""
    key = (int(n_objects), int(world), str(device))
    perm = _PERM.get(key)
    if perm is None:
        n_max = (n_objects + world - 1) // world
        j = torch.arange(n_objects, dtype=torch.int64)
        perm = ((j % world) * n_max + j // world).to(device)
        _PERM[key] = perm
    return perm


def gather_records(local_records, n_objects, rank, world, device=None, force_collective=None, validate=False):
    """The single end-of-batch collective.  local_records: f64[n_local, RECORD], row s = object `rank + s * world` (the order
    shard_objects() hands the objects out in; object id in column 15).  Returns f64[n_objects, RECORD] in object order on
    every rank (all_gather: the same cost as a gather at this size and every rank can continue with the poses).

    The layout is FIXED: every rank contributes ceil(n_objects / world) rows (unused ones zero), the result is the rank-major
    concatenation, and object j is row (j mod W) * n_max + j // W of it -- one index_select with a cached permutation, no
    data-dependent filter, no sort, no host synchronisation.  validate=True (opt-in, the same for every world size) re-checks on the
    host that row j carries object id j; all-zero rows -- objects a caller skipped -- pass.
    force_collective (default: CPPF_FORCE_DIST when a group exists): run the collective with a single rank too."""
    if force_collective is None:
        force_collective = forced() and dist.is_initialized()
    n_local = len(shard_objects(n_objects, rank, world))
    if local_records.shape[0] < n_local:
        raise ValueError(f"rank {rank} holds {local_records.shape[0]} records, its share of {n_objects} objects is {n_local}")
    if world == 1 and not force_collective:
        out = local_records[:n_objects]    # (the shortcut re-orders nothing: a caller unsure of its row order passes validate=True)
    else:
        n_max = (n_objects + world - 1) // world
        out_dev = device if device is not None else local_records.device
        dev = collective_device(out_dev)      # gloo (CPU tests; two ranks sharing one GPU): staged through the host
        buf = torch.zeros((n_max, RECORD), dtype=torch.float64, device=dev)
        buf[:n_local] = local_records[:n_local].to(dev)
        allb = torch.empty((world * n_max, RECORD), dtype=torch.float64, device=dev)   # rank-major concatenation
        dist.all_gather_into_tensor(allb, buf)
        out = allb.index_select(0, _object_order(n_objects, world, dev)).to(out_dev)
    if validate:
        host = out.cpu()
        ids, filled = host[:, 15], (host != 0).any(dim=1)
        if not torch.equal(ids[filled], torch.arange(n_objects, dtype=torch.float64)[filled]):
            raise AssertionError(f"gathered records are not in object order: {ids.tolist()}")
    return out


# ----------------------------------------------------------------------------------------------------------
# Intra-object pair sharding (optional row; BASELINE.json configs[4] "8-GPU shard"): for a scene with
# fewer instances than GPUs the pairs of ONE object are split across ranks, every rank votes its slice into a private full
# grid, and the grids are summed with one all-reduce before the arg-max.  This is the only place the path has a real
# exchange step.  What is exchanged is the vote's EXACT INTEGER image (i64 quanta per cell, cppf_vote_grid_raw; 0.4-9.8 MB,
# i.e. tens of microseconds over xGMI): integer addition is associative, every rank quantises with the same fixed-point
# bits, so the summed grid -- and with it the arg-max -- is the single-GPU grid bit for bit whatever order the collective
# adds in (north_star: "bit-exact vote-grid argmax indices").  Round 1-3 all-reduced fp32 grids: equal to fp32 rounding only.
def shard_pairs(n_pairs, rank, world):
    """contiguous, balanced slice [lo, hi) of the pair list for this rank"""
    base, rem = divmod(int(n_pairs), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_grid(grid, world, force_collective=None):
    """sum the per-rank vote grids in place (RCCL over xGMI with backend nccl; gloo in the CPU tests, staged through the host
    when the grid lives on a GPU).  i64 grids (vote_grid_raw) sum exactly; f32 grids to rounding."""
    if force_collective is None:
        force_collective = forced() and dist.is_initialized()
    if world > 1 or force_collective:
        dev = collective_device(grid.device)
        if dev == grid.device:
            dist.all_reduce(grid, op=dist.ReduceOp.SUM)
        else:
            host = grid.to(dev)
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            grid.copy_(host)
    return grid


def vote_sharded(pc, outputs, point_idxs, corner, dims, res, n_pairs_total, world, num_rots=72, adaptive=True, probs=None,
                 force_collective=None):
    """This rank's slice of the pairs (outputs / point_idxs rows) voted as integers, all-reduced, converted once.
    Returns (out_idx i64[1], out_val f32[1], grid f32[dims], quantum f32[1]); identical on every rank and identical to the
    same call with world = 1 on the whole list.  quantum == 0 flags a launch that could not vote in integers."""
    from .models import voting
    dev = pc.device
    # HARD LIMIT, checked before any collective runs (a rank that raised later would leave its peers waiting in the all-reduce):
    # the integer image exists on the tiled vote only -- grids of up to 64 LDS tiles (1.9 M cells; every category of the reference
    # at its resolution needs <= 16) and probs that are finite and non-negative.  Larger grids: vote unsharded (cppf_vote_argmax
    # falls back to global fp32 atomics there) or coarsen `res`.
    if int(_lib.lib().cppf_vote_tiles(int(dims[0]), int(dims[1]), int(dims[2]))) == 0:
        raise _lib.CppfError(f"vote_sharded: a grid of {tuple(int(d) for d in dims)} cells needs more than 64 LDS tiles; the pair-sharded "
                             "vote all-reduces the tiled vote's integer image and has no form for such grids")
    bits = voting.vote_fixed_point_bits(n_pairs_total, num_rots, dims)      # of the WHOLE list: safe for every slice
    raw = torch.empty(tuple(int(d) for d in dims), dtype=torch.int64, device=dev)
    quantum = torch.empty(1, dtype=torch.float32, device=dev)
    # (an empty slice -- more ranks than pairs -- yields a zero image with quantum +inf, which the MIN below ignores)
    voting.vote_grid_raw(pc, outputs, probs, point_idxs, raw, quantum, corner, res, num_rots, adaptive, fixed_bits=bits)
    allreduce_grid(raw, world, force_collective)
    if world > 1 or force_collective or (force_collective is None and forced() and dist.is_initialized()):
        qmin = quantum.clone()                                                 # a rank that fell back to fp32 poisons the result
        qd = collective_device(dev)
        qh = qmin.to(qd)
        dist.all_reduce(qh, op=dist.ReduceOp.MIN)
        quantum = qh.to(dev)
    grid, idx, val = voting.grid_from_raw(raw, quantum)
    return idx, val, grid, quantum


def estimate_center_sharded(encoder, pc, pc_normal, feat, point_idxs, u_tr, cfg, corner, dims, rank, world,
                            num_rots=72, adaptive=True, force_collective=None):
    """estimate_center() with the pair list split across `world` ranks: PPF + MLP + decode of this rank's slice, integer
    vote, one all-reduce, arg-max.  Returns (out_idx, out_val, grid): the same bits on every rank and for every world."""
    P = point_idxs.shape[0]
    lo, hi = shard_pairs(P, rank, world)
    idx = point_idxs[lo:hi].contiguous()
    if hi > lo:
        outputs, _ = encoder.forward_decode(pc, pc_normal, feat, idx, u_tr[lo:hi].contiguous(), cfg.vote_range, None,
                                            cfg.tr_num_bins, cfg.rot_num_bins)
    else:
        outputs = torch.empty((0, 2), dtype=torch.float32, device=pc.device)
    idx_, val, grid, _ = vote_sharded(pc, outputs, idx, corner, dims, cfg.res, P, world, num_rots, adaptive,
                                      force_collective=force_collective)
    return idx_, val, grid
