"""Drop-in for the reference's models/voting.py: the three module-level kernel objects
`ppf_kernel`, `backvote_kernel`, `rot_voting_kernel`, called exactly like the CuPy RawKernels
they replace (models/voting.py:4,70,115):

    kernel((grid, 1, 1), (block, 1, 1), (arg0, arg1, ...))   ->   None

with the same positional argument order.  Differences a caller sees:
  * array arguments are torch tensors on a HIP device (replace `cp.asarray(x)` by a torch tensor) or any object exposing
    `__cuda_array_interface__` (wrapped zero-copy);
    dtype/shape/contiguity mistakes raise TypeError/ValueError instead of corrupting memory;
  * the (grid, block) tuples are accepted and ignored -- launch geometry is the library's business
    (the reference's own grid for ppf_voting is sized by N**2, nocs/inference.py:192);
  * work is enqueued on the current torch stream; outputs are updated in place, as before.
`findpeak_kernel` is not provided: it is dead code in the reference (never launched, and its source
includes a file that does not exist).
"""
import torch

from .. import _lib
from .._torch_util import dev_tensor, require_cuda, scalar, stream_ptr, workspace

__all__ = ["ppf_kernel", "backvote_kernel", "rot_voting_kernel", "vote_argmax", "vote_argmax_dyn", "grid_argmax",
           "vote_grid_raw", "grid_from_raw", "vote_fixed_point_bits"]

F32, I32 = torch.float32, torch.int32


def _as_device_tensor(a):
    """array arguments may be any object that exposes device memory -- a torch tensor, or anything with
    `__cuda_array_interface__` (CuPy-style arrays, numba device arrays, rocm-aware buffers): wrapped zero-copy, so in-place
    outputs (grid_obj, output_ocs, candidates) land in the caller's memory.  Scalars and host objects pass through unchanged."""
    if isinstance(a, torch.Tensor) or not hasattr(a, "__cuda_array_interface__"):
        return a
    # (no device argument: torch reads the device of the pointer from the interface, so an array that lives on another GPU than
    # the current one is wrapped where it is instead of being COPIED to the current device -- in-place outputs would land in a
    # temporary and the caller's array stay zero)
    t = torch.as_tensor(a)
    ptr = a.__cuda_array_interface__["data"][0]
    if not t.is_cuda or (t.numel() > 0 and t.data_ptr() != ptr):
        raise ValueError("could not wrap the __cuda_array_interface__ array zero-copy (a strided or read-only view?): pass a "
                         "C-contiguous device array")
    return t


class _Kernel:
    def __init__(self, name, nargs, fn):
        self.name, self._nargs, self._fn = name, nargs, fn

    def __call__(self, grid, block, args, **kw):
        if kw:
            raise TypeError(f"{self.name}: unexpected keyword arguments {sorted(kw)}")
        if len(args) != self._nargs:
            raise TypeError(f"{self.name}: expected {self._nargs} kernel arguments, got {len(args)}")
        require_cuda()
        self._fn(*(_as_device_tensor(a) for a in args))
        return None

    def __repr__(self):
        return f"<cppf_amd kernel {self.name}>"


def _dims(gx, gy, gz, grid=None):
    gx, gy, gz = int(scalar(gx)), int(scalar(gy)), int(scalar(gz))
    if min(gx, gy, gz) < 1:
        raise ValueError(f"grid dims must be positive, got {(gx, gy, gz)}")
    if grid is not None and grid.numel() != gx * gy * gz:
        raise ValueError(f"grid_obj has {grid.numel()} cells, dims say {gx}x{gy}x{gz}")
    return gx, gy, gz


def _ppf_voting(points, outputs, probs, point_idxs, grid_obj, corner, res, n_ppfs, n_rots, gx, gy, gz, adaptive):
    """models/voting.py:8-66 (launch: nocs/inference.py:197-205)"""
    dev = dev_tensor(points, F32, "points", (3,)).device
    n_ppfs, n_rots = int(scalar(n_ppfs)), int(scalar(n_rots))
    dev_tensor(outputs, F32, "outputs", (2,), dev)
    dev_tensor(probs, F32, "probs", None, dev)
    dev_tensor(point_idxs, I32, "point_idxs", (2,), dev)
    dev_tensor(grid_obj, F32, "grid_obj", None, dev)
    dev_tensor(corner, F32, "corner", None, dev)
    gx, gy, gz = _dims(gx, gy, gz, grid_obj)
    if probs.numel() != points.shape[0]:
        raise ValueError("probs must have one entry per point")
    if outputs.shape[0] < n_ppfs or point_idxs.shape[0] < n_ppfs or n_ppfs < 0:
        raise ValueError("n_ppfs exceeds the outputs/point_idxs arrays")
    L = _lib.lib()
    need = L.cppf_vote_workspace_bytes(n_ppfs, n_rots, gx, gy, gz)
    if need == 0:
        raise ValueError(f"n_rots must be in 1..360, got {n_rots}")
    ws = workspace(need, dev, "vote", zero=True)
    with torch.cuda.device(dev):
        rc = L.cppf_ppf_voting(points.data_ptr(), outputs.data_ptr(), probs.data_ptr(), point_idxs.data_ptr(),
                               grid_obj.data_ptr(), corner.data_ptr(), float(scalar(res)), points.shape[0], n_ppfs,
                               n_rots, gx, gy, gz, 1 if bool(scalar(adaptive)) else 0, ws.data_ptr(), ws.numel(),
                               stream_ptr(dev))
    _lib.check(rc, "cppf_ppf_voting")


def _vote_flags(accumulate, workgroups):
    """the flags word of cppf_vote_argmax / _dyn (include/cppf.h): CPPF_VOTE_ACCUMULATE | CPPF_VOTE_WORKGROUPS(n)"""
    w = int(workgroups or 0)
    if w and not 64 <= w <= 256:
        raise ValueError(f"workgroups must be 0 (one per CU) or in 64..256, got {workgroups}")
    return (1 if accumulate else 0) | (w << 8)


def vote_argmax(points, outputs, probs, point_idxs, grid_obj, corner, res, n_rots, adaptive, out_idx=None,
                out_val=None, accumulate=True, workgroups=0):
    """ppf_voting + np.argmax (nocs/inference.py:197-208) without the host round trip.
    accumulate=True: grid_obj += votes (reference semantics, grid zero-initialised by the caller);
    accumulate=False: grid_obj = votes (no memset needed).
    workgroups: 0 = one vote workgroup per CU (fastest for one instance on an idle chip); 64..256 = at most that many, for
    callers that keep several instances in flight on different streams (cppf.h: CPPF_VOTE_WORKGROUPS).
    point_idxs may be int32 (as the reference passes it) or the original int64 pair list.
    Returns (out_idx i64[1], out_val f32[1]) device tensors."""
    dev = dev_tensor(points, F32, "points", (3,)).device
    dev_tensor(outputs, F32, "outputs", (2,), dev)
    if probs is not None:                         # None: all ones (nocs/inference.py:201), no tensor is read
        dev_tensor(probs, F32, "probs", None, dev)
    i64 = isinstance(point_idxs, torch.Tensor) and point_idxs.dtype == torch.int64
    dev_tensor(point_idxs, torch.int64 if i64 else I32, "point_idxs", (2,), dev)
    dev_tensor(grid_obj, F32, "grid_obj", None, dev)
    dev_tensor(corner, F32, "corner", None, dev)
    if grid_obj.dim() != 3:
        raise ValueError("grid_obj must be [gx,gy,gz]")
    if probs is not None and probs.numel() != points.shape[0]:
        raise ValueError("probs must have one entry per point")
    gx, gy, gz = grid_obj.shape
    n_ppfs = point_idxs.shape[0]
    if out_idx is None:
        out_idx = torch.empty(1, dtype=torch.int64, device=dev)
    if out_val is None:
        out_val = torch.empty(1, dtype=F32, device=dev)
    L = _lib.lib()
    need = L.cppf_vote_workspace_bytes(n_ppfs, int(n_rots), gx, gy, gz)
    if need == 0:
        raise ValueError(f"n_rots must be in 1..360, got {n_rots}")
    ws = workspace(need, dev, "vote", zero=True)
    with torch.cuda.device(dev):
        rc = L.cppf_vote_argmax(points.data_ptr(), outputs.data_ptr(), None if probs is None else probs.data_ptr(), point_idxs.data_ptr(),
                                1 if i64 else 0, grid_obj.data_ptr(), corner.data_ptr(), float(scalar(res)), points.shape[0], n_ppfs,
                                int(n_rots), gx, gy, gz, 1 if adaptive else 0, _vote_flags(accumulate, workgroups),
                                out_idx.data_ptr(), out_val.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr(dev))
    _lib.check(rc, "cppf_vote_argmax")
    return out_idx, out_val


def vote_argmax_dyn(points, outputs, probs, point_idxs, grid_flat, shape, corner, res, n_rots, adaptive, out_idx, out_val,
                    many_tiles=False, accumulate=False, workgroups=0):
    """vote_argmax for a captured, shape-polymorphic chain (cppf_vote_argmax_dyn): `shape` is a device i32[4]
    {n_points, gx, gy, gz}; `points`/`probs` are capacity-sized (rows beyond n_points are never read), `grid_flat` is a
    flat f32 buffer whose first gx*gy*gz cells receive the grid in the usual C order.  Same results as vote_argmax on
    the real shape.  A record exceeding a capacity yields out_idx = -1."""
    dev = dev_tensor(points, F32, "points", (3,)).device
    dev_tensor(outputs, F32, "outputs", (2,), dev)
    if probs is not None:
        dev_tensor(probs, F32, "probs", None, dev)
    i64 = point_idxs.dtype == torch.int64
    dev_tensor(point_idxs, torch.int64 if i64 else I32, "point_idxs", (2,), dev)
    dev_tensor(grid_flat, F32, "grid_flat", None, dev)
    dev_tensor(shape, I32, "shape", None, dev)
    dev_tensor(corner, F32, "corner", None, dev)
    if shape.numel() < 4 or (probs is not None and probs.numel() != points.shape[0]):
        raise ValueError("shape must be i32[4]; probs must have one entry per (capacity) point")
    L = _lib.lib()
    many_tiles = _lib.tile_class(many_tiles)
    ws = workspace(L.cppf_vote_workspace_bytes_dyn_pairs(many_tiles, int(point_idxs.shape[0])), dev, "vote_dyn", zero=True)
    with torch.cuda.device(dev):
        rc = L.cppf_vote_argmax_dyn(points.data_ptr(), outputs.data_ptr(), None if probs is None else probs.data_ptr(), point_idxs.data_ptr(),
                                    1 if i64 else 0, grid_flat.data_ptr(), grid_flat.numel(), corner.data_ptr(),
                                    float(scalar(res)), points.shape[0], point_idxs.shape[0], int(n_rots), shape.data_ptr(),
                                    many_tiles, 1 if adaptive else 0, _vote_flags(accumulate, workgroups),
                                    out_idx.data_ptr(), out_val.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr(dev))
    _lib.check(rc, "cppf_vote_argmax_dyn")
    return out_idx, out_val


def vote_batch_workgroups(n_items, workgroups=0):
    """vote workgroups each object of a vote_argmax_batch call gets (256 / n_items, at least 64, unless `workgroups` says otherwise)"""
    w = int(workgroups or 0)
    if w and not 64 <= w <= 256:
        raise ValueError(f"workgroups must be 0 or in 64..256, got {workgroups}")
    return int(_lib.lib().cppf_vote_batch_workgroups(int(n_items), w << 8))


def vote_argmax_batch(items, n_rots, adaptive, accumulate=False, workgroups=0, ws_tag="vote_b", workspaces_out=None):
    """vote_argmax / vote_argmax_dyn for up to 8 objects enqueued together (cppf_vote_argmax_batch): the objects share ONE vote
    launch and ONE reduce launch, each on 256 / len(items) workgroups (at least 64; `workgroups` overrides); an object whose grid
    needs >= 4 LDS tiles (a posed object) gets its own binning launch first.  `items`: dicts
    {points, outputs, point_idxs, grid, corner, res, out_idx, out_val[, probs][, shape, many_tiles]} -- `grid` f32[gx,gy,gz], or with
    `shape` (device i32[4] {n_points, gx, gy, gz}) a flat capacity buffer as for vote_argmax_dyn.  Every object keeps its own vote
    workspace (scratch tag ws_tag + index in the current workspace scope).  Same grids, arg-max and peaks as the single calls, at
    any width (the fixed-point scale does not follow the width)."""
    if not 1 <= len(items) <= 8:
        raise ValueError("1 to 8 objects per call")
    L = _lib.lib()
    w = int(workgroups or 0)
    if w and not 64 <= w <= 256:
        raise ValueError(f"workgroups must be 0 or in 64..256, got {workgroups}")
    flags = (1 if accumulate else 0) | (w << 8)
    arr = (_lib.VoteItem * len(items))()
    keep = []
    dev = items[0]["points"].device
    for i, it in enumerate(items):
        points = dev_tensor(it["points"], F32, "points", (3,), dev)
        outputs = dev_tensor(it["outputs"], F32, "outputs", (2,), dev)
        probs = it.get("probs")
        if probs is not None:
            dev_tensor(probs, F32, "probs", None, dev)
            if probs.numel() != points.shape[0]:
                raise ValueError("probs must have one entry per point")
        idx = it["point_idxs"]
        i64 = isinstance(idx, torch.Tensor) and idx.dtype == torch.int64
        dev_tensor(idx, torch.int64 if i64 else I32, "point_idxs", (2,), dev)
        grid = dev_tensor(it["grid"], F32, "grid", None, dev)
        corner = dev_tensor(it["corner"], F32, "corner", None, dev)
        out_idx = dev_tensor(it["out_idx"], torch.int64, "out_idx", None, dev)
        out_val = dev_tensor(it["out_val"], F32, "out_val", None, dev)
        shape = it.get("shape")
        n_ppfs = idx.shape[0]
        a = arr[i]
        if shape is not None:
            dev_tensor(shape, I32, "shape", None, dev)
            many = _lib.tile_class(it.get("many_tiles") or 0)
            ws = workspace(L.cppf_vote_workspace_bytes_dyn_pairs(many, int(n_ppfs)), dev, f"{ws_tag}dyn{i}", zero=True)
            a.shape_dev, a.grid_capacity, a.many_tiles = shape.data_ptr(), grid.numel(), many
            a.gx = a.gy = a.gz = 1
        else:
            if grid.dim() != 3:
                raise ValueError("grid must be [gx,gy,gz] (or pass `shape` for a capacity buffer)")
            gx, gy, gz = (int(v) for v in grid.shape)
            need = L.cppf_vote_workspace_bytes(n_ppfs, int(n_rots), gx, gy, gz)
            if need == 0:
                raise ValueError(f"n_rots must be in 1..360, got {n_rots}")
            ws = workspace(need, dev, f"{ws_tag}{i}", zero=True)
            a.shape_dev, a.grid_capacity, a.many_tiles = None, 0, 0
            a.gx, a.gy, a.gz = gx, gy, gz
        a.points, a.outputs, a.probs = points.data_ptr(), outputs.data_ptr(), (None if probs is None else probs.data_ptr())
        a.point_idxs, a.idx_is_i64, a.grid, a.corner = idx.data_ptr(), (1 if i64 else 0), grid.data_ptr(), corner.data_ptr()
        a.out_idx, a.out_val, a.workspace, a.workspace_bytes = out_idx.data_ptr(), out_val.data_ptr(), ws.data_ptr(), ws.numel()
        a.n_points, a.n_ppfs, a.res = points.shape[0], n_ppfs, float(scalar(it["res"]))
        keep.append(ws)
    import ctypes as C
    with torch.cuda.device(dev):
        rc = L.cppf_vote_argmax_batch(len(items), C.cast(arr, C.c_void_p), int(n_rots), 1 if adaptive else 0, flags, stream_ptr(dev))
    _lib.check(rc, "cppf_vote_argmax_batch")
    if workspaces_out is not None:  # (the back-vote loads the rotation table a vote left in its workspace)
        workspaces_out[:] = keep
    return [(it["out_idx"], it["out_val"]) for it in items]


def vote_fixed_point_bits(n_ppfs, n_rots, dims):
    """fixed-point bits the tiled vote would choose (a lower bound on the binned path) for a pair list of n_ppfs pairs"""
    return int(_lib.lib().cppf_vote_fixed_point_bits(int(n_ppfs), int(n_rots), int(dims[0]), int(dims[1]), int(dims[2])))


def vote_grid_raw(points, outputs, probs, point_idxs, grid_raw, quantum, corner, res, n_rots, adaptive, fixed_bits=0,
                  accumulate=False):
    """The centre vote as exact integers (cppf_vote_grid_raw): grid_raw i64[gx,gy,gz] = every cell's sum of deposited
    quanta, quantum f32[1] = value of one quantum (0: not valid).  Ranks voting slices of one pair list pass the same
    fixed_bits, all-reduce grid_raw as integers and convert once (grid_from_raw): the multi-GPU grid is then the
    single-GPU grid bit for bit (cppf_amd/sharding.py)."""
    dev = dev_tensor(points, F32, "points", (3,)).device
    dev_tensor(outputs, F32, "outputs", (2,), dev)
    if probs is not None:
        dev_tensor(probs, F32, "probs", None, dev)
    i64 = point_idxs.dtype == torch.int64
    dev_tensor(point_idxs, torch.int64 if i64 else I32, "point_idxs", (2,), dev)
    dev_tensor(grid_raw, torch.int64, "grid_raw", None, dev)
    dev_tensor(quantum, F32, "quantum", None, dev)
    dev_tensor(corner, F32, "corner", None, dev)
    if grid_raw.dim() != 3:
        raise ValueError("grid_raw must be [gx,gy,gz]")
    gx, gy, gz = grid_raw.shape
    n_ppfs = point_idxs.shape[0]
    L = _lib.lib()
    need = L.cppf_vote_workspace_bytes(n_ppfs, int(n_rots), gx, gy, gz)
    if need == 0:
        raise ValueError(f"n_rots must be in 1..360, got {n_rots}")
    ws = workspace(need, dev, "vote", zero=True)
    with torch.cuda.device(dev):
        rc = L.cppf_vote_grid_raw(points.data_ptr(), outputs.data_ptr(), None if probs is None else probs.data_ptr(),
                                  point_idxs.data_ptr(), 1 if i64 else 0, grid_raw.data_ptr(), quantum.data_ptr(), corner.data_ptr(),
                                  float(scalar(res)), points.shape[0], n_ppfs, int(n_rots), gx, gy, gz, 1 if adaptive else 0,
                                  1 if accumulate else 0, int(fixed_bits), ws.data_ptr(), ws.numel(), stream_ptr(dev))
    _lib.check(rc, "cppf_vote_grid_raw")


def grid_from_raw(grid_raw, quantum, grid=None, out_idx=None, out_val=None):
    """grid = (float)(grid_raw * quantum) -- the one rounding vote_argmax applies -- and its arg-max: (grid, idx, val)"""
    dev = dev_tensor(grid_raw, torch.int64, "grid_raw").device
    dev_tensor(quantum, F32, "quantum", None, dev)
    if grid is None:
        grid = torch.empty(grid_raw.shape, dtype=F32, device=dev)
    dev_tensor(grid, F32, "grid", None, dev)
    if grid.numel() != grid_raw.numel():
        raise ValueError("grid and grid_raw must have the same number of cells")
    if out_idx is None:
        out_idx = torch.empty(1, dtype=torch.int64, device=dev)
    if out_val is None:
        out_val = torch.empty(1, dtype=F32, device=dev)
    ws = workspace(256, dev, "argmax")
    with torch.cuda.device(dev):
        rc = _lib.lib().cppf_grid_from_raw(grid_raw.data_ptr(), grid_raw.numel(), quantum.data_ptr(), grid.data_ptr(),
                                           out_idx.data_ptr(), out_val.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr(dev))
    _lib.check(rc, "cppf_grid_from_raw")
    return grid, out_idx, out_val


def grid_argmax(grid, out_idx=None, out_val=None):
    """np.argmax(grid) on device (nocs/inference.py:208): (idx i64[1], val f32[1]) device tensors."""
    dev = dev_tensor(grid, F32, "grid").device
    if out_idx is None:
        out_idx = torch.empty(1, dtype=torch.int64, device=dev)
    if out_val is None:
        out_val = torch.empty(1, dtype=F32, device=dev)
    ws = workspace(256, dev, "argmax")
    with torch.cuda.device(dev):
        rc = _lib.lib().cppf_grid_argmax(grid.data_ptr(), grid.numel(), out_idx.data_ptr(), out_val.data_ptr(),
                                         ws.data_ptr(), ws.numel(), stream_ptr(dev))
    _lib.check(rc, "cppf_grid_argmax")
    return out_idx, out_val


def _backvote(points, outputs, out_offsets, point_idxs, corner, res, n_ppfs, n_rots, gx, gy, gz, gt_center, tol):
    """models/voting.py:74-112 (launch: nocs/inference.py:219-228)"""
    dev = dev_tensor(points, F32, "points", (3,)).device
    n_ppfs, n_rots = int(scalar(n_ppfs)), int(scalar(n_rots))
    dev_tensor(outputs, F32, "outputs", (2,), dev)
    dev_tensor(out_offsets, F32, "out_offsets", (3,), dev)
    dev_tensor(point_idxs, I32, "point_idxs", (2,), dev)
    dev_tensor(corner, F32, "corner", None, dev)
    dev_tensor(gt_center, F32, "gt_center", None, dev)
    gx, gy, gz = _dims(gx, gy, gz)
    if min(outputs.shape[0], point_idxs.shape[0], out_offsets.shape[0]) < n_ppfs or n_ppfs < 0:
        raise ValueError("n_ppfs exceeds the outputs/point_idxs/out_offsets arrays")
    with torch.cuda.device(dev):
        rc = _lib.lib().cppf_backvote(points.data_ptr(), outputs.data_ptr(), out_offsets.data_ptr(),
                                      point_idxs.data_ptr(), corner.data_ptr(), float(scalar(res)), n_ppfs, n_rots, gx,
                                      gy, gz, gt_center.data_ptr(), float(scalar(tol)), None, stream_ptr(dev))
    _lib.check(rc, "cppf_backvote")


def _rot_voting(points, not_used, preds_rot, outputs_up, point_idxs, corner, res, n_ppfs, n_rots, gx, gy, gz):
    """models/voting.py:119-147 (launch: nocs/inference.py:268-275); `not_used`, corner, res and the
    grid dims are ignored by the reference kernel too."""
    dev = dev_tensor(points, F32, "points", (3,)).device
    n_ppfs, n_rots = int(scalar(n_ppfs)), int(scalar(n_rots))
    dev_tensor(preds_rot, F32, "preds_rot", None, dev)
    dev_tensor(outputs_up, F32, "outputs_up", None, dev)
    dev_tensor(point_idxs, I32, "point_idxs", (2,), dev)
    if preds_rot.numel() < n_ppfs or point_idxs.shape[0] < n_ppfs or n_ppfs < 0:
        raise ValueError("n_ppfs exceeds the preds_rot/point_idxs arrays")
    if outputs_up.numel() < n_ppfs * n_rots * 3:
        raise ValueError("outputs_up must hold n_ppfs*n_rots*3 floats")
    with torch.cuda.device(dev):
        rc = _lib.lib().cppf_rot_voting(points.data_ptr(), preds_rot.data_ptr(), outputs_up.data_ptr(),
                                        point_idxs.data_ptr(), n_ppfs, n_rots, stream_ptr(dev))
    _lib.check(rc, "cppf_rot_voting")


ppf_kernel = _Kernel("ppf_voting", 13, _ppf_voting)
backvote_kernel = _Kernel("backvote", 13, _backvote)
rot_voting_kernel = _Kernel("rot_voting", 12, _rot_voting)
