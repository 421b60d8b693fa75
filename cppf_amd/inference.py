"""Device-resident counterpart of the per-instance hot section of the reference's inference script
(nocs/inference.py:177-335; the same lines are duplicated at sunrgbd/inference.py:136-241).

The reference bounces every stage through the host (torch -> numpy -> CuPy and back, :191-231,
:265-284).  Here the whole chain is enqueued on one HIP stream and the host reads back ONE small
record at the end:

  PPF + pair MLP + decode      PPFEncoder.forward_decode            (:182-188)
  second pass on survivors     PPFEncoder.forward_decode_sel        (:236-256)
  centre vote + arg-max        cppf_vote_argmax                     (:191-208)
  T = corner + cand * res      cppf_center_from_argmax              (:209-213)
  back-vote + compaction       cppf_backvote / cppf_compact_mask    (:216-231)
  orientation vote + count     cppf_rot_sphere_count, arg-max       (:259-284)
  axis sign                    cppf_axis_sign                       (:286-303)
  scale                        cppf_scale_sum                       (:335)

Stochastic pieces of the reference are made explicit inputs: `u_tr`/`u_rot` (uniforms replacing
torch.multinomial at :186/:250/:254, indexed by ORIGINAL pair; the second MLP pass of :236 runs on the
surviving pairs only, like the reference's, through cppf_pair_mlp_decode_sel), and the random
10 000-pair subset of :277-280 is the first `max_rot_pairs` survivors in pair order (pairs are
i.i.d. uniform, so a prefix is distributed exactly like a shuffled subset).
"""
import os

import numpy as np
import torch

from . import _lib
from ._torch_util import copy_words, release_scope, require_cuda, stream_ptr, workspace, workspace_scope
from .models import voting

F32, I32 = torch.float32, torch.int32


def grid_shape(pc_host, res):
    """nocs/inference.py:194-195 on the host copy of the cloud: corners, int32((max-min)/res)+1 (cppf_host_grid_shape: one pass
    over the cloud in C; the numpy form below -- same types, same results -- serves anything that is not a C-contiguous f32[N,3])."""
    a = np.asarray(pc_host)
    if a.dtype == np.float32 and a.ndim == 2 and a.shape[1] == 3 and a.shape[0] > 0 and a.flags.c_contiguous:
        corners = np.empty((2, 3), np.float32)
        dims = np.empty(3, np.int32)
        rc = _lib.lib().cppf_host_grid_shape(a.ctypes.data, a.shape[0], float(np.float32(res)), corners.ctypes.data, dims.ctypes.data)
        if rc == 0:
            return corners, (int(dims[0]), int(dims[1]), int(dims[2]))
        if rc == -4:      # CPPF_ENONFINITE; the reference fails at np.zeros(grid_res) with the dims such a cloud gives (:196)
            raise ValueError("the cloud holds NaN / inf coordinates: no vote grid can be laid over it")
    t = np.ascontiguousarray(np.asarray(pc_host, dtype=np.float32).T)    # [3,N]: numpy reduces the long axis 20x faster than axis 0 of [N,3]
    corners = np.stack([t.min(1), t.max(1)])
    if not np.isfinite(corners).all():
        raise ValueError("the cloud holds NaN / inf coordinates: no vote grid can be laid over it")
    grid_res = ((corners[1] - corners[0]) / np.float32(res)).astype(np.int32) + 1
    return corners, tuple(int(v) for v in grid_res)


class PoseWorkspace:
    """Per-object device buffers, allocated once and reused across calls of the same size."""

    def __init__(self, device, n_pairs, dims, n_sphere, grid=None):
        self.device, self.n_pairs, self.dims, self.n_sphere = device, n_pairs, None if dims is None else tuple(dims), n_sphere
        self.grid = grid if grid is not None else torch.empty(self.dims, dtype=F32, device=device)
        self.out_idx = torch.empty(1, dtype=torch.int64, device=device)
        self.out_val = torch.empty(1, dtype=F32, device=device)
        # the 21-double result record; T64 / best_dir / sign / scale are views into it, so the kernels write the
        # record directly and the read-back is one copy
        # (the record and the sphere-bin counts share one buffer: both start every tail zeroed, with one fill)
        # layout: record 176 B | counts 2 x n_sphere i32 | ticket 16 B | survivors per chunk of 1024 pairs i32[...]; 16 B multiple
        c0 = 176 + ((2 * n_sphere * 4 + 15) & ~15)
        n_chunks = (n_pairs + 1023) // 1024
        self._tail0 = torch.zeros(c0 + 16 + ((4 * n_chunks + 15) & ~15), dtype=torch.uint8, device=device)
        self.ticket = self._tail0[c0:c0 + 16].view(I32)
        self.chunk_counts = self._tail0[c0 + 16:c0 + 16 + 4 * n_chunks].view(I32)
        self.rec = self._tail0[:168].view(torch.float64)
        self.T64 = self.rec[0:3]
        self.T32 = torch.empty(3, dtype=F32, device=device)
        self.mask = torch.empty(n_pairs, dtype=torch.uint8, device=device)
        self.surv = torch.empty(n_pairs, dtype=I32, device=device)
        self.count = torch.empty(1, dtype=I32, device=device)
        self.counts = self._tail0[176:176 + 2 * n_sphere * 4].view(I32).view(2, n_sphere)
        self.best_idx = torch.empty(2, dtype=torch.int64, device=device)
        self.best_dir = self.rec[3:9].view(2, 3)
        self.sign = self.rec[9:15].view(2, 3)
        self.scale = self.rec[15:19]
        # {theta_up, theta_right, aux_up, aux_right, sx, sy, sz, 0} per pair, written by the second MLP pass for the pairs
        # that survive the back-vote (nocs/inference.py:236-256); other rows keep whatever an earlier instance left
        self.heads = torch.zeros((n_pairs, 8), dtype=F32, device=device)
        self.probs = None
        self._sph_key = None

    def sphere(self, sph64):
        """Device copies of the sphere bins (fp32 for the count, fp64 for best_dir) and whether the bins
        are unit vectors with a monotone y column (true for fibonacci_sphere), which lets
        cppf_rot_sphere_count search a band of bins instead of all of them."""
        key = (sph64.shape, sph64.tobytes())
        if self._sph_key != key:
            s32 = sph64.astype(np.float32)
            unit = bool(np.all(np.abs(np.linalg.norm(s32.astype(np.float64), axis=-1) - 1.0) < 1e-4))
            dy = np.diff(s32[:, 1])
            self._sorted_y = (1 if np.all(dy <= 0) else (-1 if np.all(dy >= 0) else 0)) if unit else 0
            self._sph32 = torch.from_numpy(s32).to(self.device)
            self._sph64 = torch.from_numpy(sph64).to(self.device)
            self._sph_key = key
        return self._sph32, self._sph64, self._sorted_y


def estimate_center(encoder, pc, pc_normal, feat, point_idxs, u_tr, cfg, corner, dims, num_rots=72, adaptive=True,
                    u_rot=None, ws=None, idx32=None, probs=None):
    """PPF -> MLP -> decode -> centre vote -> arg-max, all on device (the benchmarked chain).
    Returns (out_idx i64[1], out_val f32[1], outputs f32[P,2], heads f32[P,8] | None, grid)."""
    require_cuda()
    dev = pc.device
    P = point_idxs.shape[0]
    if ws is None:
        ws = PoseWorkspace(dev, P, dims, 1)
    if idx32 is None:
        idx32 = point_idxs.to(I32)
    # probs=None: all ones (nocs/inference.py:201) -- the vote then reads no probs at all
    outputs, heads = encoder.forward_decode(pc, pc_normal, feat, point_idxs, u_tr, cfg.vote_range, u_rot,
                                            cfg.tr_num_bins, cfg.rot_num_bins)
    # grid = votes (overwrite mode replaces the zero-initialisation of :196)
    voting.vote_argmax(pc, outputs, probs, idx32, ws.grid, corner, cfg.res, num_rots, adaptive, ws.out_idx,
                       ws.out_val, accumulate=False)
    return ws.out_idx, ws.out_val, outputs, heads, ws.grid


_grid_class_cache = {}


def grid_class(dims):
    """(tiles, tile capacity class, cell capacity) of the shape-polymorphic vote for a grid of `dims`: launches come in three
    classes -- 0: < 4 LDS tiles (the fused vote, <= 3 tiles of cells), 16: up to 16 tiles (bin + vote with queues for 16 tiles: posed
    NOCS objects need 9-12, the C5 grid 16), 1: up to 64 (queues for 64) -- the value is what the *_dyn entry points take as
    `many_tiles` (truthy = a binned class); tiles == 0: the grid needs more than the tiled vote serves -- only the exact-shape
    pipeline runs it."""
    key = (int(dims[0]), int(dims[1]), int(dims[2]))
    hit = _grid_class_cache.get(key)
    if hit is None:
        L = _lib.lib()
        T = int(L.cppf_vote_tiles(*key))
        many = 0 if T < 4 else (16 if T <= 16 and not os.environ.get("CPPF_TILE_CLASS_64") else 1)
        hit = (T, many, _lib.tiles_cap(many) * int(L.cppf_vote_tile_cells()))
        if len(_grid_class_cache) < 4096:
            _grid_class_cache[key] = hit
    return hit


class CenterPipeline:
    """The centre chain (PPF -> MLP -> decode -> vote -> arg-max) for a fixed problem shape, with static
    device buffers and -- by default -- the three kernel launches captured once in a hipGraph, so that a
    replay costs one launch on the host instead of ~0.25 ms of Python per object.

        pipe = CenterPipeline(encoder, cfg, n_points, n_pairs, dims, device)
        pipe.load(pc, normals, feat, point_idxs, u_tr, u_rot, corner)     # host or device arrays
        idx, val = pipe.run()                                             # device i64[1], f32[1]

    `outputs` (mu, nu), `heads` and `grid` of the last run stay available as attributes.  With
    `point_encoder=` (a cppf_amd PointEncoder) the per-point features are computed on device at the head of the
    chain (kNN + SPRIN, nocs/inference.py:180-181) and `load(feat=None)` is enough.

    dynamic=True makes the pipeline SHAPE-POLYMORPHIC: `n_points` is then a capacity and `dims` either a capacity
    class (`many_tiles` bool) or any grid of that class; every instance with N <= n_points points (N >= k of the point
    encoder) and a grid of that class runs on the same captured graph -- its real shape {N, gx, gy, gz} travels in the
    device record `shape` (written by load()) and the *_dyn kernels read it.  Real scenes give every instance its own N
    (voxel de-duplication, nocs/inference.py:140-142) and its own grid (:194-195); the number of pairs is the caller's
    constant (:177).  Results equal the exact-shape pipeline's bit for bit.

    The encoders' weight images are re-checked before every run: a parameter update (optimizer step, load_state_dict)
    is re-packed into the same device buffer, which the captured launches read; a moved / resized image re-captures."""

    def __init__(self, encoder, cfg, n_points, n_pairs, dims, device, num_rots=72, adaptive=True, with_heads=True,
                 use_graph=True, point_encoder=None, dynamic=False, vote_workgroups=0, idx_i32=False):
        """vote_workgroups: 0 = the vote launches one workgroup per CU (fastest for ONE instance on an idle chip); 64..256 = at
        most that many (cppf.h: CPPF_VOTE_WORKGROUPS) -- for callers that keep several pipelines in flight on different streams,
        where fewer, longer-lived vote workgroups leave the rest of the chip to the other streams (bench.py, BatchPoseRunner).
        idx_i32: the pipeline's pair list is int32[P,2] instead of the reference's int64 (nocs/inference.py:177) -- for lists that are
        drawn on the device (a staged chain: cppf_stage_batch) every kernel that streams the list moves half the index bytes, and
        the tail needs no int32 copy; a host list loaded into it is narrowed (indices are < n_points)."""
        require_cuda()
        self.vote_workgroups = int(vote_workgroups or 0)
        self.encoder, self.cfg, self.device = encoder, cfg, device
        self.point_encoder = point_encoder
        self.num_rots, self.adaptive, self.with_heads = num_rots, adaptive, with_heads
        self.n_points, self.n_pairs, self.dynamic = int(n_points), int(n_pairs), bool(dynamic)
        F = (encoder.ppffcs[0] - 4) // 2
        z = lambda *shape, dtype=F32: torch.zeros(shape, dtype=dtype, device=device)
        # the small per-instance inputs live in ONE buffer -- points | normals | corner | shape record -- so that a caller can
        # stage and upload them with a single copy (BatchPoseRunner); pc / nrm / corner / shape are views into it
        n = self.n_points
        self._in = z(6 * n + 8)
        self.pc, self.nrm = self._in[:3 * n].view(n, 3), self._in[3 * n:6 * n].view(n, 3)
        self.feat = z(n_points, F)
        self.idx = z(n_pairs, 2, dtype=I32 if idx_i32 else torch.int64)
        self.idx32 = self.idx if idx_i32 else z(n_pairs, 2, dtype=I32)
        self._u = z(2, n_pairs, 2)                     # one buffer: device-side sampling fills both with one launch
        self.u_tr, self.u_rot = self._u[0], self._u[1]
        self.corner = self._in[6 * n:6 * n + 3]
        self.probs = None       # all ones (nocs/inference.py:201): the vote takes None for that and reads nothing
        if self.dynamic:
            if isinstance(dims, (bool, np.bool_, int, np.integer)):       # a capacity class (_lib.tile_class)
                self.many_tiles = _lib.tile_class(bool(dims) if isinstance(dims, (bool, np.bool_)) else int(dims))
            else:
                T, self.many_tiles, _ = grid_class(dims)
                if T == 0:
                    raise _lib.CppfError(f"grid {tuple(dims)} needs more LDS tiles than the shape-polymorphic vote serves")
            self.grid_flat = z(_lib.tiles_cap(self.many_tiles) * int(_lib.lib().cppf_vote_tile_cells()))
            self.shape = self._in[6 * n + 4:6 * n + 8].view(I32)               # {n_points, gx, gy, gz}, read by the *_dyn kernels
            self.shape_host = (0, 0, 0, 0)
            self.dims = None
            if point_encoder is not None:
                self._nbrs = z(n_points, point_encoder.k, dtype=I32)
                self._feat_out = z(n_points, point_encoder.out_dim + point_encoder.out_dim // 4)
        else:
            self.dims = tuple(int(v) for v in dims)
            self.grid = torch.empty(self.dims, dtype=F32, device=device)
        # arg-max index and value side by side in one 16-byte record, so a caller that logs every step moves them with
        # one small copy: `result` u8[16] = {i64 flat index, f32 peak, 4 bytes unused}
        self.result = torch.zeros(16, dtype=torch.uint8, device=device)
        self.out_idx = self.result[:8].view(torch.int64)
        self.out_val = self.result[8:12].view(F32)
        self.outputs = self.heads = None
        self._graph = None
        self._use_graph = use_graph
        self._images = None

    def set_shape(self, n_points, dims, shape_src=None, upload=True):
        """dynamic pipelines: the next run's real shape.  `shape_src`: a (pinned) host i32[4] tensor the caller has filled
        with {n_points, gx, gy, gz} -- copied asynchronously; without it a small synchronous upload is made;
        upload=False: the caller writes the record itself (it is part of the `_in` buffer)."""
        if not self.dynamic:
            raise _lib.CppfError("set_shape() is for dynamic pipelines")
        T, many, cap = grid_class(dims)
        k_min = self.point_encoder.k if self.point_encoder is not None else 1
        if not (k_min <= n_points <= self.n_points) or T == 0 or T > _lib.tiles_cap(self.many_tiles) or \
                int(dims[0]) * int(dims[1]) * int(dims[2]) > self.grid_flat.numel():
            raise _lib.CppfError(f"instance shape N={n_points}, grid {tuple(dims)} does not fit this pipeline "
                                 f"(N in {k_min}..{self.n_points}, many_tiles={self.many_tiles})")
        self.shape_host = (int(n_points), int(dims[0]), int(dims[1]), int(dims[2]))
        self.dims = self.shape_host[1:]
        if not upload:
            return
        if shape_src is None:
            shape_src = torch.tensor(self.shape_host, dtype=I32)
        self.shape.copy_(shape_src, non_blocking=True)

    @property
    def grid_view(self):
        """the vote grid of the last run as f32[gx,gy,gz]"""
        if not self.dynamic:
            return self.grid
        gx, gy, gz = self.dims
        return self.grid_flat[:gx * gy * gz].view(gx, gy, gz)

    def load(self, pc, pc_normal, feat, point_idxs, u_tr, u_rot, corner, dims=None):
        """host or device arrays -> the static buffers (None = keep).  Dynamic pipelines take clouds of any N <= capacity
        and need `dims` (grid_shape(pc, res)[1]) together with `pc`."""
        n = None if pc is None else int(pc.shape[0])
        if self.dynamic and pc is not None:
            if dims is None:
                raise ValueError("a dynamic pipeline needs dims with every cloud")
            self.set_shape(n, dims)
        for dst, src, per_point in ((self.pc, pc, True), (self.nrm, pc_normal, True), (self.feat, feat, True),
                                    (self.idx, point_idxs, False), (self.u_tr, u_tr, False), (self.u_rot, u_rot, False),
                                    (self.corner, corner, False)):
            if src is None:
                continue
            src = torch.as_tensor(src)
            if self.dynamic and per_point:
                dst = dst[:src.shape[0]]
            dst.copy_(src, non_blocking=True)

    def _chain(self):
        shape = self.shape if self.dynamic else None
        if self.point_encoder is not None:                                    # nocs/inference.py:180-181, no N x N matrix
            if self.dynamic:
                self.feat = self.point_encoder.forward_dyn(self.pc, self.nrm, shape, out=self._feat_out, nbrs=self._nbrs,
                                                           nbrs_ready=getattr(self, "nbrs_ready", False))
            else:
                self.feat = self.point_encoder(self.pc[None], self.nrm[None])[0]
        self.outputs, self.heads = self.encoder.forward_decode(
            self.pc, self.nrm, self.feat, self.idx, self.u_tr, self.cfg.vote_range,
            self.u_rot if self.with_heads else None, self.cfg.tr_num_bins, self.cfg.rot_num_bins)
        # the vote reads the int64 pair list directly (the reference copies it to int32 first, nocs/inference.py:202)
        if self.dynamic:
            voting.vote_argmax_dyn(self.pc, self.outputs, None, self.idx, self.grid_flat, shape, self.corner,
                                   self.cfg.res, self.num_rots, self.adaptive, self.out_idx, self.out_val,
                                   many_tiles=self.many_tiles, accumulate=False, workgroups=self.vote_workgroups)
        else:
            voting.vote_argmax(self.pc, self.outputs, None, self.idx, self.grid, self.corner, self.cfg.res,
                               self.num_rots, self.adaptive, self.out_idx, self.out_val, accumulate=False,
                               workgroups=self.vote_workgroups)

    def _weight_images(self):
        """(re)build the encoders' weight images if a parameter changed; returns their identity (addresses)"""
        ids = [self.encoder._packed_weights(self.device).data_ptr()]
        if self.point_encoder is not None:
            ids.append(self.point_encoder._packed_weights(self.device)[0].data_ptr())
        return tuple(ids)

    def _image_ptrs(self):
        """the addresses of the weight images as they are NOW, without looking at the parameters (a caller that has just refreshed
        the images of its encoders -- BatchPoseRunner, once per batch -- only needs to know whether an image MOVED since capture)"""
        enc, penc = self.encoder, self.point_encoder
        if enc._packed is None or (penc is not None and getattr(penc, "_packed", None) is None):
            return self._weight_images()
        return (enc._packed.data_ptr(),) if penc is None else (enc._packed.data_ptr(), penc._packed[0].data_ptr())

    def run(self, check_weights=True):
        """check_weights=None: the caller refreshed the weight images itself, only their addresses are compared (_image_ptrs).
        check_weights=False skips the per-run look at the encoders' parameters (a caller that runs many instances
        between parameter updates checks once per batch: BatchPoseRunner)."""
        # scratch requested by the chain belongs to this pipeline (see workspace_scope): pipelines replay concurrently
        with torch.no_grad(), workspace_scope(id(self)):
            if not self._use_graph:
                self._chain()
                return self.out_idx, self.out_val
            # a changed parameter is re-packed here, on this stream, into the buffer the captured launches read; if the
            # image itself moved (device change, other size) the captured addresses are stale: capture again
            images = (self._weight_images() if check_weights or self._graph is None else
                      (self._image_ptrs() if check_weights is None else self._images))
            if self._graph is not None and images != self._images:
                self._graph = None
            if self._graph is None:
                # warm up on a side stream (lazy attribute setting, weight packing, scratch allocation), then
                # capture; the captured launches read the static buffers, so later loads just change the data
                s = torch.cuda.Stream(device=self.device)
                s.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(s):
                    self._chain()
                    self._chain()
                torch.cuda.current_stream(self.device).wait_stream(s)
                self._graph = torch.cuda.CUDAGraph()
                # thread_local: other threads (the RCCL watchdog of a multi-rank run) may touch the runtime meanwhile
                with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                    self._chain()
                self._images = images
            self._await_images()
            self._graph.replay()
            self._note_images_read()
        return self.out_idx, self.out_val

    def _await_images(self):
        """another stream may have rebuilt a weight image in place since this pipeline last replayed: wait for it"""
        seen = self.__dict__.setdefault("_image_gens", {})     # per (encoder, stream): a pipeline replayed on a NEW stream must wait too
        sid = torch.cuda.current_stream(self.device).cuda_stream
        for enc in (self.encoder, self.point_encoder):
            if enc is not None:
                seen[(id(enc), sid)] = enc._await_image(self.device, seen.get((id(enc), sid)))

    def _note_images_read(self):
        """... and a later rebuild (on any stream) must wait for this replay"""
        for enc in (self.encoder, self.point_encoder):
            if enc is not None:
                enc._note_image_read(self.device)

    def set_vote_workgroups(self, n):
        """change the vote's launch width (see __init__); the chain is captured again on the next run"""
        n = int(n or 0)
        if n != self.vote_workgroups:
            self.vote_workgroups = n
            self._graph = None
            if hasattr(self, "_graphs"):
                self._graphs = {}

    def release(self):
        """drop the captured graph and this pipeline's scratch buffers (BatchPoseRunner's cache eviction)"""
        self._graph = None
        release_scope(id(self))

    def __del__(self):
        try:
            release_scope(id(self))      # scratch is keyed by id(self): it must not outlive the object
        except Exception:
            pass


def estimate_pose(encoder, pc, pc_normal, feat, point_idxs, u_tr, u_rot, cfg, sphere_pts, pc_host=None, num_rots=72,
                  adaptive=True, angle_tol=1.5, max_rot_pairs=10000, ws=None, rng=None, rot_order=None):
    """Full per-instance pose (nocs/inference.py:177-339 minus dataset I/O and the laptop segmenter).

    encoder: cppf_amd.models.model.PPFEncoder on `pc.device`, eval mode.
    pc, pc_normal f32[N,3], feat f32[N,F], point_idxs i64/i32[P,2], u_tr/u_rot f32[P,2]: device tensors.
    sphere_pts: fp64 array [S,3] (np.array(fibonacci_sphere(S)), :100-102).
    rot_order: optional i32 array / device tensor of positions in the survivor list -- the reference's shuffled subsample for
    the orientation vote (:277-280: `np.random.shuffle(arange(P'))[:10000]`); entries beyond the survivor count are skipped,
    so a caller that does not know P' yet may pass a longer list.  None: the first `max_rot_pairs` survivors.
    Returns a dict of host values: T f64[3], up/right f64[3], R f64[3,3], scale f64[3], scale_norm,
    argmax (flat grid index), peak, n_surv, counts_up/right (i32[S])."""
    require_cuda()
    dev = pc.device
    L = _lib.lib()
    st = stream_ptr(dev)
    P = point_idxs.shape[0]
    if pc_host is None:
        pc_host = pc.detach().cpu().numpy()
    corners, dims = grid_shape(pc_host, cfg.res)                              # :194-195
    corner = torch.from_numpy(corners[0].copy()).to(dev)
    sph64 = np.asarray(sphere_pts, dtype=np.float64)
    S = sph64.shape[0]
    if ws is None or ws.n_pairs != P or ws.dims != dims or ws.n_sphere != S or ws.device != dev:
        ws = PoseWorkspace(dev, P, dims, S)
    sph32_d, sph64_d, sorted_y = ws.sphere(sph64)                             # :276 (uploaded once per workspace)
    idx32 = point_idxs.to(I32)
    if rot_order is not None:
        rot_order = torch.as_tensor(rot_order).to(device=dev, dtype=I32).contiguous()

    # centre ------------------------------------------------------------------------------------------
    split = encoder.fused_decode_supported(cfg.tr_num_bins, cfg.rot_num_bins)
    _, _, outputs, heads, _ = estimate_center(encoder, pc, pc_normal, feat, point_idxs, u_tr, cfg, corner, dims,
                                              num_rots, adaptive, None if split else u_rot, ws, idx32)
    if split:       # the reference's own order: a second pass for the survivors (:236-256)
        heads = ws.heads
    # (other bin counts / architectures: every head of every pair came from the first pass, through the logits + decode kernels)
    _enqueue_tail(ws, pc, pc_normal, idx32, outputs, heads, corner, cfg, dims, num_rots, angle_tol, max_rot_pairs,
                  sph32_d, sph64_d, sorted_y, second_pass=(encoder, feat, point_idxs, u_rot) if split else None,
                  rot_order=rot_order)
    out = _assemble(ws.rec.cpu().numpy(), cfg, rng)                           # the one read-back
    out.update(dims=dims, corner=corners[0], ws=ws, outputs=outputs, heads=heads)
    return out


def _enqueue_tail(ws, pc, pc_normal, idx32, outputs, heads, corner, cfg, dims, num_rots, angle_tol, max_rot_pairs,
                  sph32_d, sph64_d, sorted_y, shape=None, second_pass=None, idx64=None, rot_order=None):
    """nocs/inference.py:209-303,335 after the centre vote, all on the current stream; leaves the 21-double
    result record in ws.rec (T[3], best_dir[2,3], sign sums[2,3], scale sums[4], argmax, peak).  `shape`: the device
    dims record of a shape-polymorphic pipeline (then `dims` is unused).  `second_pass` = (encoder, feat, point_idxs, u_rot):
    the heads of the surviving pairs are computed here, after the compaction, like the reference's second
    ppf_encoder call (:236-256); without it `heads` must already hold them.  `idx64`: the pair list as int64 -- then `idx32`
    is an OUTPUT, filled by the back-vote launch for the launches after it (no conversion pass).  `rot_order`: device i32[m],
    positions in the survivor list whose pairs feed the orientation vote (the reference's shuffled subsample, :277-280);
    None = the first `max_rot_pairs` survivors."""
    dev = pc.device
    L = _lib.lib()
    st = stream_ptr(dev)
    P, S = idx32.shape[0], sph32_d.shape[0]
    with torch.cuda.device(dev):
        # the vote's workspace (same scope, same tag as models/voting.py): its rotation table is reused by the back-vote
        vws = workspace(256, dev, "vote_dyn" if shape is not None else "vote")
        vws_ptr = vws.data_ptr() if vws.numel() >= 32768 else None
        shape_ptr = None if shape is None else shape.data_ptr()
        gx, gy, gz = (1, 1, 1) if shape is not None else dims
        # T = corner + unravel(argmax) * res (:209-210); the same launch zeroes the record, the bin counts, the chunk counts
        # and the ticket of the launches below
        _lib.check(L.cppf_pose_tail_begin(ws.out_idx.data_ptr(), corner.data_ptr(), float(cfg.res), gy, gz, shape_ptr,
                                          ws.T64.data_ptr(), ws.T32.data_ptr(), ws.out_val.data_ptr(),
                                          ws.rec[19:21].data_ptr(), ws._tail0.data_ptr(), ws._tail0.numel(), st),
                   "cppf_pose_tail_begin")
        # back-vote filter (:216-231) --------------------------------------------------------------
        # mask only: the offsets themselves (:220-228) are consumed nowhere else, so no buffer is zeroed or written
        tol = float(np.float32(3 * cfg.res))
        fused = 0 < P <= 8192 * 1024
        if idx64 is not None and not fused:
            idx32.copy_(idx64)
        if fused:                    # survivors counted per chunk by the back-vote itself: the compaction is one launch
            if idx64 is not None:
                _lib.check(L.cppf_backvote_count64(pc.data_ptr(), outputs.data_ptr(), idx64.data_ptr(), idx32.data_ptr(),
                                                   corner.data_ptr(), float(cfg.res), P, num_rots, gx, gy, gz, shape_ptr,
                                                   ws.T32.data_ptr(), tol, ws.mask.data_ptr(), ws.chunk_counts.data_ptr(),
                                                   vws_ptr, st), "cppf_backvote_count64")
            else:
                _lib.check(L.cppf_backvote_count(pc.data_ptr(), outputs.data_ptr(), idx32.data_ptr(), corner.data_ptr(),
                                                 float(cfg.res), P, num_rots, gx, gy, gz, shape_ptr, ws.T32.data_ptr(), tol,
                                                 ws.mask.data_ptr(), ws.chunk_counts.data_ptr(), vws_ptr, st),
                           "cppf_backvote_count")
            _lib.check(L.cppf_compact_scatter(ws.mask.data_ptr(), P, ws.chunk_counts.data_ptr(), ws.surv.data_ptr(),
                                              ws.count.data_ptr(), st), "cppf_compact_scatter")
        else:
            _lib.check(L.cppf_backvote_ws(pc.data_ptr(), outputs.data_ptr(), None, idx32.data_ptr(), corner.data_ptr(),
                                          float(cfg.res), P, num_rots, gx, gy, gz, shape_ptr, ws.T32.data_ptr(), tol,
                                          ws.mask.data_ptr(), vws_ptr, st), "cppf_backvote_ws")
            cws = workspace(L.cppf_compact_workspace_bytes(P), dev, "compact")
            _lib.check(L.cppf_compact_mask(ws.mask.data_ptr(), P, ws.surv.data_ptr(), ws.count.data_ptr(),
                                           cws.data_ptr(), cws.numel(), st), "cppf_compact_mask")
        if second_pass is not None:                                          # :236-256, survivors only
            enc2, feat2, idxs2, u_rot2 = second_pass
            enc2.forward_decode_sel(pc, pc_normal, feat2, idxs2, u_rot2, ws.surv, ws.count, heads, max_sel=P,
                                    tr_num_bins=cfg.tr_num_bins, rot_num_bins=cfg.rot_num_bins)
        # orientation (:259-303) and scale (:335) ---------------------------------------------------
        # heads row = {theta_up, theta_right, aux_up, aux_right, sx, sy, sz, 0}: both directions' candidates are counted in
        # one launch; np.argmax(counts), sphere_pts[...] (:283-284), the sign sums (:287-301) and the scale sums in another
        thr = float(np.float32(np.cos(angle_tol / 180 * np.pi)))
        n_dirs = 2 if cfg.regress_right else 1
        if rot_order is None:
            _lib.check(L.cppf_rot_sphere_count_dirs(pc.data_ptr(), heads.data_ptr(), 8, 1, n_dirs, idx32.data_ptr(),
                                                    ws.surv.data_ptr(), ws.count.data_ptr(), P, max_rot_pairs, num_rots,
                                                    sph32_d.data_ptr(), S, thr, sorted_y, ws.counts.data_ptr(), S, st),
                       "cppf_rot_sphere_count_dirs")
        else:
            _lib.check(L.cppf_rot_sphere_count_dirs_order(pc.data_ptr(), heads.data_ptr(), 8, 1, n_dirs, idx32.data_ptr(),
                                                          ws.surv.data_ptr(), ws.count.data_ptr(), P, rot_order.data_ptr(),
                                                          rot_order.numel(), max_rot_pairs, num_rots, sph32_d.data_ptr(), S,
                                                          thr, sorted_y, ws.counts.data_ptr(), S, st),
                       "cppf_rot_sphere_count_dirs_order")
        pws = workspace(L.cppf_pose_sums_workspace_bytes(), dev, "pose_sums")
        _lib.check(L.cppf_pose_sums(pc.data_ptr(), pc_normal.data_ptr(), idx32.data_ptr(), ws.surv.data_ptr(),
                                    ws.count.data_ptr(), P, heads.data_ptr() + 4 * 2, 8, n_dirs, ws.counts.data_ptr(), S, S,
                                    sph64_d.data_ptr(), heads.data_ptr() + 4 * 4, 8, ws.best_idx.data_ptr(),
                                    ws.best_dir.data_ptr(), ws.sign.data_ptr(), ws.scale.data_ptr(), pws.data_ptr(),
                                    pws.numel(), ws.ticket.data_ptr(), st), "cppf_pose_sums")
    # (T64, best_dir, sign, scale, arg-max index and value were written into ws.rec by the kernels above)


def _assemble(rec, cfg, rng=None):
    """Host end of nocs/inference.py:299-339 from the 21-double record."""
    T = rec[0:3].copy()
    best = rec[3:9].reshape(2, 3)
    sign = rec[9:15].reshape(2, 3)
    ssum = rec[15:19]
    flat, peak = int(rec[19]), float(rec[20])
    if flat < 0:    # cppf_vote_argmax_dyn: the instance's shape record exceeded the captured launch's capacities
        raise _lib.CppfError("the instance shape did not fit the shape-polymorphic pipeline it ran on (arg-max index -1)")
    n_surv = int(ssum[3])
    n_dirs = 2 if cfg.regress_right else 1

    dirs = []
    for j in range(n_dirs):
        n = max(sign[j, 2], 1.0)
        up_loss, down_loss = sign[j, 0] / n, sign[j, 1] / n
        dirs.append(-best[j] if down_loss < up_loss else best[j].copy())      # :299-302
    up = dirs[0]
    if cfg.regress_right:                                                     # :305-312
        right = dirs[1]
        right = right - np.dot(up, right) * up
        right = right / (np.linalg.norm(right) + 1e-9)
    else:
        right = np.array([0, -up[2], up[1]])
        right = right / (np.linalg.norm(right) + 1e-9)
    if np.linalg.norm(right) < 1e-7:                                          # :325-328
        rng = rng or np.random.default_rng(0)
        right = rng.standard_normal(3)
        right -= right.dot(up) * up
        right /= np.linalg.norm(right)
    if cfg.z_right:                                                           # :330-333
        R = np.stack([np.cross(up, right), up, right], -1)
    else:
        R = np.stack([right, up, np.cross(right, up)], -1)
    mean = (ssum[:3] / max(n_surv, 1)).astype(np.float32)                     # torch mean is fp32
    pred_scale = np.exp(mean).astype(np.float64) * np.asarray(cfg.scale_mean, np.float64) * 2   # :335
    scale_norm = float(np.linalg.norm(pred_scale))
    return dict(T=T, up=up, right=right, R=R, scale=pred_scale, scale_norm=scale_norm, argmax=flat, peak=peak,
                n_surv=n_surv, best_up=best[0].copy(), best_right=best[1].copy() if n_dirs > 1 else None,
                losses=sign.copy())


def nocs_result(poses, res=None):
    """The per-image result record of nocs/inference.py:114-117,213,338-342 from a list of pose dicts (estimate_pose /
    PosePipeline.run / _assemble): `pred_RTs` f32[n,4,4] (identity-initialised, translation in the last column, rotation
    scaled by |scale|) and `pred_scales` f32[n,3] (scale / |scale|), written into `res` (the detection dict the script
    pickles, :344-345) or a new dict -- the format nocs/eval.py reads."""
    n = len(poses)
    RTs = np.zeros((n, 4, 4), dtype=np.float32)
    for i in range(n):
        RTs[i] = np.eye(4)
    scales = np.ones((n, 3), dtype=np.float32)
    for i, p in enumerate(poses):
        if p is None:                                   # skipped instance (:121-123): identity pose, unit scale
            continue
        RTs[i][:3, -1] = p["T"]
        assert p["scale_norm"] > 0
        RTs[i][:3, :3] = p["R"] * p["scale_norm"]
        scales[i, :] = p["scale"] / p["scale_norm"]
    res = {} if res is None else res
    res["pred_RTs"] = RTs
    res["pred_scales"] = scales
    return res


class CenterBatchPipeline:
    """Several CenterPipelines replayed as ONE captured chain: the pair lists of all members in one launch of the pair kernel
    (models.model.forward_decode_batch: the ~9 us a launch spends before its first MFMA are paid once, 70.9 -> 64.0 us per C2
    list in threes), then the members' votes -- by default in ONE vote launch and ONE reduce launch as well
    (models.voting.vote_argmax_batch: every object on 256 / n workgroups, a quarter of the partial-tile traffic per object in
    fours), `vote_batch=False`: a vote + reduce launch per member at the member's own width.  Members keep their buffers and results
    (`pipes[i].out_idx`, `.result`, `.outputs`, `.grid`): load them as usual, run the batch instead of the members.  Static-shape
    members on one device, no point encoder in front, all with the rotation heads or none, the same num_rots / adaptive; up to 8.
    vote_workgroups: workgroups per object of the batched vote (None / 0: 256 / n, at least 64; 64..256)."""

    def __init__(self, pipes, use_graph=True, vote_batch=True, vote_workgroups=None, own_results=True):
        """own_results=False: the members keep the result records they have (a pipeline that is a member of a second, shorter
        chain -- bench.py's remainder chains -- keeps the views the first chain gave it; `results` is then None)"""
        pipes = list(pipes)
        if not 1 <= len(pipes) <= 8:
            raise ValueError("1 to 8 pipelines per batch")
        if any(p.dynamic or p.point_encoder is not None or p.device != pipes[0].device or p.with_heads != pipes[0].with_heads
               or p.num_rots != pipes[0].num_rots or p.adaptive != pipes[0].adaptive for p in pipes):
            raise _lib.CppfError("CenterBatchPipeline takes static-shape CenterPipelines on one device, without a point encoder, "
                                 "with the same num_rots / adaptive")
        self.pipes, self.device = pipes, pipes[0].device
        self.vote_batch, self.vote_workgroups = bool(vote_batch), int(vote_workgroups or 0)
        self._use_graph, self._graph, self._images = use_graph, None, None
        # the members' 16-byte result records side by side in ONE block (`results` u8[n,16]): a caller that logs every step moves
        # a chain's results with one small copy; the members' `result` / `out_idx` / `out_val` become views into it (their own
        # captured graphs, which wrote the old records, are dropped)
        self.results = torch.zeros((len(pipes), 16), dtype=torch.uint8, device=self.device) if own_results else None
        for i, p in enumerate(pipes if own_results else []):
            p.result = self.results[i]
            p.out_idx, p.out_val = p.result[:8].view(torch.int64), p.result[8:12].view(F32)
            p._graph = None
            if hasattr(p, "ws"):
                p.ws.out_idx, p.ws.out_val = p.out_idx, p.out_val

    def _capture_key(self):
        """what the captured launches bake in besides the weight images: the vote widths and the members' buffers"""
        return (self.vote_batch, self.vote_workgroups) + tuple((p.vote_workgroups, p.idx.data_ptr(), p.grid.data_ptr(), p.result.data_ptr())
                                                               for p in self.pipes)

    def _chain(self):
        from .models.model import forward_decode_batch
        p0 = self.pipes[0]
        items = []
        for p in self.pipes:
            it = dict(encoder=p.encoder, pc=p.pc, pc_normal=p.nrm, feat=p.feat, idxs=p.idx, u_tr=p.u_tr, vote_range=p.cfg.vote_range)
            if p.with_heads:
                it["u_rot"] = p.u_rot
            items.append(it)
        outs = forward_decode_batch(items, p0.cfg.tr_num_bins, p0.cfg.rot_num_bins)
        for p, (o, h) in zip(self.pipes, outs):
            p.outputs, p.heads = o, h
        if self.vote_batch:
            voting.vote_argmax_batch([dict(points=p.pc, outputs=p.outputs, point_idxs=p.idx, grid=p.grid, corner=p.corner, res=p.cfg.res,
                                           out_idx=p.out_idx, out_val=p.out_val) for p in self.pipes],
                                     p0.num_rots, p0.adaptive, accumulate=False, workgroups=self.vote_workgroups)
        else:
            for p in self.pipes:
                voting.vote_argmax(p.pc, p.outputs, None, p.idx, p.grid, p.corner, p.cfg.res, p.num_rots, p.adaptive, p.out_idx, p.out_val,
                                   accumulate=False, workgroups=p.vote_workgroups)

    def run(self, check_weights=True):
        """-> [(out_idx, out_val)] of the members (device tensors, as CenterPipeline.run returns them)"""
        with torch.no_grad(), workspace_scope(id(self)):
            if not self._use_graph:
                self._chain()
            else:
                # the vote widths / member buffers the launches baked in are compared on every run (cheap), the weight images
                # when asked: a moved image or a member's set_vote_workgroups() captures again
                images = ((tuple(tuple(p._weight_images()) for p in self.pipes) if check_weights or self._graph is None else self._images[0]),
                          self._capture_key())
                if self._graph is not None and images != self._images:
                    self._graph = None
                if self._graph is None:
                    s = torch.cuda.Stream(device=self.device)
                    s.wait_stream(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(s):
                        self._chain()
                        self._chain()
                    torch.cuda.current_stream(self.device).wait_stream(s)
                    self._graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                        self._chain()
                    self._images = images
                for p in self.pipes:
                    p._await_images()
                self._graph.replay()
                for p in self.pipes:
                    p._note_images_read()
        return [(p.out_idx, p.out_val) for p in self.pipes]

    def release(self):
        self._graph = None
        release_scope(id(self))

    def __del__(self):
        try:
            release_scope(id(self))
        except Exception:
            pass


class PosePipeline(CenterPipeline):
    """Full per-instance pose for a fixed problem shape (or, with dynamic=True, for every shape up to its capacities, see
    CenterPipeline): the centre chain plus the pose tail, captured together in one hipGraph; `run()` replays it and
    reads back the 21-double record (one sync)."""

    def __init__(self, encoder, cfg, n_points, n_pairs, dims, device, sphere_pts, num_rots=72, adaptive=True,
                 angle_tol=1.5, max_rot_pairs=10000, use_graph=True, point_encoder=None, dynamic=False, rot_order_len=0,
                 vote_workgroups=0, idx_i32=False):
        """rot_order_len > 0: the pipeline owns a static i32[rot_order_len] buffer `rot_order` (positions in the survivor
        list, see estimate_pose) that the captured orientation vote reads -- fill it before run() to reproduce the reference's
        shuffled subsample (nocs/inference.py:277-280); it starts as 0, 1, 2, ... (= the first survivors)."""
        super().__init__(encoder, cfg, n_points, n_pairs, dims, device, num_rots, adaptive, False, use_graph, point_encoder,
                         dynamic, vote_workgroups, idx_i32)
        self.rot_order = (torch.arange(int(rot_order_len), dtype=I32, device=device) if rot_order_len else None)
        sph64 = np.asarray(sphere_pts, dtype=np.float64)
        self.ws = PoseWorkspace(device, n_pairs, self.dims, sph64.shape[0], grid=self.grid_flat if self.dynamic else self.grid)
        self.ws.out_idx, self.ws.out_val = self.out_idx, self.out_val
        self._sph = self.ws.sphere(sph64)
        self.angle_tol, self.max_rot_pairs = angle_tol, max_rot_pairs
        # bin counts / architectures the fused second pass (cppf_pair_mlp_decode_sel) does not serve run the full-first form
        # only: every head of every pair from the first pass, whose logits + decode-kernel fallback serves any configuration
        self._split_ok = encoder.fused_decode_supported(cfg.tr_num_bins, cfg.rot_num_bins)
        self.full_first = not self._split_ok
        self._graphs = {}

    # Two captured forms of the same computation.  Split (the reference's own order, nocs/inference.py:182-256): the first MLP
    # pass decodes the two centre heads only, the survivors of the back-vote get a second pass for the orientation / scale heads
    # -- 86 + 8 us at C2 when 0.4 % survive (a network that has not learnt the object), but 86 + 94 us when all do (a trained
    # one).  Full first: the first pass decodes every head of every pair (107 us) and there is no second pass.  Per pair the
    # arithmetic is the same, so the heads rows of the survivors, hence the pose, are identical; the choice follows the share
    # of survivors of the instance that ran last on this pipeline (break-even ~0.15, with hysteresis), seen in the record the
    # host reads back anyway.
    FULL_FIRST_ON, FULL_FIRST_OFF = 0.25, 0.10

    def adapt(self, n_surv):
        """choose the form for the next instance from the survivors of the last one"""
        share = float(n_surv) / max(self.idx.shape[0], 1)
        want = self.full_first
        if share > self.FULL_FIRST_ON:
            want = True
        elif share < self.FULL_FIRST_OFF:
            want = False
        if not self._split_ok:
            want = True
        if want != self.full_first:
            # a captured form owns the tensors its launches write: outputs / heads / feat are part of what is switched
            self._graphs[self.full_first] = (self._graph, self._images, self.outputs, self.heads, self.feat)
            self.full_first = want
            self._graph, self._images, outputs, heads, feat = self._graphs.get(want, (None, self._images, None, None, None))
            if self._graph is not None:
                self.outputs, self.heads, self.feat = outputs, heads, feat

    def release(self):
        self._graphs = {}
        super().release()

    def _chain(self):
        self.with_heads = self.full_first
        super()._chain()                          # (full first: self.heads = every pair's heads row, from the first pass)
        if not self.full_first:
            self.heads = self.ws.heads
        _enqueue_tail(self.ws, self.pc, self.nrm, self.idx32, self.outputs, self.heads, self.corner, self.cfg,
                      self.dims, self.num_rots, self.angle_tol, self.max_rot_pairs, *self._sph,
                      shape=self.shape if self.dynamic else None,
                      second_pass=None if self.full_first else (self.encoder, self.feat, self.idx, self.u_rot),
                      idx64=self.idx if self.idx.dtype == torch.int64 else None,   # the tail's kernels take int32 indices: written
                                                                                  # by the back-vote launch unless the list is int32 already
                      rot_order=self.rot_order)

    def run(self, rng=None, check_weights=True):
        super().run(check_weights)
        out = _assemble(self.ws.rec.cpu().numpy(), self.cfg, rng)
        out.update(dims=self.dims, ws=self.ws, outputs=self.outputs, heads=self.heads)
        self.adapt(out["n_surv"])
        return out

    def run_async(self, record_out, check_weights=True):
        """Replay the graph and copy the 21-double record into `record_out` (device f64[21]) on the current stream:
        no host synchronisation, so a batch of instances runs back to back (BatchPoseRunner reads all records back at
        once and assembles the poses with `assemble_record`)."""
        super().run(check_weights)
        if record_out is not None:          # (None: the caller collects the records itself, e.g. several pipelines' with one launch)
            record_out.copy_(self.ws.rec, non_blocking=True)

    def sample_inputs(self, seed, n_points=None):
        """Draw the pair list and the bin-sampling uniforms on the device (the reference draws the pairs with np.random.randint on
        the host and the bins with torch.multinomial, nocs/inference.py:177,186): 17 MB per instance at C2 that never cross PCIe.
        One launch (cppf_sample_pairs: Philox keyed by `seed`, counter = pair index; a torch.Generator is accepted for its
        initial_seed()): 6 us where torch's two generator kernels took 38.  Same distribution as the reference's draws, not the
        same stream of numbers -- parity tests pass explicit arrays."""
        if self.idx.dtype != torch.int64:
            raise _lib.CppfError("sample_inputs() draws int64 pair lists (cppf_sample_pairs); an int32 pipeline is filled by a staged chain")
        n = self.shape_host[0] if self.dynamic else self.pc.shape[0]
        n = int(n_points) if n_points is not None else n
        if isinstance(seed, torch.Generator):
            seed = seed.initial_seed()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().cppf_sample_pairs(self.idx.data_ptr(), self.u_tr.data_ptr(), self.u_rot.data_ptr(), self.idx.shape[0], n,
                                                    None, int(seed) & 0xFFFFFFFFFFFFFFFF, None, stream_ptr(self.device)), "cppf_sample_pairs")


class PoseChain:
    """Up to 8 PosePipelines -- the instances of a frame on one HIP stream (the reference loops over them, nocs/inference.py:120) --
    replayed as ONE captured chain with the launches SHARED between the members: their point encoders one after the other, then ONE
    launch of the pair kernel for all pair lists (forward_decode_batch), ONE vote + ONE reduce launch (vote_argmax_batch) and the six
    launches of the batched tail (cppf_pose_tail_batch: T, back-vote, compaction, second pass, orientation vote, sums) -- about
    10 + n launches for n instances instead of ~15 each.  At the reference's own size (100 000 pairs, :177) a launch is half prologue,
    so the launches were what an instance cost.  Members keep their buffers (load / stage them as usual) and their result records
    (`pipes[i].ws.rec`); records equal the members' own runs bit for bit (at the chain's vote width).

    Members: pose pipelines of ONE device with the standard fused pair encoder, the same num_rots / adaptive / angle_tol /
    max_rot_pairs / sphere bins, no rot_order, grids of the tiled vote; static or shape-polymorphic, any categories.  Like a
    PosePipeline the chain has two captured forms -- split (second pass on the survivors) and full-first (every head in the first
    pass) -- chosen from the survivor share of its last run (adapt)."""

    FULL_FIRST_ON, FULL_FIRST_OFF = PosePipeline.FULL_FIRST_ON, PosePipeline.FULL_FIRST_OFF

    def __init__(self, pipes, use_graph=True, vote_workgroups=0, prestages=None, staged=False):
        """prestages: optional list of callables, one per member (or None): launches enqueued at the head of the chain, in front of
        the member's point encoder -- e.g. the count-driven pre-processing that fills the member's cloud, normals, corner and shape
        record from a depth frame (cppf_amd.frames.FrameRunner).
        staged=True: the chain serves objects that are ALREADY on the device (run_staged): its first launch (cppf_stage_batch) copies
        each member's cloud / normals / features from where the caller keeps them, sets up its grid and draws its pairs and bin
        uniforms -- all read from a 48-byte device descriptor per member, rewritten with one small copy per replay -- and its last
        launch assembles the members' FINISHED records (`records` f64[n, sharding.RECORD], the host end of nocs/inference.py:299-339
        on the device): nothing comes back to the host."""
        pipes = list(pipes)
        if not 1 <= len(pipes) <= 8:
            raise ValueError("1 to 8 pipelines per chain")
        self.prestages = list(prestages) if prestages is not None else [None] * len(pipes)
        if len(self.prestages) != len(pipes):
            raise ValueError("one prestage (or None) per pipeline")
        p0 = pipes[0]
        for p in pipes:
            if (not isinstance(p, PosePipeline) or p.device != p0.device or not p._split_ok or p.rot_order is not None
                    or (p.num_rots, p.adaptive, p.angle_tol, p.max_rot_pairs) != (p0.num_rots, p0.adaptive, p0.angle_tol, p0.max_rot_pairs)
                    or p.ws._sph_key != p0.ws._sph_key or p._sph[2] == 0):
                raise _lib.CppfError("PoseChain takes PosePipelines of one device with the standard pair encoder and the same vote / "
                                     "orientation settings (no rot_order)")
        if len({id(p) for p in pipes}) != len(pipes):
            raise ValueError("a pipeline can be a member once")
        self.pipes, self.device = pipes, p0.device
        self.vote_workgroups = int(vote_workgroups or 0)
        self.full_first = False
        self._use_graph, self._graphs, self._owned, self.tensors = use_graph, {}, {}, None
        self.staged = bool(staged)
        if self.staged:
            from .sharding import RECORD
            n, W = len(pipes), _lib.STAGE_DESC_WORDS
            self.desc = torch.zeros((n, W), dtype=torch.int64, device=self.device)       # CppfStageDesc per member
            self.records = torch.zeros((n, RECORD), dtype=torch.float64, device=self.device)
            # a ring of pinned descriptor blocks: a block is rewritten only after the copy that last read it has executed
            self._desc_ring = []
            for _ in range(4):
                h = torch.zeros((n, W), dtype=torch.int64).pin_memory()
                self._desc_ring.append((h, h.numpy().view(np.uint64), torch.cuda.Event()))
            self._ring_pos = 0

    def _stage(self):
        """the chain's first launch (staged chains): clouds in, grids set up, pairs and uniforms drawn (cppf_stage_batch)"""
        pipes = self.pipes
        arr = (_lib.StageItem * len(pipes))()
        for i, p in enumerate(pipes):
            a = arr[i]
            a.desc, a.pc, a.nrm, a.corner = self.desc[i].data_ptr(), p.pc.data_ptr(), p.nrm.data_ptr(), p.corner.data_ptr()
            a.feat = p.feat.data_ptr() if p.point_encoder is None else None
            a.shape = p.shape.data_ptr() if p.dynamic else None
            a.idx, a.u_tr, a.u_rot = p.idx.data_ptr(), p.u_tr.data_ptr(), p.u_rot.data_ptr()
            a.n_pairs, a.n_cap, a.F, a.res = p.idx.shape[0], p.pc.shape[0], p.feat.shape[1], float(np.float32(p.cfg.res))
            a.idx_is_i64 = 1 if p.idx.dtype == torch.int64 else 0
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().cppf_stage_batch(len(pipes), arr, stream_ptr(self.device)), "cppf_stage_batch")

    def _chain(self):
        from .models.model import forward_decode_batch
        import ctypes as C
        pipes, p0 = self.pipes, self.pipes[0]
        L = _lib.lib()
        if self.staged:                                                          # nocs/inference.py:177,194-196 for resident objects
            self._stage()
        for pre in self.prestages:                                               # nocs/inference.py:131-142 (FrameRunner)
            if pre is not None:
                pre()
        feats = []            # (the members' own attributes are left alone: a member may also run on its own captured graph)
        # nocs/inference.py:180-181.  The shape-polymorphic members' encoders share three launches (search, convolution,
        # GlobalInfoProp for all of them: a cloud of 700-2000 points fills a quarter of the chip); others run one after the other.
        enc_members = [p for p in pipes if p.point_encoder is not None and p.dynamic]
        batched = None
        if len(enc_members) > 1 and not os.environ.get("CPPF_NO_POINT_BATCH"):      # (the knob: A/B measurements)
            from .models.model import point_encoder_forward_batch
            batched = point_encoder_forward_batch([dict(encoder=p.point_encoder, pc=p.pc, nrm=p.nrm, n_dev=p.shape, out=p._feat_out,
                                                        nbrs=p._nbrs, nbrs_ready=getattr(p, "nbrs_ready", False)) for p in enc_members])
        for p in pipes:
            if p.point_encoder is None:
                feats.append(p.feat)
            elif p.dynamic:
                feats.append(p._feat_out if batched is not None else
                             p.point_encoder.forward_dyn(p.pc, p.nrm, p.shape, out=p._feat_out, nbrs=p._nbrs,
                                                         nbrs_ready=getattr(p, "nbrs_ready", False)))
            else:
                feats.append(p.point_encoder(p.pc[None], p.nrm[None])[0])
        items = []
        for p, f in zip(pipes, feats):
            it = dict(encoder=p.encoder, pc=p.pc, pc_normal=p.nrm, feat=f, idxs=p.idx, u_tr=p.u_tr, vote_range=p.cfg.vote_range)
            if self.full_first:
                it["u_rot"] = p.u_rot
            items.append(it)
        tables = []
        outs = forward_decode_batch(items, p0.cfg.tr_num_bins, p0.cfg.rot_num_bins, tables_out=tables)          # :182-188
        outputs = [o for o, _ in outs]
        heads = [(h if self.full_first else p.ws.heads) for p, (_, h) in zip(pipes, outs)]
        self.tensors = [dict(outputs=o, heads=h, feat=f) for o, h, f in zip(outputs, heads, feats)]
        vws = []
        voting.vote_argmax_batch([dict(points=p.pc, outputs=o, point_idxs=p.idx, corner=p.corner, res=p.cfg.res, out_idx=p.out_idx,
                                       out_val=p.out_val, **(dict(grid=p.grid_flat, shape=p.shape, many_tiles=p.many_tiles) if p.dynamic
                                                             else dict(grid=p.grid))) for p, o in zip(pipes, outputs)],
                                 p0.num_rots, p0.adaptive, accumulate=False, workgroups=self.vote_workgroups, workspaces_out=vws)   # :191-208
        arr = (_lib.PoseTailItem * len(pipes))()
        keep = []
        sph32, sph64, sorted_y = p0._sph
        for i, p in enumerate(pipes):
            ws, a = p.ws, arr[i]
            pws = workspace(L.cppf_pose_sums_workspace_bytes(), self.device, f"pose_sums{i}")
            packed = p.encoder._packed_weights(self.device)
            a.pc, a.nrm, a.feat, a.idx32 = p.pc.data_ptr(), p.nrm.data_ptr(), feats[i].data_ptr(), p.idx32.data_ptr()
            a.idx64 = p.idx.data_ptr() if p.idx.dtype == torch.int64 else None          # (None: the list is int32, idx32 is the input)
            a.outputs, a.u_rot, a.heads, a.corner = outputs[i].data_ptr(), p.u_rot.data_ptr(), heads[i].data_ptr(), p.corner.data_ptr()
            a.shape_dev = p.shape.data_ptr() if p.dynamic else None
            a.argmax_idx, a.peak = p.out_idx.data_ptr(), p.out_val.data_ptr()
            a.packed, a.mlp_workspace, a.mlp_workspace_bytes = packed.data_ptr(), tables[i].data_ptr(), tables[i].numel()
            a.vote_workspace = vws[i].data_ptr() if vws[i].numel() >= 32768 else None
            a.rec, a.T32, a.tail0, a.tail0_bytes = ws.rec.data_ptr(), ws.T32.data_ptr(), ws._tail0.data_ptr(), ws._tail0.numel()
            a.mask, a.chunk_counts, a.surv, a.count = ws.mask.data_ptr(), ws.chunk_counts.data_ptr(), ws.surv.data_ptr(), ws.count.data_ptr()
            a.counts, a.best_idx, a.ticket = ws.counts.data_ptr(), ws.best_idx.data_ptr(), ws.ticket.data_ptr()
            a.sums_workspace, a.sums_workspace_bytes = pws.data_ptr(), pws.numel()
            a.n_points, a.n_pairs = p.pc.shape[0], p.idx.shape[0]
            a.res64, a.res, a.tol = float(p.cfg.res), float(p.cfg.res), float(np.float32(3 * p.cfg.res))
            a.gx, a.gy, a.gz = (1, 1, 1) if p.dynamic else p.dims
            a.n_dirs, a.second_pass = (2 if p.cfg.regress_right else 1), (0 if self.full_first else 1)
            if self.staged:                      # the finished record, assembled by the last launch (:299-339)
                a.record_out, a.object_id_dev = self.records[i].data_ptr(), self.desc[i].data_ptr() + 40
                a.scale_mean = (C.c_double * 3)(*[float(v) for v in p.cfg.scale_mean])
                a.regress_right = 1 if p.cfg.regress_right else 0
            keep.append((pws, packed))
        dims = (C.c_int * len(p0.encoder.ppffcs))(*p0.encoder.ppffcs)
        thr = float(np.float32(np.cos(p0.angle_tol / 180 * np.pi)))
        with torch.cuda.device(self.device):
            rc = L.cppf_pose_tail_batch(len(pipes), C.cast(arr, C.c_void_p), feats[0].shape[1], dims, len(p0.encoder.ppffcs) - 1,
                                        p0.encoder.out_dim, p0.cfg.tr_num_bins, p0.cfg.rot_num_bins, p0.num_rots, sph32.data_ptr(),
                                        sph64.data_ptr(), sph32.shape[0], sorted_y, thr, p0.max_rot_pairs, stream_ptr(self.device))
        _lib.check(rc, "cppf_pose_tail_batch")                                                                  # :209-303,335

    def _key(self):
        return (self.vote_workgroups,) + tuple((p.idx.data_ptr(), p.pc.data_ptr()) for p in self.pipes)

    def run_staged(self, objs, seeds, ids, rows_out, check_weights=None, capture=True):
        """Staged chains: objs[i] = dict(pc, normals[, feat]) of DEVICE f32 tensors (contiguous; they must stay alive and unchanged
        until the chain has run), seeds[i] the Philox key of member i's pair / uniform draws, ids[i] its object id; rows_out: device
        f64[n, RECORD] that receives the members' finished records (one device copy).  No host synchronisation.
        capture=False: a chain that has no captured graph yet runs its launches eagerly this time (a one-off combination)."""
        n = len(self.pipes)
        host, words, ev = self._desc_ring[self._ring_pos % len(self._desc_ring)]
        self._ring_pos += 1
        ev.synchronize()
        for i, (o, p) in enumerate(zip(objs, self.pipes)):
            feat = o.get("feat") if p.point_encoder is None else None
            if feat is None and p.point_encoder is None:
                raise ValueError(f"member {i}: no `feat` and no point encoder in its pipeline (the chain would reuse the last object's features)")
            if o["pc"].shape[0] > p.pc.shape[0] or (feat is not None and feat.shape != (o["pc"].shape[0], p.feat.shape[1])):
                raise ValueError(f"member {i}: cloud of {o['pc'].shape[0]} points / features {None if feat is None else tuple(feat.shape)} "
                                 f"do not fit a pipeline of {p.pc.shape[0]} x {p.feat.shape[1]}")
            words[i] = (o["pc"].data_ptr(), o["normals"].data_ptr(), 0 if feat is None else feat.data_ptr(), o["pc"].shape[0],
                        int(seeds[i]) & 0xFFFFFFFFFFFFFFFF, int(ids[i]))
        copy_words(self.desc, host, self.device)          # (kernels, not copy engines: see cppf_copy_words)
        ev.record(torch.cuda.current_stream(self.device))
        self.run_async(None, check_weights, eager=not capture and self._graphs.get(self.full_first, (None, None))[0] is None)
        copy_words(rows_out, self.records[:rows_out.shape[0]], self.device)

    def run_async(self, records_out, check_weights=True, eager=False):
        """Replay the chain and copy every member's 21-double record into records_out[i] (device f64[21]) on the current stream
        (records_out None: no copies).  eager: launch instead of capturing / replaying."""
        with torch.no_grad(), workspace_scope(id(self)):
            if not self._use_graph or eager:
                self._chain()
            else:
                form = self.full_first
                graph, images = self._graphs.get(form, (None, None))
                now = ((tuple(tuple(p._weight_images()) for p in self.pipes) if check_weights or graph is None else
                        (tuple(tuple(p._image_ptrs()) for p in self.pipes) if check_weights is None else images[0])), self._key())
                if graph is not None and now != images:
                    graph = None
                if graph is None:
                    s = torch.cuda.Stream(device=self.device)
                    s.wait_stream(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(s):
                        self._chain()
                        self._chain()
                    torch.cuda.current_stream(self.device).wait_stream(s)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                        self._chain()
                    # (a captured form owns the tensors its launches write: `tensors` = outputs / heads / feat per member of THIS form)
                    self._graphs[form] = (graph, now)
                    self._owned[form] = self.tensors
                self.tensors = self._owned[form]
                for p in self.pipes:
                    p._await_images()
                graph.replay()
                for p in self.pipes:
                    p._note_images_read()
        for p, r in zip(self.pipes, records_out or ()):
            r.copy_(p.ws.rec, non_blocking=True)

    def run(self, check_weights=True):
        """-> the members' pose dicts (one host synchronisation for all of them)"""
        recs = torch.empty((len(self.pipes), 21), dtype=torch.float64, device=self.device)
        self.run_async(list(recs), check_weights)
        host = recs.cpu().numpy()
        out = []
        for p, r in zip(self.pipes, host):
            d = _assemble(r, p.cfg)
            d.update(dims=p.dims, ws=p.ws)
            out.append(d)
        for d, tn in zip(out, self.tensors):
            d.update(outputs=tn["outputs"], heads=tn["heads"])
        self.adapt([d["n_surv"] for d in out])
        return out

    def adapt(self, n_survs):
        """choose the form of the next run from the survivor shares of the last one (see PosePipeline.adapt)"""
        share = float(np.mean([float(n) / max(p.idx.shape[0], 1) for n, p in zip(n_survs, self.pipes)]))
        if share > self.FULL_FIRST_ON:
            self.full_first = True
        elif share < self.FULL_FIRST_OFF:
            self.full_first = False

    def release(self):
        self._graphs, self._owned, self.tensors = {}, {}, None
        release_scope(id(self))

    def __del__(self):
        try:
            release_scope(id(self))
        except Exception:
            pass


def assemble_batch(recs, cfgs, object_ids, n_cols):
    """_assemble + sharding.pack_record for a whole batch in one numpy pass: recs f64[n,21] (read back from the device),
    cfgs a list of n category configs -> f64[n, n_cols] = {T, up, right, scale, argmax, peak, n_surv, object id}.
    Same arithmetic per row as _assemble (nocs/inference.py:299-339); a row whose `right` degenerates (|right| < 1e-7, :325)
    goes through _assemble itself for its random fallback."""
    recs = np.asarray(recs, dtype=np.float64).reshape(-1, 21)
    n = recs.shape[0]
    out = np.zeros((n, n_cols), np.float64)
    if n == 0:
        return out
    if np.any(recs[:, 19] < 0):
        raise _lib.CppfError("an instance shape did not fit the shape-polymorphic pipeline it ran on (arg-max index -1)")
    rr = np.array([bool(c.regress_right) for c in cfgs])
    sm = np.array([c.scale_mean for c in cfgs], np.float64)
    best = recs[:, 3:9].reshape(n, 2, 3)
    sign = recs[:, 9:15].reshape(n, 2, 3)
    cnt = np.maximum(sign[:, :, 2], 1.0)
    flip = (sign[:, :, 1] / cnt) < (sign[:, :, 0] / cnt)                  # down_loss < up_loss  (:299-302)
    dirs = np.where(flip[:, :, None], -best, best)
    up = dirs[:, 0]
    right_r = dirs[:, 1] - np.sum(up * dirs[:, 1], -1, keepdims=True) * up  # :305-312
    right_n = np.stack([np.zeros(n), -up[:, 2], up[:, 1]], -1)
    right = np.where(rr[:, None], right_r, right_n)
    right = right / (np.linalg.norm(right, axis=-1, keepdims=True) + 1e-9)
    n_surv = recs[:, 18].astype(np.int64)
    mean = (recs[:, 15:18] / np.maximum(n_surv, 1)[:, None]).astype(np.float32)     # torch mean is fp32
    scale = np.exp(mean).astype(np.float64) * sm * 2                                  # :335
    out[:, 0:3], out[:, 3:6], out[:, 6:9], out[:, 9:12] = recs[:, 0:3], up, right, scale
    out[:, 12], out[:, 13], out[:, 14], out[:, 15] = recs[:, 19], recs[:, 20], n_surv, np.asarray(object_ids, np.float64)
    for i in np.nonzero(np.linalg.norm(right, axis=-1) < 1e-7)[0]:                    # :325-328, random fallback
        p = _assemble(recs[i], cfgs[i])
        out[i, 3:6], out[i, 6:9] = p["up"], p["right"]
    return out


def assemble_record(rec, cfg, rng=None):
    """pose dict from one 21-double record read back from a PosePipeline (see PosePipeline.run_async)"""
    return _assemble(np.asarray(rec, dtype=np.float64), cfg, rng)
