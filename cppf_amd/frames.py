"""One NOCS frame through the instance loop of nocs/inference.py:108-142,177-339: depth image + instance masks -> per-instance
clouds on the device (back-projection, /1000, jitter, axis flips, voxel de-duplication, PCA normals) -> kNN + SPRIN features ->
pose.  The detector that makes the masks (Mask R-CNN results pickles, :92-97) is upstream of the path and not part of it."""
import numpy as np
import torch

from .config import CATEGORIES
from .inference import estimate_pose
from .utils.util import backproject, estimate_normals, fibonacci_sphere, num_sphere_bins, sparse_quantize

NOCS_INTRINSICS = np.array([[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]])      # nocs/inference.py:98


def instance_cloud(depth_dev, intrinsics, mask, cfg, jitter=None):
    """nocs/inference.py:131-142 for one instance, on the device: (pc f32[N,3], normals f32[N,3]) device tensors.
    depth_dev: the frame's depth (int16 bits of the uint16 millimetres, uploaded once per frame); mask: [H,W] bool/uint8 (numpy
    or device); jitter: f32[n_masked,3] standard-normal draws for :134 (None: no augmentation; the reference draws
    np.random.randn)."""
    pts, _ = backproject(depth_dev, intrinsics, mask, return_device=True)          # :131  (fp64, masked pixels with depth > 0)
    if pts.shape[0] == 0:                                                          # (an empty mask: nothing to de-duplicate or fit)
        return pts.float(), pts.float()
    pc = pts / 1000.0                                                              # :132
    if jitter is not None:                                                         # :134
        j = torch.as_tensor(jitter, dtype=torch.float64, device=pc.device)[:pc.shape[0]]
        pc = pc + torch.clamp(cfg.res / 4 * j, -cfg.res / 2, cfg.res / 2)
    pc = torch.stack([-pc[:, 0], -pc[:, 1], pc[:, 2]], -1)                         # :136-137
    _, keep = sparse_quantize(pc.float(), return_index=True, quantization_size=cfg.res)   # :140 (ME.utils.sparse_quantize)
    pc = pc[keep].float().contiguous()                                             # :141
    return pc, estimate_normals(pc, cfg.knn)                                       # :142


def pair_seed(seed, i):
    """the Philox key of instance i of a frame drawn with `seed`"""
    return (int(seed) * 1000003 + int(i)) & 0x7FFFFFFFFFFFFFFF


def draw_pairs(seed, i, n_pairs, dev, n_points, idx=None, u=None):
    """The pair list of nocs/inference.py:177 (uniform over [0, n_points)) and the bin uniforms f32[2, n_pairs, 2] (stand-ins for
    torch.multinomial's draws, :186,250) of instance i of a frame, drawn on the device by cppf_sample_pairs: a function of
    (seed, i, pair index) alone, so the eager loop (N known on the host) and a captured chain (N in a device record, the seed in
    device memory) draw the same numbers."""
    from . import _lib
    from ._torch_util import stream_ptr
    idx = torch.empty((n_pairs, 2), dtype=torch.int64, device=dev) if idx is None else idx
    u = torch.empty((2, n_pairs, 2), dtype=torch.float32, device=dev) if u is None else u
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().cppf_sample_pairs(idx.data_ptr(), u[0].data_ptr(), u[1].data_ptr(), n_pairs, int(n_points), None,
                                                pair_seed(seed, i), None, stream_ptr(dev)), "cppf_sample_pairs")
    return idx, u


def frame_poses(depth, instances, encoders, point_encoders, intrinsics=NOCS_INTRINSICS, n_pairs=100000, seed=0, device=None,
                angle_tol=1.5, num_rots=72, cfgs=None, index_of=None):
    """depth: uint16 [H,W]; instances: list of (category name, mask [H,W]); encoders / point_encoders: {category: module on the
    device}; cfgs: {category: CategoryConfig} (default: cppf_amd.config.CATEGORIES, the reference's yaml values).  Returns a list of pose dicts (None for an instance with fewer points than the kNN needs, like the reference's
    skip at :121-123), each with `n_points` added.  index_of: the instances' positions in their frame (default 0, 1, ...): the pair
    draws are a function of (seed, position)."""
    dev = device or torch.device("cuda", 0)
    d_dev = torch.from_numpy(np.ascontiguousarray(depth).view(np.int16)).to(dev)   # one upload per frame
    sphere = np.array(fibonacci_sphere(num_sphere_bins(angle_tol)))                # :100-102
    out = []
    for i, (cat, mask) in enumerate(instances):
        enc, penc, cfg = encoders[cat], point_encoders[cat], (cfgs or CATEGORIES)[cat]
        pc, nrm = instance_cloud(d_dev, intrinsics, mask, cfg)
        n = pc.shape[0]
        if n < cfg.knn + 1:
            out.append(None)
            continue
        idx, u = draw_pairs(seed, i if index_of is None else index_of[i], n_pairs, dev, n)       # :177, and the draws of :186,250
        with torch.no_grad():
            feat = penc(pc[None], nrm[None])[0]                                    # :180-181
            pose = estimate_pose(enc, pc, nrm, feat, idx, u[0], u[1], cfg, sphere, num_rots=num_rots, angle_tol=angle_tol)
        pose["n_points"] = n
        out.append(pose)
    return out


class FrameRunner:
    """frame_poses() through captured chains: depth and ONE label image (bit i = instance i's mask) uploaded per frame, every
    instance's pre-processing (back-projection, /1000, flips, voxel de-duplication, PCA normals, grid set-up: cppf_frame_cloud_dyn,
    count-driven on the device) at the head of a shape-polymorphic PosePipeline, the instances of a lane sharing their launches
    (inference.PoseChain), one read-back per frame.  The reference's loop is nocs/inference.py:108-142,177-339.

        runner = FrameRunner(encoders, point_encoders, device)
        poses = runner.run(depth_u16, [(category, mask), ...])        # list of pose dicts / None, as frame_poses

    Poses equal frame_poses' (the eager path: same kernels, same pairs) bit for bit.  An instance the captured chain cannot serve
    (a cloud whose grid needs >= 4 vote tiles; more than `max_instances` instances) goes through the eager path; an instance with
    fewer points than the kNN needs is skipped like the reference's (:121-123)."""

    def __init__(self, encoders, point_encoders, device, intrinsics=NOCS_INTRINSICS, n_pairs=100000, angle_tol=1.5, num_rots=72,
                 cfgs=None, n_lanes=3, chain_len=None, cap_bucket=4096, max_members=48, max_chains=32, batch_prestage=True):
        """max_members: pipelines kept (each with its captured graph and ~0.1 GB of buffers at 100 000 pairs); max_chains: captured
        chains kept (at least the groups of one frame: a chain of the running frame is never evicted).
        chain_len: instances per captured chain (None: two chains of equal length from four instances on, at most 8 members each);
        batch_prestage=False: round 5's form, every member's own sixteen pre-processing launches at the head of its chain and
        chains of ceil(n / n_lanes)."""
        from collections import OrderedDict
        self.encoders, self.point_encoders, self.device = encoders, point_encoders, device
        self.intrinsics = np.asarray(intrinsics, np.float64)
        self.kinv = np.ascontiguousarray(np.linalg.inv(self.intrinsics))
        self.n_pairs, self.angle_tol, self.num_rots = int(n_pairs), angle_tol, num_rots
        self.cfgs = cfgs or CATEGORIES
        self.n_lanes, self.chain_len, self.cap_bucket = max(1, int(n_lanes)), chain_len, int(cap_bucket)
        self.sphere = np.array(fibonacci_sphere(num_sphere_bins(angle_tol)))
        self.batch_prestage = bool(batch_prestage)
        self.max_instances = 32                       # bits of the u32 label image
        # members: pipelines (+ pre-processing stage) keyed by what their buffers and launches are sized for -- (category, point
        # capacity, grid class) -- and handed to the instances of a frame from a pool, first free first: WHICH instance a member
        # serves (its label bit, its Philox key) is written to the member's device record per frame, not baked into its launches,
        # so a video whose instance count, order or mask sizes change keeps replaying the graphs it has
        self._members, self.max_members = OrderedDict(), int(max_members)          # id(pipe) -> member, least recently used first
        self._pool = {}                                                            # class key -> [member, ...]
        self._chains, self._seen, self.max_chains = OrderedDict(), {}, max(1, int(max_chains))
        self._many_tile_cats = set()
        self._hw = None
        from ._torch_util import lane_streams
        self._streams = lane_streams(device, self.n_lanes)        # (streams that really run beside each other: distinct hardware queues)
        # {label bit, Philox key} of the instance a member serves in the running frame: one row per member in ONE device table, sent
        # with one copy per frame (a copy per member was 6-8 small transfers at the head of every frame)
        self._slot_rows = max(64, 2 * self.max_members)
        self._slots_dev = torch.zeros((self._slot_rows, 2), dtype=torch.int64, device=device)
        self._free_rows = list(range(self._slot_rows - 1, -1, -1))

    def _frame_buffers(self, H, W):
        if self._hw != (H, W):
            self._hw = (H, W)
            torch.cuda.synchronize(self.device)            # (frames still in flight read the buffers that go away here)
            # two pinned sets alternate (submit() of frame k + 1 fills one while the copies of frame k still read the other); a set is
            # rewritten only after the copies that last read it have executed (`sent`)
            self._host_sets = [dict(depth=torch.empty((H, W), dtype=torch.int16).pin_memory(),
                                    labels=torch.empty((H, W), dtype=torch.int32).pin_memory(),
                                    slots=torch.zeros((self._slot_rows, 2), dtype=torch.int64).pin_memory(),
                                    raw=torch.zeros((self.max_instances, 21), dtype=torch.float64).pin_memory(),
                                    shapes=torch.zeros((self.max_instances, 4), dtype=torch.int32).pin_memory(),
                                    sent=torch.cuda.Event(), done=torch.cuda.Event(), owner=None) for _ in range(2)]
            self._set_pos = 0
            self._depth = torch.empty((H, W), dtype=torch.int16, device=self.device)
            self._labels = torch.empty((H, W), dtype=torch.int32, device=self.device)
            for ch in self._chains.values():
                ch.release()
            for mem in self._members.values():
                mem["pipe"].release()
            self._members.clear()
            self._pool.clear()
            self._chains.clear()
            self._free_rows = list(range(self._slot_rows - 1, -1, -1))

    def _member(self, cat, n_mask, used):
        """A member of class (category, capacity for n_mask label pixels, grid class) that no instance of this frame holds yet
        (`used`: ids taken), created when the pool has none.  Grid class: few tiles (every NOCS category at its own resolution on a
        tight mask) until an instance of the category came back with "shape beyond the launch's capacities" -- from then on the
        category's members are of the many-tile class (any grid of up to 64 tiles)."""
        from . import _lib
        from ._torch_util import stream_ptr
        from .inference import PosePipeline
        cap = max(self.cap_bucket, 1 << int(np.ceil(np.log2(max(n_mask, 1)))))
        key = (cat, cap, cat in self._many_tile_cats)
        for mem in self._pool.get(key, ()):
            if id(mem["pipe"]) not in used:
                self._members.move_to_end(id(mem["pipe"]))
                return mem
        while len(self._members) >= self.max_members:
            victim = next((k for k in self._members if k not in used), None)
            if victim is None:
                break                                   # every member is in use by this frame: the bound yields
            torch.cuda.synchronize(self.device)
            old = self._members.pop(victim)
            self._free_rows.append(old["row"])
            self._pool[old["key"]].remove(old)
            for ck in [ck for ck in self._chains if victim in ck]:
                self._chains.pop(ck).release()
            old["pipe"].release()
        cfg, H, W = self.cfgs[cat], self._hw[0], self._hw[1]
        pipe = PosePipeline(self.encoders[cat], cfg, cap, self.n_pairs, key[2], self.device, self.sphere, num_rots=self.num_rots,
                            angle_tol=self.angle_tol, point_encoder=self.point_encoders[cat], dynamic=True)
        L = _lib.lib()
        ws = torch.empty(int(L.cppf_frame_cloud_workspace_bytes(H, W, cap, cfg.knn)), dtype=torch.uint8, device=self.device)
        if not self._free_rows:                       # (more live members than rows: only when the bound yielded to a huge frame)
            raise RuntimeError("FrameRunner: no free member record; raise max_members")
        row = self._free_rows.pop()
        slot = self._slots_dev[row]                   # {label bit (low 32 bits), Philox key}: written per frame
        dev, depth, labels, kinv = self.device, self._depth, self._labels, self.kinv
        # the normals are fitted on the k = cfg.knn neighbour sets; a point encoder with the same k (config/config.yaml: 60 for both)
        # reuses them instead of searching again (the stage writes them straight into the pipeline's neighbour buffer)
        share = pipe.point_encoder is not None and pipe.point_encoder.k == cfg.knn
        pipe.nbrs_ready = share
        nbrs_ptr = pipe._nbrs.data_ptr() if share else None

        def prestage():
            with torch.cuda.device(dev):
                _lib.check(L.cppf_frame_cloud_dyn_bit(depth.data_ptr(), 1, labels.data_ptr(), 4, slot.data_ptr(), H, W, kinv.ctypes.data,
                                                      1000.0, float(cfg.res), cfg.knn, cfg.knn + 1, cap, pipe.pc.data_ptr(),
                                                      pipe.nrm.data_ptr(), pipe.corner.data_ptr(), pipe.shape.data_ptr(), nbrs_ptr,
                                                      ws.data_ptr(), ws.numel(), stream_ptr(dev)), "cppf_frame_cloud_dyn_bit")
                # pairs and bin uniforms: N from the shape record the stage just wrote, the key from the member's record
                _lib.check(L.cppf_sample_pairs(pipe.idx.data_ptr(), pipe.u_tr.data_ptr(), pipe.u_rot.data_ptr(), pipe.idx.shape[0], 1,
                                               pipe.shape.data_ptr(), 0, slot.data_ptr() + 8, stream_ptr(dev)), "cppf_sample_pairs")
        mem = dict(pipe=pipe, pre=prestage, slot=slot, row=row, ws=ws, key=key, cfg=cfg, cap=cap, nbrs_ptr=nbrs_ptr)
        self._members[id(pipe)] = mem
        self._pool.setdefault(key, []).append(mem)
        return mem

    def _batch_prestage(self, mems):
        """the frame stage of a chain's members in eight launches (cppf_frame_cloud_dyn_batch) instead of sixteen each: a callable
        for the head of their captured chain"""
        import ctypes as C
        from . import _lib
        from ._torch_util import stream_ptr
        L, dev, H, W = _lib.lib(), self.device, self._hw[0], self._hw[1]
        depth, labels, kinv = self._depth, self._labels, self.kinv
        arr = (_lib.FrameCloudItem * len(mems))()
        for a, m in zip(arr, mems):
            pipe, cfg = m["pipe"], m["cfg"]
            a.label_bit_dev, a.seed_dev = m["slot"].data_ptr(), m["slot"].data_ptr() + 8
            a.pc_out, a.nrm_out, a.corner_out, a.shape_out = pipe.pc.data_ptr(), pipe.nrm.data_ptr(), pipe.corner.data_ptr(), pipe.shape.data_ptr()
            a.nbrs_out, a.idx, a.u_tr, a.u_rot = m["nbrs_ptr"], pipe.idx.data_ptr(), pipe.u_tr.data_ptr(), pipe.u_rot.data_ptr()
            a.workspace, a.workspace_bytes, a.res, a.n_pairs = m["ws"].data_ptr(), m["ws"].numel(), float(cfg.res), pipe.idx.shape[0]
            a.knn_k, a.k_min, a.n_cap, a.idx_is_i64 = cfg.knn, cfg.knn + 1, m["cap"], 1 if pipe.idx.dtype == torch.int64 else 0

        def prestage():
            with torch.cuda.device(dev):
                _lib.check(L.cppf_frame_cloud_dyn_batch(len(mems), C.cast(arr, C.c_void_p), depth.data_ptr(), 1, labels.data_ptr(), 4, H, W,
                                                        kinv.ctypes.data, 1000.0, stream_ptr(dev)), "cppf_frame_cloud_dyn_batch")
        return prestage

    def _chain_for(self, pipes, pres, busy):
        """the captured chain of this combination of members (second sighting on), or None: its members run their own graphs.
        `busy`: chains of the running frame, which an eviction must not touch."""
        from .inference import PoseChain
        key = tuple(id(p) for p in pipes)
        ch = self._chains.get(key)
        if ch is not None:
            self._chains.move_to_end(key)
            busy.add(key)
            return ch
        n = self._seen[key] = self._seen.get(key, 0) + 1
        if len(self._seen) > 4096:
            self._seen.clear()
        if n < 2:
            return None
        while len(self._chains) >= self.max_chains:
            victim = next((k for k in self._chains if k not in busy), None)
            if victim is None:
                return None                              # (more groups in this frame than max_chains)
            torch.cuda.synchronize(self.device)
            self._chains.pop(victim).release()
        ch = self._chains[key] = PoseChain(pipes, prestages=pres() if callable(pres) else pres)
        busy.add(key)
        return ch

    def run(self, depth, instances, seed=0):
        """-> the frame's poses (list of pose dicts / None, as frame_poses): submit() + result()"""
        return self.submit(depth, instances, seed).result()

    def submit(self, depth, instances, seed=0):
        """Enqueue a frame and return at once: -> PendingFrame, whose result() waits for the frame's one read-back and returns the
        poses.  A video loop that submits frame k + 1 before it asks for frame k's result overlaps the host's share of a frame (the
        label image, the members' bookkeeping, the enqueue: ~0.6 of the demo frame's 1.45 ms) with the device's.  Frames execute in
        submission order; at most two may be pending (a third submit() first collects the oldest's result)."""
        from ._torch_util import gather_words
        dev = self.device
        depth = np.ascontiguousarray(depth)
        H, W = depth.shape
        self._frame_buffers(H, W)
        hs = self._host_sets[self._set_pos % 2]
        self._set_pos += 1
        if hs["owner"] is not None:
            hs["owner"].result()                     # (its result arrays live in this set)
        hs["sent"].synchronize()                     # the uploads that last read this set have executed
        n_inst = len(instances)
        on_chain = list(range(min(n_inst, self.max_instances)))
        labels = hs["labels"].numpy().view(np.uint32)
        labels[...] = 0
        counts = []
        for i in on_chain:
            m = np.asarray(instances[i][1])
            if m.dtype != np.bool_:
                m = m != 0
            np.bitwise_or(labels, np.uint32(1 << i), out=labels, where=m)      # (no fancy indexing: 0.1 ms per 480 x 640 mask)
            counts.append(int(np.count_nonzero(m)))
        hs["depth"].numpy()[...] = depth.view(np.int16)
        main = torch.cuda.current_stream(dev)
        # (on the caller's stream, i.e. behind the previous frame's join: the chains' launches have these two buffers' addresses baked in)
        self._depth.copy_(hs["depth"], non_blocking=True)               # one upload per frame (two images)
        self._labels.copy_(hs["labels"], non_blocking=True)
        for cat in {instances[i][0] for i in on_chain}:
            self.encoders[cat]._packed_weights(dev)
            self.point_encoders[cat]._packed_weights(dev)
        # (zero-filled on the caller's stream BEFORE the lanes fork from it: a lane's record copy must not race the fill)
        raw = torch.zeros((max(len(on_chain), 1), 21), dtype=torch.float64, device=dev)
        shapes = torch.zeros((max(len(on_chain), 1), 4), dtype=torch.int32, device=dev)
        if self.chain_len:
            Lc = self.chain_len
        elif self.batch_prestage:          # two chains of equal length on two lanes (measured on 6 instances: 3 + 3 1.48 ms per
            n_ch = max(2 if len(on_chain) >= 4 and self.n_lanes > 1 else 1, -(-len(on_chain) // 8))     # frame, one chain 1.58, 2 + 2 + 2 1.50)
            Lc = max(1, -(-len(on_chain) // n_ch))
        else:
            Lc = max(1, min(8, -(-len(on_chain) // self.n_lanes)))
        groups = [on_chain[g:g + Lc] for g in range(0, len(on_chain), Lc)]
        ran, used, busy = [], set(), set()
        # which member serves which instance of this frame: decided for the whole frame first, written to the members' rows of the
        # record table and sent with one copy on the caller's stream, before the lanes fork from it
        slots_tbl, members_of = hs["slots"].numpy(), []
        for slots_g in groups:
            mems = []
            for i in slots_g:
                mem = self._member(instances[i][0], counts[i], used)
                used.add(id(mem["pipe"]))
                slots_tbl[mem["row"]] = (i, pair_seed(seed, i))
                mems.append(mem)
            members_of.append(mems)
        if on_chain:
            self._slots_dev.copy_(hs["slots"], non_blocking=True)
        hs["sent"].record(main)
        for st in self._streams:
            st.wait_stream(main)
        for gi, (slots_g, mems) in enumerate(zip(groups, members_of)):
            lane = gi % self.n_lanes
            with torch.cuda.stream(self._streams[lane]):
                pipes, pres = [m["pipe"] for m in mems], [m["pre"] for m in mems]
                # (a chain's members share the launches of their frame stage; without a chain every member runs its own)
                ch = self._chain_for(pipes, (lambda: [self._batch_prestage(mems)] + [None] * (len(mems) - 1)) if self.batch_prestage else pres, busy)
                # (check_weights=None: the images were refreshed above, once per frame; a pipeline compares their addresses, so a
                # weight image that moved -- an encoder re-created or resized -- re-captures instead of reading a stale address)
                if ch is not None:
                    ch.run_async(None, check_weights=None)
                else:
                    for pipe, pre in zip(pipes, pres):
                        pre()
                        pipe.run_async(None, check_weights=None)
                # the members' records and shape words, each in its own pipeline's buffers: one launch each into the frame's arrays
                gather_words(raw[slots_g[0]:slots_g[-1] + 1], [p.ws.rec for p in pipes], dev)
                gather_words(shapes[slots_g[0]:slots_g[-1] + 1], [p.shape for p in pipes], dev)
                ran.append((ch, pipes, slots_g))
        for st in self._streams:
            main.wait_stream(st)
        n = max(len(on_chain), 1)
        hs["raw"][:n].copy_(raw, non_blocking=True)                     # the frame's one read-back (two small arrays), asynchronous
        hs["shapes"][:n].copy_(shapes, non_blocking=True)
        hs["done"].record(main)
        pend = PendingFrame(self, hs, depth, instances, seed, on_chain, ran, (raw, shapes))
        hs["owner"] = pend
        return pend

    def _finish(self, pend):
        """result() of a pending frame: wait for its read-back, adapt the chains, assemble the poses, run what the chains could not serve"""
        from .inference import assemble_record
        hs, instances, on_chain, ran = pend._hs, pend._instances, pend._on_chain, pend._ran
        hs["done"].synchronize()
        n_inst = len(instances)
        out = [None] * n_inst
        host, shp = hs["raw"].numpy().copy(), hs["shapes"].numpy().copy()
        hs["owner"] = None
        eager = list(range(len(on_chain), n_inst))
        self.last = {"captured": 0, "eager": len(eager), "skipped": 0}   # how the last frame's instances were served
        for ch, pipes, slots in ran:
            if ch is not None:
                ch.adapt([host[i, 18] for i in slots])
            for pipe, i in zip(pipes, slots):
                if ch is None:
                    pipe.adapt(host[i, 18])
                n = int(shp[i, 0])
                if n == 0:
                    self.last["skipped"] += 1
                    continue                                             # fewer points than the kNN needs: skipped (:121-123)
                if host[i, 19] < 0:
                    eager.append(i)                                      # the chain could not serve this shape: a grid of >= 4 tiles
                    self._many_tile_cats.add(instances[i][0])            # (the category's next pipelines are of the many-tile class)
                    self.last["eager"] += 1
                    continue
                self.last["captured"] += 1
                pose = assemble_record(host[i], pipe.cfg)
                pose.update(n_points=n, dims=tuple(int(v) for v in shp[i, 1:4]))
                out[i] = pose
        if eager:
            sub = frame_poses(pend._depth, [instances[i] for i in eager], self.encoders, self.point_encoders, self.intrinsics, self.n_pairs,
                              pend._seed, self.device, self.angle_tol, self.num_rots, self.cfgs, index_of=eager)
            for i, p in zip(eager, sub):
                out[i] = p
        return out


class PendingFrame:
    """a frame FrameRunner.submit() enqueued; result() -> its poses (waits for the frame's read-back; idempotent)"""

    def __init__(self, runner, hs, depth, instances, seed, on_chain, ran, keep):
        self._runner, self._hs, self._depth, self._instances, self._seed = runner, hs, depth, instances, seed
        self._on_chain, self._ran, self._keep, self._out = on_chain, ran, keep, None

    def result(self):
        if self._out is None:
            self._out = self._runner._finish(self)
            self._ran = self._keep = None
        return self._out
