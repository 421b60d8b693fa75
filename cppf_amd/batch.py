"""A batch of object instances (BASELINE.json config 4: mixed NOCS categories) on one or many GPUs.

Objects are independent, so the batch is sharded round-robin over the ranks (object j -> rank j mod W,
cppf_amd.sharding), each rank runs its objects back to back on its GPU -- one hipGraph replay per object -- and ONE
all_gather of the fixed-size records closes the batch.  Mirrors the per-instance loop of nocs/inference.py:120-339
(different categories use different encoders and configs, :124-128).

Real instances almost never repeat a shape: N is whatever voxel de-duplication leaves (nocs/inference.py:140-142) and
the grid is the cloud's bounding box over `res` (:194-195).  The runner therefore keeps SHAPE-POLYMORPHIC pipelines
(inference.PosePipeline(dynamic=True)): one captured graph per (category, N rounded up to `n_bucket`, number of pairs,
grid class, lane); an instance's real {N, gx, gy, gz} travels in a 16-byte device record next to its cloud.  The cache
is bounded (least recently used pipelines are released with their buffers and scratch).  Grids beyond the tiled vote
(> 64 LDS tiles) run on exact-shape pipelines, which share the same bounded cache."""
from collections import OrderedDict

import numpy as np
import torch

from . import sharding
from ._torch_util import copy_words, lane_streams
from .inference import PoseChain, PosePipeline, assemble_batch, grid_class, grid_shape
from .utils.util import fibonacci_sphere, num_sphere_bins


def _shape_error(obj, pipe):
    return ValueError(f"object of shape N={obj['pc'].shape[0]}, grid {tuple(obj['dims'])} on an exact-shape pipeline built for "
                      f"N={pipe.n_points}, grid {tuple(pipe.dims)}")


class BatchPoseRunner:
    def __init__(self, encoders, device, num_rots=72, adaptive=True, angle_tol=1.5, max_rot_pairs=10000,
                 use_graph=True, point_encoders=None, n_bucket=1024, max_pipelines=192, dynamic=True, n_lanes=3,
                 max_scratch_bytes=64 << 30, vote_workgroups=None, chain_len=None, max_chains=24, staged_host=True,
                 overlap_batches=False, idx_i32=True):
        """encoders: {category name: PPFEncoder on `device`} (the reference keeps one per category,
        nocs/inference.py:79-90).  point_encoders: optional {category name: PointEncoder}; objects of those
        categories need no `feat` -- kNN + SPRIN run at the head of the captured graph (:180-181).
        n_bucket: point capacities are multiples of it; max_pipelines: bound of the pipeline cache;
        dynamic=False: one exact-shape pipeline per distinct instance shape (fixed-shape workloads only);
        max_scratch_bytes: second bound of the cache, on the pipelines' estimated device footprint (buffers + the vote's workspace,
        whose pair -> tile queues are sized for the worst case, every pair in every tile of the grid class: 16 x 12 + 48 B per pair
        for grids of 4-16 tiles (0.13 GB per pipeline at 524 288 pairs, 0.5 GB at 2 M), 64 x 12 + 48 B beyond (0.43 / 1.7 GB);
        INTEGRATION.md "Memory");
        n_lanes: instances in flight (HIP streams, each with its own pipelines).  Three measured best on a ragged batch of
        small instances (N 400-2000, 100 k pairs: 0.216 / 0.136 / 0.115 / 0.144 ms per instance with 1 / 2 / 3 / 4 lanes): the
        neighbours fill the gaps between an instance's ~15 short dependent launches.
        vote_workgroups: width of the vote launches (CenterPipeline).  None = chosen per pipeline: with more than one lane, 128 for
        few-tile grids of up to half a million pairs -- every vote workgroup pays for a 113 KB tile whatever it deposits, so with
        neighbours in flight half the chip per vote moves more instances per second (profiles/r4_vote_workgroups.txt) -- and one
        workgroup per CU otherwise (longer pair lists, many-tile grids, a single lane).
        chain_len: instances whose launches are SHARED (inference.PoseChain: one pair-kernel launch, one vote + one reduce launch and
        six tail launches for the whole chain instead of ~15 launches per instance).  None = chains only when the batch is large
        enough to keep every lane two chains deep -- min(8, instances / (2 n_lanes)) -- because a small batch is bounded by the host
        staging it, and a chain cannot start before its last member is staged (measured on 8 / 16 / 32 / 64 C2-size objects:
        no chains 0.189 / 0.163 / 0.164 / 0.158 ms per object, chains 0.193 / 0.164 / 0.161 / 0.140); 1 = every instance its own
        captured pipeline (round 4).  A pipeline serves one (category, point bucket, pair count, grid class, lane, chain position):
        `max_pipelines` / `max_scratch_bytes` bound the cache (a mixed batch of six categories in chains of 8 on 3 lanes wants up to
        144 pipelines of ~0.1 GB at C2 size; a cache that is too small re-captures on every batch).  A chain is captured the SECOND time
        its combination of pipelines (category, point bucket, pair count, grid class per position) comes up -- a one-off combination
        runs its members' own graphs -- and at most `max_chains` captured chains are kept (least recently used first out).
        staged_host: host objects WITHOUT their own pair lists (pairs drawn on the device: `n_pairs`) are uploaded with one copy each and
        take the chains of device-resident objects (put(), _run_resident): records assembled on the device, nothing read back -- the
        batch's records are then returned on the device whatever the world size.  False: round 5's per-instance path for them too.
        overlap_batches (staged chains only): a batch's chains wait for THEIR inputs -- the event put() / the packed upload recorded,
        the weight images, the previous use of their own pipelines -- instead of for everything the caller's stream holds, which
        includes the previous batch's join: lane 0 then starts batch k + 1 while lane 1 still finishes batch k (records are
        double-buffered).  The contract that buys it: objects returned by put() are snapshots -- work the caller enqueues on its
        stream AFTER put() that rewrites them in place is not waited for.  (Host arrays are copied at call time, so their batches
        overlap regardless.)  The records run() returns are ordered on the caller's stream as always.
        idx_i32: the pair lists a staged chain draws are int32 (half the index bytes through every kernel of the chain; the draws are the
        same numbers).  Measured on 8 resident C2-size objects: 0.1338 -> 0.1301 ms per object at 4 batches per timed region, 0.1294 ->
        0.1256 at 8 (profiles/r6_resident_probe.txt)."""
        self.encoders, self.device = encoders, device
        self.point_encoders = point_encoders or {}
        self.n_lanes = max(1, int(n_lanes))
        self.vote_workgroups = None if vote_workgroups is None else int(vote_workgroups)
        self.kw = dict(num_rots=num_rots, adaptive=adaptive, angle_tol=angle_tol, max_rot_pairs=max_rot_pairs,
                       use_graph=use_graph)
        self.sphere = np.array(fibonacci_sphere(num_sphere_bins(angle_tol)))      # :100-102
        self.n_bucket, self.max_pipelines, self.dynamic = int(n_bucket), int(max_pipelines), bool(dynamic)
        self.max_scratch_bytes, self._bytes = int(max_scratch_bytes), {}
        self.chain_len, self.max_chains = (None if chain_len is None else max(1, min(8, int(chain_len)))), int(max_chains)
        self._chains, self._chain_seen = OrderedDict(), {}     # LRU: tuple of member ids -> PoseChain; sightings of a combination
        self._pipes = OrderedDict()    # LRU: key -> PosePipeline
        self._staging = {}         # pinned host staging sets for the small per-instance arrays, see _stage()
        self.staged_host, self.overlap_batches, self.idx_i32 = bool(staged_host), bool(overlap_batches), bool(idx_i32)
        self._stage_pos = 0
        self._streams = None

    # ------------------------------------------------------------------ pipeline cache
    def _pipe(self, cfg, n_points, n_pairs, dims, lane=0, slot=0, idx_i32=False):
        """The pipeline that serves this instance shape at position `slot` of a chain on this lane (created on first use, LRU-bounded).
        idx_i32: its pair list is int32 (staged chains, whose pairs are drawn on the device)."""
        T, many, _ = grid_class(dims)
        dyn = self.dynamic and T > 0
        if dyn:
            n_cap = -(-int(n_points) // self.n_bucket) * self.n_bucket
            key = (cfg.category, n_cap, n_pairs, many, lane, slot, idx_i32)
        else:
            key = (cfg.category, n_points, n_pairs, tuple(dims), lane, slot, idx_i32)
        pipe = self._pipes.get(key)
        if pipe is None:
            need = self.footprint_bytes(n_cap if dyn else n_points, n_pairs, many if dyn else None, dims)
            while self._pipes and (len(self._pipes) >= self.max_pipelines or sum(self._bytes.values()) + need > self.max_scratch_bytes):
                # the victim may still have a replay in flight on its lane's stream: drain before its buffers go back
                # to the allocator (evictions are rare: a new shape bucket beyond the cache bound)
                torch.cuda.synchronize(self.device)
                old_key, old = self._pipes.popitem(last=False)
                self._bytes.pop(old_key, None)
                for ck in [ck for ck in self._chains if id(old) in ck]:      # chains the victim is a member of go with it
                    self._chains.pop(ck).release()
                old.release()
            width = self.vote_workgroups
            if width is None:
                few = (not many) if dyn else (0 < T < 4)
                width = 128 if (self.n_lanes > 1 and few and n_pairs <= (1 << 19)) else 0
            if dyn:
                pipe = PosePipeline(self.encoders[cfg.category], cfg, n_cap, n_pairs, many, self.device, self.sphere,
                                    point_encoder=self.point_encoders.get(cfg.category), dynamic=True, vote_workgroups=width,
                                    idx_i32=idx_i32, **self.kw)
            else:
                pipe = PosePipeline(self.encoders[cfg.category], cfg, n_points, n_pairs, dims, self.device,
                                    self.sphere, point_encoder=self.point_encoders.get(cfg.category), vote_workgroups=width,
                                    idx_i32=idx_i32, **self.kw)
            self._pipes[key] = pipe
            self._bytes[key] = need
        else:
            self._pipes.move_to_end(key)
        return pipe

    def _lanes(self):
        """(the lanes' streams, the upload stream): streams that run beside each other (lane_streams: distinct hardware queues).  The
        queues are few (4 besides the caller's), so the upload stream is picked right after the first two lanes -- small batches run on
        two lanes, and an upload stream that shares a lane's queue parks its wait for an older batch in front of that lane's chains
        (the reference-default batch from host arrays: 0.164 against 0.140 ms per instance)."""
        if self.__dict__.get("_lane_set") is None:
            s = lane_streams(self.device, self.n_lanes + 1)
            k = min(2, self.n_lanes)
            self._lane_set = (s[:k] + s[k + 1:], s[k])
        return self._lane_set

    def _chain_for(self, pipes, staged=False):
        """the captured chain of these pipelines, or None (a single instance; a member the chain cannot take; a combination seen for
        the first time: its members run their own graphs).  staged=True (objects resident on the device, _run_resident): always a
        chain -- of one member too -- returned with `capture`: False on a combination's first sighting (it then runs eagerly)."""
        if any(not p._split_ok or p.rot_order is not None or p._sph[2] == 0 for p in pipes):
            if staged:
                raise ValueError("device-resident objects need the standard pair encoder and sphere bins sorted by y (PoseChain)")
            return None
        if len(pipes) < 2 and not staged:
            return None
        key = tuple(id(p) for p in pipes) + (("staged",) if staged else ())
        chain = self._chains.get(key)
        if chain is not None:
            self._chains.move_to_end(key)
            return (chain, True) if staged else chain
        seen = self._chain_seen.get(key, 0) + 1
        if len(self._chain_seen) > 4096:
            self._chain_seen.clear()
        self._chain_seen[key] = seen
        if seen < 2 and not staged:
            return None
        while len(self._chains) >= self.max_chains:
            torch.cuda.synchronize(self.device)
            self._chains.popitem(last=False)[1].release()
        chain = self._chains[key] = PoseChain(pipes, use_graph=self.kw["use_graph"], staged=staged,
                                              vote_workgroups=0 if self.vote_workgroups is None else self.vote_workgroups)
        return (chain, seen >= 2) if staged else chain

    @staticmethod
    def footprint_bytes(n_points, n_pairs, many_tiles, dims):
        """estimated device bytes of one pipeline: per-pair buffers (pairs i64 + i32, uniforms, (mu, nu), heads, masks, survivor
        list: ~100 B per pair), per-point buffers, the grid, and the vote's workspace (many_tiles None: an exact-shape pipeline)"""
        from . import _lib
        L = _lib.lib()
        if many_tiles is None:
            ws = L.cppf_vote_workspace_bytes(int(n_pairs), 72, int(dims[0]), int(dims[1]), int(dims[2]))
            grid = 4 * int(dims[0]) * int(dims[1]) * int(dims[2])
        else:
            ws = L.cppf_vote_workspace_bytes_dyn_pairs(_lib.tile_class(many_tiles), int(n_pairs))
            grid = 4 * _lib.tiles_cap(many_tiles) * int(L.cppf_vote_tile_cells())
        return int(ws) + grid + 100 * int(n_pairs) + 1400 * int(n_points)

    _RING = 4

    def _stage(self, pipe, pc, normals, feat, corner, dims):
        """cloud, normals, grid corner and (dynamic pipelines) the shape record -> the pipeline's input buffer with ONE copy
        through PINNED host memory, so the copy is truly asynchronous (a copy from pageable memory blocks the host until
        everything queued before it has run, i.e. until the previous instance has finished); features, when the caller
        supplies them, with a second one.  A ring of staging sets per point capacity, each guarded by an event."""
        n, cap = pc.shape[0], pipe.n_points
        key = (cap, None if feat is None else feat.shape[1])
        ring = self._staging.get(key)
        if ring is None:
            mk = lambda *shape: torch.empty(shape, dtype=torch.float32).pin_memory()
            ring = []
            for _ in range(self._RING):
                buf = mk(6 * cap + 8)
                buf.zero_()
                arr = buf.numpy()
                ring.append(dict(buf=buf, pc=arr[:3 * cap].reshape(cap, 3), nrm=arr[3 * cap:6 * cap].reshape(cap, 3),
                                 corner=arr[6 * cap:6 * cap + 3], shape=arr[6 * cap + 4:6 * cap + 8].view(np.int32),
                                 feat=None if feat is None else mk(cap, feat.shape[1]), ev=torch.cuda.Event()))
            self._staging[key] = ring
        st = ring[self._stage_pos % self._RING]
        self._stage_pos += 1
        st["ev"].synchronize()                          # the copies that last read this set have executed
        st["pc"][:n] = pc
        st["nrm"][:n] = normals
        st["corner"][...] = corner
        if pipe.dynamic:
            st["shape"][...] = (n,) + tuple(dims)
            pipe.set_shape(n, dims, upload=False)
        pipe._in.copy_(st["buf"], non_blocking=True)    # (rows behind the cloud carry stale values nobody reads)
        if feat is not None:
            st["feat"].numpy()[:n] = feat
            pipe.feat[:n].copy_(st["feat"][:n], non_blocking=True)
        st["ev"].record(torch.cuda.current_stream(self.device))   # (the lane's stream: _stage is called inside its context)

    def _check(self, j, obj):
        cat = obj["cfg"].category
        if cat not in self.encoders:
            raise ValueError(f"object {j}: no pair encoder for category {cat!r}")
        if obj.get("feat") is None and cat not in self.point_encoders:
            raise ValueError(f"object {j} ({cat}) has no `feat` and the runner has no point encoder for that category")
        if obj["normals"].shape != obj["pc"].shape:
            raise ValueError(f"object {j}: normals {obj['normals'].shape} vs points {obj['pc'].shape}")

    def put(self, objects):
        """Upload a batch ONCE: -> the same objects with `pc`, `normals` and `feat` as device tensors and the grid's `dims` (host ints)
        beside them -- the form SURVEY.md 8(d) times ("inputs already resident on device").  run() takes such objects through chains
        whose first launch reads them where they are (cppf_stage_batch): no staging copies, no host arithmetic, no read-back; the
        records it returns stay on the device.  Host arrays, or device tensors (then `dims` is computed on the device, one read-back
        here).  An object needs `n_pairs` (pairs and bin uniforms are drawn on the device, as for host objects without `point_idxs`)."""
        out = []
        for j, obj in enumerate(objects):
            self._check(j, obj)
            if obj.get("point_idxs") is not None or "n_pairs" not in obj:
                raise ValueError(f"object {j}: a device-resident object draws its pairs on the device: give n_pairs, not point_idxs")
            o = dict(obj)
            if torch.is_tensor(obj["pc"]) and obj["pc"].is_cuda:
                for k in ("pc", "normals", "feat"):
                    if obj.get(k) is not None:
                        o[k] = obj[k].to(device=self.device, dtype=torch.float32).contiguous()
                if o.get("dims") is None:            # cppf_grid_setup: the arithmetic the chain's own grid set-up uses (a torch
                    from . import _lib               # division by a scalar multiplies by its reciprocal: other dims at the edges)
                    from ._torch_util import stream_ptr
                    cd = torch.empty(8, dtype=torch.int32, device=self.device)
                    with torch.cuda.device(self.device):
                        _lib.check(_lib.lib().cppf_grid_setup(o["pc"].data_ptr(), o["pc"].shape[0], float(np.float32(obj["cfg"].res)),
                                                              cd[:3].view(torch.float32).data_ptr(), cd[4:7].data_ptr(),
                                                              stream_ptr(self.device)), "cppf_grid_setup")
                    o["dims"] = tuple(int(v) for v in cd[4:7].tolist())
            else:
                _, o["dims"] = grid_shape(obj["pc"], obj["cfg"].res)
                for k in ("pc", "normals", "feat"):
                    if obj.get(k) is not None:
                        o[k] = torch.from_numpy(np.ascontiguousarray(obj[k], dtype=np.float32)).to(self.device)
            o["dims"] = tuple(int(v) for v in o["dims"])
            out.append(o)
        ready = torch.cuda.Event()             # the uploads above (or whatever produced the caller's device tensors on this stream) are done
        ready.record(torch.cuda.current_stream(self.device))
        for o in out:
            o["_ready"] = ready
        return out

    def _upload_batch(self, objects, mine, host_dims):
        """Host objects on their way into staged chains: the clouds, normals and (when the caller supplies them) features of ALL of a
        rank's objects packed into ONE pinned block and sent with ONE asynchronous copy on a copy stream of its own -- so the copy of
        batch k + 1 runs under the chains of batch k (two device blocks alternate; a block is overwritten only after the chains that
        read it two batches ago have been joined) -- and the chains' first launch reads each object there like any resident one.
        One copy per BATCH, not per object: a 36 KB copy (a 1 500-point cloud) costs the host 35 us on this stack, eight of them made
        the reference-default batch host-bound.  -> ([the objects as put() would return them], event the lanes wait for)"""
        dev = self.device
        metas, total = [], 0
        for j in mine:
            obj = objects[j]
            self._check(j, obj)
            feat = obj.get("feat") if obj["cfg"].category not in self.point_encoders else None
            n, F = obj["pc"].shape[0], 0 if feat is None else feat.shape[1]
            metas.append((obj, feat, n, F, total))
            total += ((6 + F) * n + 63) & ~63                       # (256-byte aligned objects)
        st = self.__dict__.get("_up")
        if st is None or st["words"] < total:
            if st is not None:
                torch.cuda.synchronize(dev)
            words = max(total, 1 << 16) * 5 // 4
            st = self._up = dict(words=words, pos=0, stream=(st or {}).get("stream") or self._lanes()[1], blocks=[])
            for _ in range(2):
                h = torch.empty(words, dtype=torch.float32).pin_memory()
                st["blocks"].append(dict(host=h, arr=h.numpy(), dev=torch.empty(words, dtype=torch.float32, device=dev),
                                         sent=torch.cuda.Event(), free=torch.cuda.Event()))
        blk = st["blocks"][st["pos"] % 2]
        st["pos"] += 1
        blk["sent"].synchronize()                                    # the copy that last read this pinned block has executed
        arr, dblk, out = blk["arr"], blk["dev"], []
        for obj, feat, n, F, off in metas:
            arr[off:off + 3 * n].reshape(n, 3)[...] = obj["pc"]
            arr[off + 3 * n:off + 6 * n].reshape(n, 3)[...] = obj["normals"]
            if F:
                arr[off + 6 * n:off + (6 + F) * n].reshape(n, F)[...] = feat
            o = dict(obj, pc=dblk[off:off + 3 * n].view(n, 3), normals=dblk[off + 3 * n:off + 6 * n].view(n, 3))
            o["feat"] = dblk[off + 6 * n:off + (6 + F) * n].view(n, F) if F else None
            out.append(o)
        for o, j in zip(out, mine):
            o["dims"] = tuple(int(v) for v in host_dims[j])
        with torch.cuda.stream(st["stream"]):
            st["stream"].wait_event(blk["free"])                     # the chains that read this device block two batches ago are done
            dblk[:total].copy_(blk["host"][:total], non_blocking=True)
            blk["sent"].record(st["stream"])
        return out, blk

    def _run_resident(self, objects, mine, rank, world, seed, host_dims=None):
        """run() for objects put() on the device -- and for host objects whose pairs are drawn on the device, which are uploaded on
        the way (_upload_batch): per chain one 48-byte-per-member descriptor copy, one graph replay, one device copy
        of the finished records; nothing is read back (the survivor counts that choose a chain's form for the NEXT batch are copied
        to pinned memory asynchronously and looked at when the next batch starts)."""
        dev, n, n_total = self.device, len(mine), len(objects)
        self._adapt_resident()
        # two record buffers alternate: with overlap_batches a lane may write batch k + 1's rows while the caller's stream still reads
        # batch k's (its clone / gather); a buffer is rewritten only after the read of two batches ago has been enqueued AND waited for
        lb = self.__dict__.get("_locals")
        if lb is None or lb["bufs"][0].shape[0] < max(n, 1):
            if lb is not None:
                torch.cuda.synchronize(dev)
            lb = self._locals = dict(bufs=[torch.zeros((max(n, 1), sharding.RECORD), dtype=torch.float64, device=dev) for _ in range(2)],
                                     read=[torch.cuda.Event(), torch.cuda.Event()], pos=0)
        which = lb["pos"] % 2
        lb["pos"] += 1
        local, read_done = lb["bufs"][which][:max(n, 1)], lb["read"][which]
        if self._streams is None:
            self._streams = self._lanes()[0]
        main = torch.cuda.current_stream(dev)
        for cat in {objects[j]["cfg"].category for j in mine}:
            if cat in self.encoders:
                self.encoders[cat]._packed_weights(dev)
            if cat in self.point_encoders:
                self.point_encoders[cat]._packed_weights(dev)
        blk = None
        if host_dims is not None:           # host objects: one packed copy for the whole batch, on the copy stream (_upload_batch)
            uploaded, blk = self._upload_batch(objects, mine, host_dims)
            objects = dict(zip(mine, uploaded))
        readies = {id(objects[j].get("_ready")): objects[j].get("_ready") for j in mine} if host_dims is None else {}
        # (host arrays are snapshotted into the pinned block by _upload_batch at call time: nothing on the caller's stream can matter
        # to them, so their batches always overlap; resident objects only under the overlap_batches contract)
        overlap = host_dims is not None or (self.overlap_batches and None not in readies.values())
        for st in self._streams:
            if overlap:                    # the inputs' own events + the record buffer's last reader, not the caller's whole stream
                for ev in readies.values():
                    st.wait_event(ev)
                st.wait_event(read_done)
            else:
                st.wait_stream(main)
            if blk is not None:
                st.wait_event(blk["sent"])
        # chains of 1, 2, 4 or 8 members -- lists of equal length in those numbers keep each list on its own XCDs in the pair kernel
        # (cppf_pair_mlp_batch_plan) -- the smallest such length that gives every lane at most one chain, capped at 8: 8 objects on 3
        # lanes run as 4 + 4 (0.145 ms per object; 3 + 3 + 2: 0.151, 2 + 2 + 2 + 2: 0.161), 16 as 8 + 8, 24 as 8 + 8 + 8
        # (profiles/r6_resident_probe.txt)
        L = self.chain_len or min(8, 1 << max(0, (-(-n // self.n_lanes) - 1).bit_length()))
        groups = [list(range(g, min(g + L, n))) for g in range(0, n, L)]
        ran = []
        for gi, slots in enumerate(groups):
            lane = gi % self.n_lanes
            with torch.cuda.stream(self._streams[lane]):
                pipes, objs = [], []
                for q, slot in enumerate(slots):
                    obj = objects[mine[slot]]
                    pipe = self._pipe(obj["cfg"], obj["pc"].shape[0], int(obj["n_pairs"]), obj["dims"], lane, q, idx_i32=self.idx_i32)
                    if pipe.dynamic:
                        pipe.set_shape(obj["pc"].shape[0], obj["dims"], upload=False)      # (capacity check + host bookkeeping)
                    elif obj["pc"].shape[0] != pipe.n_points or tuple(obj["dims"]) != tuple(pipe.dims):
                        raise _shape_error(obj, pipe)
                    pipes.append(pipe)
                    objs.append(obj)
                chain, capture = self._chain_for(pipes, staged=True)
                chain.run_staged(objs, [int(seed) * 1000003 + mine[slot] for slot in slots], [mine[slot] for slot in slots],
                                 local[slots[0]:slots[-1] + 1], check_weights=None, capture=capture)
                ran.append((chain, slots))
        for st in self._streams:
            main.wait_stream(st)
        if blk is not None:
            blk["free"].record(main)        # (joined: every chain that read the device block has finished before this point of `main`)
        if self.__dict__.get("own_done") is not None:      # a caller's timing event: this rank's chains, before the gather (bench.py)
            self.own_done.record(main)
        snap = self.__dict__.get("_snap")
        if snap is None or snap[0].shape[0] < n:
            snap = self._snap = (torch.zeros((max(n, 1), sharding.RECORD), dtype=torch.float64).pin_memory(), torch.cuda.Event())
        copy_words(snap[0][:n], local[:n], dev)             # (into pinned memory by a kernel: no copy engine in a batch's steady state)
        snap[1].record(main)
        self._pending = ran
        if world == 1 and not (sharding.forced() and sharding.dist.is_initialized()):
            out = local[:n_total].clone()                  # (a later batch overwrites `local`: hand the caller its own copy)
        else:
            out = sharding.gather_records(local, n_total, rank, world, dev, validate=False)
        read_done.record(main)                             # `local` has been read (in stream order) up to here
        return out

    def _stageable(self, objects, mine):
        """can these host objects take the staged chains (the standard fused pair encoder; grids of the tiled vote)?  -> {object index:
        grid dims} (computed once here, used by _upload_batch), or None"""
        out = {}
        for j in mine:
            obj = objects[j]
            enc = self.encoders.get(obj["cfg"].category)
            if enc is None or not enc.fused_decode_supported(obj["cfg"].tr_num_bins, obj["cfg"].rot_num_bins):
                return None
            _, dims = grid_shape(obj["pc"], obj["cfg"].res)
            if grid_class(dims)[0] == 0:
                return None
            out[j] = dims
        return out

    def _adapt_resident(self):
        """the split / full-first form of the chains of the last resident batch, from its survivor counts (PoseChain.adapt), once
        their asynchronous copy has landed"""
        ran = self.__dict__.get("_pending")
        if not ran or not self._snap[1].query():
            return
        host = self._snap[0].numpy()
        if np.any(host[[s for _, slots in ran for s in slots], 12] < 0):
            self._pending = None
            from ._lib import CppfError
            raise CppfError("an instance of the last batch did not fit the shape-polymorphic pipeline it ran on (arg-max index -1)")
        for chain, slots in ran:
            chain.adapt([host[s, 14] for s in slots])
        self._pending = None

    def run_object(self, obj):
        """obj: dict(pc, normals, feat, point_idxs, u_tr, u_rot, cfg) of host arrays -> pose dict."""
        self._check(0, obj)
        corners, dims = grid_shape(obj["pc"], obj["cfg"].res)
        pipe = self._pipe(obj["cfg"], obj["pc"].shape[0], obj["point_idxs"].shape[0], dims)
        feat = obj.get("feat") if pipe.point_encoder is None else None
        pipe.load(obj["pc"], obj["normals"], feat, obj["point_idxs"], obj["u_tr"], obj["u_rot"], corners[0].copy(), dims=dims)
        return pipe.run()

    def run(self, objects, rank=0, world=1, seed=0):
        """objects: the WHOLE batch (list, same on every rank).  Returns f64[n_objects, RECORD] in object
        order on every rank (sharding.pack_record layout): on the device when a process group gathered them, on the host for
        a single rank without a group (the records were just assembled there: no upload only to be read back).

        The rank's instances are enqueued back to back -- inputs, graph replay, a 21-double device copy of the result --
        and all results are read back once at the end (one synchronisation per batch instead of one per instance).  An
        object without `point_idxs` gets its pairs (n_pairs = obj["n_pairs"]) and bin uniforms drawn on the device from
        `seed` and its index: then only the cloud itself crosses PCIe."""
        mine = sharding.shard_objects(len(objects), rank, world)
        resident = [torch.is_tensor(objects[j]["pc"]) and objects[j]["pc"].is_cuda for j in mine]
        if any(resident):
            if not all(resident) or any(objects[j].get("dims") is None for j in mine):
                raise ValueError("a batch is either host arrays or objects returned by put(): device tensors need `dims` beside them")
            return self._run_resident(objects, mine, rank, world, seed)
        # host objects whose pairs are drawn on the device take the same staged chains (uploaded on the way): the host only packs and
        # sends the clouds.  Explicit pair lists (parity tests, callers with their own draws) keep the per-instance path below.
        if self.staged_host and mine and all(objects[j].get("point_idxs") is None and "n_pairs" in objects[j] for j in mine):
            host_dims = self._stageable(objects, mine)
            if host_dims is not None:
                return self._run_resident(objects, mine, rank, world, seed, host_dims)
        raw = self.__dict__.get("_raw")           # (every row in use is overwritten by its instance's record copy)
        if raw is None or raw.shape[0] < max(len(mine), 1):
            raw = self._raw = torch.zeros((max(len(mine), 1), 21), dtype=torch.float64, device=self.device)
        raw = raw[:max(len(mine), 1)]
        cfgs, used = [], []
        # n_lanes instances in flight: consecutive instances rotate over the HIP streams, each with its own pipelines
        # (buffers + captured graph), so one instance's head overlaps the previous one's tail
        if self._streams is None:
            self._streams = self._lanes()[0]
        main = torch.cuda.current_stream(self.device)
        # weight images: looked at ONCE per batch, here on the caller's stream, before the lanes fan out -- a parameter update
        # since the last batch is re-packed (in place) now, and every lane's replays are ordered after it by the wait below
        # (a lane that found the change itself would rebuild the image on its own stream while its neighbours replay)
        for cat in {objects[j]["cfg"].category for j in mine}:
            if cat in self.encoders:
                self.encoders[cat]._packed_weights(self.device)
            if cat in self.point_encoders:
                self.point_encoders[cat]._packed_weights(self.device)
        for st in self._streams:
            st.wait_stream(main)
        L = self.chain_len or max(1, min(8, len(mine) // (2 * self.n_lanes)))
        groups = [list(range(g, min(g + L, len(mine)))) for g in range(0, len(mine), L)]     # consecutive instances share a chain
        ran = []                 # (chain or None, pipelines, slots) per group, for adapt()
        for gi, slots in enumerate(groups):
            lane = gi % self.n_lanes
            pipes = []
            with torch.cuda.stream(self._streams[lane]):
                for q, slot in enumerate(slots):
                    j = mine[slot]
                    obj = objects[j]
                    self._check(j, obj)
                    corners, dims = grid_shape(obj["pc"], obj["cfg"].res)
                    on_device = obj.get("point_idxs") is None
                    n_pairs = int(obj["n_pairs"]) if on_device else obj["point_idxs"].shape[0]
                    pipe = self._pipe(obj["cfg"], obj["pc"].shape[0], n_pairs, dims, lane, q)
                    self._stage(pipe, obj["pc"], obj["normals"], obj.get("feat") if pipe.point_encoder is None else None,
                                corners[0], dims)
                    if not on_device:
                        pipe.load(None, None, None, obj["point_idxs"], obj["u_tr"], obj["u_rot"], None)
                    else:
                        pipe.sample_inputs(int(seed) * 1000003 + j, n_points=obj["pc"].shape[0])      # a function of (seed, object index)
                    pipes.append(pipe)
                    cfgs.append(obj["cfg"])
                chain = self._chain_for(pipes)
                # (check_weights=None: the images were refreshed above, once per batch; a pipeline only compares their addresses)
                if chain is not None:
                    chain.run_async([raw[slot] for slot in slots], check_weights=None)
                else:
                    for pipe, slot in zip(pipes, slots):
                        pipe.run_async(raw[slot], check_weights=None)
                ran.append((chain, pipes, slots))
        for st in self._streams:
            main.wait_stream(st)
        host = raw.cpu().numpy()                       # the batch's only synchronisation
        for chain, pipes, slots in ran:                # split / full-first form of the next run (PosePipeline.adapt / PoseChain.adapt)
            if chain is not None:
                chain.adapt([host[slot, 18] for slot in slots])
            else:
                for pipe, slot in zip(pipes, slots):
                    pipe.adapt(host[slot, 18])
        local = torch.from_numpy(assemble_batch(host[:len(mine)], cfgs, mine, sharding.RECORD))
        if world > 1 or sharding.dist.is_initialized():       # the gather needs them on the device; a single rank returns them as they are
            local = local.to(self.device)
        return sharding.gather_records(local, len(objects), rank, world, self.device, validate=False)   # (rows built in object order above)
