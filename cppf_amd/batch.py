"""A batch of object instances (BASELINE.json config 4: mixed NOCS categories) on one or many GPUs.

Objects are independent, so the batch is sharded round-robin over the ranks (object j -> rank j mod W,
cppf_amd.sharding), each rank runs its objects back to back on its GPU -- one hipGraph replay per object,
pipelines cached per (category, N, P, grid dims) -- and ONE all_gather of the fixed-size records closes
the batch.  Mirrors the per-instance loop of nocs/inference.py:120-339 (different categories use different
encoders and configs, :124-128)."""
import numpy as np
import torch

from . import sharding
from .inference import PosePipeline, assemble_record, grid_shape
from .utils.util import fibonacci_sphere, num_sphere_bins


class BatchPoseRunner:
    def __init__(self, encoders, device, num_rots=72, adaptive=True, angle_tol=1.5, max_rot_pairs=10000,
                 use_graph=True, point_encoders=None):
        """encoders: {category name: PPFEncoder on `device`} (the reference keeps one per category,
        nocs/inference.py:79-90).  point_encoders: optional {category name: PointEncoder}; objects of those
        categories need no `feat` -- kNN + SPRIN run at the head of the captured graph (:180-181)."""
        self.encoders, self.device = encoders, device
        self.point_encoders = point_encoders or {}
        self.kw = dict(num_rots=num_rots, adaptive=adaptive, angle_tol=angle_tol, max_rot_pairs=max_rot_pairs,
                       use_graph=use_graph)
        self.sphere = np.array(fibonacci_sphere(num_sphere_bins(angle_tol)))      # :100-102
        self._pipes = {}
        self._staging = {}         # pinned host staging sets for the small per-instance arrays, see _stage()
        self._stage_pos = 0
        self._streams = None

    def _pipe(self, cfg, n_points, n_pairs, dims, lane=0):
        key = (cfg.category, n_points, n_pairs, tuple(dims), lane)
        if key not in self._pipes:
            self._pipes[key] = PosePipeline(self.encoders[cfg.category], cfg, n_points, n_pairs, dims, self.device,
                                            self.sphere, point_encoder=self.point_encoders.get(cfg.category), **self.kw)
        return self._pipes[key]

    _RING = 4

    def _stage(self, pipe, pc, normals, feat, corner):
        """cloud, normals, (features,) grid corner -> the pipeline's device buffers through PINNED host memory, so the copies
        are truly asynchronous (a copy from pageable memory blocks the host until everything queued before it has run, i.e.
        until the previous instance has finished).  A ring of staging sets, each guarded by an event."""
        n = pc.shape[0]
        key = (n, None if feat is None else feat.shape[1])
        ring = self._staging.get(key)
        if ring is None:
            mk = lambda *shape: torch.empty(shape, dtype=torch.float32).pin_memory()
            ring = [dict(pc=mk(n, 3), nrm=mk(n, 3), corner=mk(3), feat=None if feat is None else mk(n, feat.shape[1]),
                         ev=torch.cuda.Event()) for _ in range(self._RING)]
            self._staging[key] = ring
        st = ring[self._stage_pos % self._RING]
        self._stage_pos += 1
        st["ev"].synchronize()                          # the copies that last read this set have executed
        st["pc"].numpy()[...] = pc
        st["nrm"].numpy()[...] = normals
        st["corner"].numpy()[...] = corner
        pipe.pc.copy_(st["pc"], non_blocking=True)
        pipe.nrm.copy_(st["nrm"], non_blocking=True)
        pipe.corner.copy_(st["corner"], non_blocking=True)
        if feat is not None:
            st["feat"].numpy()[...] = feat
            pipe.feat.copy_(st["feat"], non_blocking=True)
        st["ev"].record(torch.cuda.current_stream(self.device))   # (the lane's stream: _stage is called inside its context)

    def run_object(self, obj):
        """obj: dict(pc, normals, feat, point_idxs, u_tr, u_rot, cfg) of host arrays -> pose dict."""
        corners, dims = grid_shape(obj["pc"], obj["cfg"].res)
        pipe = self._pipe(obj["cfg"], obj["pc"].shape[0], obj["point_idxs"].shape[0], dims)
        pipe.load(obj["pc"], obj["normals"], obj.get("feat"), obj["point_idxs"], obj["u_tr"], obj["u_rot"],
                  corners[0].copy())
        return pipe.run()

    def run(self, objects, rank=0, world=1, seed=0):
        """objects: the WHOLE batch (list, same on every rank).  Returns f64[n_objects, RECORD] in object
        order on every rank (sharding.pack_record layout).

        The rank's instances are enqueued back to back -- inputs, graph replay, a 21-double device copy of the result --
        and all results are read back once at the end (one synchronisation per batch instead of one per instance).  An
        object without `point_idxs` gets its pairs (n_pairs = obj["n_pairs"]) and bin uniforms drawn on the device from
        `seed` and its index: then only the cloud itself crosses PCIe."""
        mine = sharding.shard_objects(len(objects), rank, world)
        raw = torch.zeros((max(len(mine), 1), 21), dtype=torch.float64, device=self.device)
        cfgs = []
        # two instances in flight: consecutive instances alternate between two HIP streams, each with its own pipelines
        # (buffers + captured graph), so one instance's head overlaps the previous one's tail
        if self._streams is None:
            self._streams = [torch.cuda.Stream(device=self.device) for _ in range(2)]
        main = torch.cuda.current_stream(self.device)
        for st in self._streams:
            st.wait_stream(main)
        for slot, j in enumerate(mine):
            obj = objects[j]
            corners, dims = grid_shape(obj["pc"], obj["cfg"].res)
            on_device = obj.get("point_idxs") is None
            n_pairs = int(obj["n_pairs"]) if on_device else obj["point_idxs"].shape[0]
            lane = slot & 1
            pipe = self._pipe(obj["cfg"], obj["pc"].shape[0], n_pairs, dims, lane)
            with torch.cuda.stream(self._streams[lane]):
                self._stage(pipe, obj["pc"], obj["normals"], obj.get("feat") if pipe.point_encoder is None else None, corners[0])
                if not on_device:
                    pipe.load(None, None, None, obj["point_idxs"], obj["u_tr"], obj["u_rot"], None)
                if on_device:
                    gen = torch.Generator(device=self.device)
                    gen.manual_seed(int(seed) * 1000003 + j)
                    pipe.sample_inputs(gen)
                pipe.run_async(raw[slot])
            cfgs.append(obj["cfg"])
        for st in self._streams:
            main.wait_stream(st)
        host = raw.cpu().numpy()                       # the batch's only synchronisation
        recs = [sharding.pack_record(j, assemble_record(host[slot], cfgs[slot])) for slot, j in enumerate(mine)]
        local = torch.stack(recs).to(self.device) if recs else \
            torch.zeros((0, sharding.RECORD), dtype=torch.float64, device=self.device)
        return sharding.gather_records(local, len(objects), rank, world, self.device)
