"""A batch of object instances (BASELINE.json config 4: mixed NOCS categories) on one or many GPUs.

Objects are independent, so the batch is sharded round-robin over the ranks (object j -> rank j mod W,
cppf_amd.sharding), each rank runs its objects back to back on its GPU -- one hipGraph replay per object,
pipelines cached per (category, N, P, grid dims) -- and ONE all_gather of the fixed-size records closes
the batch.  Mirrors the per-instance loop of nocs/inference.py:120-339 (different categories use different
encoders and configs, :124-128)."""
import numpy as np
import torch

from . import sharding
from .inference import PosePipeline, grid_shape
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

    def _pipe(self, cfg, n_points, n_pairs, dims):
        key = (cfg.category, n_points, n_pairs, tuple(dims))
        if key not in self._pipes:
            self._pipes[key] = PosePipeline(self.encoders[cfg.category], cfg, n_points, n_pairs, dims, self.device,
                                            self.sphere, point_encoder=self.point_encoders.get(cfg.category), **self.kw)
        return self._pipes[key]

    def run_object(self, obj):
        """obj: dict(pc, normals, feat, point_idxs, u_tr, u_rot, cfg) of host arrays -> pose dict."""
        corners, dims = grid_shape(obj["pc"], obj["cfg"].res)
        pipe = self._pipe(obj["cfg"], obj["pc"].shape[0], obj["point_idxs"].shape[0], dims)
        pipe.load(obj["pc"], obj["normals"], obj.get("feat"), obj["point_idxs"], obj["u_tr"], obj["u_rot"],
                  corners[0].copy())
        return pipe.run()

    def run(self, objects, rank=0, world=1):
        """objects: the WHOLE batch (list, same on every rank).  Returns f64[n_objects, RECORD] in object
        order on every rank (sharding.pack_record layout)."""
        mine = sharding.shard_objects(len(objects), rank, world)
        recs = [sharding.pack_record(j, self.run_object(objects[j])) for j in mine]
        local = torch.stack(recs).to(self.device) if recs else \
            torch.zeros((0, sharding.RECORD), dtype=torch.float64, device=self.device)
        return sharding.gather_records(local, len(objects), rank, world, self.device)
