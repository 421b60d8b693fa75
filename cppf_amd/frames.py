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
    pc = pts / 1000.0                                                              # :132
    if jitter is not None:                                                         # :134
        j = torch.as_tensor(jitter, dtype=torch.float64, device=pc.device)[:pc.shape[0]]
        pc = pc + torch.clamp(cfg.res / 4 * j, -cfg.res / 2, cfg.res / 2)
    pc = torch.stack([-pc[:, 0], -pc[:, 1], pc[:, 2]], -1)                         # :136-137
    _, keep = sparse_quantize(pc.float(), return_index=True, quantization_size=cfg.res)   # :140 (ME.utils.sparse_quantize)
    pc = pc[keep].float().contiguous()                                             # :141
    return pc, estimate_normals(pc, cfg.knn)                                       # :142


def frame_poses(depth, instances, encoders, point_encoders, intrinsics=NOCS_INTRINSICS, n_pairs=100000, seed=0, device=None,
                angle_tol=1.5, num_rots=72, cfgs=None):
    """depth: uint16 [H,W]; instances: list of (category name, mask [H,W]); encoders / point_encoders: {category: module on the
    device}; cfgs: {category: CategoryConfig} (default: cppf_amd.config.CATEGORIES, the reference's yaml values).  Returns a list of pose dicts (None for an instance with fewer points than the kNN needs, like the reference's
    skip at :121-123), each with `n_points` added."""
    dev = device or torch.device("cuda", 0)
    d_dev = torch.from_numpy(np.ascontiguousarray(depth).view(np.int16)).to(dev)   # one upload per frame
    sphere = np.array(fibonacci_sphere(num_sphere_bins(angle_tol)))                # :100-102
    gen = torch.Generator(device=dev)
    out = []
    for i, (cat, mask) in enumerate(instances):
        enc, penc, cfg = encoders[cat], point_encoders[cat], (cfgs or CATEGORIES)[cat]
        pc, nrm = instance_cloud(d_dev, intrinsics, mask, cfg)
        n = pc.shape[0]
        if n < cfg.knn + 1:
            out.append(None)
            continue
        gen.manual_seed(seed * 1000003 + i)
        idx = torch.randint(0, n, (n_pairs, 2), device=dev, generator=gen)         # :177
        u = torch.rand((2, n_pairs, 2), device=dev, generator=gen)                 # stands in for torch.multinomial's draws (:186,250)
        with torch.no_grad():
            feat = penc(pc[None], nrm[None])[0]                                    # :180-181
            pose = estimate_pose(enc, pc, nrm, feat, idx, u[0], u[1], cfg, sphere, num_rots=num_rots, angle_tol=angle_tol)
        pose["n_points"] = n
        out.append(pose)
    return out
