"""The utilities of the reference's utils/util.py that sit on (or right in front of) the hot path."""
import numpy as np


def fibonacci_sphere(samples):
    """Sphere bins of the orientation vote (reference utils/util.py:102-118, used at
    nocs/inference.py:100-102): point i has y = 1 - 2i/(samples-1) and azimuth i times the golden
    angle.  Returns a list of (x, y, z) tuples in fp64 like the reference (callers do np.array(..))."""
    i = np.arange(samples, dtype=np.float64)
    y = 1.0 - (i / float(samples - 1)) * 2.0
    radius = np.sqrt(1.0 - y * y)
    theta = (np.pi * (3.0 - np.sqrt(5.0))) * i
    pts = np.stack([np.cos(theta) * radius, y, np.sin(theta) * radius], -1)
    return [tuple(float(v) for v in p) for p in pts]


def num_sphere_bins(angle_tol_deg):
    """nocs/inference.py:100-101"""
    return int(4 * np.pi / (angle_tol_deg / 180 * np.pi))


# --------------------------------------------------------------------------- pre-processing on device (row f3)
def _device_cloud(pc):
    """numpy or torch [N,3] -> contiguous f32 tensor on the HIP device (+ whether the caller passed numpy)"""
    import torch
    from .._torch_util import require_cuda
    require_cuda()
    was_np = isinstance(pc, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(pc, dtype=np.float32)) if was_np else pc
    if t.dim() != 2 or t.shape[1] != 3:
        raise ValueError(f"pc must be [N,3], got {tuple(t.shape)}")
    if not t.is_cuda:
        t = t.cuda()
    return t.detach().float().contiguous(), was_np


def estimate_normals(pc, knn):
    """utils/util.py:61-65 `estimate_normals(pc, knn)` (open3d PCA normals over the knn nearest neighbours, the point
    itself included) on the device: neighbour selection (csrc/sprin.hip) + a 3x3 Jacobi per point (csrc/preproc.hip).
    numpy in -> numpy f32[N,3] out, torch in -> torch out.  open3d leaves the sign of a normal unspecified; here the
    component of largest magnitude is positive (parity with open3d is unpinned, see oracle/preproc_oracle.c)."""
    import torch
    from .. import _lib
    from .._torch_util import stream_ptr
    t, was_np = _device_cloud(pc)
    N, k = t.shape[0], min(int(knn), t.shape[0])
    nbrs = torch.empty((N, k), dtype=torch.int32, device=t.device)
    out = torch.empty((N, 3), dtype=torch.float32, device=t.device)
    L = _lib.lib()
    with torch.cuda.device(t.device):
        _lib.check(L.cppf_knn(t.data_ptr(), None, N, k, nbrs.data_ptr(), stream_ptr(t.device)), "cppf_knn")
        _lib.check(L.cppf_estimate_normals(t.data_ptr(), nbrs.data_ptr(), N, k, out.data_ptr(), stream_ptr(t.device)),
                   "cppf_estimate_normals")
    return out.cpu().numpy() if was_np else out


def sparse_quantize(pc, return_index=True, quantization_size=1.0):
    """The call of nocs/inference.py:140, `ME.utils.sparse_quantize(pc, return_index=True, quantization_size=res)`, on
    the device: one representative per occupied voxel floor(p / res).  Returns (discrete_coords i32[M,3], indices
    i64[M]) like MinkowskiEngine (numpy in -> numpy out).  The representative is the lowest original index and the
    output is in index order (MinkowskiEngine's choice is hash-order dependent: parity unpinned)."""
    import torch
    from .. import _lib
    from .._torch_util import stream_ptr, workspace
    t, was_np = _device_cloud(pc)
    N = t.shape[0]
    keep = torch.empty(N, dtype=torch.int32, device=t.device)
    count = torch.zeros(1, dtype=torch.int32, device=t.device)
    L = _lib.lib()
    ws = workspace(L.cppf_voxel_dedupe_workspace_bytes(N), t.device, "voxel")
    with torch.cuda.device(t.device):
        _lib.check(L.cppf_voxel_dedupe(t.data_ptr(), N, float(quantization_size), keep.data_ptr(), count.data_ptr(),
                                       ws.data_ptr(), ws.numel(), stream_ptr(t.device)), "cppf_voxel_dedupe")
    idx = keep[:int(count.item())].long()
    coords = torch.floor(t[idx].double() / float(quantization_size)).to(torch.int32)
    if was_np:
        idx, coords = idx.cpu().numpy(), coords.cpu().numpy()
    return (coords, idx) if return_index else coords
