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


def backproject(depth, intrinsics, instance_mask, return_device=False):
    """utils/util.py:598-631 `backproject(depth, intrinsics, instance_mask)` (nocs/inference.py:131) on the device: the
    pixels of the instance mask with depth > 0, in row-major order, through inv(intrinsics) in fp64, x and y negated like
    the reference.  depth: [H,W] uint16 (NOCS depth PNGs) / float32 / anything castable, numpy or device tensor;
    instance_mask: [H,W] bool / uint8.  Returns (pts f64[n,3], (rows, cols)) as numpy like the reference, or device tensors
    (pts, flat pixel index i32[n]) with return_device=True (no host round trip: the cloud can go straight into
    sparse_quantize / estimate_normals / the encoders).  The depth image of a frame can be uploaded once and reused for
    all of its instances (pass the device tensor)."""
    import torch
    from .. import _lib
    from .._torch_util import require_cuda, stream_ptr, workspace
    require_cuda()

    def dev(x, kinds):
        t = torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x
        if t.dtype not in kinds:
            t = t.to(kinds[-1])
        return (t if t.is_cuda else t.cuda()).contiguous()

    d = depth
    if isinstance(d, np.ndarray) and d.dtype == np.uint16:
        d = torch.from_numpy(np.ascontiguousarray(d).view(np.int16))       # torch has no uint16 arithmetic; bits are what matter
    d = dev(d, (torch.int16, torch.float32))
    m = dev(instance_mask, (torch.uint8, torch.bool, torch.uint8))
    if m.dtype == torch.bool:
        m = m.view(torch.uint8)
    if d.dim() != 2 or tuple(m.shape) != tuple(d.shape):
        raise ValueError(f"depth {tuple(d.shape)} and instance_mask {tuple(m.shape)} must be the same [H,W]")
    H, W = d.shape
    kinv = np.ascontiguousarray(np.linalg.inv(np.asarray(intrinsics, np.float64)))
    pts = torch.empty((H * W, 3), dtype=torch.float64, device=d.device)
    pix = torch.empty(H * W, dtype=torch.int32, device=d.device)
    count = torch.zeros(1, dtype=torch.int32, device=d.device)
    L = _lib.lib()
    ws = workspace(L.cppf_backproject_workspace_bytes(H, W), d.device, "backproject")
    with torch.cuda.device(d.device):
        _lib.check(L.cppf_backproject(d.data_ptr(), 1 if d.dtype == torch.int16 else 0, m.data_ptr(), H, W, kinv.ctypes.data,
                                      pts.data_ptr(), pix.data_ptr(), count.data_ptr(), ws.data_ptr(), ws.numel(),
                                      stream_ptr(d.device)), "cppf_backproject")
    n = int(count.item())
    if return_device:
        return pts[:n], pix[:n]
    p = pix[:n].cpu().numpy().astype(np.int64)
    return pts[:n].cpu().numpy(), (p // W, p % W)


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


# --------------------------------------------------------------------------- depth image I/O
def read_depth_png(path):
    """The depth read of nocs/inference.py:110 -- `cv2.imread(path, -1)` on a NOCS `*_depth.png` -- without OpenCV: an
    8- or 16-bit greyscale, non-interlaced PNG -> uint8 / uint16 [H,W] (millimetres in the NOCS frames).  PNG filtering
    (types 0-4) undone with numpy; anything else the format allows (palette, colour, interlacing) is refused."""
    import struct
    import zlib
    with open(path, "rb") as f:
        b = f.read()
    if b[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(b):
        n, = struct.unpack(">I", b[pos:pos + 4])
        typ = b[pos + 4:pos + 8]
        body = b[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
        pos += 12 + n
    if hdr is None:
        raise ValueError(f"{path}: no IHDR chunk")
    W, H, depth, colour, _, _, interlace = hdr
    if colour != 0 or interlace != 0 or depth not in (8, 16):
        raise ValueError(f"{path}: only non-interlaced 8/16-bit greyscale PNGs are depth images (colour type {colour}, "
                         f"bit depth {depth}, interlace {interlace})")
    bpp = depth // 8
    stride = W * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8)
    if raw.size != H * (stride + 1):
        raise ValueError(f"{path}: {raw.size} bytes of image data, expected {H * (stride + 1)}")
    raw = raw.reshape(H, stride + 1)
    out = np.zeros((H, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(H):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:                                   # Up
            cur = (line + prev) & 255
        elif ft == 1:                                   # Sub: a running sum per byte lane
            cur = (np.cumsum(line.reshape(-1, bpp), axis=0) & 255).reshape(-1)
        elif ft in (3, 4):                              # Average / Paeth depend on the reconstructed left neighbour: byte by byte
            cur = np.zeros(stride, dtype=np.int32)
            for i in range(stride):
                a = int(cur[i - bpp]) if i >= bpp else 0
                up = int(prev[i])
                if ft == 3:
                    pred = (a + up) >> 1
                else:
                    c = int(prev[i - bpp]) if i >= bpp else 0
                    p = a + up - c
                    pa, pb, pc_ = abs(p - a), abs(p - up), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc_) else (up if pb <= pc_ else c)
                cur[i] = (int(line[i]) + pred) & 255
        else:
            raise ValueError(f"{path}: bad filter type {ft} in row {y}")
        out[y] = cur
        prev = cur
    if bpp == 1:
        return out
    return (out[:, 0::2].astype(np.uint16) << 8) | out[:, 1::2].astype(np.uint16)     # PNG samples are big-endian
