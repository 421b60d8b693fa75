"""torch plumbing for the C ABI: pointer extraction, argument checks, stream, scratch cache."""
import threading

import numpy as np
import torch

from . import _lib

_ws_cache = {}


def stream_ptr(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda():
    if not torch.cuda.is_available():
        raise _lib.CppfError("cppf_amd needs a HIP device (torch.cuda.is_available() is False); "
                             "there is no CPU fallback")


def dev_tensor(x, dtype, name, shape_tail=None, device=None):
    """Validate a device array argument the way the reference's CuPy launch would need it:
    right dtype, C-contiguous, on a HIP device.  Raises TypeError/ValueError otherwise."""
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor on a HIP device, got {type(x).__name__}")
    if not x.is_cuda:
        raise ValueError(f"{name}: tensor must live on a HIP device (got {x.device})")
    if x.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {x.dtype}")
    if not x.is_contiguous():
        raise ValueError(f"{name}: tensor must be C-contiguous")
    if shape_tail is not None and tuple(x.shape[-len(shape_tail):]) != tuple(shape_tail):
        raise ValueError(f"{name}: expected trailing shape {tuple(shape_tail)}, got {tuple(x.shape)}")
    if device is not None and x.device != device:
        raise ValueError(f"{name}: on {x.device}, expected {device}")
    return x


def scalar(v):
    """cp.float32 / np.float32 / python numbers / 0-d tensors passed by value."""
    if isinstance(v, torch.Tensor):
        return v.item()
    if isinstance(v, np.generic):
        return v.item()
    return v


_ws_tls = threading.local()          # the enclosing scope is per host thread: two threads may drive two pipelines


class workspace_scope:
    """Scratch buffers requested inside the block belong to `owner` instead of to the current stream.  A captured
    pipeline needs this: all graph captures run on torch's one capture stream, so per-stream scratch would be SHARED by
    every captured pipeline -- fine while they replay one after the other, a data race once two of them are in flight on
    two streams (the per-point table, the vote's partial tiles, the reduction scratch)."""

    def __init__(self, owner):
        self.owner = owner

    def __enter__(self):
        self.prev, _ws_tls.scope = getattr(_ws_tls, "scope", None), self.owner
        return self

    def __exit__(self, *exc):
        _ws_tls.scope = self.prev
        return False


def release_scope(owner):
    """Drop every scratch buffer that belongs to workspace_scope(owner)."""
    for key in [k for k in _ws_cache if k[1] == ("scope", owner)]:
        del _ws_cache[key]


def workspace(nbytes, device, tag="ws", zero=False):
    """Grow-only scratch per (device, owner, tag); owner = the enclosing workspace_scope, else the current stream (reuse on
    one stream is stream-ordered).  zero=True: a fresh allocation starts zero-filled (the vote's workspace keeps a header
    between calls that must start at zero, include/cppf.h: cppf_vote_workspace_init_bytes)."""
    scope = getattr(_ws_tls, "scope", None)
    owner = ("scope", scope) if scope is not None else torch.cuda.current_stream(device).cuda_stream
    key = (device, owner, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        n = max(int(nbytes), 256)
        if zero:
            buf = torch.empty(n, dtype=torch.uint8, device=device)
            buf[:min(n, int(_lib.lib().cppf_vote_workspace_init_bytes()))].zero_()
        else:
            buf = torch.empty(n, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf
