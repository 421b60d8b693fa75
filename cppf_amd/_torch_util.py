"""torch plumbing for the C ABI: pointer extraction, argument checks, stream, scratch cache."""
import threading

import numpy as np
import torch

from . import _lib

_ws_cache = {}


def stream_ptr(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def copy_words(dst, src, device):
    """dst <- src (tensors of 8-byte elements, contiguous, the same number of elements; either may be PINNED host memory) by a kernel
    of the current stream of `device`, not by a copy engine (cppf_copy_words: small copies on the SDMA queues can wait behind another
    stream's transfers)"""
    n = dst.numel()
    assert src.numel() == n and dst.element_size() == 8 and src.element_size() == 8 and dst.is_contiguous() and src.is_contiguous()
    with torch.cuda.device(device):
        _lib.check(_lib.lib().cppf_copy_words(dst.data_ptr(), src.data_ptr(), n, stream_ptr(device)), "cppf_copy_words")


def gather_words(dst, srcs, device):
    """dst[r] <- srcs[r] (dst: [len(srcs), W] of 8-byte elements or an equivalent view; srcs: device tensors of at least W such words
    each, in different allocations) with ONE launch on the current stream (cppf_gather_words) instead of a small copy per row"""
    import ctypes as C
    n = len(srcs)
    if n == 0:
        return
    words = dst.numel() * dst.element_size() // 8 // n
    assert dst.is_contiguous() and dst.numel() * dst.element_size() == n * words * 8
    for i in range(0, n, 32):
        part = srcs[i:i + 32]
        ptrs = (C.c_void_p * len(part))(*[t_.data_ptr() for t_ in part])
        with torch.cuda.device(device):
            _lib.check(_lib.lib().cppf_gather_words(len(part), ptrs, words, dst.data_ptr() + i * words * 8, stream_ptr(device)), "cppf_gather_words")


_lane_cache = {}


def _runs_beside(ref, cand, device, spin_cycles):
    """does work on `cand` execute WHILE `ref` is busy?  ref spins for a few hundred microseconds; an event recorded on cand right
    after must complete well before the spin ends.  Streams that share a hardware queue are served in order, so it would not."""
    e_ref, e_c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    with torch.cuda.stream(ref):
        t0.record()
        torch.cuda._sleep(spin_cycles)
        t1.record()
    with torch.cuda.stream(cand):
        e_c.record()
    torch.cuda.synchronize(device)
    spin, lag = t0.elapsed_time(t1), t0.elapsed_time(e_c)
    return spin > 0.05 and lag < 0.5 * spin


def lane_streams(device, n):
    """n HIP streams for work that must overlap.  The HIP runtime multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4
    by default) chosen when a stream is created, and streams that share a queue are served IN ORDER: two lanes on one queue run one
    after the other (measured: 8-object batches 0.13 against 0.16 ms per object, the headline 6.0 against 5.75 G pairs/s, decided
    by how many streams the process happened to create before -- a pool stream's queue is its creation index modulo the queues).
    Which queue a stream got cannot be asked, so it is measured: candidates are probed pairwise with a spinning kernel
    (_runs_beside: profiles/r6_stream_queues.txt holds such a matrix) and a set whose members run beside each other is kept (per
    device and caller stream; as many as the queues allow, the rest filled with whatever is left).  The caller's stream is tested one
    way only -- its work must not queue behind a lane's; nothing runs beside a kernel of the legacy default stream in this probe."""
    require_cuda()
    import os
    if os.environ.get("CPPF_NO_LANE_PROBE"):          # (development knob: plain pool streams, whatever queues they got)
        return [torch.cuda.Stream(device=device) for _ in range(n)]
    main = torch.cuda.current_stream(device)
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), main.cuda_stream)
    have = _lane_cache.setdefault(key, [])
    if len(have) >= n:
        return have[:n]
    spin = 400_000                                    # ~0.2-0.4 ms of s_sleep at the shader clock
    cands = [torch.cuda.Stream(device=device) for _ in range(12)]
    good, rest = list(have), []
    for c in cands:
        if len(good) >= n:
            break
        try:
            # (the caller's stream must not be stuck behind a lane either: its joins and the records' hand-over live there)
            ok = _runs_beside(c, main, device, spin) and all(_runs_beside(g, c, device, spin) and _runs_beside(c, g, device, spin) for g in good)
        except (RuntimeError, AttributeError):       # (no spin kernel in this torch build, a probe disturbed: take the stream as it is)
            ok = False
        (good if ok else rest).append(c)
    while len(good) < n and rest:                     # fewer independent queues than lanes: the remaining lanes share
        good.append(rest.pop(0))
    while len(good) < n:
        good.append(torch.cuda.Stream(device=device))
    _lane_cache[key] = good
    return good[:n]


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
