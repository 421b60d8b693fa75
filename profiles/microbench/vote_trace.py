#!/usr/bin/env python3
"""Per-workgroup wall-clock phases of v3_vote_kernel (development aid; needs a library built with -DV3_TRACE:
   hipcc ... -DV3_TRACE -c vote.hip; CPPF_SO=<that .so> python profiles/microbench/vote_trace.py [c2|c5] [known-answer|uniform-bin]).
Stamps (100 MHz wall clock): entry, prologue done, every wave's main-loop exit, barrier, end -- written into the unused tail of the
vote workspace's extra plane."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cppf_amd.synthetic as syn                                # noqa: E402
from cppf_amd.inference import PoseWorkspace, grid_shape       # noqa: E402
from cppf_amd.models import voting                              # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    tag = sys.argv[2] if len(sys.argv) > 2 else "known-answer"
    N, K, res = {"c2": (4096, 128, None), "c5": (8192, 256, 2e-3)}[name]
    dev = torch.device("cuda:0")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ob = syn.make_object("bottle", N, 0)
    cfg = ob["cfg"]
    r = res or cfg.res
    idx = syn.make_pairs(N, K, 0)
    P = idx.shape[0]
    corners, dims = grid_shape(ob["pc"], r)
    if tag == "known-answer":
        out = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
    else:
        kb = np.random.default_rng(1).integers(0, 32, (P, 2))
        out = np.stack([kb[:, 0] / 31 * 2 * cfg.vote_range[0] - cfg.vote_range[0], kb[:, 1] / 31 * cfg.vote_range[1]], -1).astype(np.float32)
    pc, idx_d, corner, o = d(ob["pc"]), d(idx), d(corners[0]), d(out)
    ws = PoseWorkspace(dev, P, dims, 1)
    fn = lambda: voting.vote_argmax(pc, o, None, idx_d, ws.grid, corner, r, 72, True, ws.out_idx, ws.out_val, accumulate=False)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    from cppf_amd import _lib, _torch_util
    vws = [buf for key, buf in _torch_util._ws_cache.items() if key[2] == "vote"][0]
    raw = vws.cpu().numpy().view(np.uint8)
    init = int(_lib.lib().cppf_vote_workspace_init_bytes())
    plane_bytes = 64 * 30720 * 8
    plane_off = init - plane_bytes
    tr = raw[plane_off + 1500000 * 8: plane_off + 1500000 * 8 + 1024 * 32 * 8].view(np.uint64).reshape(1024, 32).astype(np.int64)
    live = tr[:, 3] > 0
    wg_index = np.nonzero(live)[0]
    tr = tr[live]
    t0 = tr[:, 0].min()
    us = lambda x: (x - t0) / 100.0
    print(f"{name} {tag}: {len(tr)} workgroups traced; kernel span {us(tr[:, 3].max()):.1f} us")
    for label, col in (("entry", 0), ("prologue done", 1), ("main loop done (barrier)", 2), ("end", 3)):
        v = us(tr[:, col])
        print(f"  {label:28s} min {v.min():7.1f}  median {np.median(v):7.1f}  max {v.max():7.1f} us")
    w = us(tr[:, 8:24])
    spread = w.max(1) - w.min(1)
    print(f"  wave exit spread inside a workgroup: median {np.median(spread):.1f}  max {spread.max():.1f} us")
    for name_, groups in (("wave w on SIMD w mod 4", [list(range(q, 16, 4)) for q in range(4)]), ("wave w on SIMD w div 4", [list(range(4 * q, 4 * q + 4)) for q in range(4)])):
        fin = np.stack([w[:, g].max(1) for g in groups], 1)          # when a SIMD's last wave left the main loop
        first = np.stack([w[:, g].min(1) for g in groups], 1)
        print(f"  [{name_}] SIMD finish spread in a workgroup: median {np.median(fin.max(1) - fin.min(1)):.1f} us; mean idle of a SIMD before the barrier "
              f"{np.mean(fin.max(1)[:, None] - fin):.1f} us; mean time a SIMD runs with < 4 waves {np.mean(fin - first):.1f} us")
    hw = tr[:, 24:32]
    print("  HW_ID simd field ([5:4]) of waves 0..7 in the first workgroups:", [[int((h >> 4) & 3) for h in row] for row in hw[:4]])
    main_t = us(tr[:, 2]) - us(tr[:, 1])
    for x in range(8):
        m = wg_index % 8 == x
        print(f"  workgroups = {x} mod 8 (one XCD): main loop min {main_t[m].min():.1f} median {np.median(main_t[m]):.1f} max {main_t[m].max():.1f} us")
    order = np.argsort(main_t)
    print("  slowest workgroups:", [(int(wg_index[i]), round(float(main_t[i]), 1)) for i in order[-8:]], " fastest:", [(int(wg_index[i]), round(float(main_t[i]), 1)) for i in order[:8]])
    for t in np.unique(tr[:, 4]):
        m = tr[:, 4] == t
        print(f"  tile {t}: passes per workgroup {np.median(tr[m, 6]):.0f}, batches {np.median(tr[m, 7]):.0f}, us per 1000 passes {np.median(main_t[m] / np.maximum(tr[m, 6], 1)) * 1e3:.1f}")
        print(f"  tile {t}: {m.sum()} workgroups, main loop min {main_t[m].min():.1f} median {np.median(main_t[m]):.1f} max {main_t[m].max():.1f} us; "
              f"dump median {np.median(us(tr[m, 3]) - us(tr[m, 2])):.1f} us")


if __name__ == "__main__":
    main()

