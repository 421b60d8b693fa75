#!/usr/bin/env python3
"""Vote + reduce + arg-max alone, per regime and configuration (ms per call, HIP events, the smallest of three brackets):
   c2 / c5 grids  x  known-answer (a trained network) / uniform-bin (a random-weight network) inputs,
plus the back-vote and the whole known-answer tail.  Kernel-development aid: python profiles/microbench/vote_regimes.py [c2|c5|all]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cppf_amd.synthetic as syn                                # noqa: E402
from cppf_amd.inference import PoseWorkspace, grid_shape       # noqa: E402
from cppf_amd.models import voting                              # noqa: E402


def bracket(fn, n=10):
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / n
        best = t if best is None else min(best, t)
    return best


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    regime = sys.argv[2] if len(sys.argv) > 2 else "all"       # known-answer | uniform-bin
    dev = torch.device("cuda:0")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # c2posed: the C2 object in an arbitrary pose (a real scene's instances are not upright): a bounding box of 9-12 tiles
    for name, (N, K, res) in (("c2", (4096, 128, None)), ("c5", (8192, 256, 2e-3)), ("c2posed", (4096, 128, None))):
        if which not in ("all", name):
            continue
        ob = syn.make_posed_object("bottle", N, 900101, rotate=True) if name == "c2posed" else syn.make_object("bottle", N, 0)
        cfg = ob["cfg"]
        r = res or cfg.res
        idx = syn.make_pairs(N, K, 0)
        P = idx.shape[0]
        corners, dims = grid_shape(ob["pc"], r)
        ka = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
        rng = np.random.default_rng(1)
        kb = rng.integers(0, 32, (P, 2))
        un = np.stack([kb[:, 0] / 31 * 2 * cfg.vote_range[0] - cfg.vote_range[0], kb[:, 1] / 31 * cfg.vote_range[1]], -1).astype(np.float32)
        pc, idx_d, corner = d(ob["pc"]), d(idx), d(corners[0])
        ws = PoseWorkspace(dev, P, dims, 1)
        for tag, out in (("known-answer", ka), ("uniform-bin", un)):
            if regime not in ("all", tag):
                continue
            o = d(out)
            fn = lambda: voting.vote_argmax(pc, o, None, idx_d, ws.grid, corner, r, 72, True, ws.out_idx, ws.out_val, accumulate=False)
            t = bracket(fn)
            g = ws.grid.double()
            print(f"{name} {tag:13s} grid {dims} P={P}: vote+reduce+argmax {t * 1e3:8.1f} us   argmax {int(ws.out_idx.item())} "
                  f"mass {float(g.sum()):.3f} checksum {float((g * torch.arange(g.numel(), device=dev).reshape(g.shape).double()).sum()):.6e}")


if __name__ == "__main__":
    main()
