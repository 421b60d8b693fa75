#!/usr/bin/env python3
"""Duration of the pair stage (point_proj_kernel + pair_mlp_kernel, centre heads) against the number of pairs: the fixed part of a
launch (prologue, pipeline fill, tail) and the per-pair rate.  python profiles/microbench/pair_scaling.py  (on an MI355X)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cppf_amd.synthetic as syn                                # noqa: E402
from cppf_amd.models.model import PPFEncoder                    # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    enc = PPFEncoder([84, 32, 32, 16], 141).to(dev).eval()
    N = 4096
    ob = syn.make_object("bottle", N, 0)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    pc, nrm, feat = d(ob["pc"]), d(ob["normals"]), d(ob["feat"])
    rows = []
    for K in (8, 16, 32, 64, 128, 256, 512):
        idx = d(syn.make_pairs(N, K, 0))
        P = idx.shape[0]
        u = torch.rand((P, 2), device=dev)
        fn = lambda: enc.forward_decode(pc, nrm, feat, idx, u, ob["cfg"].vote_range)
        for _ in range(3):
            fn()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        rows.append((P, best * 1e3))
        print(f"P = {P:8d}: {best * 1e3:8.1f} us per call (projection + pair kernel + the host side of the call)")
    P = np.array([r[0] for r in rows[3:]], float)
    t = np.array([r[1] for r in rows[3:]])
    b, a = np.polyfit(P, t, 1)
    print(f"fit over P >= {int(P[0])}: {a:.1f} us + {b * 1e6:.1f} us per million pairs")


if __name__ == "__main__":
    main()
