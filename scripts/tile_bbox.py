"""Pricing aid (CPU, oracle): would dumping only the BOUNDING BOX of the cells a vote workgroup touched shrink its 113 KB partial
tile?  For the C2 object: one fused-path workgroup's share of the pair list (blocks c, c + C, ... of 64 pairs, C = 128 chunks per
tile: csrc/vote.hip v3_fused_chunk_pairs) is voted by the oracle, and the bounding box of the touched cells inside each of the two
tiles (26 x 38 x 26 owned cells) is compared with the tile -- on known-answer inputs (a trained network) and on the uniform-bin
inputs of a random-weight one.  python scripts/tile_bbox.py > profiles/r6_tile_bbox.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cppf_amd.synthetic as syn                     # noqa: E402
from oracle import oracle as O                       # noqa: E402

N, K, C = 4096, 128, 128
ob = syn.make_object("bottle", N, 0)
cfg = ob["cfg"]
idx = syn.make_pairs(N, K, 0)
P = idx.shape[0]
corner, dims = O.grid_setup(ob["pc"], cfg.res)
ka = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
rng = np.random.default_rng(1)
kb = rng.integers(0, 32, (P, 2))
un = np.stack([kb[:, 0] / 31 * 2 * cfg.vote_range[0] - cfg.vote_range[0], kb[:, 1] / 31 * cfg.vote_range[1]], -1).astype(np.float32)
ty = (int(dims[1]) + 1) // 2
print(f"C2 object: grid {tuple(int(d) for d in dims)}, two tiles of {int(dims[0])} x {ty} x {int(dims[2])} owned cells, {P} pairs, {C} chunks per tile")
for tag, out in (("known-answer", ka), ("uniform-bin", un)):
    fracs, filled = [], []
    for c in (0, 17, 63, 127):
        blocks = np.arange(c, (P + 63) // 64, C)
        sel = (blocks[:, None] * 64 + np.arange(64)[None]).reshape(-1)
        sel = sel[sel < P]
        _, cnt = O.ppf_voting_f64(ob["pc"], out[sel], np.ones(N, np.float32), idx[sel].astype(np.int32), dims, corner, cfg.res, 72, True)
        for t in range(2):
            sub = cnt[:, t * ty:min((t + 1) * ty, int(dims[1])), :]
            nz = np.argwhere(sub > 0)
            if len(nz) == 0:
                fracs.append(0.0)
                filled.append(0.0)
                continue
            ext = nz.max(0) - nz.min(0) + 1
            fracs.append(float(np.prod(ext)) / sub.size)
            filled.append(float((sub > 0).mean()))
    print(f"{tag:13s}: bounding box of a workgroup's touched cells = {min(fracs):.2f} .. {max(fracs):.2f} of its tile "
          f"(cells actually touched: {min(filled):.2f} .. {max(filled):.2f}) over chunks 0 / 17 / 63 / 127 of both tiles")
print("=> a dense sub-box dump would write the whole tile in both regimes: the pairs of a chunk are random pairs of the whole cloud")
