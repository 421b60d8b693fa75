#!/usr/bin/env python3
"""What the arithmetic freedom of the reference's toolchain can change (oracle/voting_variants.c): per-sample flip rates of every
discrete outcome of ppf_voting, and the pose computed with every vote kernel under each variant, against this repo's oracle
(variant 0 = what the HIP kernels reproduce bit for bit).  CPU only.

    python tests/fmad_sensitivity.py [--full] > profiles/r3_fmad_sensitivity.txt

--full adds the BASELINE.json config sizes C2 (N=4096 K=128) and C5 (N=8192 K=256, res 2e-3); without it only the sizes the
CPU test suite uses."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cppf_amd.synthetic as syn          # noqa: E402  (numpy only)
from oracle import oracle as O            # noqa: E402


def make_case(cat, N, K, seed, mode, res=None):
    """inputs of the vote stage: (mu, nu) per pair + decoded heads per pair
    mode 'ka': closed-form (mu, nu) w.r.t. the true centre, quantised to the 32 bins (what a trained network emits);
         'uniform': every bin equally likely (what a random-weight network emits)"""
    ob = syn.make_object(cat, N, seed)
    cfg = ob["cfg"]
    idx = syn.make_pairs(N, K, seed)
    P = idx.shape[0]
    rng = np.random.default_rng(seed + 17)
    if mode == "ka":
        out = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
    else:
        k = rng.integers(0, 32, (P, 2))
        out = np.stack([k[:, 0] / 31 * 2 * cfg.vote_range[0] - cfg.vote_range[0], k[:, 1] / 31 * cfg.vote_range[1]], -1)
        out = out.astype(np.float32)
    if mode == "ka":        # ... and the orientation / scale heads of a perfectly trained network (utils/dataset.py:38-60)
        heads = syn.closed_form_heads(ob["pc"], ob["normals"], idx, cfg, quantise=True, seed=seed)
    else:
        heads = np.zeros((P, 8), np.float32)
        heads[:, 0:2] = (rng.integers(0, 36, (P, 2)) / 35 * np.pi).astype(np.float32)      # theta = k / 35 * pi
        heads[:, 2:4] = rng.standard_normal((P, 2)).astype(np.float32)                      # aux logits
        heads[:, 4:7] = (0.1 * rng.standard_normal((P, 3))).astype(np.float32)              # log-scale
    ocfg = dict(res=res or cfg.res, tr_num_bins=32, rot_num_bins=36, vote_range=cfg.vote_range, scale_mean=cfg.scale_mean,
                regress_right=cfg.regress_right, ppffcs=[84, 32, 32, 16], out_dim=141, up_sym=cfg.up_sym)
    return ob, idx, out, heads, ocfg


def study(cat, N, K, seed, mode, res=None, variants=None, out=sys.stdout):
    """returns {variant name: dict of flip counts and pose differences}; prints one line per variant"""
    ob, idx, outputs, heads, ocfg = make_case(cat, N, K, seed, mode, res)
    sph = O.fibonacci_sphere(480)
    t0 = time.time()
    base = O.pose_tail_variant(ob["pc"], ob["normals"], idx, outputs, heads, ocfg, sph, 0)
    n0 = int(base["mask"].sum())
    print(f"## {cat} N={N} K={K} P={idx.shape[0]} res={ocfg['res']} inputs={mode}: grid {tuple(int(d) for d in base['dims'])}, "
          f"arg-max {base['argmax']}, peak {base['peak']:.3f}, top1-top2 margin {base['margin']:.4f}, survivors {n0}, "
          f"sphere arg-max {base['sphere_argmax']}", file=out)
    res_all = {}
    idx32 = idx.astype(np.int32)
    for name, v in (variants or O.VARIANTS).items():
        fl = O.vote_flips(ob["pc"], outputs, idx32, base["dims"], base["corner"], ocfg["res"], 72, True, v)
        p = O.pose_tail_variant(ob["pc"], ob["normals"], idx, outputs, heads, ocfg, sph, v)
        r = dict(fl)
        r.update(mode=mode, argmax_same=p["argmax"] == base["argmax"], max_grid_diff=float(np.abs(p["grid"] - base["grid"]).max()),
                 margin=base["margin"], n_surv0=n0, n_surv=int(p["mask"].sum()),
                 mask_flips=int((p["mask"] != base["mask"]).sum()), sphere_same=p["sphere_argmax"] == base["sphere_argmax"],
                 T_diff=float(np.abs(p["T"] - base["T"]).max()),
                 up_diff=float(np.abs(p["up"] - base["up"]).max()) if p["up"] is not None else 0.0,
                 # categories with an up/down symmetry (config/category/*.yaml: up_sym) fold theta_up (utils/dataset.py:50-51): the
                 # two poles of the axis collect the same votes and which of them wins the count is a coin toss in the reference
                 # too -- the axis is what is defined, compare modulo its sign
                 up_diff_mod_sym=float(min(np.abs(p["up"] - base["up"]).max(), np.abs(p["up"] + base["up"]).max()))
                 if (p["up"] is not None and ocfg["up_sym"]) else (float(np.abs(p["up"] - base["up"]).max()) if p["up"] is not None else 0.0),
                 scale_rel=float(np.abs(p["scale"] / base["scale"] - 1).max()) if n0 else 0.0)
        res_all[name] = r
        ig = max(fl["in_grid"], 1)
        print(f"  {name:22s} trip {fl['trip_count_flips']} degen {fl['degenerate_flips']} | in-grid flips {fl['in_grid_flips']} "
              f"({fl['in_grid_flips'] / ig:.1e}) floor-cell flips {fl['floor_cell_flips']} ({fl['floor_cell_flips'] / ig:.1e}) of "
              f"{fl['in_grid']} | max |dcoord| {fl['max_coord_diff_cells']:.1e} cells | arg-max same {r['argmax_same']} "
              f"(max |dgrid| {r['max_grid_diff']:.4f}) | survivors {r['n_surv']} ({r['mask_flips']} flips) | sphere same "
              f"{r['sphere_same']} | dT {r['T_diff']:.1e} dup {r['up_diff']:.1e} (mod up_sym {r['up_diff_mod_sym']:.1e}) dscale {r['scale_rel']:.1e}", file=out)
    print(f"  ({time.time() - t0:.1f} s)", file=out)
    return res_all


CASES_SMALL = [("bottle", 1024, 64, 0, "ka", None), ("bottle", 1024, 64, 0, "uniform", None), ("camera", 1024, 64, 1, "ka", None),
               ("laptop", 1024, 64, 2, "ka", None), ("mug", 1024, 64, 3, "ka", None), ("bowl", 1024, 64, 4, "uniform", None)]
CASES_FULL = [("bottle", 4096, 128, 0, "ka", None), ("bottle", 4096, 128, 0, "uniform", None), ("camera", 4096, 128, 7, "ka", None),
              ("bottle", 8192, 256, 3, "ka", 2e-3), ("bottle", 8192, 256, 3, "uniform", 2e-3)]

if __name__ == "__main__":
    print(__doc__)
    print(f"variants: {O.VARIANTS}\n")
    for c in CASES_SMALL + (CASES_FULL if "--full" in sys.argv else []):
        study(*c)
