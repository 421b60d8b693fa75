#!/usr/bin/env python3
"""Generate tests/golden/eval_map.npz: outputs of the reference's own NOCS evaluation
on seeded synthetic results (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eval.py /root/reference

`utils/util.py` cannot be imported here (it pulls open3d and cv2 at module level for unrelated helpers), so the
metric functions are taken from it one by one: the file is parsed, the function definitions this row needs are
compiled AS THEY ARE into a namespace that holds what they use (numpy, math, os, pickle, tqdm, matplotlib with the Agg
backend, and the reference's own utils.box / utils.iou, which do import).  Nothing of the reference is written to the
repository: the stored DATA are the synthetic inputs and the values those functions return --
  * 3D IoU of random oriented box pairs (compute_3d_iou, utils/util.py:181-216), with and without the up-axis symmetry;
  * rotation / translation errors (compute_RT_degree_cm_symmetry, :219-255);
  * iou_3d_aps, pose_aps, pose_pred_matches, pose_gt_matches of compute_degree_cm_mAP (:709-1008) called exactly like
    nocs/eval.py:44-49 on 24 synthetic images.
"""
import ast
import os
import sys
import tempfile

import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out_dir = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, ref)

import math  # noqa: E402
import pickle  # noqa: E402

import matplotlib  # noqa: E402
matplotlib.use("Agg")
import matplotlib.pyplot as plt  # noqa: E402
from tqdm import tqdm  # noqa: E402
from utils.box import Box  # noqa: E402
from utils.iou import IoU  # noqa: E402

WANT = {"compute_3d_iou", "compute_RT_degree_cm_symmetry", "trim_zeros", "compute_3d_matches",
        "compute_ap_from_matches_scores", "compute_RT_overlaps", "compute_match_from_degree_cm", "compute_degree_cm_mAP"}
src = open(os.path.join(ref, "utils", "util.py")).read()
tree = ast.parse(src)
ns = {"np": np, "math": math, "os": os, "pickle": pickle, "tqdm": tqdm, "plt": plt, "Box": Box, "IoU": IoU}
mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANT], type_ignores=[])
assert {n.name for n in mod.body} == WANT
exec(compile(mod, os.path.join(ref, "utils", "util.py"), "exec"), ns)

SYNSET = ["BG", "bottle", "bowl", "camera", "can", "laptop", "mug"]   # nocs/inference.py:19-27
UP_SYM = {"bottle", "bowl", "can"}                                    # nocs/eval.py:29-33


def rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def small_rot(rng, deg):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    a = np.deg2rad(deg)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def rt(R, t, s=1.0):
    M = np.eye(4)
    M[:3, :3] = R * s
    M[:3, 3] = t
    return M


rng = np.random.default_rng(20240)

# ---- box pairs: generic, near-identical, touching, disjoint, and exactly identical
pairs = []
for k in range(60):
    R1, t1, s1 = rot(rng), rng.normal(0, 0.05, 3), rng.uniform(0.05, 0.3, 3)
    kind = k % 6
    if kind == 0:
        R2, t2, s2 = rot(rng), t1 + rng.normal(0, 0.05, 3), rng.uniform(0.05, 0.3, 3)
    elif kind == 1:
        R2, t2, s2 = small_rot(rng, rng.uniform(0, 20)) @ R1, t1 + rng.normal(0, 0.01, 3), s1 * rng.uniform(0.8, 1.2, 3)
    elif kind == 2:
        R2, t2, s2 = R1.copy(), t1 + R1 @ (np.array([1.0, 0, 0]) * s1[0] * rng.uniform(0.2, 0.9)), s1.copy()
    elif kind == 3:
        R2, t2, s2 = rot(rng), t1 + np.array([2.0, 0, 0]), rng.uniform(0.05, 0.3, 3)          # disjoint
    elif kind == 4:
        R2, t2, s2 = R1.copy(), t1.copy(), s1.copy()                                          # identical
    else:
        R2, t2, s2 = small_rot(rng, 90) @ R1, t1.copy(), s1[[2, 1, 0]] * rng.uniform(0.5, 1.5)
    pairs.append((rt(R1, t1, rng.uniform(0.5, 2.0)), rt(R2, t2, rng.uniform(0.5, 2.0)), s1, s2))
iou_plain = np.array([ns["compute_3d_iou"](a.copy(), b.copy(), s1, s2, False, "mug", "mug") for a, b, s1, s2 in pairs])
iou_sym = np.array([ns["compute_3d_iou"](a.copy(), b.copy(), s1, s2, True, "can", "can") for a, b, s1, s2 in pairs])
err_plain = np.array([ns["compute_RT_degree_cm_symmetry"](a, b, False) for a, b, _, _ in pairs])
err_sym = np.array([ns["compute_RT_degree_cm_symmetry"](a, b, True) for a, b, _, _ in pairs])

# ---- synthetic images: ground truth instances + predictions (perturbed copies, misses, false positives, wrong classes)
results = []
for img in range(24):
    n_gt = int(rng.integers(1, 6))                      # (the reference cannot take an image without ground truth: np.stack([]))
    gt_cls = rng.integers(1, 7, n_gt).astype(np.int32)
    gt_RTs = np.array([rt(rot(rng), rng.normal(0, 0.3, 3) + [0, 0, 1.0], rng.uniform(0.1, 0.4)) for _ in range(n_gt)]).reshape(-1, 4, 4)
    gt_scales = rng.uniform(0.3, 1.0, (n_gt, 3))
    gt_vis = np.ones(n_gt, dtype=np.int32)
    for i, c in enumerate(gt_cls):
        if SYNSET[c] == "mug" and rng.random() < 0.4:
            gt_vis[i] = 0
    up = np.array([(v == 0) or (SYNSET[c] in UP_SYM) for c, v in zip(gt_cls, gt_vis)], dtype=bool)
    pr_cls, pr_RTs, pr_scales, pr_scores = [], [], [], []
    for i in range(n_gt):
        r = rng.random()
        if r < 0.12:
            continue                                                             # missed
        s0 = np.cbrt(np.linalg.det(gt_RTs[i][:3, :3]))
        R0 = gt_RTs[i][:3, :3] / s0
        quality = rng.choice([1.0, 4.0, 12.0, 40.0])
        Rp = small_rot(rng, rng.uniform(0, quality)) @ R0
        tp = gt_RTs[i][:3, 3] + rng.normal(0, 0.004 * quality, 3)
        sp = s0 * rng.uniform(0.9, 1.1)
        pr_RTs.append(rt(Rp, tp, sp))
        pr_scales.append(gt_scales[i] * rng.uniform(0.85, 1.15, 3))
        pr_cls.append(int(gt_cls[i]) if r > 0.2 else int(rng.integers(1, 7)))   # sometimes the wrong class
        pr_scores.append(rng.uniform(0.3, 1.0))
    for _ in range(int(rng.integers(0, 3))):                                     # false positives
        pr_RTs.append(rt(rot(rng), rng.normal(0, 0.3, 3) + [0, 0, 1.0], rng.uniform(0.1, 0.4)))
        pr_scales.append(rng.uniform(0.3, 1.0, 3))
        pr_cls.append(int(rng.integers(1, 7)))
        pr_scores.append(rng.uniform(0.0, 0.6))
    n_pr = len(pr_cls)
    results.append({
        "gt_class_ids": gt_cls, "gt_RTs": gt_RTs, "gt_scales": gt_scales, "gt_handle_visibility": gt_vis, "gt_up_syms": up,
        "pred_class_ids": np.array(pr_cls, dtype=np.int32), "pred_RTs": np.array(pr_RTs).reshape(-1, 4, 4),
        "pred_scales": np.array(pr_scales).reshape(-1, 3), "pred_scores": np.array(pr_scores, dtype=np.float64),
        "pred_bboxes": np.zeros((n_pr, 4), dtype=np.int32)})

import copy  # noqa: E402
with tempfile.TemporaryDirectory() as tmp:
    iou_aps, pose_aps, pose_pred_matches, pose_gt_matches = ns["compute_degree_cm_mAP"](
        copy.deepcopy(results), SYNSET, tmp, degree_thresholds=[5, 10, 15], shift_thresholds=[5, 10, 15],
        iou_3d_thresholds=np.linspace(0, 1, 101), iou_pose_thres=0.1, use_matches_for_pose=True)
    iou_aps2, pose_aps2, _, _ = ns["compute_degree_cm_mAP"](
        copy.deepcopy(results), SYNSET, tmp, degree_thresholds=[5, 10], shift_thresholds=[2, 5],
        iou_3d_thresholds=[0.1, 0.25, 0.5], iou_pose_thres=0.1, use_matches_for_pose=False)

store = {"pair_a": np.array([p[0] for p in pairs]), "pair_b": np.array([p[1] for p in pairs]),
         "pair_s1": np.array([p[2] for p in pairs]), "pair_s2": np.array([p[3] for p in pairs]),
         "iou_plain": iou_plain, "iou_sym": iou_sym, "err_plain": err_plain, "err_sym": err_sym,
         "n_images": np.array(len(results)), "iou_aps": iou_aps, "pose_aps": pose_aps, "pose_pred_matches": pose_pred_matches,
         "pose_gt_matches": pose_gt_matches, "iou_aps_detection": iou_aps2, "pose_aps_detection": pose_aps2}
for i, r in enumerate(results):
    for k, v in r.items():
        store[f"img{i}::{k}"] = np.asarray(v)
np.savez_compressed(os.path.join(out_dir, "eval_map.npz"), **store)
print("eval_map.npz:", len(results), "images; mean 3D IoU AP@25/50 = %.3f / %.3f; 5deg5cm = %.3f, 10deg10cm = %.3f" % (
    iou_aps[-1, 25], iou_aps[-1, 50], pose_aps[-1, 0, 0], pose_aps[-1, 1, 1]))
