"""NOCS evaluation of pose predictions: 3D-IoU AP and degree / centimetre AP over the
result dictionaries `nocs/inference.py:338-345` pickles (see cppf_amd.inference.nocs_result for this package's side).

Host-side numpy, like the reference's (`utils/util.py:181-255,342-527,709-1008`, driven by `nocs/eval.py:16-49`): this is
offline metric code, there is nothing here for the GPU.  Same entry points, arguments, return values and matching rules;
the implementation is this repository's own:

* the intersection volume of two oriented boxes is computed with the divergence theorem over the clipped faces of
  both boxes (each face of one box clipped by the six half-spaces of the other), not by collecting intersection points
  and taking their convex hull (`utils/iou.py`); the two agree to rounding (tests/golden/eval_map.npz: <= 1e-6);
* no plots are drawn (the reference's figures are a side effect; the AP tables it pickles next to them are written when
  `log_dir` is given).

Parity is pinned on outputs of the reference's own functions for seeded synthetic results (tests/golden/
make_golden_eval.py -> eval_map.npz; tests/test_evaluation.py)."""
import math
import os
import pickle

import numpy as np

SYNSET_NAMES = ["BG", "bottle", "bowl", "camera", "can", "laptop", "mug"]      # nocs/inference.py:19-27
_UP_SYMMETRIC = ("bowl", "bottle", "can")                                       # nocs/eval.py:32

# ------------------------------------------------------------------------------------------------ oriented boxes
_SIGNS = np.array([[sx, sy, sz] for sx in (-1.0, 1.0) for sy in (-1.0, 1.0) for sz in (-1.0, 1.0)])
_EPS = 1e-9


def _frame(RT, scales):
    """(centre, unit axes as columns, half extents) of the box `Box.from_transformation(R / cbrt(det R), t, scales)`
    (utils/util.py:188-191); a residual scale in a column of R goes into that axis' extent."""
    R = np.asarray(RT, dtype=np.float64)[:3, :3]
    R = R / np.cbrt(np.linalg.det(R))
    norms = np.linalg.norm(R, axis=0)
    return np.asarray(RT, dtype=np.float64)[:3, 3].copy(), R / norms, 0.5 * np.asarray(scales, dtype=np.float64) * norms


def _faces(c, U, h):
    """the six faces of a box: (outward unit normal, plane offset n.x = d, 4 corners in cyclic order)"""
    out = []
    for k in range(3):
        a, b = (k + 1) % 3, (k + 2) % 3
        for s in (1.0, -1.0):
            ctr = c + s * h[k] * U[:, k]
            ea, eb = h[a] * U[:, a], h[b] * U[:, b]
            quad = np.array([ctr - ea - eb, ctr + ea - eb, ctr + ea + eb, ctr - ea + eb])
            n = s * U[:, k]
            out.append((n, float(n @ ctr), quad))
    return out


def _clip(poly, n, d, tol):
    """part of the convex polygon `poly` ([m,3]) inside the half-space n.x <= d; `tol` only decides on which side a
    vertex that lies on the plane (to rounding) falls -- crossings are computed on the plane itself"""
    if len(poly) == 0:
        return poly
    dist = poly @ n - d
    inside = dist <= tol
    if inside.all():
        return poly
    if not inside.any():
        return poly[:0]
    out = []
    m = len(poly)
    for i in range(m):
        j = (i + 1) % m
        if inside[i]:
            out.append(poly[i])
        if inside[i] != inside[j]:
            t = min(1.0, max(0.0, dist[i] / (dist[i] - dist[j])))
            out.append(poly[i] + t * (poly[j] - poly[i]))
    return np.array(out)


def _flux(poly, n, d):
    """(n . x) * area of a planar convex polygon lying in the plane n.x = d"""
    if len(poly) < 3:
        return 0.0
    v = poly[1:] - poly[0]
    area = 0.5 * np.linalg.norm(np.cross(v[:-1], v[1:]).sum(0))
    return d * area


def box_intersection_volume(c1, U1, h1, c2, U2, h2):
    """volume of the intersection of two oriented boxes: V = 1/3 * sum over the faces of the intersection polytope of
    (n . x) * area.  Those faces are the faces of box 1 clipped by box 2 and vice versa; a face the two boxes share
    (coincident planes) is counted once: box 1's faces are clipped by closed half-spaces, box 2's by open ones."""
    f1, f2 = _faces(c1, U1, h1), _faces(c2, U2, h2)
    scale = float(max(h1.max(), h2.max()))
    tol = 1e-9 * scale
    total = 0.0
    for mine, other, t in ((f1, f2, tol), (f2, f1, -tol)):
        for n, d, quad in mine:
            poly = quad
            for n2, d2, _ in other:
                poly = _clip(poly, n2, d2, t)
                if len(poly) < 3:
                    break
            total += _flux(poly, n, d)
    return max(total / 3.0, 0.0)


def _iou_frames(a, b):
    v = box_intersection_volume(*a, *b)
    va, vb = 8.0 * np.prod(a[2]), 8.0 * np.prod(b[2])
    if v <= _EPS * min(va, vb):
        return 0.0
    return float(v / (va + vb - v))


def _rot_y(theta):
    c, s = math.cos(theta), math.sin(theta)
    return np.array([[c, 0.0, s, 0.0], [0.0, 1.0, 0.0, 0.0], [-s, 0.0, c, 0.0], [0.0, 0.0, 0.0, 1.0]])


def compute_3d_iou(RT_1, RT_2, scales_1, scales_2, up_sym, class_name_1, class_name_2):
    """utils/util.py:181-216: IoU of the boxes (RT_k normalised to a rotation, extents scales_k); for an up-symmetric
    ground truth of the same class the best of 20 rotations of box 1 about its own y axis.  -1 when a pose is missing."""
    if RT_1 is None or RT_2 is None:
        return -1
    try:
        fb = _frame(RT_2, scales_2)
        if class_name_1 == class_name_2 and up_sym:
            return max(_iou_frames(_frame(np.asarray(RT_1, dtype=np.float64) @ _rot_y(2.0 * math.pi * i / 20.0), scales_1), fb)
                       for i in range(20))
        return _iou_frames(_frame(RT_1, scales_1), fb)
    except (np.linalg.LinAlgError, FloatingPointError, ValueError):
        return 0


def compute_RT_degree_cm_symmetry(RT_1, RT_2, up_sym):
    """utils/util.py:219-255: [rotation error in degrees, translation error in centimetres]; with `up_sym` only the
    angle between the two y axes counts."""
    if RT_1 is None or RT_2 is None:
        return -1
    RT_1, RT_2 = np.asarray(RT_1, dtype=np.float64), np.asarray(RT_2, dtype=np.float64)
    if not (np.array_equal(RT_1[3], RT_2[3]) and np.array_equal(RT_1[3], [0, 0, 0, 1])):
        raise ValueError(f"homogeneous rows differ from [0 0 0 1]: {RT_1[3]} {RT_2[3]}")
    R1 = RT_1[:3, :3] / np.cbrt(np.linalg.det(RT_1[:3, :3]))
    R2 = RT_2[:3, :3] / np.cbrt(np.linalg.det(RT_2[:3, :3]))
    if up_sym:
        y1, y2 = R1[:, 1], R2[:, 1]
        cosv = y1.dot(y2) / (np.linalg.norm(y1) * np.linalg.norm(y2))
    else:
        cosv = (np.trace(R1 @ R2.T) - 1.0) / 2.0
    theta = math.degrees(math.acos(min(1.0, max(-1.0, float(cosv)))))     # (the reference lets rounding above 1 become NaN)
    return np.array([theta, np.linalg.norm(RT_1[:3, 3] - RT_2[:3, 3]) * 100.0])


# ------------------------------------------------------------------------------------------------ matching
def compute_3d_matches(gt_class_ids, gt_RTs, gt_scales, gt_up_syms, synset_names, pred_boxes, pred_class_ids, pred_scores,
                       pred_RTs, pred_scales, iou_3d_thresholds, score_threshold=0):
    """utils/util.py:342-416.  Predictions in descending score order claim, per IoU threshold, the unclaimed ground truth
    of their class with the highest IoU above the threshold (a candidate below it ends the search).  Returns
    (gt_matches [T, n_gt], pred_matches [T, n_pred] in sorted order, overlaps [n_pred, n_gt] float32, the sort order)."""
    n_pred, n_gt = len(pred_class_ids), len(gt_class_ids)
    order = np.zeros(0)
    if n_pred:
        order = np.argsort(pred_scores)[::-1]
        pred_class_ids, pred_scales, pred_RTs = pred_class_ids[order], pred_scales[order], pred_RTs[order]
    overlaps = np.zeros((n_pred, n_gt), dtype=np.float32)
    for i in range(n_pred):
        for j in range(n_gt):
            overlaps[i, j] = compute_3d_iou(pred_RTs[i], gt_RTs[j], pred_scales[i], gt_scales[j], gt_up_syms[j],
                                            synset_names[pred_class_ids[i]], synset_names[gt_class_ids[j]])
    n_thr = len(iou_3d_thresholds)
    pred_matches = -np.ones((n_thr, n_pred))
    gt_matches = -np.ones((n_thr, n_gt))
    ranked = [np.argsort(overlaps[i])[::-1] for i in range(n_pred)]
    for i in range(n_pred):
        keep = overlaps[i, ranked[i]] >= score_threshold
        if not keep.all():
            ranked[i] = ranked[i][:int(np.argmin(keep))]
    for s, thr in enumerate(iou_3d_thresholds):
        for i in range(n_pred):
            for j in ranked[i]:
                if gt_matches[s, j] > -1:
                    continue
                iou = overlaps[i, j]
                if iou < thr:
                    break
                if pred_class_ids[i] != gt_class_ids[j]:
                    continue
                if iou > thr:
                    gt_matches[s, j] = i
                    pred_matches[s, i] = j
                    break
    return gt_matches, pred_matches, overlaps, order


def compute_RT_overlaps(gt_class_ids, gt_RTs, gt_up_syms, pred_class_ids, pred_RTs):
    """utils/util.py:447-467: [n_pred, n_gt, 2] (degrees, centimetres)"""
    out = np.zeros((len(pred_class_ids), len(gt_class_ids), 2))
    for i in range(out.shape[0]):
        for j in range(out.shape[1]):
            out[i, j] = compute_RT_degree_cm_symmetry(pred_RTs[i], gt_RTs[j], gt_up_syms[j])
    return out


def compute_match_from_degree_cm(overlaps, pred_class_ids, gt_class_ids, degree_thres_list, shift_thres_list):
    """utils/util.py:470-517: per (degree, shift) threshold pair every prediction takes the unclaimed ground truth of
    its class with the smallest degree + centimetre sum that is inside both thresholds."""
    n_pred, n_gt = len(pred_class_ids), len(gt_class_ids)
    pred_matches = -np.ones((len(degree_thres_list), len(shift_thres_list), n_pred))
    gt_matches = -np.ones((len(degree_thres_list), len(shift_thres_list), n_gt))
    if n_pred == 0 or n_gt == 0:
        return gt_matches, pred_matches
    if overlaps.shape != (n_pred, n_gt, 2):
        raise ValueError(f"overlaps {overlaps.shape} for {n_pred} predictions and {n_gt} ground truths")
    ranked = [np.argsort(overlaps[i].sum(-1)) for i in range(n_pred)]
    same = np.asarray(pred_class_ids)[:, None] == np.asarray(gt_class_ids)[None, :]
    for d, deg in enumerate(degree_thres_list):
        for s, sh in enumerate(shift_thres_list):
            for i in range(n_pred):
                for j in ranked[i]:
                    if gt_matches[d, s, j] > -1 or not same[i, j]:
                        continue
                    if overlaps[i, j, 0] > deg or overlaps[i, j, 1] > sh:
                        continue
                    gt_matches[d, s, j] = i
                    pred_matches[d, s, i] = j
                    break
    return gt_matches, pred_matches


def compute_ap_from_matches_scores(pred_match, pred_scores, gt_match):
    """utils/util.py:419-444: VOC-style AP (precision made monotone from the right, summed over recall steps)"""
    if pred_match.shape[0] != pred_scores.shape[0]:
        raise ValueError("one score per prediction")
    hit = (pred_match[np.argsort(pred_scores)[::-1]] > -1)
    tp = np.cumsum(hit)
    precisions = np.concatenate([[0.0], tp / (np.arange(len(hit)) + 1), [0.0]])
    recalls = np.concatenate([[0.0], tp.astype(np.float32) / len(gt_match), [1.0]])
    precisions = np.maximum.accumulate(precisions[::-1])[::-1]
    steps = np.where(recalls[:-1] != recalls[1:])[0] + 1
    return np.sum((recalls[steps] - recalls[steps - 1]) * precisions[steps])


# ------------------------------------------------------------------------------------------------ the metric
def _unit_scale(RTs, scales, eps):
    """RTs with the scale taken out of the rotation block, and the extents multiplied by it (utils/util.py:754-768)"""
    RTs = np.array(RTs, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)
    if len(RTs) == 0:
        return RTs.reshape(0, 4, 4), scales.reshape(0, 3)
    s = np.cbrt(np.linalg.det(RTs[:, :3, :3]))
    RTs[:, :3, :3] = RTs[:, :3, :3] / (s[:, None, None] + eps)
    return RTs, scales * s[:, None]


def compute_degree_cm_mAP(final_results, synset_names, log_dir, degree_thresholds=[360], shift_thresholds=[100],
                          iou_3d_thresholds=[0.1], iou_pose_thres=0.1, use_matches_for_pose=False):
    """utils/util.py:709-1008 (called by nocs/eval.py:44-49).  final_results: one dict per image with gt_class_ids,
    gt_RTs, gt_scales, gt_up_syms, pred_class_ids, pred_RTs, pred_scales, pred_scores (pred_bboxes is ignored, as there).
    Returns (iou_3d_aps [C+1, T], pose_aps [C+1, D+1, S+1], pose_pred_matches, pose_gt_matches [D+1, S+1, images, 20]);
    row C of the AP tables is the mean over the classes, the extra threshold is 360 degrees / 100 cm."""
    n_cls = len(synset_names)
    deg_list = list(degree_thresholds) + [360]
    sh_list = list(shift_thresholds) + [100]
    iou_list = list(iou_3d_thresholds)
    nD, nS, nT = len(deg_list), len(sh_list), len(iou_list)
    if use_matches_for_pose and iou_pose_thres not in iou_list:
        raise ValueError("iou_pose_thres must be one of iou_3d_thresholds")
    iou_pm = [np.zeros((nT, 0)) for _ in range(n_cls)]
    iou_ps = [np.zeros((nT, 0)) for _ in range(n_cls)]
    iou_gm = [np.zeros((nT, 0)) for _ in range(n_cls)]
    pose_pm = [np.zeros((nD, nS, 0)) for _ in range(n_cls)]
    pose_ps = [np.zeros((nD, nS, 0)) for _ in range(n_cls)]
    pose_gm = [np.zeros((nD, nS, 0)) for _ in range(n_cls)]
    pose_gt_matches = np.full((nD, nS, len(final_results), 20), -1, dtype=int)
    pose_pred_matches = np.full((nD, nS, len(final_results), 20), -1, dtype=int)

    for img, res in enumerate(final_results):
        gt_cls = np.asarray(res["gt_class_ids"]).astype(np.int32)
        gt_RTs, gt_scales = _unit_scale(res["gt_RTs"], res["gt_scales"], 0.0)
        gt_sym = np.asarray(res["gt_up_syms"])
        pr_cls = np.asarray(res["pred_class_ids"])
        pr_scores = np.asarray(res["pred_scores"])
        pr_RTs, pr_scales = _unit_scale(res["pred_RTs"], res["pred_scales"], 1e-9)
        if len(gt_cls) == 0 and len(pr_cls) == 0:
            continue
        for c in range(1, n_cls):
            g_sel = np.where(gt_cls == c)[0] if len(gt_cls) else np.zeros(0, dtype=int)
            p_sel = np.where(pr_cls == c)[0] if len(pr_cls) else np.zeros(0, dtype=int)
            c_gt_cls, c_gt_RTs, c_gt_scales, c_gt_sym = gt_cls[g_sel], gt_RTs[g_sel], gt_scales[g_sel], gt_sym[g_sel]
            c_pr_cls, c_pr_RTs, c_pr_scales, c_pr_scores = pr_cls[p_sel], pr_RTs[p_sel], pr_scales[p_sel], pr_scores[p_sel]
            gm, pm, _, order = compute_3d_matches(c_gt_cls, c_gt_RTs, c_gt_scales, c_gt_sym, synset_names, None, c_pr_cls,
                                                  c_pr_scores, c_pr_RTs, c_pr_scales, iou_list)
            if len(order):
                p_sel, c_pr_cls, c_pr_RTs, c_pr_scores = p_sel[order], c_pr_cls[order], c_pr_RTs[order], c_pr_scores[order]
            iou_pm[c] = np.concatenate((iou_pm[c], pm), -1)
            iou_ps[c] = np.concatenate((iou_ps[c], np.tile(c_pr_scores, (nT, 1))), -1)
            iou_gm[c] = np.concatenate((iou_gm[c], gm), -1)
            if use_matches_for_pose:                       # only the instances matched at iou_pose_thres are scored for pose
                k = iou_list.index(iou_pose_thres)
                keep_p, keep_g = pm[k] > -1, gm[k] > -1
                p_sel, c_pr_cls, c_pr_RTs, c_pr_scores = p_sel[keep_p], c_pr_cls[keep_p], c_pr_RTs[keep_p], c_pr_scores[keep_p]
                g_sel, c_gt_cls, c_gt_RTs, c_gt_sym = g_sel[keep_g], c_gt_cls[keep_g], c_gt_RTs[keep_g], c_gt_sym[keep_g]
            errs = compute_RT_overlaps(c_gt_cls, c_gt_RTs, c_gt_sym, c_pr_cls, c_pr_RTs)
            p_gm, p_pm = compute_match_from_degree_cm(errs, c_pr_cls, c_gt_cls, deg_list, sh_list)
            # per-image tables in the image's own instance numbering
            for i in range(p_pm.shape[2]):
                m = p_pm[:, :, i].astype(int)
                pose_pred_matches[:, :, img, p_sel[i]] = np.where(m >= 0, g_sel[np.maximum(m, 0)] if len(g_sel) else -1, -1)
            for i in range(p_gm.shape[2]):
                m = p_gm[:, :, i].astype(int)
                pose_gt_matches[:, :, img, g_sel[i]] = np.where(m >= 0, p_sel[np.maximum(m, 0)] if len(p_sel) else -1, -1)
            pose_pm[c] = np.concatenate((pose_pm[c], p_pm), -1)
            pose_ps[c] = np.concatenate((pose_ps[c], np.tile(c_pr_scores, (nD, nS, 1))), -1)
            pose_gm[c] = np.concatenate((pose_gm[c], p_gm), -1)

    iou_3d_aps = np.zeros((n_cls + 1, nT))
    pose_aps = np.zeros((n_cls + 1, nD, nS))
    for c in range(1, n_cls):
        for s in range(nT):
            iou_3d_aps[c, s] = compute_ap_from_matches_scores(iou_pm[c][s], iou_ps[c][s], iou_gm[c][s])
        for d in range(nD):
            for s in range(nS):
                pose_aps[c, d, s] = compute_ap_from_matches_scores(pose_pm[c][d, s], pose_ps[c][d, s], pose_gm[c][d, s])
    iou_3d_aps[-1] = iou_3d_aps[1:-1].mean(0)
    pose_aps[-1] = pose_aps[1:-1].mean(0)
    if log_dir:                                             # the two tables the reference pickles next to its figures
        os.makedirs(log_dir, exist_ok=True)
        with open(os.path.join(log_dir, "IoU_3D_AP_{}-{}.pkl".format(iou_list[0], iou_list[-1])), "wb") as f:
            pickle.dump({"thres_list": iou_list, "aps": iou_3d_aps}, f)
        prefix = "Pose_Only_" if use_matches_for_pose else "Pose_Detection_"
        name = prefix + "AP_{}-{}degree_{}-{}cm.pkl".format(deg_list[0], deg_list[-2], sh_list[0], sh_list[-2])
        with open(os.path.join(log_dir, name), "wb") as f:
            pickle.dump({"degree_thres": deg_list, "shift_thres_list": sh_list, "aps": pose_aps}, f)
    return iou_3d_aps, pose_aps, pose_pred_matches, pose_gt_matches


# ------------------------------------------------------------------------------------------------ nocs/eval.py
def mark_up_symmetry(result, synset_names=SYNSET_NAMES):
    """nocs/eval.py:25-33: gt_up_syms from the class and the handle visibility (a mug whose handle is hidden counts as
    symmetric about its up axis)"""
    vis, cls = np.asarray(result["gt_handle_visibility"]), np.asarray(result["gt_class_ids"])
    if len(vis) != len(cls):
        raise ValueError(f"{len(vis)} handle flags for {len(cls)} instances")
    sym = np.zeros(len(cls), dtype=bool)
    for i, (c, v) in enumerate(zip(cls, vis)):
        name = synset_names[c]
        if v == 0:
            if name != "mug":
                raise ValueError(f"hidden handle on a {name}")
            sym[i] = True
        elif name in _UP_SYMMETRIC:
            sym[i] = True
    result["gt_up_syms"] = sym
    return result


def evaluate_prediction_dir(pred_dir, stride=10, synset_names=SYNSET_NAMES):
    """nocs/eval.py:16-49: every `stride`-th results_*.pkl of a prediction directory -> the four arrays of
    compute_degree_cm_mAP with the thresholds the paper reports (5/10/15 degrees, 5/10/15 cm, IoU 0..1 in steps of 0.01)."""
    import glob
    files = sorted(glob.glob(os.path.join(pred_dir, "results_*.pkl")))[::stride]
    if not files:
        raise FileNotFoundError(f"no results_*.pkl under {pred_dir}")
    results = []
    for path in files:
        with open(path, "rb") as f:
            r = pickle.load(f)
        for one in (r if isinstance(r, list) else [r]):
            results.append(mark_up_symmetry(one, synset_names))
    return compute_degree_cm_mAP(results, synset_names, pred_dir + "_map", degree_thresholds=[5, 10, 15],
                                 shift_thresholds=[5, 10, 15], iou_3d_thresholds=np.linspace(0, 1, 101), iou_pose_thres=0.1,
                                 use_matches_for_pose=True)
