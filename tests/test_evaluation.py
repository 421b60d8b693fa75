"""NOCS evaluation against outputs of the reference's own functions on seeded synthetic
results (tests/golden/make_golden_eval.py -> eval_map.npz).  Host-side numpy, no GPU involved (the reference's is too)."""
import copy
import pickle

import numpy as np
import pytest

from cppf_amd import evaluation as E


@pytest.fixture(scope="module")
def z(golden):
    return golden("eval_map.npz")


def _results(z):
    out = []
    for i in range(int(z["n_images"])):
        out.append({k.split("::", 1)[1]: z[k] for k in z.files if k.startswith(f"img{i}::")})
    return out


def test_box_iou_matches_the_reference_hull_based_iou(z):
    """generic, nearly identical, sliding, disjoint, identical and axis-swapped box pairs; RTs carry a scale that has to
    be taken out of the rotation first (utils/util.py:188-189)"""
    for sym, key in ((False, "iou_plain"), (True, "iou_sym")):
        got = np.array([E.compute_3d_iou(a, b, s1, s2, sym, "can", "can")
                        for a, b, s1, s2 in zip(z["pair_a"], z["pair_b"], z["pair_s1"], z["pair_s2"])])
        np.testing.assert_allclose(got, z[key], atol=1e-6, rtol=0)
    ident = np.arange(60) % 6 == 4
    assert np.allclose(z["iou_plain"][ident], 1.0) and np.all(z["iou_plain"][np.arange(60) % 6 == 3] == 0)
    # different classes: the up symmetry is not applied (utils/util.py:200)
    a, b, s1, s2 = z["pair_a"][1], z["pair_b"][1], z["pair_s1"][1], z["pair_s2"][1]
    assert E.compute_3d_iou(a, b, s1, s2, True, "can", "mug") == pytest.approx(z["iou_plain"][1], abs=1e-6)
    assert E.compute_3d_iou(None, b, s1, s2, False, "can", "can") == -1


def test_box_volume_identities():
    """closed forms: nested boxes, a half overlap along one axis, a 45 degree turn of a square prism about its axis"""
    rng = np.random.default_rng(1)
    I = np.eye(4)
    assert E.compute_3d_iou(I, I, [1, 1, 1], [0.5, 0.5, 0.5], False, "a", "a") == pytest.approx(0.125, abs=1e-12)
    T = np.eye(4); T[0, 3] = 0.5
    assert E.compute_3d_iou(I, T, [1, 2, 3], [1, 2, 3], False, "a", "a") == pytest.approx(1 / 3, abs=1e-12)
    c = np.sqrt(0.5)
    Ry = np.array([[c, 0, c, 0], [0, 1, 0, 0], [-c, 0, c, 0], [0, 0, 0, 1]])
    inter = 2 * (np.sqrt(2) - 1)                       # regular octagon inscribed in the unit square
    assert E.compute_3d_iou(I, Ry, [1, 1, 1], [1, 1, 1], False, "a", "a") == pytest.approx(inter / (2 - inter), abs=1e-12)
    # up symmetry: box 1 is turned in steps of 18 degrees; the closest it gets to 45 is 36 / 54, i.e. 9 degrees apart, and two
    # unit squares turned by phi about a common centre share 2 / (1 + sin phi + cos phi)
    a9 = 2 / (1 + np.sin(np.deg2rad(9)) + np.cos(np.deg2rad(9)))
    assert E.compute_3d_iou(I, Ry, [1, 1, 1], [1, 1, 1], True, "a", "a") == pytest.approx(a9 / (2 - a9), abs=1e-12)
    # invariance under a common rigid motion and a scale inside the RTs
    for _ in range(5):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        w, x, y, zq = q
        R = np.array([[1 - 2 * (y * y + zq * zq), 2 * (x * y - zq * w), 2 * (x * zq + y * w)],
                      [2 * (x * y + zq * w), 1 - 2 * (x * x + zq * zq), 2 * (y * zq - x * w)],
                      [2 * (x * zq - y * w), 2 * (y * zq + x * w), 1 - 2 * (x * x + y * y)]])
        M = np.eye(4); M[:3, :3] = R; M[:3, 3] = rng.normal(size=3)
        S = np.diag([3.0, 3.0, 3.0, 1.0])
        assert E.compute_3d_iou(M @ I @ S, M @ T, [1, 2, 3], [1, 2, 3], False, "a", "a") == pytest.approx(1 / 3, abs=1e-9)


def test_pose_errors(z):
    for sym, key in ((False, "err_plain"), (True, "err_sym")):
        got = np.array([E.compute_RT_degree_cm_symmetry(a, b, sym) for a, b in zip(z["pair_a"], z["pair_b"])])
        ref = z[key]
        nan = np.isnan(ref[:, 0])                      # arccos of 1 + rounding in the reference: identical rotations
        np.testing.assert_allclose(got[~nan], ref[~nan], atol=1e-6, rtol=1e-9)
        # near 0 degrees arccos amplifies rounding: 1e-8 in the cosine is 8e-3 degrees
        assert np.all(got[nan, 0] < 1e-2)
        np.testing.assert_allclose(got[nan, 1], ref[nan, 1], atol=1e-9)
    with pytest.raises(ValueError):
        bad = np.eye(4); bad[3, 0] = 1
        E.compute_RT_degree_cm_symmetry(bad, np.eye(4), False)


def test_map_tables_equal_the_reference(z, tmp_path):
    """the call of nocs/eval.py:44-49 (pose AP on the instances matched at IoU 0.1) and the detection variant"""
    res = _results(z)
    iou_aps, pose_aps, ppm, pgm = E.compute_degree_cm_mAP(
        copy.deepcopy(res), E.SYNSET_NAMES, str(tmp_path / "log"), degree_thresholds=[5, 10, 15], shift_thresholds=[5, 10, 15],
        iou_3d_thresholds=np.linspace(0, 1, 101), iou_pose_thres=0.1, use_matches_for_pose=True)
    np.testing.assert_allclose(iou_aps, z["iou_aps"], atol=1e-12)
    np.testing.assert_allclose(pose_aps, z["pose_aps"], atol=1e-12)
    assert np.array_equal(ppm, z["pose_pred_matches"]) and np.array_equal(pgm, z["pose_gt_matches"])
    assert iou_aps.shape == (8, 101) and pose_aps.shape == (8, 4, 4) and ppm.shape == (4, 4, len(res), 20)
    with open(tmp_path / "log" / "Pose_Only_AP_5-15degree_5-15cm.pkl", "rb") as f:
        assert np.array_equal(pickle.load(f)["aps"], pose_aps)
    iou2, pose2, _, _ = E.compute_degree_cm_mAP(copy.deepcopy(res), E.SYNSET_NAMES, None, degree_thresholds=[5, 10],
                                                shift_thresholds=[2, 5], iou_3d_thresholds=[0.1, 0.25, 0.5], iou_pose_thres=0.1,
                                                use_matches_for_pose=False)
    np.testing.assert_allclose(iou2, z["iou_aps_detection"], atol=1e-12)
    np.testing.assert_allclose(pose2, z["pose_aps_detection"], atol=1e-12)


def test_ap_and_matching_rules():
    # AP: three predictions (hit, miss, hit) against two ground truths -> recall steps at 0.5 and 1.0
    ap = E.compute_ap_from_matches_scores(np.array([0, -1, 1]), np.array([0.9, 0.8, 0.7]), np.array([0, 2]))
    assert ap == pytest.approx(0.5 * 1.0 + 0.5 * (2 / 3))
    # degree / cm matching: the closer ground truth wins, classes must agree, a claimed ground truth is skipped
    errs = np.array([[[3.0, 1.0], [1.0, 1.0]], [[2.0, 2.0], [50.0, 1.0]]])
    gm, pm = E.compute_match_from_degree_cm(errs, np.array([1, 1]), np.array([1, 1]), [5, 360], [5, 100])
    assert pm[0, 0].tolist() == [1, 0] and gm[0, 0].tolist() == [1, 0]
    gm, pm = E.compute_match_from_degree_cm(errs, np.array([1, 2]), np.array([1, 1]), [5], [5])
    assert pm[0, 0].tolist() == [1, -1]
    gm, pm = E.compute_match_from_degree_cm(np.zeros((0, 2, 2)), np.zeros(0), np.array([1, 1]), [5], [5])
    assert pm.shape == (1, 1, 0) and np.all(gm == -1)


def test_prediction_dir_round_trip(z, tmp_path):
    """results_*.pkl files in the layout nocs/inference.py:338-345 writes -> evaluate_prediction_dir (nocs/eval.py)"""
    res = _results(z)
    d = tmp_path / "pred"
    d.mkdir()
    for i, r in enumerate(res):
        r = {k: v for k, v in r.items() if k != "gt_up_syms"}
        with open(d / f"results_{i:04d}.pkl", "wb") as f:
            pickle.dump(r, f)
    iou_aps, pose_aps, _, _ = E.evaluate_prediction_dir(str(d), stride=1)
    np.testing.assert_allclose(iou_aps, z["iou_aps"], atol=1e-12)
    np.testing.assert_allclose(pose_aps, z["pose_aps"], atol=1e-12)
    with pytest.raises(FileNotFoundError):
        E.evaluate_prediction_dir(str(tmp_path / "none"))


def test_pipeline_records_feed_the_metric():
    """pose dicts in the shape estimate_pose / PosePipeline return -> nocs_result (nocs/inference.py:338-342 layout) ->
    the metric: predictions equal to the ground truth score AP 1 at every threshold, a 20 degree / 8 cm miss does not"""
    from cppf_amd.inference import nocs_result
    rng = np.random.default_rng(3)
    poses, gts = [], []
    for k in range(4):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        w, x, y, zq = q
        R = np.array([[1 - 2 * (y * y + zq * zq), 2 * (x * y - zq * w), 2 * (x * zq + y * w)],
                      [2 * (x * y + zq * w), 1 - 2 * (x * x + zq * zq), 2 * (y * zq - x * w)],
                      [2 * (x * zq - y * w), 2 * (y * zq + x * w), 1 - 2 * (x * x + y * y)]])
        scale = rng.uniform(0.1, 0.3, 3)
        poses.append({"T": rng.normal(0, 0.2, 3), "R": R, "scale": scale, "scale_norm": float(np.linalg.norm(scale))})
    cls = np.array([1, 3, 5, 6], dtype=np.int32)
    res = nocs_result(poses, {"pred_class_ids": cls, "pred_scores": np.array([0.9, 0.8, 0.7, 0.6]),
                              "pred_bboxes": np.zeros((4, 4), np.int32)})
    assert res["pred_RTs"].shape == (4, 4, 4) and res["pred_scales"].shape == (4, 3)
    res.update(gt_class_ids=cls, gt_RTs=res["pred_RTs"].astype(np.float64), gt_scales=res["pred_scales"].astype(np.float64),
               gt_handle_visibility=np.ones(4, np.int32))
    E.mark_up_symmetry(res)
    iou_aps, pose_aps, _, _ = E.compute_degree_cm_mAP([res], E.SYNSET_NAMES, None, [5, 10], [5, 10], [0.25, 0.5, 0.75], 0.25, True)
    assert np.all(iou_aps[cls] == 1.0) and np.all(pose_aps[cls] == 1.0)
    off = copy.deepcopy(res)
    c20, s20 = np.cos(np.deg2rad(20)), np.sin(np.deg2rad(20))
    off["pred_RTs"][1][:3, :3] = off["pred_RTs"][1][:3, :3] @ np.array([[1, 0, 0], [0, c20, -s20], [0, s20, c20]], np.float32)
    off["pred_RTs"][1][:3, 3] += np.array([0.08, 0, 0], np.float32)
    _, pose_off, _, _ = E.compute_degree_cm_mAP([off], E.SYNSET_NAMES, None, [5, 10], [5, 10], [0.1], 0.1, True)
    assert pose_off[3, 0, 0] == 0.0 and pose_off[3, -1, -1] == 1.0 and pose_off[1, 0, 0] == 1.0
