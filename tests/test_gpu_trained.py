"""Train -> infer -> pose recovered: the one functional claim parity-with-an-oracle cannot make.  The two networks of
train.py:34-35 are trained with the loss of train.py:68-87 on posed synthetic objects -- forward AND backward of both encoders on
the HIP kernels -- and nocs/inference.py:177-339 (cppf_amd.inference.estimate_pose) then recovers centre, axes and size of
held-out objects in poses and sizes no training step saw.  tests/golden/trained_<category>.npz hold the weights
scripts/train_synthetic.py produced on one MI355X (10 000 steps each); bench.py's trained regime runs them."""
import os

import numpy as np
import pytest
import torch

import cppf_amd.synthetic as syn
from conftest import GOLDEN
from cppf_amd import training

pytestmark = pytest.mark.gpu


def _held_out(penc, enc, cat, dev, n=8, n_points=1536):
    errs = []
    for j in range(n):
        ob = syn.make_posed_object(cat, n_points, 900000 + j)      # seeds no training step uses (training: 10000 + step)
        pose = training.infer(penc, enc, ob, dev, seed=j)
        e = training.pose_errors(pose, ob)
        e["n_surv"] = pose["n_surv"]
        errs.append(e)
    return errs


def _med(errs, key):
    return float(np.median([e[key] for e in errs]))


def test_training_runs_on_the_hip_kernels_and_the_pose_comes_out(dev):
    """from scratch, a few seconds: 1 500 steps on necked bottles; both encoders' forward / backward are the HIP autograd Functions"""
    cfg = syn.CATEGORIES["bottle"]
    penc, enc = training.new_encoders(cfg, dev)
    ob = syn.make_posed_object("bottle", 512, 1)
    pcs, nrms = torch.from_numpy(ob["pc"][None]).to(dev), torch.from_numpy(ob["normals"][None]).to(dev)
    feat = penc.train()(pcs, nrms, torch.cdist(pcs, pcs))
    out = enc.train()(pcs, nrms, feat, idxs=torch.randint(0, 512, (1000, 2), device=dev))
    assert "PointEncoderFunction" in type(feat.grad_fn).__name__ + str(feat.grad_fn.next_functions)
    assert "PairMlpFunction" in type(out.grad_fn).__name__ + str(out.grad_fn.next_functions)
    penc, enc, losses = training.train("bottle", dev, steps=1500, n_points=(768, 2048), encoders=(penc, enc))
    assert losses[-1] < 0.6 * losses[0], losses
    errs = _held_out(penc, enc, "bottle", dev, n=6)
    assert _med(errs, "t_cells") <= 4.0 and _med(errs, "up_deg_mod_sign") <= 10.0 and _med(errs, "scale_rel") <= 0.2, errs
    assert min(e["n_surv"] for e in errs) > 1000


@pytest.mark.parametrize("cat", ["bottle", "mug", "laptop"])
def test_committed_trained_weights_recover_held_out_poses(dev, cat):
    cfg = syn.CATEGORIES[cat]
    penc, enc = training.load_weights(os.path.join(GOLDEN, f"trained_{cat}.npz"), cfg, dev)
    errs = _held_out(penc, enc, cat, dev)
    print(cat, {k: round(_med(errs, k), 3) for k in ("t_cells", "up_deg_mod_sign", "scale_rel")})
    assert _med(errs, "t_cells") <= 2.0 and max(e["t_cells"] for e in errs) <= 4.0, errs          # centre: within 1-2 cells
    assert _med(errs, "up_deg_mod_sign") <= 5.0 and max(e["up_deg_mod_sign"] for e in errs) <= 12.0, errs
    assert _med(errs, "scale_rel") <= 0.10 and max(e["scale_rel"] for e in errs) <= 0.2, errs
    if cat == "bottle":                                       # the neck tells up from down: the sign head must get it right
        assert sum(e["up_deg"] < 15 for e in errs) >= len(errs) - 1, errs
    if cfg.regress_right:
        assert _med(errs, "right_deg_mod_sign") <= 10.0, errs
    # a trained network's back-vote keeps a large share of the pairs (a random-weight one: 0.4 %)
    assert min(e["n_surv"] for e in errs) > 0.05 * 100000
