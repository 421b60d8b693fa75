#!/usr/bin/env python3
"""Training targets and loss from the reference itself (run in the build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train.py /root/reference

  * utils/dataset.py:generate_target and utils/util.py:real2prob are executed from their own source (FunctionDef extracted with
    `ast`, exec'd with the real numpy / torch: the modules' top-level imports -- open3d, cv2, hydra ... -- are absent here);
  * the soft-bin construction of utils/dataset.py:232-243 and the loss of train.py:68-87 are script / method-level statements:
    they are re-typed below with the same library calls (np.clip, real2prob, nn.KLDivLoss(reduction='batchmean'),
    nn.BCEWithLogitsLoss, F.mse_loss) on the reference functions' outputs.
Only data is written (train_targets.npz): inputs, the reference functions' outputs, the loss values."""
import ast
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ref = next((a for a in sys.argv[1:] if not a.startswith("--")), "/root/reference")
out_dir = os.path.dirname(os.path.abspath(__file__))


# Trust: the two FunctionDefs below are exec'd from the reference checkout with this user's privileges.  The checkout is untrusted
# content, so the files are pinned by hash -- the versions that were read (functions are pure numpy / torch arithmetic) -- and a
# modified checkout is refused (--trust-modified-reference overrides, for a reviewed upgrade of the reference).
PINNED = {"utils/dataset.py": "5abbd20c0b137c9c5bf35997cda5eb210ab6e8017baba2537bb96d400c7035a6",
          "utils/util.py": "d6b3854a81d35c3eb0139899f0e2a687095ba37e3294bcb084ec8f4e8d4dddf0"}


def extract(path, name, env):
    import hashlib
    rel = os.path.relpath(path, ref)
    digest = hashlib.sha256(open(path, "rb").read()).hexdigest()
    if digest != PINNED.get(rel) and "--trust-modified-reference" not in sys.argv:
        raise SystemExit(f"{rel}: sha256 {digest} is not the pinned version; refusing to exec code from it")
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), env)
            return env[name]
    raise KeyError(name)


gt = extract(os.path.join(ref, "utils/dataset.py"), "generate_target", {"np": np})
real2prob = extract(os.path.join(ref, "utils/util.py"), "real2prob", {"np": np, "torch": torch})

NP = 400
rng = np.random.default_rng(12)
n = 300
th = rng.uniform(0, 2 * np.pi, n)
pc = np.stack([0.05 * np.cos(th), rng.uniform(-0.15, 0.15, n), 0.05 * np.sin(th)], -1) + rng.normal(0, 1e-3, (n, 3))
nrm = np.stack([np.cos(th), np.zeros(n), np.sin(th)], -1) + rng.normal(0, 0.05, (n, 3))
nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
pc, nrm = pc.astype(np.float32), nrm.astype(np.float32)
out = dict(pc=pc, nrm=nrm)
tr_bins, rot_bins, vote_range, scale_mean = 32, 36, [0.25, 0.25], np.array([0.05, 0.15, 0.05])
kldiv, bce = nn.KLDivLoss(reduction="batchmean"), nn.BCEWithLogitsLoss()
for tag, (up_sym, z_right, regress_right) in (("plain", (False, False, True)), ("upsym", (True, False, False)), ("zright", (False, True, True))):
    np.random.seed(7)
    tr, rot, aux, pidx = gt(pc, nrm.copy(), up_sym=up_sym, right_sym=False, z_right=z_right, subsample=NP)      # utils/dataset.py:230
    tr_soft = np.stack([real2prob(np.clip(tr[:, 0] + vote_range[0], 0, 2 * vote_range[0]), 2 * vote_range[0], tr_bins, circular=False),
                        real2prob(np.clip(tr[:, 1], 0, vote_range[1]), vote_range[1], tr_bins, circular=False)], 1)   # :232-237
    rot_soft = np.stack([real2prob(rot[:, 0], np.pi, rot_bins, circular=False),
                         real2prob(rot[:, 1], np.pi, rot_bins, circular=False)], 1)                                   # :239-243
    half_extents = np.array([0.055, 0.14, 0.048])
    scale_t = (np.log(half_extents.astype(np.float32)) - np.log(scale_mean)).astype(np.float32)                       # :246-248
    torch.manual_seed(3)
    preds = torch.randn(1, NP, 2 * tr_bins + 2 * rot_bins + 2 + 3)
    targets_tr, targets_rot, targets_rot_aux = (torch.from_numpy(a)[None] for a in (tr_soft, rot_soft, aux))
    targets_scale = torch.from_numpy(scale_t)[None]
    preds_tr = preds[..., :2 * tr_bins].reshape(-1, 2, tr_bins)                                                       # train.py:68-75
    preds_up = preds[..., 2 * tr_bins:2 * tr_bins + rot_bins]
    preds_right = preds[..., 2 * tr_bins + rot_bins:2 * tr_bins + 2 * rot_bins]
    loss_tr = kldiv(F.log_softmax(preds_tr[:, 0], dim=-1), targets_tr[0, :, 0]) + kldiv(F.log_softmax(preds_tr[:, 1], dim=-1), targets_tr[0, :, 1])
    loss_up = kldiv(F.log_softmax(preds_up[0], dim=-1), targets_rot[0, :, 0])                                          # :77-80
    loss_up_aux = bce(preds[..., -5][0], targets_rot_aux[0, :, 0])
    loss_scale = F.mse_loss(preds[..., -3:], targets_scale[:, None])
    loss = loss_up + loss_tr + loss_up_aux + loss_scale
    if regress_right:                                                                                                 # :82-86
        loss = loss + kldiv(F.log_softmax(preds_right[0], dim=-1), targets_rot[0, :, 1]) + bce(preds[..., -4][0], targets_rot_aux[0, :, 1])
    out.update({f"{tag}.point_idxs": pidx, f"{tag}.tr": tr, f"{tag}.rot": rot, f"{tag}.aux": aux, f"{tag}.tr_soft": tr_soft.astype(np.float32),
                f"{tag}.rot_soft": rot_soft.astype(np.float32), f"{tag}.scale": scale_t,
                f"{tag}.loss": np.float64(loss.item()), f"{tag}.flags": np.array([up_sym, z_right, regress_right])})
out["half_extents"] = half_extents
out["reference_sha256"] = np.array([f"{k}:{v}" for k, v in sorted(PINNED.items())])      # which reference sources produced this file
out["preds"] = preds.numpy()              # (the same seeded draw for every tag)
np.savez_compressed(os.path.join(out_dir, "train_targets.npz"), **out)
print("train_targets.npz written")
