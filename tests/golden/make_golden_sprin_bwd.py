#!/usr/bin/env python3
"""Generate tests/golden/sprin_bwd.npz: parameter gradients torch autograd computes through the reference's own
PointEncoder (models/model.py:36-61, models/sprin.py:40-107), as train.py:62-64,91 does (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sprin_bwd.py /root/reference

The module is built like train.py:34 builds it (k neighbours, spfcs [32, 64, 32, 32], one layer, out_dim 32), its
LayerNorm affine parameters re-drawn so that they are not the identity; for a fixed upstream gradient R the loss
(out * R).sum() is back-propagated.  Stored DATA: state_dict, cloud, the neighbour sets torch.topk chose, R, the
output and d/d(parameter) for every parameter.  Points and normals carry no gradient (train.py:58-60).
"""
import os
import sys

import numpy as np
import torch

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out_dir = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, ref)
from models.model import PointEncoder  # noqa: E402

torch.set_num_threads(1)
n, k, seed = 200, 60, 11
rng = np.random.default_rng(seed)
th = rng.uniform(0, 2 * np.pi, n)
h = rng.uniform(-0.15, 0.15, n)
pc = (np.stack([0.05 * np.cos(th), h, 0.05 * np.sin(th)], -1) + rng.normal(0, 1e-3, (n, 3))).astype(np.float32)
nrm = np.stack([np.cos(th), np.zeros(n), np.sin(th)], -1) + rng.normal(0, 0.05, (n, 3))
nrm = (nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)).astype(np.float32)
torch.manual_seed(seed)
enc = PointEncoder(k=k, spfcs=[32, 64, 32, 32], num_layers=1, out_dim=32)
with torch.no_grad():
    for name, p in enc.named_parameters():
        if "layer_norm" in name or (".kernel." in name and p.ndim == 1 and name.endswith("weight")):
            p.copy_(1.0 + 0.2 * torch.randn_like(p))
        elif ".kernel." in name and name.endswith("bias"):
            p.add_(0.1 * torch.randn_like(p))
pcs, nrms = torch.from_numpy(pc[None]), torch.from_numpy(nrm[None])
R = rng.normal(0, 1, (n, 40)).astype(np.float32)
with torch.no_grad():
    dist = torch.cdist(pcs, pcs)
    nbrs = torch.topk(dist, k, largest=False, sorted=False)[1]
out = enc(pcs, nrms, dist)
(out[0] * torch.from_numpy(R)).sum().backward()
ds = np.sort(dist[0].numpy(), -1)
data = {"pc": pc, "nrm": nrm, "k": np.int32(k), "R": R, "out": out[0].detach().numpy(),
        "nbrs_topk": np.sort(nbrs[0].numpy(), -1).astype(np.int16), "kth_gap": (ds[:, k] - ds[:, k - 1]).astype(np.float32)}
for key, v in enc.state_dict().items():
    data["sd::" + key] = v.numpy().copy()
for name, p in enc.named_parameters():
    data["grad::" + name] = p.grad.numpy().copy()
np.savez_compressed(os.path.join(out_dir, "sprin_bwd.npz"), **data)
print("sprin_bwd.npz:", len([1 for _ in enc.parameters()]), "parameter tensors, min kth gap", float(data["kth_gap"].min()),
      "max |grad|", max(float(p.grad.abs().max()) for p in enc.parameters()))
