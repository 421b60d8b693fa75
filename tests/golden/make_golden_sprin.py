#!/usr/bin/env python3
"""Generate tests/golden/sprin_*.npz from the reference's own PointEncoder (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sprin.py /root/reference

models/model.py + models/sprin.py import as-is on CPU (torch/numpy only).  For each configuration the
reference module is built under a fixed seed (LayerNorm affine parameters are re-drawn so they are not
the identity), run as the scripts run it (nocs/inference.py:180-181: dist = torch.cdist(pcs, pcs);
point_encoder(pcs, pc_normals, dist)), and the following DATA is stored: the state_dict, the inputs, the
neighbour sets torch.topk chose, the gap between the k-th and (k+1)-th distance of every row (so tests
know which rows have an unambiguous neighbour set), and the [N, 40] output.
"""
import os
import sys

import numpy as np
import torch

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out_dir = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, ref)
from models.model import PointEncoder  # noqa: E402  (the reference's own class)

torch.set_num_threads(1)


def cloud(rng, n):
    th = rng.uniform(0, 2 * np.pi, n)
    h = rng.uniform(-0.15, 0.15, n)
    pc = np.stack([0.05 * np.cos(th), h, 0.05 * np.sin(th)], -1) + rng.normal(0, 1e-3, (n, 3))
    nrm = np.stack([np.cos(th), np.zeros(n), np.sin(th)], -1) + rng.normal(0, 0.05, (n, 3))
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    return pc.astype(np.float32), nrm.astype(np.float32)


for tag, n, k, spfcs, layers, seed in (("l1", 256, 60, [32, 64, 32, 32], 1, 1), ("l2", 96, 16, [16, 24], 2, 2)):
    torch.manual_seed(seed)
    enc = PointEncoder(k=k, spfcs=spfcs, num_layers=layers, out_dim=32).eval()
    with torch.no_grad():
        for name, p in enc.named_parameters():
            if "layer_norm" in name or (".kernel." in name and p.ndim == 1 and name.endswith("weight")):
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
            elif ".kernel." in name and name.endswith("bias"):
                p.add_(0.1 * torch.randn_like(p))
    pc, nrm = cloud(np.random.default_rng(seed), n)
    pcs, nrms = torch.from_numpy(pc[None]), torch.from_numpy(nrm[None])
    with torch.no_grad():
        dist = torch.cdist(pcs, pcs)
        out = enc(pcs, nrms, dist)
        nbrs = torch.topk(dist, k, largest=False, sorted=False)[1]
        out_nbrs = enc.forward_nbrs(pcs, nrms, nbrs)
    assert torch.equal(out, out_nbrs)
    ds = np.sort(dist[0].numpy(), -1)
    data = {"pc": pc, "nrm": nrm, "k": np.int32(k), "num_layers": np.int32(layers), "spfcs": np.asarray(spfcs, np.int32),
            "dist": dist[0].numpy().astype(np.float32) if n <= 128 else np.zeros((0,), np.float32),
            "nbrs_topk": np.sort(nbrs[0].numpy(), -1).astype(np.int16), "kth_gap": (ds[:, k] - ds[:, k - 1]).astype(np.float32),
            "out": out[0].numpy()}
    for key, v in enc.state_dict().items():
        data["sd::" + key] = v.numpy().copy()
    np.savez_compressed(os.path.join(out_dir, f"sprin_{tag}.npz"), **data)
    print(tag, "params", sum(v.numel() for v in enc.state_dict().values()), "out", tuple(out.shape),
          "min kth gap", float(data["kth_gap"].min()))
