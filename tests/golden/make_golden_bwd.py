#!/usr/bin/env python3
"""Generate tests/golden/bwd_*.npz: gradients torch autograd computes through the reference's own
PPFEncoder.forward_with_idx (models/model.py:117-137), as train.py:66,91 does (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_bwd.py /root/reference

For a seeded module, cloud and pair list, the upstream gradient dL/dlogits = R (fixed random matrix) is pushed
through `(logits * R).sum().backward()`.  Stored DATA: state_dict, inputs, R, d/d(param) for every parameter,
d/d(feat).  `feat` is a leaf that requires grad, as sprin_feat is in train.py:64.
"""
import os
import sys

import numpy as np
import torch

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out_dir = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, ref)
from models.model import PPFEncoder  # noqa: E402

torch.set_num_threads(1)


def cloud(rng, n, f):
    th = rng.uniform(0, 2 * np.pi, n)
    h = rng.uniform(-0.15, 0.15, n)
    pc = np.stack([0.05 * np.cos(th), h, 0.05 * np.sin(th)], -1) + rng.normal(0, 1e-3, (n, 3))
    nrm = np.stack([np.cos(th), np.zeros(n), np.sin(th)], -1) + rng.normal(0, 0.05, (n, 3))
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    return pc.astype(np.float32), nrm.astype(np.float32), rng.normal(0, 1, (n, f)).astype(np.float32)


for tag, ppffcs, out_dim, n, p, seed in (("141", [84, 32, 32, 16], 141, 128, 300, 5), ("generic", [44, 24, 24], 10, 64, 130, 6)):
    torch.manual_seed(seed)
    enc = PPFEncoder(ppffcs, out_dim)
    rng = np.random.default_rng(seed)
    pc, nrm, feat = cloud(rng, n, (ppffcs[0] - 4) // 2)
    idxs = rng.integers(0, n, (p, 2)).astype(np.int64)
    idxs[3] = idxs[2]                      # a repeated pair
    idxs[5, 1] = idxs[5, 0]                # a == b (degenerate PPF, models/model.py:122 adds 1e-7)
    R = rng.normal(0, 1, (p, out_dim)).astype(np.float32)
    f = torch.from_numpy(feat).requires_grad_(True)
    logits = enc.forward_with_idx(torch.from_numpy(pc), torch.from_numpy(nrm), f, torch.from_numpy(idxs))
    (logits * torch.from_numpy(R)).sum().backward()
    data = {"pc": pc, "nrm": nrm, "feat": feat, "idxs": idxs, "R": R, "grad_feat": f.grad.numpy().copy(),
            "logits": logits.detach().numpy().copy()}
    for k, v in enc.state_dict().items():
        data["sd." + k] = v.numpy().copy()
    for k, v in enc.named_parameters():
        data["grad." + k] = v.grad.numpy().copy()
    np.savez_compressed(os.path.join(out_dir, f"bwd_{tag}.npz"), **data)
    print(tag, "params", sum(v.numel() for v in enc.parameters()), "pairs", p)
