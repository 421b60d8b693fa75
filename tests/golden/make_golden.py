#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference itself (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py /root/reference

What is executed from the reference, and how:
  * models/model.py is imported as-is on CPU (it needs only torch/numpy): PPFEncoder with seeded
    default init -> state_dict + logits of forward(..., idxs=...) / forward_with_idx.   (mlp_*.npz)
  * utils/util.py:fibonacci_sphere and utils/dataset.py:generate_target are pure Python/numpy
    functions inside modules whose top-level imports (open3d, cv2, ...) are absent here.  Their
    FunctionDef source is extracted with `ast` and exec'd with the real math/numpy -- the
    reference's own code runs, no stand-in modules are written.                (sphere.npz, targets.npz)
  * the bin->value affine maps of nocs/inference.py:187-188,252,256 are script-level statements;
    they are re-typed here with the same torch float ops on the bin indices.        (binvals.npz)
The vote kernels (models/voting.py) are CUDA text behind `import cupy`: not executable here, no
fixture (see DESIGN.md "Oracle").  Only data is written: inputs and expected outputs.
"""
import ast
import math
import os
import sys

import numpy as np
import torch

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out_dir = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, ref)
from models.model import PPFEncoder  # noqa: E402  (the reference's own class)

torch.set_num_threads(1)


def extract(path, name, env):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), env)
            return env[name]
    raise KeyError(name)


def cloud(rng, n):
    """points on a cylinder-ish surface + unit normals + 40-d features (inputs only)"""
    th = rng.uniform(0, 2 * np.pi, n)
    h = rng.uniform(-0.15, 0.15, n)
    pc = np.stack([0.05 * np.cos(th), h, 0.05 * np.sin(th)], -1) + rng.normal(0, 1e-3, (n, 3))
    nrm = np.stack([np.cos(th), np.zeros(n), np.sin(th)], -1) + rng.normal(0, 0.05, (n, 3))
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    feat = rng.normal(0, 1, (n, 40))
    return pc.astype(np.float32), nrm.astype(np.float32), feat.astype(np.float32)


# ---------------------------------------------------------------- F1/F2: weights + logits
for tag, out_dim in (("141", 2 * 32 + 2 * 36 + 2 + 3), ("9", 2 + 2 + 2 + 3)):
    torch.manual_seed(0)
    enc = PPFEncoder([84, 32, 32, 16], out_dim).eval()
    sd = {k: v.detach().numpy().copy() for k, v in enc.state_dict().items()}
    rng = np.random.default_rng(1)
    N, P = 256, 512
    pc, nrm, feat = cloud(rng, N)
    idxs = rng.integers(0, N, (P, 2)).astype(np.int64)
    idxs[:8, 1] = idxs[:8, 0]            # a == b (degenerate pairs still get a row)
    idxs[8:16] = idxs[16:24]             # duplicates
    with torch.no_grad():
        y = enc(torch.from_numpy(pc)[None], torch.from_numpy(nrm)[None], torch.from_numpy(feat)[None],
                idxs=idxs)               # numpy int64 idxs, as nocs/inference.py:182
        y2 = enc.forward_with_idx(torch.from_numpy(pc), torch.from_numpy(nrm), torch.from_numpy(feat),
                                  torch.from_numpy(idxs))   # LongTensor idxs, as train.py:66
    assert y.shape == (1, P, out_dim) and torch.equal(y[0], y2)
    np.savez_compressed(os.path.join(out_dir, f"mlp_{tag}.npz"), pc=pc, nrm=nrm, feat=feat, idxs=idxs,
                        logits=y[0].numpy(), **{"sd." + k: v for k, v in sd.items()})

# a non-default architecture (generic path): ppffcs=[44,24,24], F=20, out_dim=10
torch.manual_seed(3)
enc = PPFEncoder([44, 24, 24], 10).eval()
sd = {k: v.detach().numpy().copy() for k, v in enc.state_dict().items()}
rng = np.random.default_rng(4)
pc, nrm, feat = cloud(rng, 128)
feat = feat[:, :20].copy()
idxs = rng.integers(0, 128, (300, 2)).astype(np.int64)
with torch.no_grad():
    y = enc(torch.from_numpy(pc)[None], torch.from_numpy(nrm)[None], torch.from_numpy(feat)[None], idxs=idxs)
np.savez_compressed(os.path.join(out_dir, "mlp_generic.npz"), pc=pc, nrm=nrm, feat=feat, idxs=idxs,
                    logits=y[0].numpy(), **{"sd." + k: v for k, v in sd.items()})

# ---------------------------------------------------------------- sphere bins
fib = extract(os.path.join(ref, "utils/util.py"), "fibonacci_sphere", {"math": math})
num_samples = int(4 * np.pi / (1.5 / 180 * np.pi))          # nocs/inference.py:100-101
np.savez_compressed(os.path.join(out_dir, "sphere.npz"), n=num_samples, pts=np.array(fib(num_samples)),
                    pts7=np.array(fib(7)))

# ---------------------------------------------------------------- closed-form targets
gt = extract(os.path.join(ref, "utils/dataset.py"), "generate_target", {"np": np})
rng = np.random.default_rng(2)
pc, nrm, _ = cloud(rng, 200)
np.random.seed(5)                                            # generate_target draws its own pairs
tr, rot, aux, pidx = gt(pc, nrm.copy(), subsample=1000)
np.savez_compressed(os.path.join(out_dir, "targets.npz"), pc=pc, nrm=nrm, point_idxs=pidx, target_tr=tr,
                    target_rot=rot, target_rot_aux=aux)

# ---------------------------------------------------------------- bin -> value maps
k32 = torch.arange(32).float()
k36 = torch.arange(36).float()
vr = [0.25, 0.25]
mu = k32 / (32 - 1) * 2 * vr[0] - vr[0]                      # nocs/inference.py:187
nu = k32 / (32 - 1) * vr[1]                                  # :188
th = k36 / (36 - 1) * np.pi                                  # :252
np.savez_compressed(os.path.join(out_dir, "binvals.npz"), mu=mu.numpy(), nu=nu.numpy(), theta=th.numpy(),
                    vote_range=np.array(vr))
print("golden fixtures written to", out_dir)
