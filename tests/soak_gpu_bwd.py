"""Randomised parity soak of the two backward paths on the GPU: every gradient bit-exact against the oracle
(oracle/backward_oracle.c, oracle/sprin_bwd_oracle.c) for random sizes, index widths, head widths and neighbour counts.
`python tests/soak_gpu_bwd.py [seconds] [seed]`; tests/test_gpu_backward.py::test_randomised_backward_soak runs it briefly."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(budget, seed, dev=None):
    from cppf_amd.models.model import PPFEncoder, PointEncoder
    from cppf_amd.models.sprin import pack_point_encoder
    from oracle import oracle as O
    O.build()
    O.lib()
    dev = dev or torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    t0 = time.time()
    n_pair = n_point = 0
    while time.time() - t0 < budget:
        # ---- pair encoder
        n = int(rng.choice([3, 40, 300, 2048]))
        p = int(rng.choice([1, 2, 63, 64, 65, 500, 4097, 30000]))
        od = int(rng.choice([141, 141, 9, 16, 33, 128, 144]))
        i32 = bool(rng.integers(2))
        pc = rng.normal(0, 0.1, (n, 3)).astype(np.float32)
        nrm = rng.normal(0, 1, (n, 3)); nrm = (nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)).astype(np.float32)
        feat = rng.normal(0, 1, (n, 40)).astype(np.float32)
        idxs = rng.integers(0, n, (p, 2)).astype(np.int64)
        if rng.integers(3) == 0:
            idxs[:, 1] = idxs[:, 0]                       # degenerate pairs (a == b): zero distance
        R = (rng.normal(0, 1, (p, od)) * rng.choice([1e-3, 1.0, 50.0])).astype(np.float32)
        torch.manual_seed(int(rng.integers(1 << 30)))
        enc = PPFEncoder([84, 32, 32, 16], od)
        sd = {k: v.detach().numpy().copy() for k, v in enc.state_dict().items()}
        enc = enc.to(dev).train()
        f = torch.from_numpy(feat).to(dev).requires_grad_(True)
        it = torch.from_numpy(idxs.astype(np.int32) if i32 else idxs).to(dev)
        enc.forward_with_idx(torch.from_numpy(pc).to(dev), torch.from_numpy(nrm).to(dev), f, it).backward(torch.from_numpy(R).to(dev))
        _, gf_o, flat_o = O.pair_mlp_backward(pc, nrm, feat, idxs, sd, [84, 32, 32, 16], od, R)
        flat = torch.cat([q.grad.reshape(-1) for q in enc._ordered_params()]).cpu().numpy()
        tag = ("pair", n, p, od, i32)
        assert np.array_equal(flat, flat_o), tag
        assert np.array_equal(f.grad.cpu().numpy(), gf_o), tag
        n_pair += 1
        # ---- point encoder
        n = int(rng.choice([1, 5, 64, 257, 1100]))
        k = int(min(n, rng.choice([1, 3, 16, 17, 40, 60, 64])))
        th = rng.uniform(0, 2 * np.pi, n)
        pc = (np.stack([0.05 * np.cos(th), rng.uniform(-0.15, 0.15, n), 0.05 * np.sin(th)], -1) + rng.normal(0, 1e-3, (n, 3))).astype(np.float32)
        nrm = rng.normal(0, 1, (n, 3)); nrm = (nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)).astype(np.float32)
        R = rng.normal(0, 1, (n, 40)).astype(np.float32)
        torch.manual_seed(int(rng.integers(1 << 30)))
        penc = PointEncoder(k=k, spfcs=[32, 64, 32, 32], num_layers=1, out_dim=32)
        with torch.no_grad():
            for q in penc.parameters():
                if q.ndim == 1:
                    q.add_(0.1 * torch.randn_like(q))
        psd = {kk: v.detach().numpy().copy() for kk, v in penc.state_dict().items()}
        penc = penc.to(dev).train()
        pcd = torch.from_numpy(pc).to(dev)
        nbrs = penc.neighbours(pcd).cpu().numpy()
        penc.zero_grad()
        out = penc.forward_nbrs(pcd[None], torch.from_numpy(nrm[None]).to(dev), torch.from_numpy(nbrs[None]).to(dev))
        out[0].backward(torch.from_numpy(R).to(dev))
        packed, _ = pack_point_encoder(psd, 1)
        g_o, _ = O.point_encoder_backward(pc, nrm, nbrs, packed, R)
        for name, q in penc.named_parameters():
            assert np.array_equal(q.grad.cpu().numpy(), g_o[name]), ("point", n, k, name)
        n_point += 1
    return n_pair, n_point


if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("backward soak ok: %d pair-encoder, %d point-encoder cases" % run(seconds, s))
