"""Randomised parity soak on the GPU (vote grids against the exact-sum bound and the arg-max, back-vote offsets
bit-exact, kNN sets exact), all against the oracle.  `python tests/soak_gpu.py [seconds] [seed]` for a long run;
tests/test_gpu_parity.py::test_randomised_soak runs it for a few seconds.  A 240 s run covers ~4 600 cases."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

CATS = ["bottle", "bowl", "camera", "can", "laptop", "mug", "bed", "sofa"]


def run(budget, seed, dev=None):
    import test_gpu_parity as T
    import cppf_amd.synthetic as syn
    from cppf_amd import _lib
    from cppf_amd._torch_util import stream_ptr
    from cppf_amd.models import voting
    from cppf_amd.models.model import PointEncoder
    from oracle import oracle as O
    O.build()
    O.lib()
    dev = dev or torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    t0 = time.time()
    n_v = n_b = n_k = n_i = 0
    L = _lib.lib()
    while time.time() - t0 < budget:
        cat = CATS[rng.integers(len(CATS))]
        n = int(rng.choice([64, 200, 512, 1024, 2048]))
        k = int(rng.choice([4, 16, 48]))
        case_seed = int(rng.integers(1 << 30))
        ob = syn.make_object(cat, n, case_seed)
        cfg = ob["cfg"]
        res = float(cfg.res * rng.choice([0.35, 0.5, 0.7, 1.0, 1.5, 3.0]))
        idx = syn.make_pairs(n, k, case_seed)
        P = idx.shape[0]
        idx32 = idx.astype(np.int32)
        # output regimes: known answer (quantised / exact), arbitrary (incl. negative nu, far-off mu), half and half
        mode = int(rng.integers(4))
        if mode == 0:
            outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
        elif mode == 1:
            outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=False)
        elif mode == 2:
            outputs = np.stack([rng.uniform(-0.4, 0.4, P), rng.uniform(-0.05, 0.4, P)], -1).astype(np.float32)
        else:
            outputs = syn.closed_form_outputs(ob["pc"], ob["center"], idx, cfg, quantise=True)
            sel = rng.random(P) < 0.5
            outputs[sel] = np.stack([rng.uniform(-0.3, 0.3, sel.sum()), rng.uniform(0, 0.3, sel.sum())], -1)
        corner, dims = O.grid_setup(ob["pc"], res)
        corner = (corner + np.float32(rng.choice([0.0, 0.0, 0.03, -0.05]))).astype(np.float32)
        if int(np.prod(dims)) > 3_000_000:
            continue
        n_rots = int(rng.choice([1, 7, 36, 72, 72, 72, 73, 90, 144, 200, 360]))    # > 72: passes over windows of 72 rotations
        adaptive = bool(rng.integers(2))
        probs = None if rng.integers(3) else rng.uniform(0.1, 2.0, n).astype(np.float32)
        tag = (cat, n, k, case_seed, res, mode, n_rots, adaptive)

        # centre vote: every cell within the fixed-point bound of the exact sum; arg-max when the peak is unique
        go, _ = T.oracle_vote(O, ob["pc"], outputs, idx32, corner, dims, res, n_rots, adaptive, probs)
        gg, flat, _ = T.run_vote(dev, ob["pc"], outputs, idx32, corner, dims, res, n_rots, adaptive, probs)
        T.check_grid(O, gg, ob["pc"], outputs, idx32, corner, dims, res, n_rots, adaptive, probs)
        srt = np.sort(go.reshape(-1))
        if srt[-1] > 0 and srt[-1] - srt[-2] > 1e-4 * srt[-1]:
            assert flat == O.grid_argmax(go)[0], tag
        n_v += 1
        if rng.integers(3) == 0:    # the same vote launched narrower (CPPF_VOTE_WORKGROUPS): longer chunks, possibly a coarser quantum
            w = int(rng.choice([64, 128, 200]))
            gw, flat_w, _ = T.run_vote(dev, ob["pc"], outputs, idx32, corner, dims, res, n_rots, adaptive, probs, workgroups=w)
            T.check_grid(O, gw, ob["pc"], outputs, idx32, corner, dims, res, n_rots, adaptive, probs, bits_slack=2)
            if srt[-1] > 0 and srt[-1] - srt[-2] > 1e-4 * srt[-1]:
                assert flat_w == O.grid_argmax(go)[0], tag + (w,)
        # the same vote as exact integers (tiled grids): EQUAL to the oracle's fixed-point statement, cell by cell
        if L.cppf_vote_tiles(int(dims[0]), int(dims[1]), int(dims[2])) > 0 and P > 0:
            raw = torch.full(tuple(int(d) for d in dims), -3, dtype=torch.int64, device=dev)
            q = torch.zeros(1, dtype=torch.float32, device=dev)
            fb = int(rng.choice([0, 0, 12, 24]))
            voting.vote_grid_raw(T.t(ob["pc"], dev), T.t(outputs, dev), None if probs is None else T.t(probs, dev), T.t(idx32, dev), raw, q,
                                 T.t(corner, dev), res, n_rots, adaptive, fixed_bits=fb)
            qv = float(q)
            pr = np.ones(n, np.float32) if probs is None else probs
            p2 = 2.0 ** np.ceil(np.log2(float(pr.max())))
            used = int(round(np.log2(p2 / qv)))
            assert qv > 0 and (fb == 0 or used == fb), tag
            want, qo = O.ppf_voting_fixed(ob["pc"], outputs, pr, idx32, dims, corner, res, n_rots, adaptive, used)
            assert qo == qv and np.array_equal(raw.cpu().numpy(), want), tag
            n_i += 1

        # back-vote around a plausible centre: offsets bit-exact
        center = (ob["center"] + rng.normal(0, 1.0, 3) * res).astype(np.float32)
        tol = np.float32(3 * res)
        nr = min(n_rots, 72)
        oo, _ = O.backvote(ob["pc"], outputs, idx32, corner, res, nr, dims, center, tol)
        out_d = torch.zeros((P, 3), dtype=torch.float32, device=dev)
        voting.backvote_kernel((1, 1, 1), (512, 1, 1),
                               (T.t(ob["pc"], dev), T.t(outputs, dev), out_d, T.t(idx32, dev), T.t(corner, dev), np.float32(res),
                                P, nr, int(dims[0]), int(dims[1]), int(dims[2]), T.t(center, dev), tol))
        assert np.array_equal(out_d.cpu().numpy(), oo), tag
        # the pipelines' form: mask + survivors per chunk of 1 024 pairs, then the scatter that needs no scan
        nch = (P + 1023) // 1024
        m_d = torch.empty(P, dtype=torch.uint8, device=dev)
        cc_d = torch.zeros(nch, dtype=torch.int32, device=dev)
        sv_d = torch.empty(P, dtype=torch.int32, device=dev)
        ct_d = torch.empty(1, dtype=torch.int32, device=dev)
        pc_d, o_d, i_d, c_d, g_d = T.t(ob["pc"], dev), T.t(outputs, dev), T.t(idx32, dev), T.t(corner, dev), T.t(center, dev)
        _lib.check(L.cppf_backvote_count(pc_d.data_ptr(), o_d.data_ptr(), i_d.data_ptr(), c_d.data_ptr(), float(np.float32(res)), P, nr,
                                         int(dims[0]), int(dims[1]), int(dims[2]), None, g_d.data_ptr(), float(tol), m_d.data_ptr(),
                                         cc_d.data_ptr(), None, stream_ptr(dev)), "backvote_count")
        _lib.check(L.cppf_compact_scatter(m_d.data_ptr(), P, cc_d.data_ptr(), sv_d.data_ptr(), ct_d.data_ptr(), stream_ptr(dev)),
                   "compact_scatter")
        ref = np.nonzero(np.any(oo != 0, -1))[0]
        assert int(ct_d.item()) == ref.size and np.array_equal(sv_d.cpu().numpy()[:ref.size], ref), tag
        n_b += 1

        # neighbour sets, with a few exact duplicates
        kk = min(int(rng.choice([1, 5, 16, 60, 64])), n)
        pc2 = ob["pc"].copy()
        dup = int(rng.integers(0, 8))
        if dup:
            pc2[-dup:] = pc2[:dup]
        enc = PointEncoder(k=kk, spfcs=[32, 64, 32, 32], num_layers=1, out_dim=32)
        got = enc.neighbours(torch.from_numpy(pc2).to(dev)).cpu().numpy()
        assert np.array_equal(got, O.knn(pc2, kk)), ("knn", n, kk, case_seed)
        n_k += 1
    return n_v, n_b, n_k, n_i


if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("soak ok: %d vote, %d back-vote, %d kNN, %d integer-image cases" % run(seconds, s))
