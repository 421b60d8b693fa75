"""Randomised soak of the round-5 batched launches against the oracle chain: PoseChains of 1..8 members of random categories,
cloud sizes, pair counts, weights (few ... many survivors), static or shape-polymorphic, both captured forms -- every member's pose
against oracle.estimate_pose -- and cppf_vote_argmax_batch on closed-form inputs against the exact fp64 vote sum.
Run by hand on a GPU box:  python tests/soak_gpu_chain.py [seconds] [seed]   (not collected by pytest)."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def run(seconds, seed):
    import test_gpu_parity as T
    import cppf_amd.synthetic as syn
    from cppf_amd.inference import PoseChain, PosePipeline, grid_class, grid_shape
    from cppf_amd.models import voting
    from oracle import oracle as O
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    sph = np.load(os.path.join(HERE, "golden", "sphere.npz"))["pts"]
    cats = ["bottle", "bowl", "camera", "can", "laptop", "mug"]
    t_end = time.time() + seconds
    n_chains = n_members = n_votes = 0
    while time.time() < t_end:
        cs = int(rng.integers(0, 2**31 - 1))
        n_mem = int(rng.integers(1, 9))
        dynamic = bool(rng.integers(0, 2))
        sd = T.seeded_sd(cs % 1000)
        gain = float(rng.choice([1.0, 4.0, 12.0]))
        for key in ("final.weight", "final.bias"):
            sd[key] = sd[key] * gain
        enc = T.make_encoder(sd, [84, 32, 32, 16], 141, dev)
        pipes, want = [], []
        for j in range(n_mem):
            cat = cats[int(rng.integers(0, len(cats)))]
            n = int(rng.choice([64, 200, 512, 777, 1024, 1500]))
            k = int(rng.choice([2, 5, 9, 16, 24]))
            s = (cs + 7919 * j) % 100000
            ob = syn.make_object(cat, n, s)
            cfg = ob["cfg"]
            idx = syn.make_pairs(n, k, s)
            u_tr, u_rot = syn.make_uniforms(idx.shape[0], s)
            corners, dims = grid_shape(ob["pc"], cfg.res)
            if dynamic:
                p = PosePipeline(enc, cfg, 2048, idx.shape[0], bool(grid_class(dims)[1]), dev, sph, dynamic=True)
            else:
                p = PosePipeline(enc, cfg, n, idx.shape[0], dims, dev, sph)
            p.load(ob["pc"], ob["normals"], ob["feat"], idx, u_tr, u_rot, corners[0].copy(), dims=dims if dynamic else None)
            ocfg = dict(res=cfg.res, tr_num_bins=32, rot_num_bins=36, vote_range=cfg.vote_range, scale_mean=cfg.scale_mean,
                        regress_right=cfg.regress_right, ppffcs=[84, 32, 32, 16], out_dim=141)
            want.append((O.estimate_pose(ob["pc"], ob["normals"], ob["feat"], idx, sd, ocfg, u_tr, u_rot, sph), (cat, n, k, s, dynamic, gain)))
            pipes.append(p)
        chain = PoseChain(pipes)
        for form in (False, True):
            chain.full_first = form
            chain.run()
            chain.full_first = form                       # (run() adapts the form: the replay is of the same one)
            poses = chain.run()                           # a replay: the accumulators start from the last run's values
            for r, (o, tag) in zip(poses, want):
                how = (form, tag)
                assert r["argmax"] == o["argmax"], how
                assert r["n_surv"] == int(o["mask"].sum()), how
                assert np.array_equal(r["ws"].mask.cpu().numpy().astype(bool), o["mask"]), how
                assert np.array_equal(r["outputs"].cpu().numpy(), o["outputs"]), how
                assert np.array_equal(r["heads"].cpu().numpy()[o["mask"]], o["heads"][o["mask"]]), how
                np.testing.assert_allclose(r["T"], o["T"], rtol=0, atol=1e-12, err_msg=str(how))
                if o["mask"].any():
                    np.testing.assert_allclose(r["up"], o["up"], atol=1e-12, err_msg=str(how))
                    np.testing.assert_allclose(r["scale"], o["scale"], rtol=1e-6, err_msg=str(how))
        chain.release()
        for p in pipes:
            p.release()
        n_chains += 1
        n_members += n_mem
        # batched votes on closed-form inputs: every cell against the exact fp64 sum, the arg-max its arg-max
        items, cases = [], []
        for j in range(int(rng.integers(1, 9))):
            cat = cats[int(rng.integers(0, len(cats)))]
            n = int(rng.choice([128, 512, 1024, 2048]))
            k = int(rng.choice([4, 16, 40]))
            s = (cs + 104729 * j) % 100000
            ob = syn.make_object(cat, n, s)
            res = float(np.float32(ob["cfg"].res * float(rng.choice([1.0, 1.0, 0.6, 2.0]))))
            idx = syn.make_pairs(n, k, s).astype(np.int64 if j % 2 else np.int32)
            out = syn.closed_form_outputs(ob["pc"], ob["center"], idx, ob["cfg"], quantise=bool(rng.integers(0, 2)))
            corners, dims = grid_shape(ob["pc"], res)
            items.append(dict(points=T.t(ob["pc"], dev), outputs=T.t(out, dev), point_idxs=T.t(idx, dev),
                              grid=torch.full(tuple(dims), float("nan"), dtype=torch.float32, device=dev), corner=T.t(corners[0], dev), res=res,
                              out_idx=torch.zeros(1, dtype=torch.int64, device=dev), out_val=torch.zeros(1, dtype=torch.float32, device=dev)))
            cases.append((ob, out, idx, corners, dims, res))
        voting.vote_argmax_batch(items, 72, True, workgroups=int(rng.choice([0, 64, 128])))
        torch.cuda.synchronize()
        for it, (ob, out, idx, corners, dims, res) in zip(items, cases):
            g64, _ = T.check_grid(O, it["grid"].cpu().numpy(), ob["pc"], out, idx.astype(np.int32), corners[0], dims, res, 72, True, bits_slack=4)
            assert int(it["out_idx"]) == int(np.argmax(g64)), (dims, res)
            n_votes += 1
    return n_chains, n_members, n_votes


if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("chain soak ok: %d chains, %d members (both forms, replayed), %d batched votes" % run(seconds, s))
