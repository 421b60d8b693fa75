"""Randomised soak of the WHOLE pose (centre chain + fused tail) against the oracle chain: random category, cloud size, pairs per
point and seeds; every case through estimate_pose (eager launches) and every fourth also through a captured PosePipeline.
Run by hand on a GPU box:  python tests/soak_gpu_pose.py [seconds] [seed]   (not collected by pytest)."""
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
    from cppf_amd.inference import PosePipeline, estimate_pose, grid_shape
    from oracle import oracle as O
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    sph = np.load(os.path.join(HERE, "golden", "sphere.npz"))["pts"]
    cats = ["bottle", "bowl", "camera", "can", "laptop", "mug"]
    t_end = time.time() + seconds
    n_cases = n_graph = 0
    while time.time() < t_end:
        cs = int(rng.integers(0, 2**31 - 1))
        cat = cats[int(rng.integers(0, len(cats)))]
        n = int(rng.choice([64, 200, 512, 777, 1024, 1500]))
        k = int(rng.choice([2, 5, 9, 16, 24]))
        tag = (cat, n, k, cs)
        ob = syn.make_object(cat, n, cs % 100000)
        cfg = ob["cfg"]
        idx = syn.make_pairs(n, k, cs % 100000)
        P = idx.shape[0]
        u_tr, u_rot = syn.make_uniforms(P, cs % 100000)
        sd = T.seeded_sd(cs % 1000)
        gain = float(rng.choice([1.0, 4.0, 12.0]))            # flat ... peaked bin distributions: few ... many survivors
        for key in ("final.weight", "final.bias"):
            sd[key] = sd[key] * gain
        enc = T.make_encoder(sd, [84, 32, 32, 16], 141, dev)
        ocfg = dict(res=cfg.res, tr_num_bins=32, rot_num_bins=36, vote_range=cfg.vote_range, scale_mean=cfg.scale_mean,
                    regress_right=cfg.regress_right, ppffcs=[84, 32, 32, 16], out_dim=141)
        o = O.estimate_pose(ob["pc"], ob["normals"], ob["feat"], idx, sd, ocfg, u_tr, u_rot, sph)

        def check(r, how):
            assert r["argmax"] == o["argmax"], (how, tag)
            assert r["n_surv"] == int(o["mask"].sum()), (how, tag)
            assert np.array_equal(r["ws"].mask.cpu().numpy().astype(bool), o["mask"]), (how, tag)
            assert np.array_equal(r["outputs"].cpu().numpy(), o["outputs"]), (how, tag)
            assert np.array_equal(r["heads"].cpu().numpy()[o["mask"]], o["heads"][o["mask"]]), (how, tag)
            np.testing.assert_allclose(r["T"], o["T"], rtol=0, atol=1e-12, err_msg=str((how, tag)))
            if o["mask"].any():
                np.testing.assert_allclose(r["up"], o["up"], atol=1e-12, err_msg=str((how, tag)))
                np.testing.assert_allclose(r["scale"], o["scale"], rtol=1e-6, err_msg=str((how, tag)))

        with torch.no_grad():
            r = estimate_pose(enc, T.t(ob["pc"], dev), T.t(ob["normals"], dev), T.t(ob["feat"], dev), T.t(idx, dev),
                              T.t(u_tr, dev), T.t(u_rot, dev), cfg, sph, pc_host=ob["pc"])
        check(r, "eager")
        n_cases += 1
        if n_cases % 4 == 0:
            corners, dims = grid_shape(ob["pc"], cfg.res)
            pp = PosePipeline(enc, cfg, n, P, dims, dev, sph, 72)
            pp.load(ob["pc"], ob["normals"], ob["feat"], idx, u_tr, u_rot, corners[0].copy())
            pp.run()
            check(pp.run(), "graph")          # a replay after the first run: the tail's accumulators start from the last run's values
            pp.release()
            n_graph += 1
    return n_cases, n_graph


if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("pose soak ok: %d cases (eager), %d of them also through a captured pipeline" % run(seconds, s))
