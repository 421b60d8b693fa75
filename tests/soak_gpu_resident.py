"""Randomised soak of round 6's device-resident batches against the oracle: batches of 1..20 objects of random categories, cloud
sizes, pair counts and weights (few ... many survivors) through BatchPoseRunner -- objects put() on the device, or host arrays
through the packed upload --, random lanes / chain lengths, several batches per runner (new objects in old pipelines, captured
replays, adapted forms).  Every record (assembled on the device) against oracle.estimate_pose on the pairs the device drew
(re-drawn on the host by cppf_amd.synthetic.philox_pairs): arg-max and survivor count bit for bit, T / up / right 1e-9, scale 1e-6.
Run by hand on a GPU box:  python tests/soak_gpu_resident.py [seconds] [seed]   (not collected by pytest)."""
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
    from cppf_amd.batch import BatchPoseRunner
    from oracle import oracle as O
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    sph = np.load(os.path.join(HERE, "golden", "sphere.npz"))["pts"]
    cats = ["bottle", "bowl", "camera", "can", "laptop", "mug"]
    t_end = time.time() + seconds
    n_runners = n_batches = n_objects = n_ties = 0
    while time.time() < t_end:
        cs = int(rng.integers(0, 2**31 - 1))
        sds, encs = {}, {}
        for c in cats:
            sd = T.seeded_sd((cs + len(c)) % 1000)
            gain = float(rng.choice([1.0, 4.0, 12.0]))
            for key in ("final.weight", "final.bias"):
                sd[key] = sd[key] * gain
            sds[c], encs[c] = sd, T.make_encoder(sd, [84, 32, 32, 16], 141, dev)
        lanes = int(rng.integers(1, 5))
        chain_len = [None, None, 1, 2, 4, 8][int(rng.integers(0, 6))]
        runner = BatchPoseRunner(encs, dev, n_lanes=lanes, chain_len=chain_len, n_bucket=int(rng.choice([256, 1024])))
        n_pairs = int(rng.choice([3000, 12000, 40000]))              # (one pair count per runner: pipelines are keyed by it)
        n_runners += 1
        for b in range(int(rng.integers(2, 5))):
            objs = []
            for j in range(int(rng.integers(1, 21))):
                cat = cats[int(rng.integers(0, len(cats)))]
                n = int(rng.choice([64, 200, 512, 777, 1024, 1500]))
                ob = syn.make_object(cat, n, (cs + 7919 * j + 104729 * b) % 100000)
                objs.append(dict(pc=ob["pc"], normals=ob["normals"], feat=ob["feat"], cfg=ob["cfg"], n_pairs=n_pairs))
            seed_b = int(rng.integers(0, 1 << 30))
            resident = bool(rng.integers(0, 2))
            batch = runner.put(objs) if resident else objs
            recs = runner.run(batch, seed=seed_b).cpu().numpy()
            if rng.integers(0, 2):                                    # ... and the same batch again: captured replays, adapted forms
                again = runner.run(batch, seed=seed_b).cpu().numpy()
                assert np.array_equal(recs, again), (cs, b, "replay differs")
            assert recs.shape == (len(objs), 20) and recs[:, 15].tolist() == list(range(len(objs)))
            for j, obj in enumerate(objs):
                cfg, tag = obj["cfg"], (cs, b, j, resident, lanes, chain_len)
                idx, u_tr, u_rot = syn.philox_pairs(seed_b * 1000003 + j, n_pairs, obj["pc"].shape[0])
                ocfg = dict(res=cfg.res, tr_num_bins=32, rot_num_bins=36, vote_range=cfg.vote_range, scale_mean=cfg.scale_mean,
                            regress_right=cfg.regress_right, ppffcs=[84, 32, 32, 16], out_dim=141)
                o = O.estimate_pose(obj["pc"], obj["normals"], obj["feat"], idx, sds[cfg.category], ocfg, u_tr, u_rot, sph)
                r = recs[j]
                if int(r[12]) != o["argmax"]:      # a tie of the reference's own fp32 summation order: the device must hold the exact arg-max
                    g64, _ = O.ppf_voting_f64(obj["pc"], o["outputs"], np.ones(obj["pc"].shape[0], np.float32), idx.astype(np.int32),
                                              o["dims"], o["corner"], cfg.res, 72, True)
                    assert int(r[12]) == int(np.argmax(g64)), tag
                    n_ties += 1
                    continue
                assert int(r[14]) == int(o["mask"].sum()), tag
                np.testing.assert_allclose(r[0:3], o["T"], rtol=0, atol=1e-12, err_msg=str(tag))
                if o["mask"].any():
                    np.testing.assert_allclose(r[3:6], o["up"], atol=1e-9, err_msg=str(tag))
                    np.testing.assert_allclose(r[9:12], o["scale"], rtol=1e-6, err_msg=str(tag))
                    if cfg.regress_right:
                        rr = o["right"] - np.dot(o["up"], o["right"]) * o["up"]
                        nr = np.linalg.norm(rr)
                        if nr > 1e-6:
                            np.testing.assert_allclose(r[6:9], rr / (nr + 1e-9), atol=1e-9, err_msg=str(tag))
                n_objects += 1
            n_batches += 1
        del runner
    return n_runners, n_batches, n_objects, n_ties


if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("resident soak ok: %d runners, %d batches, %d records equal to the oracle's poses (%d fp32-order ties resolved by the exact sum)"
          % run(seconds, s))
