"""Randomised soak of FrameRunner against the eager loop (frames.frame_poses) on the reference's demo depth frame: random
rectangular "instances" (position, size, depth window, category; overlapping, tiny and empty ones included), a new set every
frame, so that members / chains are created, captured, evicted and replayed with new inputs.  Poses must be equal bit for bit.
Run by hand on a GPU box:  python tests/soak_gpu_frames.py [seconds] [seed]   (not collected by pytest)."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def run(seconds, seed):
    from cppf_amd import training
    from cppf_amd.config import CATEGORIES
    from cppf_amd.frames import FrameRunner, frame_poses
    from cppf_amd.utils.util import read_depth_png
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    depth = read_depth_png(os.path.join(HERE, "golden", "demo_0000_depth.png"))
    nets = {}
    for cat, src in (("mug", "mug"), ("laptop", "laptop"), ("bowl", "bottle"), ("can", "bottle")):
        penc, enc = training.load_weights(os.path.join(HERE, "golden", f"trained_{src}.npz"), CATEGORIES[src], dev)
        nets[cat] = (enc, penc)
    encs = {c: v[0] for c, v in nets.items()}
    pencs = {c: v[1] for c, v in nets.items()}
    runner = FrameRunner(encs, pencs, dev, n_pairs=20000)
    cats = list(nets)
    t_end = time.time() + seconds
    frames = inst_total = captured = skipped = eager = 0
    layouts = []
    while time.time() < t_end:
        if not layouts or rng.random() < 0.5:            # a new layout half of the time, a repeated one otherwise (chains get captured and replayed)
            inst = []
            for _ in range(int(rng.integers(1, 10)) if rng.random() < 0.85 else int(rng.integers(17, 36))):   # (sometimes more than a label image's 32 bits)
                h, w = int(rng.integers(2, 200)), int(rng.integers(2, 200))
                r0, c0 = int(rng.integers(0, 480 - h)), int(rng.integers(0, 640 - w))
                m = np.zeros(depth.shape, bool)
                patch = depth[r0:r0 + h, c0:c0 + w]
                if rng.random() < 0.1 or not (patch > 0).any():
                    pass                                   # an empty mask
                else:
                    med = np.median(patch[patch > 0])
                    m[r0:r0 + h, c0:c0 + w] = np.abs(patch.astype(np.int64) - med) <= int(rng.choice([40, 90, 260]))
                inst.append((cats[int(rng.integers(0, len(cats)))], m))
            layouts.append(inst)
            layouts = layouts[-6:]
        else:
            inst = layouts[int(rng.integers(0, len(layouts)))]
        s = int(rng.integers(0, 1000))
        # a third of the frames go through submit() with one or two more frames (repeats of known layouts) enqueued behind them
        # before the first result is asked for: the pipelined form of a video loop
        batch = [(inst, s)]
        if layouts and rng.random() < 0.33:
            for _ in range(int(rng.integers(1, 3))):
                batch.append((layouts[int(rng.integers(0, len(layouts)))], int(rng.integers(0, 1000))))
            pend = [runner.submit(depth, f, seed=sd) for f, sd in batch]
            gots = [p_.result() for p_ in (reversed(pend) if rng.random() < 0.5 else pend)]
            if len(gots) and gots[0] is not pend[0].result():
                gots = [p_.result() for p_ in pend]
        else:
            gots = [runner.run(depth, inst, seed=s)]
        for (f_inst, sd), got in zip(batch, gots):
            want = frame_poses(depth, f_inst, encs, pencs, n_pairs=20000, seed=sd, device=dev)
            for i, (w, g) in enumerate(zip(want, got)):
                assert (w is None) == (g is None), (frames, i)
                if w is None:
                    continue
                assert g["n_points"] == w["n_points"] and g["argmax"] == w["argmax"] and g["n_surv"] == w["n_surv"], (frames, i, f_inst[i][0])
                for k in ("T", "up", "right", "scale"):
                    assert np.array_equal(g[k], w[k]), (frames, i, k)
            frames += 1
            inst_total += len(f_inst)
        captured += runner.last["captured"]
        skipped += runner.last["skipped"]
        eager += runner.last["eager"]
    return frames, inst_total, captured, eager, skipped, len(runner._chains)


if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("frame soak ok: %d frames, %d instances (%d through captured stages, %d eager fall-backs, %d skipped), %d chains cached" % run(seconds, s))
