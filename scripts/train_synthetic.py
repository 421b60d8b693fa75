#!/usr/bin/env python3
"""Train the two networks of train.py:34-35 on posed synthetic objects with the HIP forward + backward, check the pose of
held-out objects, and save the weights (tests/golden/trained_<category>.npz were produced by this script on one MI355X):

    python scripts/train_synthetic.py --out gpurun_out/trained --categories bottle camera --steps 600
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cppf_amd.synthetic as syn          # noqa: E402
from cppf_amd import training             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/trained")
    ap.add_argument("--categories", nargs="+", default=["bottle", "camera"])
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--n-points", type=int, nargs=2, default=[768, 2048], help="cloud size range, drawn per step")
    ap.add_argument("--n-pairs", type=int, default=60000)
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--held-out", type=int, default=8)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    dev = torch.device("cuda", 0)
    report = {}
    for cat in args.categories:
        t0 = time.perf_counter()
        penc, enc, losses = training.train(cat, dev, args.steps, tuple(args.n_points), args.n_pairs, args.lr, log=print)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        errs = []
        for j in range(args.held_out):
            ob = syn.make_posed_object(cat, 1536, 900000 + j)               # seeds no training step saw
            pose = training.infer(penc, enc, ob, dev, seed=j)
            e = training.pose_errors(pose, ob)
            e["n_surv"] = pose["n_surv"]
            errs.append(e)
            print(cat, "held-out", j, json.dumps(e))
        training.save_weights(os.path.join(args.out, f"trained_{cat}.npz"), penc, enc,
                              meta=dict(steps=args.steps, n_points=np.array(args.n_points), n_pairs=args.n_pairs, lr=args.lr,
                                        final_loss=losses[-1]))
        report[cat] = dict(train_seconds=dt, losses=losses, held_out=errs)
    with open(os.path.join(args.out, "train_report.json"), "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
