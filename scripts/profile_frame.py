"""Development aid: the demo depth frame through FrameRunner, host sections timed (run under rocprofv3 --kernel-trace --stats for
the device side)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cppf_amd import training                      # noqa: E402
from cppf_amd.config import CATEGORIES             # noqa: E402
from cppf_amd.frames import FrameRunner, frame_poses  # noqa: E402
from cppf_amd.utils.util import read_depth_png     # noqa: E402
from test_real_frame import DEPTH, GOLDEN, instances  # noqa: E402

dev = torch.device("cuda", 0)
depth = read_depth_png(DEPTH)
inst = instances(depth)
nets = {}
for cat, src in (("mug", "mug"), ("laptop", "laptop"), ("bowl", "bottle"), ("can", "bottle")):
    penc, enc = training.load_weights(os.path.join(GOLDEN, f"trained_{src}.npz"), CATEGORIES[src], dev)
    nets[cat] = (enc, penc)
encs = {c: v[0] for c, v in nets.items()}
pencs = {c: v[1] for c, v in nets.items()}
cl = os.environ.get("CHAIN_LEN")
runner = FrameRunner(encs, pencs, dev, n_lanes=int(os.environ.get("LANES", "3")), chain_len=int(cl) if cl else None,
                     batch_prestage=not os.environ.get("NO_BATCH_PRE"))
for _ in range(5):
    runner.run(depth, inst)
torch.cuda.synchronize()
n = int(os.environ.get("REPS", "20"))
t0 = time.perf_counter()
for _ in range(n):
    runner.run(depth, inst)
torch.cuda.synchronize()
print("instances_processed %d" % ((5 + n) * len(inst)))
print("FrameRunner ms per frame", (time.perf_counter() - t0) / n * 1e3, "chains", len(runner._chains),
      [m["key"] for m in runner._members.values()], runner.last)
if os.environ.get("PIPELINED"):
    t0 = time.perf_counter()
    prev = None
    for _ in range(n):
        cur = runner.submit(depth, inst)
        if prev is not None:
            prev.result()
        prev = cur
    prev.result()
    torch.cuda.synchronize()
    print("FrameRunner pipelined ms per frame", (time.perf_counter() - t0) / n * 1e3)
if os.environ.get("CPROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        runner.run(depth, inst)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
if os.environ.get("EAGER"):          # (off under rocprofv3: its launches would be summed into the per-instance kernel times)
    t0 = time.perf_counter()
    for _ in range(5):
        frame_poses(depth, inst, encs, pencs, device=dev)
    torch.cuda.synchronize()
    print("eager ms per frame", (time.perf_counter() - t0) / 5 * 1e3)
