"""Development aid: the reference-default batch (8 posed instances, N 700-2000, 100 000 pairs each, kNN + SPRIN + full pose per
instance: bench.py's `dropin_flow_reference_defaults.level3`) through BatchPoseRunner; run under rocprofv3 --kernel-trace --stats
for the per-kernel times.  MODE=c4: 8 mixed-category C2-size objects instead (bench.py's `c4_one_gpu_share`)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                       # noqa: E402
import cppf_amd.synthetic as syn                   # noqa: E402
from cppf_amd import training                      # noqa: E402
from cppf_amd.batch import BatchPoseRunner         # noqa: E402
from cppf_amd.config import NOCS_CATEGORIES        # noqa: E402
from cppf_amd.models.model import PPFEncoder       # noqa: E402

dev = torch.device("cuda", 0)
mode = os.environ.get("MODE", "level3")
lanes = int(os.environ.get("LANES", "3"))
cl = os.environ.get("CHAIN_LEN")
cl = int(cl) if cl else None
if mode == "level3":
    cats = ["bottle", "mug", "laptop"]
    nets = {c: training.load_weights(bench.TRAINED_WEIGHTS.format(c), syn.CATEGORIES[c], dev) for c in cats}
    sizes = (717, 1203, 1890, 960, 1544, 2011, 1333, 1777)
    robjs = [syn.make_posed_object(cats[j % 3], n_j, 910000 + j) for j, n_j in enumerate(sizes)]
    batch = [dict(pc=o["pc"], normals=o["normals"], cfg=o["cfg"], n_pairs=100000) for o in robjs]
    runner = BatchPoseRunner({c: nets[c][1] for c in cats}, dev, point_encoders={c: nets[c][0] for c in cats}, n_lanes=lanes, chain_len=cl,
                             overlap_batches=bool(os.environ.get("OVERLAP")), idx_i32=not os.environ.get("IDX64"))
else:
    encs = {}
    for i, c in enumerate(NOCS_CATEGORIES):
        torch.manual_seed(i)
        cfg = syn.make_object(c, 8, 0)["cfg"]
        encs[c] = PPFEncoder(cfg.ppffcs, cfg.out_dim).eval().to(dev)
    batch = bench.c4_objects(int(os.environ.get("OBJECTS", "8")), 4096, 128)
    runner = BatchPoseRunner(encs, dev, n_lanes=lanes, chain_len=cl, overlap_batches=bool(os.environ.get("OVERLAP")), idx_i32=not os.environ.get("IDX64"))
if os.environ.get("RESIDENT"):
    batch = runner.put(batch)
for _ in range(8):
    runner.run(batch)
bench.settle()
n = int(os.environ.get("REPS", "40"))
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n // 5):
        runner.run(batch)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / (n // 5) / len(batch) * 1e3)
print("instances_processed %d" % ((8 + 5 * (n // 5)) * len(batch)))
print(("overlap " if os.environ.get("OVERLAP") else "") + ("resident " if os.environ.get("RESIDENT") else "") + "%s lanes %d chain_len %s: ms per instance median %.4f [%.4f, %.4f]" % (mode, lanes, cl, sorted(ts)[2], min(ts), max(ts)))
