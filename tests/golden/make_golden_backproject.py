#!/usr/bin/env python3
"""tests/golden/backproject.npz from the reference's own utils/util.py:backproject (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_backproject.py /root/reference

utils/util.py imports open3d / cv2 / ... at module level (absent here), so -- like make_golden.py does for
fibonacci_sphere -- the FunctionDef of `backproject` is extracted with `ast` and exec'd with the real numpy: the reference's
own code runs, no stand-in modules.  Inputs: a synthetic 60 x 80 uint16 depth image with holes, a float32 copy of it, two
instance masks, and the camera matrix of nocs/inference.py:98.  Only data is written."""
import ast
import os
import sys

import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out_dir = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
path = os.path.join(ref, "utils", "util.py")
env = {"np": np}
for node in ast.parse(open(path).read()).body:
    if isinstance(node, ast.FunctionDef) and node.name == "backproject":
        exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), env)
backproject = env["backproject"]

rng = np.random.default_rng(2024)
H, W = 60, 80
yy, xx = np.mgrid[0:H, 0:W]
depth = (900 + 4 * xx + 3 * yy + rng.integers(0, 40, (H, W))).astype(np.uint16)
depth[rng.random((H, W)) < 0.15] = 0                                   # holes
intrinsics = np.array([[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]])   # nocs/inference.py:98
masks = np.stack([(xx - 40) ** 2 + (yy - 30) ** 2 < 400, rng.random((H, W)) < 0.3], 0)
out = dict(depth=depth, intrinsics=intrinsics, masks=masks)
for k, m in enumerate(masks):
    for tag, d in (("u16", depth), ("f32", depth.astype(np.float32) * np.float32(0.37))):
        pts, idxs = backproject(d, intrinsics, m)
        out[f"pts_{tag}_{k}"] = pts
        out[f"rows_{tag}_{k}"] = idxs[0]
        out[f"cols_{tag}_{k}"] = idxs[1]
np.savez_compressed(os.path.join(out_dir, "backproject.npz"), **out)
print({k: v.shape for k, v in out.items()})
