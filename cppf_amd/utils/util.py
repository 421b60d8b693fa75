"""The one utility of the reference's utils/util.py that sits on the hot path."""
import numpy as np


def fibonacci_sphere(samples):
    """Sphere bins of the orientation vote (reference utils/util.py:102-118, used at
    nocs/inference.py:100-102): point i has y = 1 - 2i/(samples-1) and azimuth i times the golden
    angle.  Returns a list of (x, y, z) tuples in fp64 like the reference (callers do np.array(..))."""
    i = np.arange(samples, dtype=np.float64)
    y = 1.0 - (i / float(samples - 1)) * 2.0
    radius = np.sqrt(1.0 - y * y)
    theta = (np.pi * (3.0 - np.sqrt(5.0))) * i
    pts = np.stack([np.cos(theta) * radius, y, np.sin(theta) * radius], -1)
    return [tuple(float(v) for v in p) for p in pts]


def num_sphere_bins(angle_tol_deg):
    """nocs/inference.py:100-101"""
    return int(4 * np.pi / (angle_tol_deg / 180 * np.pi))
