"""cppf_amd -- MI355X (gfx950) implementation of CPPF's point-pair-feature + voting hot path.

Layout mirrors the part of the reference (qq456cvb/CPPF) it replaces:
  cppf_amd.models.model   PPFEncoder / ResLayer          <- models/model.py:8-31,80-137
  cppf_amd.models.voting  ppf_kernel / backvote_kernel / rot_voting_kernel  <- models/voting.py
  cppf_amd.utils.util     fibonacci_sphere               <- utils/util.py:102-118
  cppf_amd.inference      estimate_pose (device-resident glue)  <- nocs/inference.py:177-335
  cppf_amd.sharding       one-object-per-GPU sharding + the single RCCL gather
  cppf_amd.csrc           HIP kernels + the C ABI (include/cppf.h) -> libcppf_hip.so
There is no CPU fallback: every compute entry point needs libcppf_hip.so and a HIP device.
"""
__version__ = "0.1.0"
