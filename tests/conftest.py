import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def sd_from_npz(g):
    return {k[3:]: g[k] for k in g.files if k.startswith("sd.")}


@pytest.fixture(scope="session")
def c_host(oracle, tmp_path_factory):
    """tests/c_host/vote_host.c -- a host in plain C over include/cppf.h -- compiled with gcc as C99 against the in-tree
    libcppf_hip.so (the product) and liboracle.so (its checker); returns the binary's path"""
    import subprocess
    from cppf_amd import _lib
    _lib.lib()                                                     # (fails loudly when the HIP library was not built)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    csrc, orc = os.path.join(ROOT, "cppf_amd", "csrc"), os.path.join(ROOT, "oracle")
    exe = str(tmp_path_factory.mktemp("c_host") / "vote_host")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(rocm, "include"),
           os.path.join(ROOT, "tests", "c_host", "vote_host.c"), "-o", exe, "-L" + csrc, "-lcppf_hip", "-L" + orc, "-l:liboracle.so",
           "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lm"] + ["-Wl,-rpath," + d for d in (csrc, orc, os.path.join(rocm, "lib"))]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return exe
