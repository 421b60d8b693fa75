"""The HIP path against the arithmetic family of the reference's own toolchain (oracle/voting_variants.c; CPU-side study:
tests/test_oracle_variants.py, profiles/r3_fmad_sensitivity.txt).  The kernels reproduce variant 0 bit for bit; here the device's
discrete results are compared with EVERY variant (NVRTC-style fused multiply-adds, another libm, +-1..2 ulp trigonometry) on the
vote fixtures and on the BASELINE.json C2 / C5 inputs: same arg-max cell, same translation, survivors within a band, same
orientation bin, scale within the north star's 1e-4."""
import numpy as np
import pytest
import torch

import fmad_sensitivity as FS
from cppf_amd.config import CATEGORIES
from cppf_amd.inference import PoseWorkspace, _assemble, _enqueue_tail
from cppf_amd.models import voting

pytestmark = pytest.mark.gpu


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def device_pose_from_votes(dev, ob, idx, outputs, heads, cfg, res, sph):
    """centre vote + arg-max + the whole pose tail on the device from given (mu, nu) and heads"""
    import dataclasses
    from cppf_amd.inference import grid_shape
    cfg = dataclasses.replace(cfg, res=res)
    corners, dims = grid_shape(ob["pc"], res)
    P = idx.shape[0]
    ws = PoseWorkspace(dev, P, dims, sph.shape[0])
    sph_d = ws.sphere(np.asarray(sph, np.float64))
    pc, nrm, out_d, heads_d = t(ob["pc"], dev), t(ob["normals"], dev), t(outputs, dev), t(heads, dev)
    idx_d, corner = t(idx, dev), t(corners[0], dev)
    idx32 = idx_d.to(torch.int32)
    voting.vote_argmax(pc, out_d, None, idx_d, ws.grid, corner, res, 72, True, ws.out_idx, ws.out_val, accumulate=False)
    _enqueue_tail(ws, pc, nrm, idx32, out_d, heads_d, corner, cfg, dims, 72, 1.5, 10000, *sph_d)
    r = _assemble(ws.rec.cpu().numpy(), cfg)
    r["mask"] = ws.mask.cpu().numpy().astype(bool)
    r["counts"] = ws.counts.cpu().numpy()
    return r


CASES = FS.CASES_SMALL + FS.CASES_FULL


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-N{c[1]}-K{c[2]}-{c[4]}")
def test_device_results_hold_under_every_variant(oracle, golden, dev, case):
    O = oracle
    cat, N, K, seed, mode, res = case
    ob, idx, outputs, heads, ocfg = FS.make_case(cat, N, K, seed, mode, res)
    cfg = CATEGORIES[cat]
    sph = golden("sphere.npz")["pts"]
    r = device_pose_from_votes(dev, ob, idx, outputs, heads, cfg, ocfg["res"], sph)
    base = O.pose_tail_variant(ob["pc"], ob["normals"], idx, outputs, heads, ocfg, sph, 0)
    # variant 0 is what the kernels implement: everything discrete is equal
    assert r["argmax"] == base["argmax"]
    np.testing.assert_array_equal(r["mask"], base["mask"])
    np.testing.assert_allclose(r["T"], base["T"], atol=1e-12)
    for j, c in enumerate(base["counts"]):
        np.testing.assert_array_equal(r["counts"][j], c)
    np.testing.assert_allclose(r["up"], base["up"], atol=1e-12)
    np.testing.assert_allclose(r["scale"], base["scale"], rtol=1e-6)
    n0 = r["n_surv"]
    small = N <= 1024
    variants = O.VARIANTS if small else {k: O.VARIANTS[k] for k in ("fmad_left+libm", "fmad_right+libm", "fmad_right+ulp_hash2")}
    for name, v in variants.items():
        p = O.pose_tail_variant(ob["pc"], ob["normals"], idx, outputs, heads, ocfg, sph, v)
        assert p["argmax"] == r["argmax"], name                                   # bit-exact vote-grid arg-max index
        if mode == "ka":      # a grid with a real peak: no cell moved by half the top-1 / top-2 margin -> provably the same cell
            assert np.abs(p["grid"] - base["grid"]).max() < 0.5 * base["margin"], name
        np.testing.assert_allclose(p["T"], r["T"], atol=1e-12, err_msg=name)
        assert abs(int(p["mask"].sum()) - n0) <= max(3, 2e-3 * n0), (name, int(p["mask"].sum()), n0)
        assert int((p["mask"] != r["mask"]).sum()) <= max(4, 4e-3 * n0), name
        if mode == "ka":
            d = np.abs(p["up"] - r["up"]).max()
            if cfg.up_sym:
                d = min(d, np.abs(p["up"] + r["up"]).max())
            assert d <= 1e-4, (name, p["up"], r["up"])
        if n0:
            np.testing.assert_allclose(p["scale"], r["scale"], rtol=1e-4, err_msg=name)
