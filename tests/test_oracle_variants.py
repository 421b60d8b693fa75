"""oracle/voting_variants.c: (1) its variant 0 IS the oracle (bit for bit), (2) the members of the arithmetic family the
reference's own toolchain may pick (NVRTC --fmad=true contraction, device libm within 1-2 ulp) change no trip count, almost no
in-grid / floor-cell decision, and neither the arg-max of the vote grid nor the pose.  CPU only; the same study at the
BASELINE.json sizes is profiles/r3_fmad_sensitivity.txt (tests/fmad_sensitivity.py --full) and, against the HIP path,
tests/test_gpu_variants.py."""
import numpy as np
import pytest

import cppf_amd.synthetic as syn
import fmad_sensitivity as FS


def test_variant_zero_is_the_oracle(oracle):
    O = oracle
    for cat, seed in (("bottle", 0), ("camera", 5)):
        ob, idx, outputs, heads, ocfg = FS.make_case(cat, 512, 16, seed, "ka")
        idx[::97, 1] = idx[::97, 0]                      # degenerate pairs (a == b): early return at voting.py:21/:87/:131
        outputs[::53, 1] = 1e-5                          # zero adaptive rotations
        idx32 = idx.astype(np.int32)
        corner, dims = O.grid_setup(ob["pc"], ocfg["res"])
        g64, _ = O.ppf_voting_f64(ob["pc"], outputs, np.ones(512, np.float32), idx32, dims, corner, ocfg["res"], 72, True)
        gv = O.ppf_voting_variant(ob["pc"], outputs, np.ones(512, np.float32), idx32, dims, corner, ocfg["res"], 72, True, 0,
                                  threads=1)
        np.testing.assert_array_equal(gv, g64)
        gv4 = O.ppf_voting_variant(ob["pc"], outputs, np.ones(512, np.float32), idx32, dims, corner, ocfg["res"], 72, True, 0,
                                   threads=4)
        np.testing.assert_allclose(gv4, g64, rtol=1e-13, atol=1e-12)
        T = O.center_from_argmax(int(np.argmax(g64)), dims, corner, ocfg["res"]).astype(np.float32)
        oo, m = O.backvote(ob["pc"], outputs, idx32, corner, ocfg["res"], 72, dims, T, np.float32(3 * ocfg["res"]))
        oo_v, m_v = O.backvote_variant(ob["pc"], outputs, idx32, corner, ocfg["res"], 72, dims, T, np.float32(3 * ocfg["res"]), 0)
        np.testing.assert_array_equal(oo_v, oo)
        np.testing.assert_array_equal(m_v, m)
        assert m.sum() > 100
        np.testing.assert_array_equal(O.rot_voting_variant(ob["pc"], heads[:2000, 0], idx32[:2000], 72, 0),
                                      O.rot_voting(ob["pc"], heads[:2000, 0], idx32[:2000], 72))
        fl = O.vote_flips(ob["pc"], outputs, idx32, dims, corner, ocfg["res"], 72, True, 0)
        assert fl["in_grid_flips"] == fl["floor_cell_flips"] == fl["trip_count_flips"] == 0 and fl["max_coord_diff_cells"] == 0.0


def test_variants_really_differ(oracle):
    """the study is not vacuous: every variant changes some float result"""
    O = oracle
    ob, idx, outputs, heads, ocfg = FS.make_case("camera", 512, 16, 2, "ka")
    idx32 = idx.astype(np.int32)
    base = O.rot_voting_variant(ob["pc"], heads[:4000, 0], idx32[:4000], 72, 0)
    corner, dims = O.grid_setup(ob["pc"], ocfg["res"])
    for name, v in O.VARIANTS.items():
        assert not np.array_equal(O.rot_voting_variant(ob["pc"], heads[:4000, 0], idx32[:4000], 72, v), base), name
        assert O.vote_flips(ob["pc"], outputs, idx32, dims, corner, ocfg["res"], 72, True, v)["max_coord_diff_cells"] > 0, name


@pytest.mark.parametrize("case", FS.CASES_SMALL, ids=lambda c: f"{c[0]}-{c[4]}")
def test_discrete_outcomes_and_pose_survive_the_toolchain_freedom(oracle, case):
    import io
    res = FS.study(*case, out=io.StringIO())
    for name, r in res.items():
        # by construction (no product feeds an add in either expression): never a flip
        assert r["trip_count_flips"] == 0 and r["degenerate_flips"] == 0, name
        ig = max(r["in_grid"], 1)
        assert r["in_grid_flips"] <= 1e-5 * ig + 2, (name, r)
        assert r["floor_cell_flips"] <= 1e-4 * ig + 2, (name, r)
        assert r["max_coord_diff_cells"] < 1e-3, (name, r)
        # the arg-max is provably the same cell: no cell moved by half the top-1 / top-2 margin
        assert r["argmax_same"], (name, r)
        if r["mode"] == "ka":      # (uniform-bin inputs: a noise grid whose top cells differ by less than one vote)
            assert r["max_grid_diff"] < 0.5 * r["margin"], (name, r)
        assert r["T_diff"] == 0.0
        # back-vote survivors: a band, not equality (the distance test of voting.py:101 sits on a continuous quantity)
        assert abs(r["n_surv"] - r["n_surv0"]) <= max(3, 2e-3 * r["n_surv0"]) and r["mask_flips"] <= max(4, 4e-3 * r["n_surv0"]), (name, r)
        if r["mode"] == "ka":      # (uniform-bin inputs carry random orientation heads: the 480 sphere counts have no peak to keep)
            assert r["up_diff_mod_sym"] == 0.0, (name, r)       # same sphere bin (or its antipode for up/down-symmetric categories)
        assert r["scale_rel"] <= 1e-4, (name, r)        # the north star's float tolerance
