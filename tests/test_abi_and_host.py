"""CPU-side checks: the C-ABI library loads and exports every symbol include/cppf.h declares, host
logic (packing, planning, argument validation, drop-in module surface) -- no device compute."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, sd_from_npz


def test_library_loads_and_exports_every_declared_symbol():
    from cppf_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "cppf.h")).read()
    hdr = re.sub(r"#ifdef CPPF_DEBUG_ENTRY.*?#endif", "", hdr, flags=re.S)       # profiling aids are not in the shipped library
    declared = sorted(set(re.findall(r"\b(cppf_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/cppf.h but not exported"
    assert set(declared) == set(_lib.exported_symbols())
    assert not hasattr(L, "cppf_debug_mlp_chain_only")
    assert L.cppf_abi_version() == 4 == _lib.ABI_VERSION and "#define CPPF_ABI_VERSION 4" in hdr
    assert b"workspace" in L.cppf_error_string(-2)


def test_workspace_queries_and_argument_errors_without_a_device():
    from cppf_amd import _lib
    L = _lib.lib()
    # bottle grid of BASELINE config 2 on the binned path: two tiles of 26 x 38 x 26 owned cells (+ one halo row), 128 chunks of
    # raw partial tiles, and a queue of 12-byte records per tile with room for every pair
    need = L.cppf_vote_workspace_bytes(524288, 72, 26, 76, 26)
    slot = 26 * 39 * 26
    lo = 256 * slot * 4 + 64 * 30720 * 8      # fewer than 4 tiles: no queues; + the extra plane, sized for any tiled grid
    assert lo <= need < lo + (1 << 17)
    assert 65536 < L.cppf_vote_workspace_init_bytes() <= need
    # n_rots > 72 runs the same kernels in ceil(n_rots / 72) passes: the same workspace (round 2's kernels are gone)
    assert L.cppf_vote_workspace_bytes(524288, 100, 26, 76, 26) == need == L.cppf_vote_workspace_bytes(524288, 360, 26, 76, 26)
    plan = (C.c_int32 * 10)()
    assert L.cppf_vote_plan_query(524288, 144, 26, 76, 26, plan) == 0 and plan[0] == 2 and plan[1] == 2      # fused, two tiles
    assert L.cppf_vote_plan_query(2097152, 144, 52, 152, 52, plan) == 0 and plan[0] == 3 and plan[1] == 16   # binned
    assert L.cppf_vote_plan_query(1000, 72, 400, 400, 400, plan) == 0 and plan[0] == 0                      # global atomics
    assert L.cppf_vote_workspace_bytes(100, 0, 26, 76, 26) == 0          # n_rots out of range
    assert L.cppf_vote_workspace_bytes(100, 361, 26, 76, 26) == 0
    # huge grid -> global-atomic path: the arg-max keys only, but never less than the state a tiled call would want zeroed
    assert L.cppf_vote_workspace_bytes(524288, 72, 400, 400, 400) == L.cppf_vote_workspace_init_bytes()
    assert L.cppf_compact_workspace_bytes(0) > 0 and L.cppf_reduce_workspace_bytes() > 0
    # null pointers / bad sizes are rejected before any HIP call
    assert L.cppf_ppf_voting(None, None, None, None, None, None, 0.004, 4, 10, 72, 4, 4, 4, 1, None, 0, None) == -1
    assert L.cppf_backvote(None, None, None, None, None, 0.004, 10, 72, 4, 4, 4, None, 0.01, None, None) == -1
    assert L.cppf_grid_argmax(None, 10, None, None, None, 0, None) == -1
    assert L.cppf_rot_voting(None, None, None, None, 10, 72, None) == -1
    assert L.cppf_stage_batch(0, None, None) == -1 and L.cppf_stage_batch(1, None, None) == -1
    # empty pair lists are legal no-ops (ragged inputs), with null pointers
    assert L.cppf_backvote(None, None, None, None, None, 0.004, 0, 72, 4, 4, 4, None, 0.01, None, None) == 0
    assert L.cppf_rot_voting(None, None, None, None, 0, 72, None) == 0


def test_weight_packing_follows_the_documented_lane_order(golden):
    from cppf_amd import _lib
    from cppf_amd.models.model import flatten_state_dict
    g = golden("mlp_141.npz")
    sd = sd_from_npz(g)
    params, offs = flatten_state_dict(sd, [84, 32, 32, 16])
    L = _lib.lib()
    dims = (C.c_int * 4)(84, 32, 32, 16)
    n = L.cppf_pair_mlp_packed_floats(40, dims, 3, 141)
    assert n == 23024                                                    # forward 13 152 + decode-order final 3 216 + transposed (backward) 6 656
    packed = np.zeros(n, np.float32)
    assert L.cppf_pair_mlp_pack(params.ctypes.data, offs.ctypes.data, 40, dims, 3, 141, packed.ctypes.data) == 0
    w1, w0 = sd["res_layers.0.fc1.weight"], sd["res_layers.0.fc0.weight"]
    # layer 0: one PPF k-step [64 lanes][4 blocks] at the head, the per-point projection [40][128] + bias at the tail
    for lane, ob in ((0, 0), (17, 1), (40, 2), (63, 3)):
        o, k = 16 * (ob & 1) + (lane & 15), 80 + (lane >> 4)
        assert packed[lane * 4 + ob] == (w1 if ob < 2 else w0)[o, k]
    lds_floats = 64 * 4 + 4 * (8 * 64 * 2) + 4 * 64 + 4 * 64 * 12 + 32 * 4 + 16 + 144
    wpt = packed[lds_floats:lds_floats + 40 * 128].reshape(40, 128)
    np.testing.assert_array_equal(wpt[:, :32], w1[:, :40].T)
    np.testing.assert_array_equal(wpt[:, 32:64], w0[:, :40].T)
    np.testing.assert_array_equal(wpt[:, 64:96], w1[:, 40:80].T)
    np.testing.assert_array_equal(wpt[:, 96:], w0[:, 40:80].T)
    np.testing.assert_array_equal(packed[lds_floats + 5120:lds_floats + 5120 + 64],
                                  np.concatenate([sd["res_layers.0.fc1.bias"], sd["res_layers.0.fc0.bias"]]))
    assert L.cppf_pair_mlp_workspace_bytes(4096, 40, dims, 3, 141) == 4096 * 128 * 4
    # final layer: [4][64][12], zero padded past out_dim; biases in natural order at the tail
    off_wf = 64 * 4 + 4 * (8 * 64 * 2) + 4 * 64
    wf = sd["final.weight"]
    for s, lane, ob in ((0, 0, 0), (3, 63, 8), (2, 20, 5)):
        o, k = 16 * ob + (lane & 15), 16 * (s // 4) + 4 * (lane >> 4) + s % 4
        exp = wf[o, k] if o < 141 else 0.0
        assert packed[off_wf + (s * 64 + lane) * 12 + ob] == exp
    np.testing.assert_array_equal(packed[lds_floats - 144:lds_floats - 3], sd["final.bias"])
    assert np.all(packed[lds_floats - 3:lds_floats] == 0)
    # decode-order copy of the final layer (dec_col in csrc/pair_mlp.hip): a lane group owns consecutive bins
    off_wfd = 13152
    off_bfd = off_wfd + 4 * 64 * 12

    def dec_col(ob, g, r):
        if ob < 4:
            return 32 * (ob >> 1) + 8 * g + 4 * (ob & 1) + r
        if ob < 8:
            return 64 + 36 * ((ob - 4) >> 1) + 9 * g + 4 * (ob & 1) + r
        if r < 2:
            return 64 + 36 * r + 9 * g + 8
        q = 2 * g + (r - 2)
        return 136 + q if q < 5 else -1
    cols = sorted(c for ob in range(9) for g_ in range(4) for r in range(4) if (c := dec_col(ob, g_, r)) >= 0)
    assert cols == list(range(141))                                       # a permutation of the 141 columns
    for s, lane, ob in ((0, 0, 0), (3, 63, 8), (1, 22, 5), (2, 40, 8), (0, 15, 7)):
        m, k = lane & 15, 16 * (s // 4) + 4 * (lane >> 4) + s % 4
        c = dec_col(ob, m >> 2, m & 3)
        assert packed[off_wfd + (s * 64 + lane) * 12 + ob] == (wf[c, k] if c >= 0 else 0.0)
        assert packed[off_bfd + 16 * ob + m] == (sd["final.bias"][c] if c >= 0 else 0.0)
    # backward section (csrc/pair_layout.h): TRANSPOSED weights as MFMA A operands, A[input row][k = output khid(s, g)]
    off_t0b = off_bfd + 144
    khid = lambda s, g_: 16 * (s // 4) + 4 * g_ + s % 4
    t32 = packed[off_t0b:off_t0b + 3 * 1024].reshape(3, 8, 64, 2)           # layer 0 fc2^T, layer 1 fc1^T, layer 1 fc2^T
    for which, name in enumerate(("res_layers.0.fc2.weight", "res_layers.1.fc1.weight", "res_layers.1.fc2.weight")):
        for s, lane, ib in ((0, 0, 0), (7, 63, 1), (3, 37, 0)):
            assert t32[which, s, lane, ib] == sd[name][khid(s, lane >> 4), 16 * ib + (lane & 15)]
    t2 = packed[off_t0b + 3072:off_t0b + 4096].reshape(4, 64, 4)            # layer 2: fc0^T ib 0,1 | fc1^T ib 0,1
    t2b = packed[off_t0b + 4096:off_t0b + 4352].reshape(4, 64)
    tf = packed[off_t0b + 4352:off_t0b + 4352 + 36 * 64].reshape(36, 64)
    for s, lane in ((0, 0), (3, 63), (2, 21)):
        k, m = khid(s, lane >> 4), lane & 15
        assert t2[s, lane, 0] == sd["res_layers.2.fc0.weight"][k, m] and t2[s, lane, 1] == sd["res_layers.2.fc0.weight"][k, 16 + m]
        assert t2[s, lane, 2] == sd["res_layers.2.fc1.weight"][k, m] and t2[s, lane, 3] == sd["res_layers.2.fc1.weight"][k, 16 + m]
        assert t2b[s, lane] == sd["res_layers.2.fc2.weight"][k, m]
    for s, lane in ((0, 0), (35, 63), (35, 5), (17, 40)):
        k = khid(s, lane >> 4)
        assert tf[s, lane] == (wf[k, lane & 15] if k < 141 else 0.0)
    assert off_t0b + 4352 + 36 * 64 == n
    # unsupported: layer wider than 128
    dims_bad = (C.c_int * 3)(84, 256, 16)
    assert L.cppf_pair_mlp_packed_floats(40, dims_bad, 2, 10) == 0
    # generic architecture: canonical concatenation
    gg = golden("mlp_generic.npz")
    sdg = sd_from_npz(gg)
    pg, og = flatten_state_dict(sdg, [44, 24, 24])
    dg = (C.c_int * 3)(44, 24, 24)
    ng = L.cppf_pair_mlp_packed_floats(20, dg, 2, 10)
    assert ng == pg.size
    out = np.zeros(ng, np.float32)
    assert L.cppf_pair_mlp_pack(pg.ctypes.data, og.ctypes.data, 20, dg, 2, 10, out.ctypes.data) == 0
    np.testing.assert_array_equal(out, pg)


def test_flatten_state_dict_matches_oracle_packing(oracle, golden):
    from cppf_amd.models.model import flatten_state_dict
    sd = sd_from_npz(golden("mlp_141.npz"))
    p1, o1 = flatten_state_dict(sd, [84, 32, 32, 16])
    p2, o2 = oracle.pack_params(sd, [84, 32, 32, 16])
    np.testing.assert_array_equal(p1, p2)
    np.testing.assert_array_equal(o1, o2)
    assert o1[10] == -1 and o1[4] >= 0 and o1[16] >= 0        # fc0 only where dim_in != dim_out


def test_ppfencoder_module_surface_matches_reference(golden):
    from cppf_amd.models.model import PPFEncoder, ResLayer
    g = golden("mlp_141.npz")
    sd = sd_from_npz(g)
    enc = PPFEncoder([84, 32, 32, 16], 141)
    assert sorted(enc.state_dict().keys()) == sorted(sd.keys())            # reference checkpoints load as-is
    assert sum(p.numel() for p in enc.parameters()) == 12333
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    assert isinstance(enc.res_layers[0], ResLayer) and enc.res_layers[1].fc0 is None
    # autograd path (train.py:66,91): composite torch ops, equals the reference logits, gives grads
    feat = torch.from_numpy(g["feat"]).requires_grad_(True)
    y = enc(torch.from_numpy(g["pc"])[None], torch.from_numpy(g["nrm"])[None], feat[None],
            idxs=torch.from_numpy(g["idxs"]))
    assert y.shape == (1, 512, 141)
    np.testing.assert_allclose(y[0].detach().numpy(), g["logits"], atol=2e-6)
    y.sum().backward()
    assert feat.grad is not None and enc.final.weight.grad is not None
    # inference path never falls back to the CPU
    from cppf_amd import _lib
    with torch.no_grad(), pytest.raises(_lib.CppfError):
        enc(torch.from_numpy(g["pc"])[None], torch.from_numpy(g["nrm"])[None], torch.from_numpy(g["feat"])[None],
            idxs=g["idxs"])


def test_voting_module_surface_and_loud_failure_without_device():
    from cppf_amd import _lib
    from cppf_amd.models import voting
    assert {"ppf_kernel", "backvote_kernel", "rot_voting_kernel"} <= set(dir(voting))
    assert not hasattr(voting, "findpeak_kernel")
    with pytest.raises(TypeError):
        voting.ppf_kernel((1, 1, 1), (512, 1, 1), (1, 2, 3))
    if not torch.cuda.is_available():
        args = (torch.zeros(4, 3), torch.zeros(8, 2), torch.ones(4), torch.zeros((8, 2), dtype=torch.int32),
                torch.zeros(4, 4, 4), torch.zeros(3), 0.004, 8, 72, 4, 4, 4, True)
        with pytest.raises(_lib.CppfError):
            voting.ppf_kernel((1, 1, 1), (512, 1, 1), args)
    # the flags word of cppf_vote_argmax / _dyn (cppf.h: CPPF_VOTE_ACCUMULATE, CPPF_VOTE_WORKGROUPS)
    hdr = open(os.path.join(ROOT, "include", "cppf.h")).read()
    assert "#define CPPF_VOTE_ACCUMULATE 1" in hdr and "#define CPPF_VOTE_WORKGROUPS(n) (((n) & 0x1ff) << 8)" in hdr
    assert voting._vote_flags(True, 0) == 1 and voting._vote_flags(False, 128) == 128 << 8
    assert voting._vote_flags(True, 256) == (256 << 8) | 1
    for bad in (1, 63, 257):
        with pytest.raises(ValueError):
            voting._vote_flags(False, bad)


def test_host_utilities(golden):
    from cppf_amd.utils.util import fibonacci_sphere, num_sphere_bins
    s = golden("sphere.npz")
    assert num_sphere_bins(1.5) == 480
    pts = fibonacci_sphere(480)
    assert isinstance(pts, list) and len(pts[0]) == 3
    np.testing.assert_array_equal(np.array(pts), s["pts"])
    import cppf_amd.synthetic as syn
    a, b = syn.make_object("bottle", 256, 3), syn.make_object("bottle", 256, 3)
    np.testing.assert_array_equal(a["pc"], b["pc"])
    assert a["pc"].dtype == np.float32 and a["feat"].shape == (256, 40)
    np.testing.assert_allclose(np.linalg.norm(a["normals"], axis=-1), 1.0, atol=1e-6)
    idx = syn.make_pairs(256, 4, 3)
    assert idx.shape == (1024, 2) and idx.dtype == np.int64 and idx.max() < 256
    from cppf_amd.config import CATEGORIES
    assert CATEGORIES["bottle"].out_dim == 141 and CATEGORIES["laptop"].res == 1e-2


def test_missing_library_is_a_loud_error(monkeypatch, tmp_path):
    from cppf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_SO", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.CppfError):
        _lib.lib()


def test_point_encoder_and_backward_abi_without_a_device(golden):
    """Rows f1/f2: workspace queries, argument validation (no HIP call is reached), packing, module surface."""
    from cppf_amd import _lib
    from cppf_amd.models.model import PointEncoder
    from cppf_amd.models.sprin import pack_point_encoder
    L = _lib.lib()
    hid = (C.c_int * 4)(32, 64, 32, 32)
    assert L.cppf_point_encoder_packed_floats(hid, 4, 32, 2, 32, 8, 1) == 9256 + 6912   # parameters of train.py:34 + MFMA image
    assert L.cppf_point_encoder_packed_floats((C.c_int * 2)(16, 24), 2, 32, 2, 32, 8, 1) == 3808        # other shapes: natural block only
    # 256 spare bytes (+ a [N, 40] ping buffer when layers alternate) + 32 channel maxima per conv workgroup of >= 4 points
    assert L.cppf_point_encoder_workspace_bytes(4096, 32, 8, 1) == 256 + 1024 * 32 * 4
    assert L.cppf_point_encoder_workspace_bytes(4096, 32, 8, 2) == 256 + 4096 * 40 * 4 + 1024 * 32 * 4
    assert L.cppf_knn(None, None, 10, 3, None, None) == -1                               # no points, no matrix
    assert L.cppf_knn(None, None, 0, 3, None, None) == 0                                 # empty cloud: legal no-op
    assert L.cppf_knn(None, None, 10, 0, None, None) == -1
    bad = (C.c_int * 2)(16, 24)
    one = C.c_void_p(8)                                                                  # non-null dummy, never dereferenced
    assert L.cppf_point_encoder_forward(one, one, one, 10, 3, one, bad, 2, 32, 2, 32, 8, 1, one, one, 1 << 20, None) == -3
    assert L.cppf_point_encoder_forward(one, one, one, 10, 3, one, hid, 4, 32, 2, 32, 8, 1, one, None, 0, None) == -2
    assert L.cppf_point_encoder_forward(one, one, one, 100, 65, one, hid, 4, 32, 2, 32, 8, 1, one, one, 1 << 20, None) == -3
    dims = (C.c_int * 4)(84, 32, 32, 16)
    # weight image + point table + partial gradients + per-pair rows [d(h0) | d(x1)] + sort keys/values + per-point sums + scratch
    need = L.cppf_pair_mlp_backward_workspace_bytes(200000, 4096, 40, dims, 3, 141)
    parts = 447                                                           # 3 125 tiles, seven per workgroup
    fixed = 23024 * 4 + 4096 * 128 * 4 * 2 + parts * 12333 * 4 + 200000 * 64 * 4 + 8 * 200000 * 4
    assert fixed <= need < fixed + 110 * 200000
    assert L.cppf_pair_mlp_backward_workspace_bytes(130, 64, 40, dims, 3, 141) >= 3 * 12333 * 4 + 130 * 64 * 4
    other = (C.c_int * 3)(44, 24, 24)
    assert L.cppf_pair_mlp_backward_workspace_bytes(130, 64, 20, other, 2, 10) == 0
    offs = (C.c_int64 * 20)(*range(20))
    assert L.cppf_pair_mlp_backward(one, one, one, one, 1, one, offs, 10, 20, other, 2, 5, 10, one, one, one, one, 1 << 20,
                                    None) == -3
    assert L.cppf_pair_mlp_backward(one, one, one, one, 1, one, offs, 10, 40, dims, 3, 5, 141, one, one, one, None, 0,
                                    None) == -1                                          # layer 1 has no fc0: offs[10] must be -1
    offs[10] = offs[11] = -1
    assert L.cppf_pair_mlp_backward(one, one, one, one, 1, one, offs, 10, 40, dims, 3, 5, 141, one, one, one, None, 0,
                                    None) == -2
    assert L.cppf_pair_mlp_pack_device(one, offs, 20, other, 2, 10, one, None) == -3       # device pack: MFMA path only
    # row f3 entry points
    assert L.cppf_voxel_dedupe_workspace_bytes(70000) > 70000 * (16 + 8 + 1)
    assert L.cppf_voxel_dedupe(one, 10, 0.0, one, one, one, 1 << 30, None) == -1           # res must be positive
    assert L.cppf_voxel_dedupe(one, 10, 0.01, one, one, None, 0, None) == -2
    assert L.cppf_estimate_normals(None, None, 0, 30, None, None) == 0
    assert L.cppf_estimate_normals(None, None, 10, 30, None, None) == -1
    # module surface: the reference checkpoint layout loads, and the CPU autograd composite equals the reference
    z = golden("sprin_l1.npz")
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    enc = PointEncoder(k=60, spfcs=[32, 64, 32, 32], num_layers=1, out_dim=32)
    assert sorted(enc.state_dict().keys()) == sorted(sd.keys())
    enc.load_state_dict(sd)
    pc, nrm = torch.from_numpy(z["pc"][None]), torch.from_numpy(z["nrm"][None])
    out = enc(pc, nrm, torch.cdist(pc, pc))                                              # grad enabled -> composite
    np.testing.assert_allclose(out[0].detach().numpy(), z["out"], atol=1e-6)
    with torch.no_grad(), pytest.raises(_lib.CppfError):
        enc(pc, nrm, torch.cdist(pc, pc))                                                # no CPU fallback for inference
    packed, desc = pack_point_encoder({k: v.numpy() for k, v in sd.items()}, 1)
    image = np.zeros(9256 + 6912, np.float32)
    assert L.cppf_point_encoder_pack(packed.ctypes.data, hid, 4, 32, 2, 32, 8, 1, image.ctypes.data) == 0
    assert np.array_equal(image[:9256], packed)
    W2 = sd["spconvs.0.kernel.3.weight"].numpy()                                         # Linear(32, 64)
    ob, s_, ln = 3, 5, 37                                                                # lane (row 5, g = 2) of block 3, step 5
    assert image[9256 + 256 + (ob * 8 + s_) * 64 + ln] == W2[16 * ob + (ln & 15), 16 * (s_ // 4) + 4 * (ln >> 4) + s_ % 4]
    assert packed.size == 9256 and desc == dict(hidden=[32, 64, 32, 32], rank=32, n_nbr_feats=2, n_out=32, n_glob=8,
                                                num_layers=1)


def test_nocs_result_record_format():
    """pred_RTs / pred_scales as nocs/inference.py:114-117,213,338-342 builds them (the input of nocs/eval.py)"""
    from cppf_amd.inference import nocs_result
    R = np.array([[0, 1, 0], [0, 0, 1], [1, 0, 0]], np.float64)
    scale = np.array([0.1, 0.3, 0.1])
    pose = dict(T=np.array([0.05, -0.02, 0.7]), R=R, scale=scale, scale_norm=float(np.linalg.norm(scale)))
    res = nocs_result([pose, None], {"image_path": "x"})
    assert res["image_path"] == "x" and res["pred_RTs"].shape == (2, 4, 4) and res["pred_RTs"].dtype == np.float32
    np.testing.assert_allclose(res["pred_RTs"][0][:3, :3], (R * np.linalg.norm(scale)).astype(np.float32))
    np.testing.assert_allclose(res["pred_RTs"][0][:3, 3], pose["T"].astype(np.float32))
    assert res["pred_RTs"][0][3].tolist() == [0, 0, 0, 1]
    np.testing.assert_allclose(np.linalg.norm(res["pred_scales"][0]), 1.0, atol=1e-6)
    assert np.array_equal(res["pred_RTs"][1], np.eye(4, dtype=np.float32)) and res["pred_scales"][1].tolist() == [1, 1, 1]


def test_shape_polymorphic_plan_queries_without_a_device():
    """The *_dyn launch geometry is derived on the host from these queries (cppf_amd/inference.py:grid_class)."""
    from cppf_amd import _lib
    from cppf_amd.inference import grid_class
    L = _lib.lib()
    cells = L.cppf_vote_tile_cells()
    assert cells % 4 == 0 and 20000 < cells < 40960                      # one LDS tile of a 160 KB CU
    assert L.cppf_vote_tiles(26, 76, 26) == 2                            # the bottle grid of BASELINE config 2
    assert L.cppf_vote_tiles(10, 10, 10) == 1
    assert L.cppf_vote_tiles(0, 10, 10) < 0
    assert L.cppf_vote_tiles(600, 600, 600) == 0                         # beyond the tiled vote: global atomics only
    for dims in ((26, 76, 26), (52, 152, 52), (100, 49, 76), (5, 5, 4000)):
        T, many, cap = grid_class(dims)
        assert T * cells >= dims[0] * dims[1] * dims[2] and bool(many) == (T >= 4) and many == (0 if T < 4 else (16 if T <= 16 else 1))
        assert cap == _lib.tiles_cap(many) * cells and _lib.tiles_cap(many) >= T
    assert [_lib.tiles_cap(m_) for m_ in (0, False, 1, True, 4, 16, 64)] == [3, 3, 64, 64, 4, 16, 64]
    with pytest.raises(ValueError):
        _lib.tile_class(3)
    # ... and the queues of a capacity class are sized for ITS tiles: class 16 asks for a quarter of class 1's
    q16, q64 = (L.cppf_vote_workspace_bytes_dyn_pairs(m_, 2 ** 21) - L.cppf_vote_workspace_bytes_dyn_pairs(0, 2 ** 21) for m_ in (16, 1))
    assert q16 == 16 * 2 ** 21 * 12 + 2 ** 21 * 48 and q64 == 64 * 2 ** 21 * 12 + 2 ** 21 * 48
    # the dyn workspace holds the partial tiles of ANY plan its launch geometry can meet
    few, many = L.cppf_vote_workspace_bytes_dyn_pairs(0, 524288), L.cppf_vote_workspace_bytes_dyn_pairs(1, 2 ** 21)
    assert few >= L.cppf_vote_workspace_bytes(524288, 72, 26, 76, 26) and many >= L.cppf_vote_workspace_bytes(2 ** 21, 72, 52, 152, 52)
    assert few >= L.cppf_vote_workspace_bytes(524288, 100, 26, 76, 26)     # (n_rots > 72: the same kernels, several passes)
    # entry points reject a missing shape record before any HIP call
    assert L.cppf_vote_argmax_dyn(None, None, None, None, 0, None, 100, None, 0.004, 4, 10, 72, None, 0, 1, 0, None, None,
                                  None, 0, None) == -1
    assert L.cppf_backvote_dyn(None, None, None, None, None, 0.004, 10, 72, None, None, 0.01, None, None) == -1
    assert L.cppf_knn_dyn(None, 64, None, 60, None, None) == -1


def test_weight_image_invalidation_keys():
    """the device weight images are keyed on (data_ptr, _version) of every parameter; `.data` edits bypass
    the version counter and need invalidate(); .to()/.float() re-key through _apply; replaced Parameters are picked up
    after invalidate()."""
    import torch
    from cppf_amd.models.model import PPFEncoder, PointEncoder
    for enc in (PPFEncoder([84, 32, 32, 16], 141), PointEncoder(60, [32, 64, 32, 32], 32, 1)):
        p0 = next(enc.parameters())
        k0 = enc._param_key("cpu")
        with torch.no_grad():
            p0.add_(1.0)                                  # what an optimizer step / load_state_dict does
        k1 = enc._param_key("cpu")
        assert k1 != k0
        p0.data.mul_(2.0)                                 # invisible to the version counter ...
        assert enc._param_key("cpu") == k1
        enc.invalidate()                                  # ... until told
        k2 = enc._param_key("cpu")
        assert k2 != k1
        enc.double().float()                              # _apply: parameters re-created / re-typed
        assert enc._param_key("cpu") != k2
        n_before = len(enc._param_key("cpu"))
        first = [m for m in enc.modules() if isinstance(m, torch.nn.Linear)][0]
        first.bias = torch.nn.Parameter(torch.zeros_like(first.bias))     # a replaced Parameter object
        enc.invalidate()
        assert len(enc._param_key("cpu")) == n_before and any(p is first.bias for p in enc.parameters())


def test_error_codes_surface_as_readable_cppf_errors():
    """include/cppf.h: return value 0, a NEGATIVE CPPF_E* code, or a POSITIVE hipError_t when a HIP call failed -- both kinds
    must reach Python as a CppfError that names the call and carries the runtime's own text"""
    import pytest
    from cppf_amd import _lib
    L = _lib.lib()
    _lib.check(0, "cppf_nothing")                                    # success raises nothing
    for code, word in ((-1, "invalid argument"), (-2, "workspace"), (-3, "unsupported")):
        with pytest.raises(_lib.CppfError, match=word) as e:
            _lib.check(code, "cppf_some_call")
        assert "cppf_some_call" in str(e.value) and f"({code})" in str(e.value)
    # positive codes are hipError_t values: 1 = hipErrorInvalidValue, 2 = hipErrorOutOfMemory, 98 = hipErrorInvalidDeviceFunction
    # (hipGetErrorString is a table lookup: it needs no device)
    texts = {}
    for code in (1, 2, 98, 719):
        with pytest.raises(_lib.CppfError) as e:
            _lib.check(code, "cppf_vote_argmax")
        msg = str(e.value)
        assert "cppf_vote_argmax" in msg and f"({code})" in msg
        texts[code] = msg
        raw = L.cppf_error_string(code)
        assert raw and raw.decode() in msg and raw.decode().strip() not in ("", "?", "cppf: unknown error")
    assert "invalid" in texts[1].lower() and "memory" in texts[2].lower()
    assert len(set(texts.values())) == len(texts)
    assert b"unknown" in L.cppf_error_string(-77)


def test_vote_plans_respect_their_invariants_without_a_device():
    """the round-3 tilings (owned cells + halo) over a sweep of grids: every tile with its halo fits the LDS tile, the tiles cover
    the grid, the path follows the tile count, the workspace holds one partial tile per workgroup (+ the queues for >= 4 tiles)"""
    import numpy as np
    from cppf_amd import _lib
    L = _lib.lib()
    cells = L.cppf_vote_tile_cells()
    out = (C.c_int32 * 10)()
    rng = np.random.default_rng(0)
    seen = set()
    cases = [(26, 76, 26), (52, 152, 52), (67, 34, 67), (1, 1, 1), (300, 3, 40), (3, 300, 40), (120, 120, 120), (20, 20, 3000)]
    cases += [tuple(int(v) for v in rng.integers(1, 160, 3)) for _ in range(200)]
    for gx, gy, gz in cases:
        P = int(rng.integers(1, 3_000_000))
        assert L.cppf_vote_plan_query(P, 72, gx, gy, gz, out) == 0
        path, T, tx, ty, ntx, nty, hx, hy, wgs, bits = list(out)
        seen.add(path)
        need = L.cppf_vote_workspace_bytes(P, 72, gx, gy, gz)
        if path in (2, 3):
            assert T == ntx * nty and 1 <= T <= 64 and (path == 2) == (T < 4)
            assert (tx + hx) * (ty + hy) * gz <= cells                       # a tile with its halo fits the LDS tile
            assert ntx * tx >= gx > (ntx - 1) * tx and nty * ty >= gy > (nty - 1) * ty     # the tiles cover the grid, none is empty
            assert hx == (1 if ntx > 1 else 0) and hy == (1 if nty > 1 else 0)
            assert T <= wgs <= 1024 and 8 <= bits <= 24
            slot = ((tx + hx) * (ty + hy) * gz + 3) // 4 * 4
            assert need >= wgs * slot * 4 + (T * P * 12 if path == 3 else 0)
            assert T == L.cppf_vote_tiles(gx, gy, gz)
        else:
            assert path == 0 and need == L.cppf_vote_workspace_init_bytes()  # > 64 tiles: global atomics
        # n_rots > 72 takes the same path (several passes of the same kernels), with the same workspace
        first = list(out)
        assert L.cppf_vote_plan_query(P, 100, gx, gy, gz, out) == 0 and list(out)[:9] == first[:9]
        assert L.cppf_vote_workspace_bytes(P, 100, gx, gy, gz) == need
    assert {2, 3} <= seen
    assert L.cppf_vote_plan_query(10, 72, 0, 4, 4, out) == -1 and L.cppf_vote_plan_query(10, 72, 4, 4, 4, None) == -1


def test_bench_vote_width_rule():
    """bench.py's timed pipelines: ~8 192 pairs per vote workgroup and tile when several instances are in flight (cppf.h
    CPPF_VOTE_WORKGROUPS), one workgroup per CU for one instance at a time or from a million pairs on; an explicit flag wins"""
    import sys
    import types
    sys.path.insert(0, ROOT)
    import bench
    a = types.SimpleNamespace(vote_workgroups=-1, streams=3)
    assert bench.vote_width(a, 524288, (26, 76, 26)) == 128      # C2: two tiles
    assert bench.vote_width(a, 524288, (29, 83, 29)) == 192      # three tiles
    assert bench.vote_width(a, 1048576, (26, 76, 26)) == 0       # C3: the full width
    assert bench.vote_width(a, 100000, (20, 20, 20)) == 64       # never below 64
    assert bench.vote_width(types.SimpleNamespace(vote_workgroups=-1, streams=1)) == 0
    assert bench.vote_width(types.SimpleNamespace(vote_workgroups=200, streams=3)) == 200


def test_round5_entry_points_host_logic_without_a_device():
    """the batched entry points' plans and argument checks (no launches): pair-list plan, vote widths, frame stage workspace,
    pair sampler / batched vote / batched tail argument errors, the fused vote's width-independent fixed-point bits"""
    from cppf_amd import _lib
    from cppf_amd.models.model import batch_plan
    L = _lib.lib()
    P = 524288
    for n in (1, 2, 4, 8):            # equal lists: XCD-pinned, list i on XCDs [i per_xcd, (i + 1) per_xcd)
        pl = batch_plan([P] * n)
        assert pl["per_xcd"] == 8 // n and pl["grid"] % 8 == 0 and pl["wg_begin"][0] == 0 and pl["wg_begin"][-1] >= pl["grid"]
    assert batch_plan([P] * 3)["per_xcd"] == 0 and batch_plan([P, P // 2])["per_xcd"] == 0 and batch_plan([P, P - P // 11])["per_xcd"] == 4
    ragged = batch_plan([81920, 64, 100000])
    assert ragged["per_xcd"] == 0 and all(b > a for a, b in zip(ragged["wg_begin"], ragged["wg_begin"][1:]))      # at least one each
    with pytest.raises(_lib.CppfError):
        batch_plan([0])
    # vote widths of a batch: 256 / n, never below 64 (the width the fixed-point scale is chosen for), a hint overrides
    assert [L.cppf_vote_batch_workgroups(n, 0) for n in (1, 2, 3, 4, 8)] == [256, 128, 85, 64, 64]
    assert L.cppf_vote_batch_workgroups(4, 128 << 8) == 128 and L.cppf_vote_batch_workgroups(4, 32 << 8) == 64
    assert L.cppf_vote_batch_workgroups(0, 0) == -1 and L.cppf_vote_batch_workgroups(9, 0) == -1
    # the fused vote's bits do not depend on the launch width any more: the plan query (full width) says what a 64-wide launch uses
    plan = (C.c_int32 * 10)()
    assert L.cppf_vote_plan_query(P, 72, 26, 76, 26, plan) == 0 and plan[0] == 2 and plan[8] == 256 and plan[9] == 22
    assert L.cppf_vote_fixed_point_bits(P, 72, 26, 76, 26) == 22 and L.cppf_vote_fixed_point_bits(65536, 72, 26, 76, 26) == 24
    # argument errors before any HIP call
    assert L.cppf_vote_argmax_batch(0, None, 72, 1, 0, None) == -1 and L.cppf_vote_argmax_batch(1, None, 72, 1, 0, None) == -1
    it = (_lib.VoteItem * 1)()
    assert L.cppf_vote_argmax_batch(1, C.cast(it, C.c_void_p), 0, 1, 0, None) == -1               # n_rots out of range
    assert L.cppf_vote_argmax_batch(1, C.cast(it, C.c_void_p), 72, 1, 2, None) == -1              # unknown flag bit
    dims = (C.c_int * 4)(84, 32, 32, 16)
    assert L.cppf_pose_tail_batch(0, None, 40, dims, 3, 141, 32, 36, 72, None, None, 480, 1, 0.9, 10000, None) == -1
    assert L.cppf_pair_mlp_decode_sel_batch(0, None, 40, dims, 3, 141, 32, 36, None) == -1
    assert L.cppf_sample_pairs(None, None, None, 10, 100, None, 1, None, None) == -1 and L.cppf_sample_pairs(None, None, None, 0, 100, None, 1, None, None) == 0
    assert L.cppf_mod_pairs_dyn(None, 0, None, None) == 0 and L.cppf_mod_pairs_dyn(None, 5, None, None) == -1
    # the frame stage's workspace grows with the image and with the capacity; bad arguments are refused
    w1, w2 = L.cppf_frame_cloud_workspace_bytes(480, 640, 4096, 60), L.cppf_frame_cloud_workspace_bytes(480, 640, 65536, 60)
    assert 0 < w1 < w2 < (1 << 27) and L.cppf_frame_cloud_workspace_bytes(0, 640, 4096, 60) == 0
    assert L.cppf_frame_cloud_dyn(None, 1, None, 2, 0, 480, 640, None, 1000.0, 0.004, 60, 61, 4096, None, None, None, None, None, None, 0, None) == -1


def test_bench_compact_line_fits_the_drivers_tail():
    """bench.py prints a compact line (< 4 KB) with the contract's fields and both rooflines; the full record goes to
    bench_full.json.  Checked on the committed full record of this round's default run."""
    import glob
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default_full.json")))
    if not recs:
        pytest.skip("no committed full record yet")
    full = json.load(open(recs[-1]))
    line = bench.compact(full)
    text = json.dumps(line)
    assert len(text) < 4096, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == full["value"] and line["metric"] == bench.METRIC and "workload" in line["config"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])


def test_workspace_scope_is_per_thread():
    """scratch ownership follows the host thread that entered the scope (two threads may drive two pipelines): a scope entered
    on one thread is invisible on another, and nesting restores the outer owner"""
    import threading
    from cppf_amd import _torch_util as tu
    seen, gate, done = {}, threading.Barrier(2), threading.Barrier(2)

    def worker(name):
        with tu.workspace_scope(name):
            gate.wait(timeout=10)                       # both threads are inside their scopes now
            seen[name] = getattr(tu._ws_tls, "scope", None)
            with tu.workspace_scope(name + "/inner"):
                seen[name + "/inner"] = tu._ws_tls.scope
            seen[name + "/after"] = tu._ws_tls.scope
            done.wait(timeout=10)
        seen[name + "/out"] = getattr(tu._ws_tls, "scope", None)

    ts = [threading.Thread(target=worker, args=(n,)) for n in ("a", "b")]
    for t_ in ts:
        t_.start()
    for t_ in ts:
        t_.join(20)
    assert seen == {"a": "a", "b": "b", "a/inner": "a/inner", "b/inner": "b/inner", "a/after": "a", "b/after": "b", "a/out": None, "b/out": None}
    assert getattr(tu._ws_tls, "scope", None) is None


def test_plain_c_host_compiles_and_runs_the_host_only_entry_points(c_host):
    """include/cppf.h is a C header (gcc -std=c99 -Wall -Werror) and libcppf_hip.so links into a C program: ABI version, error
    strings, the host grid shape of nocs/inference.py:194-195, launch plans, EINVAL before any device call -- no Python in the loop"""
    import subprocess
    p = subprocess.run([c_host], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "host ok: ABI 4" in p.stdout


def test_grid_shape_fails_loudly_on_non_finite_clouds():
    """nocs/inference.py:194-196 on a cloud with a NaN: numpy's min / max propagate it into the dims and np.zeros refuses them; both
    paths of cppf_amd.inference.grid_shape (the C pass and the numpy form) raise instead of laying a plausible grid over the rest"""
    from cppf_amd.inference import grid_shape
    rng = np.random.default_rng(0)
    pc = rng.random((100, 3)).astype(np.float32)
    c0, d0 = grid_shape(pc, 0.01)
    c1, d1 = grid_shape(pc.astype(np.float64), 0.01)            # (the numpy form: not a C-contiguous f32 array)
    assert d0 == d1 and np.array_equal(c0, c1)
    for bad in (np.nan, np.inf, -np.inf):
        for row in (0, 57, 99):
            q = pc.copy()
            q[row, 1] = bad
            with pytest.raises(ValueError):
                grid_shape(q, 0.01)
            with pytest.raises(ValueError):
                grid_shape(q.astype(np.float64), 0.01)


def test_workspace_bounds_at_five_million_pairs():
    """what the vote asks for at the notebook's size: a by-value launch sizes its pair -> tile queues for the
    tiles the grid HAS (67 x 34 x 67: 8); the global-atomics path keeps no queues; a shape-polymorphic many-tile pipeline sizes them
    for every tile of its class (64): 3.9 GB -- callers at this size launch by value, as the notebook does"""
    from cppf_amd import _lib
    L = _lib.lib()
    P5M = 5_000_000
    tiles = L.cppf_vote_tiles(67, 34, 67)
    assert 4 <= tiles <= 8
    by_value = L.cppf_vote_workspace_bytes(P5M, 72, 67, 34, 67)
    assert by_value <= tiles * P5M * 12 + P5M * 48 + (64 << 20)   # 12 B per (pair, tile) record worst case + a 48 B frame per pair + partial tiles
    assert by_value <= 0.65 * (1 << 30)                          # 6 tiles: 0.36 GB of queues + 0.24 GB of frames + partial tiles
    assert L.cppf_vote_workspace_bytes(P5M, 72, 330, 166, 334) == L.cppf_vote_workspace_init_bytes()      # beyond 64 tiles: no queues
    few = L.cppf_vote_workspace_bytes_dyn_pairs(0, P5M)
    assert few <= (64 << 20)                                      # few-tile class: no queues at all, whatever P
    many = L.cppf_vote_workspace_bytes_dyn_pairs(1, P5M)
    assert 64 * P5M * 12 + P5M * 48 <= many <= 64 * P5M * 12 + P5M * 48 + (64 << 20)
