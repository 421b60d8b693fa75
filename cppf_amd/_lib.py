"""ctypes binding of libcppf_hip.so (C ABI: include/cppf.h).  Fails loudly when the library is
missing or a call returns an error -- there is no fallback path."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# CPPF_SO: developer override (ablation / instrumented builds of the same ABI)
_SO = os.environ.get("CPPF_SO") or os.path.join(_HERE, "csrc", "libcppf_hip.so")
_lib = None

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

_SIGS = {
    "cppf_abi_version": (C.c_int, []),
    "cppf_error_string": (C.c_char_p, [C.c_int]),
    "cppf_vote_workspace_bytes": (sz, [i64, i32, i32, i32, i32]),
    "cppf_vote_fixed_point_bits": (C.c_int, [i64, i32, i32, i32, i32]),
    "cppf_ppf_voting": (C.c_int, [vp, vp, vp, vp, vp, vp, f32, i64, i64, i32, i32, i32, i32, i32, vp, sz, vp]),
    "cppf_vote_argmax": (C.c_int, [vp, vp, vp, vp, i32, vp, vp, f32, i64, i64, i32, i32, i32, i32, i32, i32, vp, vp, vp, sz,
                                   vp]),
    "cppf_vote_grid_raw": (C.c_int, [vp, vp, vp, vp, i32, vp, vp, vp, f32, i64, i64, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "cppf_grid_from_raw": (C.c_int, [vp, i64, vp, vp, vp, vp, vp, sz, vp]),
    "cppf_vote_tiles": (C.c_int, [i32, i32, i32]),
    "cppf_vote_tile_cells": (C.c_int, []),
    "cppf_vote_workspace_bytes_dyn_pairs": (sz, [i32, i64]),
    "cppf_vote_workspace_init_bytes": (sz, []),
    "cppf_vote_plan_query": (C.c_int, [i64, i32, i32, i32, i32, vp]),
    "cppf_vote_argmax_dyn": (C.c_int, [vp, vp, vp, vp, i32, vp, i64, vp, f32, i64, i64, i32, vp, i32, i32, i32, vp, vp, vp, sz,
                                       vp]),
    "cppf_pose_tail_batch": (C.c_int, [i32, vp, i32, C.POINTER(C.c_int), i32, i32, i32, i32, i32, vp, vp, i32, i32, f32, i64, vp]),
    "cppf_vote_batch_workgroups": (C.c_int, [i32, i32]),
    "cppf_vote_argmax_batch": (C.c_int, [i32, vp, i32, i32, i32, vp]),
    "cppf_center_from_argmax_dyn": (C.c_int, [vp, vp, C.c_double, vp, vp, vp, vp, vp, vp]),
    "cppf_backvote_ws": (C.c_int, [vp, vp, vp, vp, vp, f32, i64, i32, i32, i32, i32, vp, vp, f32, vp, vp, vp]),
    "cppf_backvote_dyn": (C.c_int, [vp, vp, vp, vp, vp, f32, i64, i32, vp, vp, f32, vp, vp]),
    "cppf_knn_dyn": (C.c_int, [vp, i32, vp, i32, vp, vp]),
    "cppf_point_encoder_forward_dyn": (C.c_int, [vp, vp, vp, i32, vp, i32, vp, C.POINTER(C.c_int), i32, i32, i32, i32, i32, i32,
                                                 vp, vp, sz, vp]),
    "cppf_gather_words": (C.c_int, [i32, vp, i64, vp, vp]),
    "cppf_frame_cloud_dyn_batch": (C.c_int, [i32, vp, vp, i32, vp, i32, i32, i32, vp, C.c_double, vp]),
    "cppf_point_encoder_forward_batch": (C.c_int, [i32, vp, i32, C.POINTER(C.c_int), i32, i32, i32, i32, i32, i32, vp]),
    "cppf_grid_argmax": (C.c_int, [vp, i64, vp, vp, vp, sz, vp]),
    "cppf_center_from_argmax": (C.c_int, [vp, vp, C.c_double, i32, i32, vp, vp, vp, vp, vp]),
    "cppf_counts_argmax_select": (C.c_int, [vp, i32, vp, vp, vp, vp]),
    "cppf_backvote": (C.c_int, [vp, vp, vp, vp, vp, f32, i64, i32, i32, i32, i32, vp, f32, vp, vp]),
    "cppf_compact_workspace_bytes": (sz, [i64]),
    "cppf_compact_mask": (C.c_int, [vp, i64, vp, vp, vp, sz, vp]),
    "cppf_rot_voting": (C.c_int, [vp, vp, vp, vp, i64, i32, vp]),
    "cppf_rot_sphere_count": (C.c_int, [vp, vp, i32, vp, vp, vp, i64, i64, i32, vp, i32, f32, i32, vp, vp]),
    "cppf_pair_mlp_packed_floats": (sz, [i32, C.POINTER(C.c_int), i32, i32]),
    "cppf_pair_mlp_pack": (C.c_int, [vp, vp, i32, C.POINTER(C.c_int), i32, i32, vp]),
    "cppf_pair_mlp_pack_device": (C.c_int, [vp, C.POINTER(C.c_int64), i32, C.POINTER(C.c_int), i32, i32, vp, vp]),
    "cppf_pair_mlp_workspace_bytes": (sz, [i64, i32, C.POINTER(C.c_int), i32, i32]),
    "cppf_pair_mlp_forward": (C.c_int, [vp, vp, vp, vp, i32, vp, i64, i32, C.POINTER(C.c_int), i32, i64, i32, vp, vp,
                                        sz, vp]),
    "cppf_pair_mlp_decode": (C.c_int, [vp, vp, vp, vp, i32, vp, i64, i32, C.POINTER(C.c_int), i32, i64, i32, i32, i32,
                                       f32, f32, vp, vp, vp, vp, vp, sz, vp]),
    "cppf_pair_mlp_decode_batch": (C.c_int, [i32, vp, i32, C.POINTER(C.c_int), i32, i32, i32, i32, vp]),
    "cppf_pair_mlp_decode_sel_batch": (C.c_int, [i32, vp, i32, C.POINTER(C.c_int), i32, i32, i32, i32, vp]),
    "cppf_pair_mlp_batch_plan": (C.c_int, [i32, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "cppf_pair_mlp_decode_sel": (C.c_int, [vp, vp, vp, vp, i32, vp, i64, i32, C.POINTER(C.c_int), i32, i64, i32, i32, i32, vp, vp, vp,
                                           i64, vp, vp, sz, vp]),
    "cppf_decode_center": (C.c_int, [vp, i64, i32, i32, f32, f32, vp, vp, vp]),
    "cppf_decode_rot": (C.c_int, [vp, i64, i32, i32, i32, i32, vp, vp, vp]),
    "cppf_reduce_workspace_bytes": (sz, []),
    "cppf_pose_tail_begin": (C.c_int, [vp, vp, C.c_double, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]),
    "cppf_backvote_count": (C.c_int, [vp, vp, vp, vp, f32, i64, i32, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp]),
    "cppf_backvote_count64": (C.c_int, [vp, vp, vp, vp, vp, f32, i64, i32, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp]),
    "cppf_compact_scatter": (C.c_int, [vp, i64, vp, vp, vp, vp]),
    "cppf_rot_sphere_count_dirs": (C.c_int, [vp, vp, i32, i32, i32, vp, vp, vp, i64, i64, i32, vp, i32, f32, i32, vp, i32, vp]),
    "cppf_rot_sphere_count_dirs_order": (C.c_int, [vp, vp, i32, i32, i32, vp, vp, vp, i64, vp, i64, i64, i32, vp, i32, f32, i32, vp,
                                                   i32, vp]),
    "cppf_pose_sums_workspace_bytes": (sz, []),
    "cppf_pose_sums": (C.c_int, [vp, vp, vp, vp, vp, i64, vp, i32, i32, vp, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, sz, vp, vp]),
    "cppf_axis_sign": (C.c_int, [vp, vp, vp, vp, vp, i64, vp, i32, vp, vp, vp, sz, vp]),
    "cppf_scale_sum": (C.c_int, [vp, i32, vp, vp, i64, vp, vp, sz, vp]),
    "cppf_grid_setup": (C.c_int, [vp, i64, f32, vp, vp, vp]),
    "cppf_pair_mlp_backward_workspace_bytes": (sz, [i64, i64, i32, C.POINTER(C.c_int), i32, i32]),
    "cppf_pair_mlp_backward": (C.c_int, [vp, vp, vp, vp, i32, vp, C.POINTER(C.c_int64), i64, i32, C.POINTER(C.c_int), i32,
                                         i64, i32, vp, vp, vp, vp, sz, vp]),
    "cppf_knn": (C.c_int, [vp, vp, i32, i32, vp, vp]),
    "cppf_frame_cloud_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "cppf_frame_cloud_dyn": (C.c_int, [vp, i32, vp, i32, i32, i32, i32, vp, C.c_double, C.c_double, i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]),
    "cppf_frame_cloud_dyn_bit": (C.c_int, [vp, i32, vp, i32, vp, i32, i32, vp, C.c_double, C.c_double, i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]),
    "cppf_sample_pairs": (C.c_int, [vp, vp, vp, i64, i64, vp, C.c_uint64, vp, vp]),
    "cppf_host_grid_shape": (C.c_int, [vp, i64, f32, vp, vp]),
    "cppf_mod_pairs_dyn": (C.c_int, [vp, i64, vp, vp]),
    "cppf_stage_batch": (C.c_int, [i32, vp, vp]),
    "cppf_copy_words": (C.c_int, [vp, vp, i64, vp]),
    "cppf_backproject_workspace_bytes": (sz, [i32, i32]),
    "cppf_backproject": (C.c_int, [vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
    "cppf_voxel_dedupe_workspace_bytes": (sz, [i64]),
    "cppf_voxel_dedupe": (C.c_int, [vp, i64, C.c_double, vp, vp, vp, sz, vp]),
    "cppf_estimate_normals": (C.c_int, [vp, vp, i64, i32, vp, vp]),
    "cppf_point_encoder_packed_floats": (sz, [C.POINTER(C.c_int), i32, i32, i32, i32, i32, i32]),
    "cppf_point_encoder_pack": (C.c_int, [vp, C.POINTER(C.c_int), i32, i32, i32, i32, i32, i32, vp]),
    "cppf_point_encoder_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "cppf_point_encoder_backward_workspace_bytes": (sz, [i32]),
    "cppf_point_encoder_pack_device": (C.c_int, [vp, C.POINTER(C.c_int), i32, i32, i32, i32, i32, i32, vp, vp]),
    "cppf_point_encoder_backward": (C.c_int, [vp, vp, vp, i32, i32, vp, C.POINTER(C.c_int), i32, i32, i32, i32, i32, i32, vp, vp,
                                              vp, vp, vp, sz, vp]),
    "cppf_point_encoder_forward_train": (C.c_int, [vp, vp, vp, i32, i32, vp, C.POINTER(C.c_int), i32, i32, i32, i32, i32, i32,
                                                   vp, vp, vp, sz, vp]),
    "cppf_point_encoder_forward": (C.c_int, [vp, vp, vp, i32, i32, vp, C.POINTER(C.c_int), i32, i32, i32, i32, i32, i32,
                                             vp, vp, sz, vp]),
}

class PairMlpItem(C.Structure):
    """include/cppf.h: CppfPairMlpItem (one pair list of cppf_pair_mlp_decode_batch)"""
    _fields_ = [("pc", C.c_void_p), ("nrm", C.c_void_p), ("feat", C.c_void_p), ("idxs", C.c_void_p), ("packed", C.c_void_p),
                ("u_tr", C.c_void_p), ("u_rot", C.c_void_p), ("outputs", C.c_void_p), ("heads", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("n_points", C.c_int64), ("n_pairs", C.c_int64),
                ("vr0", C.c_float), ("vr1", C.c_float), ("idx_is_i64", C.c_int),
                ("sel", C.c_void_p), ("n_sel_dev", C.c_void_p), ("max_sel", C.c_int64)]


class PoseTailItem(C.Structure):
    """include/cppf.h: CppfPoseTailItem (one object of cppf_pose_tail_batch)"""
    _fields_ = [("pc", C.c_void_p), ("nrm", C.c_void_p), ("feat", C.c_void_p), ("idx64", C.c_void_p), ("idx32", C.c_void_p),
                ("outputs", C.c_void_p), ("u_rot", C.c_void_p), ("heads", C.c_void_p), ("corner", C.c_void_p), ("shape_dev", C.c_void_p),
                ("argmax_idx", C.c_void_p), ("peak", C.c_void_p), ("packed", C.c_void_p), ("mlp_workspace", C.c_void_p),
                ("mlp_workspace_bytes", C.c_size_t), ("vote_workspace", C.c_void_p), ("rec", C.c_void_p), ("T32", C.c_void_p),
                ("tail0", C.c_void_p), ("tail0_bytes", C.c_size_t), ("mask", C.c_void_p), ("chunk_counts", C.c_void_p), ("surv", C.c_void_p),
                ("count", C.c_void_p), ("counts", C.c_void_p), ("best_idx", C.c_void_p), ("ticket", C.c_void_p),
                ("sums_workspace", C.c_void_p), ("sums_workspace_bytes", C.c_size_t), ("n_points", C.c_int64), ("n_pairs", C.c_int64),
                ("res64", C.c_double), ("res", C.c_float), ("tol", C.c_float), ("gx", C.c_int), ("gy", C.c_int), ("gz", C.c_int),
                ("n_dirs", C.c_int), ("second_pass", C.c_int),
                ("record_out", C.c_void_p), ("object_id_dev", C.c_void_p), ("object_id_host", C.c_longlong),
                ("scale_mean", C.c_double * 3), ("regress_right", C.c_int)]


class StageItem(C.Structure):
    """include/cppf.h: CppfStageItem (one object of cppf_stage_batch); its device-side descriptor CppfStageDesc is six 64-bit words:
    {pc_src, nrm_src, feat_src, n_points, seed, object_id}"""
    _fields_ = [("desc", C.c_void_p), ("pc", C.c_void_p), ("nrm", C.c_void_p), ("feat", C.c_void_p), ("corner", C.c_void_p),
                ("shape", C.c_void_p), ("idx", C.c_void_p), ("u_tr", C.c_void_p), ("u_rot", C.c_void_p), ("n_pairs", C.c_int64),
                ("n_cap", C.c_int64), ("F", C.c_int), ("res", C.c_float), ("idx_is_i64", C.c_int)]


STAGE_DESC_WORDS = 6


class FrameCloudItem(C.Structure):
    """include/cppf.h: CppfFrameCloudItem (one instance of cppf_frame_cloud_dyn_batch)"""
    _fields_ = [("label_bit_dev", C.c_void_p), ("seed_dev", C.c_void_p), ("pc_out", C.c_void_p), ("nrm_out", C.c_void_p),
                ("corner_out", C.c_void_p), ("shape_out", C.c_void_p), ("nbrs_out", C.c_void_p), ("idx", C.c_void_p), ("u_tr", C.c_void_p),
                ("u_rot", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("res", C.c_double),
                ("n_pairs", C.c_int64), ("knn_k", C.c_int32), ("k_min", C.c_int32), ("n_cap", C.c_int32), ("idx_is_i64", C.c_int32)]


class PointEncItem(C.Structure):
    """include/cppf.h: CppfPointEncItem (one cloud of cppf_point_encoder_forward_batch)"""
    _fields_ = [("pc", C.c_void_p), ("nrm", C.c_void_p), ("nbrs", C.c_void_p), ("n_dev", C.c_void_p), ("packed", C.c_void_p),
                ("out", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("n_cap", C.c_int32),
                ("nbrs_ready", C.c_int32)]


class VoteItem(C.Structure):
    """include/cppf.h: CppfVoteItem (one object of cppf_vote_argmax_batch)"""
    _fields_ = [("points", C.c_void_p), ("outputs", C.c_void_p), ("probs", C.c_void_p), ("point_idxs", C.c_void_p), ("grid", C.c_void_p),
                ("corner", C.c_void_p), ("out_idx", C.c_void_p), ("out_val", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("shape_dev", C.c_void_p), ("grid_capacity", C.c_int64), ("n_points", C.c_int64),
                ("n_ppfs", C.c_int64), ("res", C.c_float), ("gx", C.c_int), ("gy", C.c_int), ("gz", C.c_int), ("idx_is_i64", C.c_int),
                ("many_tiles", C.c_int)]


ABI_VERSION = 4     # include/cppf.h CPPF_ABI_VERSION: records assembled on the device (CppfPoseTailItem.record_out), cppf_stage_batch


class CppfError(RuntimeError):
    pass


def tile_class(many):
    """what the *_dyn entry points take as `many_tiles` (include/cppf.h): 0 = up to 3 LDS tiles (fused vote), 1 / True = up to 64,
    4..64 = up to that many"""
    m = 1 if many is True else int(many)
    if m in (0, 1) or 4 <= m <= 64:
        return m
    raise ValueError(f"tile capacity class {many!r}: 0, 1 (= 64 tiles) or 4..64")


def tiles_cap(many):
    """tiles a launch of that class serves"""
    m = tile_class(many)
    return 3 if m == 0 else (64 if m == 1 else m)


def library_path():
    return _SO


def build(force=False):
    """Compile libcppf_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    args = ["make", "-C", src_dir] + (["-B"] if force else []) + ["libcppf_hip.so"]
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise CppfError(f"{_SO} is missing: build it with `make -C cppf_amd/csrc` "
                            "(or __graft_entry__.build()); cppf_amd has no CPU fallback")
        # torch bundles its own HIP runtime (same soname as /opt/rocm's): import it first so that this
        # library binds to the runtime torch's tensors and streams live in.  Loading in the other
        # order puts two runtimes' worth of state in one process ("no ROCm-capable device").
        import torch  # noqa: F401
        L = C.CDLL(_SO)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError = ABI mismatch, also loud
            fn.restype, fn.argtypes = res, args
        if L.cppf_abi_version() != ABI_VERSION:
            raise CppfError(f"libcppf_hip.so has ABI version {L.cppf_abi_version()}, this binding needs {ABI_VERSION}: "
                            "rebuild with `make -C cppf_amd/csrc`")
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().cppf_error_string(rc)
        raise CppfError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def exported_symbols():
    return sorted(_SIGS)
