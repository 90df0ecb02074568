"""ctypes binding of libhsp.so (include/hsp.h).  Loaded lazily on the first device op; a missing
library is a hard error -- there is no fallback path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HSP_LIB=<path> loads another build of the same ABI (profiling variants); default: the in-tree library
LIB_PATH = os.environ.get("HSP_LIB") or os.path.join(_HERE, "libhsp.so")
_lib = None

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol declared in include/hsp.h
SIGNATURES = {
    "hsp_version": (_i, []),
    "hsp_error_string": (ctypes.c_char_p, [_i]),
    "hsp_last_hip_error": (ctypes.c_char_p, []),
    "hsp_knn_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "hsp_knn_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "hsp_nn1_f32": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    "hsp_rf_surface_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "hsp_rf_bwd_workspace_bytes": (_sz, [_i]),
    "hsp_rf_surface_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "hsp_rf_conv_wants_fwin": (_i, [_i, _i, _i]),
    "hsp_rf_conv_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "hsp_rf_conv_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "hsp_rf_bwd_scatter_workspace_bytes": (_sz, [_i, _i]),
    "hsp_rf_conv_bwd_scatter": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "hsp_rev_build": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "hsp_gather_max_bwd_csr": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "hsp_gather_max_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "hsp_pool_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "hsp_gather_max_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "hsp_points_max_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "hsp_points_max_bwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "hsp_orl_workspace_bytes": (_sz, [_i, _i, _i]),
    "hsp_knn_exact_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "hsp_knn_exact_f32": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _vp]),
    "hsp_knn_xyz_workspace_bytes": (_sz, [_i, _i]),
    "hsp_knn_xyz_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "hsp_geometry_levels_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hsp_geometry_all_workspace_bytes": (_sz, [_i, _i]),
    "hsp_geometry_all_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                  _vp]),
    "hsp_knn_quadmode_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _i, _vp]),
    "hsp_quad_outer_f32": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "hsp_center_cloud_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "hsp_orl_exact_workspace_bytes": (_sz, [_i, _i, _i]),
    "hsp_orl_global_exact_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "hsp_bn_eval_f32": (_i, [_vp, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _i, _vp, _vp]),
    "hsp_layer_out_exact_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "hsp_orl_global_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "hsp_colsum_rows": (_i, [_vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "hsp_add_relu_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _vp]),
    "hsp_residual_bias": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "hsp_concat_rows": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "hsp_gather_rows_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "hsp_gather_rows_bwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "hsp_gather_rows_bwd_csr": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "hsp_gemm_rows_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "hsp_gemm_rows_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, ctypes.c_float, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "hsp_gemm_rows_bf16": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, ctypes.c_float, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "hsp_gemm_wave_supported": (_i, [_i, _i, _i, _i, _i]),
    "hsp_gemm_wave_plan_info": (_i, [_i, _i, _i, _i, _i, _vp]),
    "hsp_gemm_wave_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, ctypes.c_float, _vp, _vp, _vp, _i, _i, _vp]),
    "hsp_split_params_x3": (_i, [_vp, _i, _i, _vp]),
    "hsp_gemm_x3_supported": (_i, [_i, _i, _i, _i]),
    "hsp_gemm_x3_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "hsp_gemm_x3_f32": (_i, [_vp, _i, _vp, _i, ctypes.c_longlong, _i, _vp, _i, _vp, _i, ctypes.c_longlong, _i, _i, _i, _vp, _vp, _i, _vp, _i,
                             ctypes.c_float, _vp, _i, _vp, _sz, _vp]),
    "hsp_small_rows_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, ctypes.c_float, _vp, _i, _vp]),
    "hsp_small_outer_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _vp]),
    "hsp_colsum_rows_xyz": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "hsp_colsum_rows_xyz_bf16": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "hsp_bn_relu_fwd_mixed": (_i, [_vp, _i, _i, _vp, _vp, ctypes.c_float, ctypes.c_float, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "hsp_bn_relu_apply_mixed": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "hsp_bn_relu_bwd_mixed": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "hsp_cast_params_bf16": (_i, [_vp, _i, _i, _vp]),
    "hsp_knn_bf16": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "hsp_rf_surface_fwd_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "hsp_rf_surface_bwd_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "hsp_rf_conv_wants_fwin_bf16": (_i, [_i, _i, _i]),
    "hsp_rf_conv_fwd_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "hsp_rf_conv_bwd_scatter_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "hsp_gather_max_fwd_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "hsp_gather_max_bwd_bf16": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "hsp_orl_global_fwd_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "hsp_colsum_rows_bf16": (_i, [_vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "hsp_concat_rows_pitched": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "hsp_concat_rows_bf16": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "hsp_gather_rows_bwd_csr_bf16": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "hsp_wgrad_bf16": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "hsp_bn_relu_fwd_bf16": (_i, [_vp, _i, _i, _vp, _vp, ctypes.c_float, ctypes.c_float, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "hsp_bn_relu_apply_bf16": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "hsp_bn_relu_bwd_bf16": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "hsp_wgrad_workspace_bytes": (_sz, [_i, _i, _i]),
    "hsp_wgrad_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "hsp_bn_workspace_bytes": (_sz, [_i, _i]),
    "hsp_bn_relu_fwd": (_i, [_vp, _i, _i, _vp, _vp, ctypes.c_float, ctypes.c_float, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "hsp_bn_relu_apply": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "hsp_bn_relu_bwd": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "hsp_gemm_x3_bn_f32": (_i, [_vp, _i, _vp, _i, ctypes.c_longlong, _i, _vp, _i, _vp, _i, ctypes.c_longlong, _i, _i, _i, _vp, _i, _vp, _i,
                                _vp, _i, _vp, _vp, _vp]),
    "hsp_gemm_x3_bn_tiles": (_i, [_i, _i]),
    "hsp_gemm_x3_bias_bn_f32": (_i, [_vp, _i, _vp, _i, ctypes.c_longlong, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "hsp_bn_relu_fwd_partials": (_i, [_vp, _i, _i, _vp, _vp, ctypes.c_float, ctypes.c_float, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                                      _vp, _vp]),
    "hsp_bn_relu_bwd2": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "hsp_pc_compact_workspace_bytes": (_sz, [_i, _i]),
    "hsp_pc_compact": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "hsp_pc_gather": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "hsp_depth_to_pcl": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "hsp_generate_rt": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "hsp_sumsq_workspace_bytes": (_sz, [ctypes.c_longlong]),
    "hsp_sumsq_f32": (_i, [_vp, ctypes.c_longlong, _vp, _vp, _sz, _vp]),
    "hsp_ranger_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                             ctypes.c_float, ctypes.c_float, _i, _i, ctypes.c_float, _i, _vp, ctypes.c_float, _vp]),
    "hsp_chamfer_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "hsp_chamfer_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "hsp_fps_workspace_bytes": (_sz, [_i, _i]),
    "hsp_fps_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "hsp_fps_f64": (_i, [_vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "hsp_axis_conf_fwd": (_i, [_vp, _i, _vp, _vp, _vp]),
    "hsp_axis_conf_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "hsp_face_split_fwd": (_i, [_vp, ctypes.c_longlong, _vp, _vp, _vp, _vp]),
    "hsp_face_split_bwd": (_i, [_vp, _vp, _vp, _vp, ctypes.c_longlong, _vp, _vp]),
    "hsp_pose_losses_workspace_bytes": (_sz, [_i]),
    "hsp_pose_losses_fwd": (_i, [_vp] * 17 + [_i, _i, _vp, _vp, _vp, _sz, _vp]),
    "hsp_pose_losses_bwd": (_i, [_vp] * 17 + [_i, _i, _vp, _vp, _vp, _sz] + [_vp] * 11 + [_vp]),
    "hsp_wgrad_partial_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _sz, _vp, _vp]),
    "hsp_wgrad_partial_bf16": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _sz, _vp, _vp]),
    "hsp_wgrad_fold": (_i, [_vp, _i, _vp]),
    "hsp_step_fold": (_i, [_vp, _i, _vp, _i, _vp]),
    "hsp_rf_surface_bwd_partial": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _vp]),
    "hsp_rf_conv_bwd_scatter_partial": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "hsp_rf_surface_bwd_partial_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _vp]),
    "hsp_rf_conv_bwd_scatter_partial_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "hsp_wgrad_partial_pair_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _sz] * 2 + [_vp, _vp]),
    "hsp_pose_augment": (_i, [_vp] * 14 + [_i, _i, _i] + [ctypes.c_float] * 4 + [_vp] * 5),
}


class HspLossCfg(ctypes.Structure):
    """include/hsp.h: HspLossCfg (a HOST struct)"""
    _fields_ = [(n, ctypes.c_float) for n in (
        "rot_1_w", "rot_2_w", "rot_regular", "tran_w", "size_w", "r_con_w", "recon_n_w", "recon_d_w", "recon_f_w",
        "recon_v_w", "recon_bb_r_w", "recon_bb_t_w", "recon_bb_s_w", "recon_bb_self_w", "geo_p_w", "prop_pm_w",
        "prop_sym_w")] + [("smooth_l1", ctypes.c_int)]


class HspSplitDesc(ctypes.Structure):
    """include/hsp.h: HspSplitDesc (the table lives in DEVICE memory; built on the host, uploaded once)"""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("rows", ctypes.c_int), ("cols", ctypes.c_int),
                ("ld", ctypes.c_int), ("transpose", ctypes.c_int), ("kp", ctypes.c_int), ("tile0", ctypes.c_int),
                ("ps", ctypes.c_longlong)]


class HspError(RuntimeError):
    pass


def lib():
    """The loaded libhsp.so with argtypes set.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HspError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C hs_pose_amd/csrc`). "
                "hs_pose_amd has no CPU / eager fallback.")
        # torch first: its wheel carries its own libamdhip64.so, and libhsp.so's NEEDED entry (same soname) must resolve to THAT copy.
        # Loaded before torch, libhsp.so pulls in /opt/rocm's runtime; torch then brings a second one, the device is initialised in one
        # and the kernels are registered in the other: every launch fails with "no ROCm-capable device is detected" (seen when
        # build() and smoke() ran in one process).
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        L = lib()
        msg = L.hsp_error_string(rc).decode()
        hip = L.hsp_last_hip_error().decode()
        raise HspError(f"{what} failed: {msg} (code {rc})" + (f" [hip: {hip}]" if hip and rc == -4 else ""))


class HspWgradPending(ctypes.Structure):
    """include/hsp.h: HspWgradPending (a HOST struct)"""
    _fields_ = [("part", ctypes.c_void_p), ("cs_part", ctypes.c_void_p), ("C", ctypes.c_void_p), ("colsum", ctypes.c_void_p),
                ("nparts", ctypes.c_int), ("M", ctypes.c_int), ("N", ctypes.c_int), ("ldc", ctypes.c_int)]


class HspDirsPending(ctypes.Structure):
    """include/hsp.h: HspDirsPending (a HOST struct)"""
    _fields_ = [("part", ctypes.c_void_p), ("dirs", ctypes.c_void_p), ("grad_dirs", ctypes.c_void_p),
                ("nparts", ctypes.c_int), ("SC", ctypes.c_int)]
