"""Loader for libmumpy_hip.so.  Fails loudly: there is no fallback implementation."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
_LIB = None

c_f = ctypes.c_void_p      # device pointers travel as void*
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_fl = ctypes.c_float
c_d = ctypes.c_double

# name -> argtypes; mirrors include/mumpy_hip.h one to one (tests/test_abi.py checks the header against this)
SIGNATURES = {
    "mumpy_layernorm_fwd": [c_f, c_f, c_f, c_f, c_l, c_i, c_fl, c_f],
    "mumpy_layernorm_bf16_fwd": [c_f, c_f, c_f, c_f, c_l, c_i, c_fl, c_f],
    "mumpy_linear_fwd": [c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_f],
    "mumpy_linear_bf16s_fwd": [c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_f],
    "mumpy_window_attention_bf16_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_f],
    "mumpy_linear_ws_fwd": [c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_f, c_l, c_f],
    "mumpy_linear_wsz_fwd": [c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_f, c_l, c_f],
    "mumpy_linear_rd_fwd": [c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_f],
    "mumpy_window_attention_bg_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_f],
    "mumpy_linear_workspace_bytes": [c_l, c_i, c_i],
    "mumpy_tuning_build": [],
    "mumpy_workspace_status": [c_f, ctypes.POINTER(ctypes.c_int)],
    "mumpy_linear_ln_tiles": [c_l, c_i, c_i],
    "mumpy_linear_lnx_fwd": [c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_f, c_l, c_f, c_f, c_i, c_f, c_fl, c_f],
    "mumpy_linear_rows_fwd": [c_f, c_l, c_l, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_f, c_l, c_f],
    "mumpy_linear_rows_kseg_fwd": [c_f, c_l, c_l, c_i, c_i, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_f, c_l, c_f],
    "mumpy_gn_stats_nhwc_fwd": [c_f, c_f, c_i, c_l, c_i, c_i, c_i, c_f],
    "mumpy_gn_apply_resample_nhwc_fwd": [c_f, c_f, c_i, c_f, c_f, c_i, c_fl, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_i, c_i,
                                         c_i, c_i, c_i, c_i, c_f],
    "mumpy_conv2d_nhwc_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_l, c_f],
    "mumpy_conv2d_workspace_bytes": [c_i, c_i, c_i, c_i, c_i, c_i, c_i],
    "mumpy_avgpool2_pad_nhwc_fwd": [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f],
    "mumpy_copy_rows_fwd": [c_f, c_l, c_f, c_l, c_l, c_i, c_f],
    "mumpy_merge_views_fwd": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f],
    "mumpy_trunk_head_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f],
    "mumpy_final_conv_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_fl, c_f],
    "mumpy_window_attention_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_f],
    "mumpy_deform_offsets_fwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f],
    "mumpy_deform_sample_fwd": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f],
    "mumpy_deform_attention_fwd": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_fl, c_f],
    "mumpy_deform_sample_kv_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f],
    "mumpy_deform_out_combine_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f],
    "mumpy_deform_combine_fwd": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f],
    "mumpy_faf_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f],
    "mumpy_patch_embed_fwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_f],
    "mumpy_patch_merge_ln_fwd": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_fl, c_f],
    "mumpy_temporal_attention_fwd": [c_f, c_f, c_l, c_i, c_i, c_i, c_fl, c_f],
    "mumpy_attention_probs_fwd": [c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_l, c_l, c_fl, c_f],
    "mumpy_temporal_attention_q_fwd": [c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_fl, c_f],
    "mumpy_sigmoid_threshold_fwd": [c_f, c_f, c_l, c_fl, c_f],
    "mumpy_add_fwd": [c_f, c_f, c_f, c_l, c_f],
    "mumpy_normalize_u8_fwd": [c_f, c_f, c_l, c_i, c_i, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), c_f],
    "mumpy_resize_nearest_table": [c_i, c_i, ctypes.POINTER(ctypes.c_int32)],
    "mumpy_resize_normalize_u8_fwd": [c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_i, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), c_f],
    "mumpy_mask_loss_workspace_bytes": [c_i, c_l],
    "mumpy_mask_loss_fwd_bwd": [c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_l, c_fl, c_fl, c_f],
    "mumpy_layernorm_bwd_workspace_bytes": [c_l, c_i],
    "mumpy_layernorm_bwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_l, c_l, c_i, c_fl, c_i, c_f],
    "mumpy_gelu_fwd": [c_f, c_f, c_l, c_f],
    "mumpy_gelu_bwd": [c_f, c_f, c_f, c_l, c_f],
    "mumpy_transpose_fwd": [c_f, c_f, c_l, c_l, c_f],
    "mumpy_col_sum_workspace_bytes": [c_l, c_i],
    "mumpy_col_sum_fwd": [c_f, c_f, c_f, c_l, c_l, c_i, c_f],
    "mumpy_final_conv_bwd_workspace_bytes": [c_i, c_i, c_i],
    "mumpy_final_conv_bwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_f],
    "mumpy_patch_gather_fwd": [c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_f],
    "mumpy_conv_weight_dgrad_fwd": [c_f, c_f, c_i, c_i, c_i, c_i, c_f],
    "mumpy_conv2d_wgrad_workspace_bytes": [c_i, c_i, c_i, c_i, c_i, c_i, c_i],
    "mumpy_conv2d_wgrad_nhwc": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_l, c_f],
    "mumpy_linear_bwd_workspace_bytes": [c_l, c_i, c_i],
    "mumpy_linear_bwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_f, c_l, c_f],
    "mumpy_window_attention_bwd_workspace_bytes": [c_i, c_i, c_i, c_i],
    "mumpy_window_attention_bwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_i, c_fl, c_i, c_f],
    "mumpy_window_attention_bwd_csr": [c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_i, c_fl, c_i, c_f],
    "mumpy_relpos_bias_expand_fwd": [c_f, c_f, c_f, c_i, c_f],
    "mumpy_gn_bwd_workspace_bytes": [c_i, c_l, c_i],
    "mumpy_gn_bwd_nhwc": [c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_l, c_i, c_i, c_fl, c_i, c_f],
    "mumpy_upsample_bwd_nhwc": [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f],
    "mumpy_temporal_attention_bwd": [c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_fl, c_f],
    "mumpy_scale_samples_fwd": [c_f, c_f, c_f, c_i, c_l, c_f],
    "mumpy_dwconv5_window_fwd": [c_f, c_f, c_f, c_f, c_l, c_i, c_f],
    "mumpy_dwconv5_window_bwd_workspace_bytes": [c_l, c_i],
    "mumpy_dwconv5_window_bwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_l, c_l, c_i, c_f],
    "mumpy_deform_attention_bwd_workspace_bytes": [c_l, c_i],
    "mumpy_deform_attention_bwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_l, c_l, c_i, c_i, c_fl, c_f],
    "mumpy_deform_sample_bwd": [c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_f],
    "mumpy_adamw_hyper": [ctypes.POINTER(ctypes.c_float), c_d, c_d, c_d, c_d, c_d, c_i, c_d],
    "mumpy_adamw_step_dev": [c_f, c_f, c_f, c_f, c_l, c_f, c_f],
    "mumpy_adamw_step": [c_f, c_f, c_f, c_f, c_l, c_d, c_d, c_d, c_d, c_d, c_i, c_d, c_f],
}
ABI_VERSION = 2


def library_path() -> str:
    """MUMPY_HIP_LIB=<file> overrides; MUMPY_TUNING=1 selects the diagnostics build (tools/gemm_shapes.py & co: the only build
    whose kernels' A/B hooks read MUMPY_GEMM_* / MUMPY_WA_* variables).  This choice is the binding's, not the library's."""
    if "MUMPY_HIP_LIB" in os.environ:
        return os.environ["MUMPY_HIP_LIB"]
    name = "libmumpy_hip_tuning.so" if os.environ.get("MUMPY_TUNING", "0") == "1" else "libmumpy_hip.so"
    return os.path.join(_PKG, "lib", name)


def tuning_library_path() -> str:
    """The diagnostics build (same sources, -DMUMPY_TUNING): its planner / kernel A/B hooks read MUMPY_* environment variables.
    Select it for a process with MUMPY_HIP_LIB=<this path>; the shipped library reads no environment variable."""
    return os.path.join(_PKG, "lib", "libmumpy_hip_tuning.so")


def build_library(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", os.path.join(_PKG, "csrc"), "-j8", "all", "tuning"], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode:
        raise RuntimeError("building libmumpy_hip.so failed (see output above)")
    return library_path()


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"libmumpy_hip.so not found at {path}: the HIP kernels are the only implementation of this package "
            "(no CPU/torch fallback). Build it with `python __graft_entry__.py` or `make -C <pkg>/csrc`.")
    lib = ctypes.CDLL(path)
    lib.mumpy_abi_version.restype = c_i
    lib.mumpy_last_error.restype = ctypes.c_char_p
    if lib.mumpy_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {lib.mumpy_abi_version()} != binding {ABI_VERSION}; rebuild")
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = the library is stale: fail loudly
        fn.argtypes = args
        fn.restype = c_l if name.endswith("_bytes") else c_i
    _LIB = lib
    return lib
