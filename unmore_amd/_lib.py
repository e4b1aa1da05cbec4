"""ctypes binding of the C-ABI library (include/umr.h).  Fails loudly when the
HIP library is missing: there is no CPU / PyTorch fallback in the product path."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UMR_LIB") or os.path.join(_HERE, "lib", "libumr.so")   # UMR_LIB: an instrumented build (tools/probe)
CSRC = os.path.join(_HERE, "csrc")

F32, BF16, BF16X3 = 0, 1, 2
EPI_BIAS, EPI_ADD_AUX, EPI_MASK_RELU, EPI_MASK_DGELU, EPI_ADD_AUX2, EPI_OUT_F32, EPI_ROWBIAS, EPI_OUT_X3 = 1, 2, 4, 8, 16, 32, 64, 128
EPI_AUX_X3, EPI_AUX2_X3, EPI_C2_X3 = 256, 512, 1024
ACT_NONE, ACT_RELU, ACT_GELU, ACT_TANH = 0, 1, 2, 3
ACT_SIGMOID = 5
F32_EXACT, F32_X3, F32_X3_FAST = 0, 1, 2   # umr_f32_mode

_vp, _i32, _i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64


class GemmDesc(ctypes.Structure):
    _fields_ = [("A", _vp), ("B", _vp), ("C", _vp), ("C2", _vp), ("bias", _vp), ("aux", _vp),
                ("aux2", _vp), ("rowbias", _vp),
                ("lda", _i64), ("ldb", _i64), ("ldc", _i64), ("ldc2", _i64), ("ldaux", _i64), ("ldaux2", _i64),
                ("M", _i32), ("N", _i32), ("K", _i32), ("dtype", _i32),
                ("flags", _i32), ("act", _i32), ("c2_mode", _i32), ("rows_per_batch", _i32),
                ("conv", _i32), ("nb", _i32), ("H", _i32), ("W", _i32), ("Cin", _i32), ("Ho", _i32), ("Wo", _i32),
                ("a_rows_in", _i32), ("a_rows_out", _i32), ("a_row_off", _i32),
                ("c_rows_in", _i32), ("c_rows_out", _i32), ("c_row_off", _i32), ("aux_mod", _i32),
                ("red_w", _vp), ("red_out", _vp), ("red_c", _i32), ("no_store", _i32)]


class GemmTnDesc(ctypes.Structure):
    _fields_ = [("dY", _vp), ("X", _vp), ("dW", _vp), ("dbias", _vp), ("workspace", _vp),
                ("workspace_bytes", _i64), ("lddy", _i64), ("ldx", _i64), ("lddw", _i64),
                ("M", _i32), ("N", _i32), ("K", _i32), ("dtype", _i32), ("accumulate", _i32),
                ("conv", _i32), ("nb", _i32), ("H", _i32), ("W", _i32), ("Cin", _i32), ("Ho", _i32), ("Wo", _i32),
                ("dy_rows_in", _i32), ("dy_rows_out", _i32), ("dy_row_off", _i32),
                ("x_rows_in", _i32), ("x_rows_out", _i32), ("x_row_off", _i32)]


class AdamPackEntry(ctypes.Structure):   # umr_adam_pack_entry
    _fields_ = [("p", _vp), ("g", _vp), ("m", _vp), ("v", _vp), ("dst_lin", _vp), ("dst_t", _vp), ("n", _i64), ("N", _i32), ("K", _i32),
                ("blk_start", _i64)]


class PermEntry(ctypes.Structure):   # umr_perm_entry
    _fields_ = [("src", _vp), ("dst", _vp), ("d", _i32 * 4), ("sstride", _i64 * 4), ("soff", _i64), ("dtype_in", _i32), ("dtype_out", _i32),
                ("blk_start", _i64), ("e", _i32 * 4), ("ord", _i32 * 4), ("rowlen", _i64)]


def build(verbose=False, jobs=8):
    """Compile every HIP source for gfx950 into unmore_amd/lib/libumr.so (hipcc
    cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, f"-j{jobs}"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-8000:])
    if r.returncode != 0:
        raise RuntimeError("building libumr.so failed")
    return LIB_PATH


_lib = None
_count = [0]     # launches checked so far (graphs.StagedCaptured skips graph segments that recorded nothing)

_f32 = ctypes.c_float
_SIGS = {
    "umr_gemm_nt": [_vp, _vp],
    "umr_gemm_nt_ws": [_vp, _vp, _i64, _vp],
    "umr_gemm_nt_workspace": [],
    "umr_gemm_nt_splits": [_vp, _i64],
    "umr_gemm_nt_rowreduce_ok": [_vp],
    "umr_label_synthesis_workspace": [_i32, _i32, _i32],
    "umr_label_synthesis": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp],
    "umr_label_synthesis_cropped": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp],
    "umr_crop_resize_batch": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_distance_transform_workspace": [_i32, _i32, _i32],
    "umr_distance_transform": [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp],
    "umr_im2col_nchw": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_maxpool3x3s2": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_bn_fold": [_vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "umr_head_out_finish": [_vp, _i32, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "umr_gemm_tn": [_vp, _vp],
    "umr_gemm_tn_workspace": [_vp],
    "umr_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _i32, _vp],
    "umr_layernorm_bwd_workspace": [_i32, _i32],
    "umr_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _i32, _vp],
    "umr_layernorm_bwd_rows": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "umr_layernorm_bwd_params": [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _vp],
    "umr_attention_fwd": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_attention_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_attention_bwd_workspace": [_i32, _i32, _i32],
    "umr_patchify": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_bilinear_fwd": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_bilinear_bwd": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_bilinear_fwd_ex": [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_bilinear_bwd_ex": [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_pixel_shuffle": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_zero_stuff2": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_permute4": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "umr_segsum": [_vp, _vp, _i32, _i32, _i64, _i64, _i32, _i32, _i32, _i32, _vp],
    "umr_fill_cls": [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp],
    "umr_cast": [_vp, _vp, _i64, _f32, _i32, _i32, _vp],
    "umr_scale_by_device_scalar": [_vp, _vp, _vp, _i64, _vp],
    "umr_head_out_fwd": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_head_out_bwd_workspace": [_i64, _i32],
    "umr_head_out_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_loss_workspace": [],
    "umr_objectness_loss": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp],
    "umr_adam_step": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _i32, _f32, _vp],
    "umr_adam_set_hyper": [_vp, _f32, _f32, _f32, _f32, _i32, _f32, _vp],
    "umr_adam_step_hyper": [_vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "umr_adam_pack_step": [_vp, _i32, _i64, _vp, _vp, _vp],
    "umr_crop_resize_bilinear": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "umr_center_peaks": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_center_peaks_certified": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, ctypes.c_double, _vp],
    "umr_boundary_deltas": [_vp, _vp, _i32, _i32, _i32, _vp],
    "umr_mask_paste_stats": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "umr_mask_paste": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "umr_mask_components": [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "umr_nms_workspace": [_i32],
    "umr_nms": [_vp, _vp, _i32, _f32, _vp, _i64, _vp, _vp, _vp],
    "umr_linear_head_fwd": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_linear_head_bwd_data": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_linear_head_bwd_weight_workspace": [_i64, _i32],
    "umr_linear_head_bwd_weight": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_linear_head_shift9_workspace": [_i64],
    "umr_linear_head_shift9": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp],
    "umr_linear_head_gather9": [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "umr_small_gemm_f32": [_vp, _vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp],
    "umr_permute4_batched": [_vp, _i32, _i64, _vp, _vp],
    "umr_split3": [_vp, _vp, _i64, _i32, _i64, _i64, _vp],
    "umr_unsplit3": [_vp, _vp, _i64, _i32, _i64, _i64, _vp],
    "umr_split3_rows": [_vp, _vp, _i64, _i32, _i64, _i64, _i32, _i32, _i32, _vp],
    "umr_gemm_nt_x3_workspace": [_vp],
    "umr_set_f32_mode": [_i32],
    "umr_get_f32_mode": [],
    "umr_set_cu_budget": [_i32],
    "umr_get_cu_budget": [],
    "umr_set_debug_option": [ctypes.c_char_p, ctypes.c_char_p],
    "umr_get_debug_option": [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)],
    "umr_gemm_nt_ws_status": [_vp, _vp, ctypes.POINTER(ctypes.c_int)],
    "umr_version": [],
    "umr_last_error_string": [],
}


def _set_argtypes(l):
    for name, sig in _SIGS.items():
        getattr(l, name).argtypes = sig


def exported_symbols():
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the product has no CPU fallback)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.umr_last_error_string.restype = ctypes.c_char_p
        _set_argtypes(_lib)
        for fn in ("umr_gemm_tn_workspace", "umr_layernorm_bwd_workspace", "umr_head_out_bwd_workspace", "umr_loss_workspace",
                   "umr_linear_head_bwd_weight_workspace", "umr_linear_head_shift9_workspace", "umr_label_synthesis_workspace", "umr_attention_bwd_workspace",
                   "umr_distance_transform_workspace", "umr_gemm_nt_workspace", "umr_gemm_nt_x3_workspace", "umr_nms_workspace"):
            getattr(_lib, fn).restype = ctypes.c_int64
    return _lib


def check(status, what):
    _count[0] += 1
    if status != 0:
        raise RuntimeError(f"{what} failed with status {status}: {lib().umr_last_error_string().decode()}")
