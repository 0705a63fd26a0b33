"""ctypes binding of libmic_hip.so (C ABI in include/mic_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmic_hip.so")

MIC_BF16, MIC_F32, MIC_FP8 = 0, 1, 2
MIC_E4M3, MIC_E5M2 = 0, 1
ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH, ACT_QUICK_GELU = 0, 1, 2, 3
ACT_IDS = {"none": 0, None: 0, "gelu": 1, "erf": 1, "gelu_erf": 1, "tanh": 2, "gelu_tanh": 2, "gelu_new": 2, "quick_gelu": 3}


class MicError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("c_dtype", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("a_kmajor", C.c_int), ("b_kmajor", C.c_int),
        ("A", C.c_void_p), ("lda", C.c_int), ("B", C.c_void_p), ("ldb", C.c_int), ("C", C.c_void_p), ("ldc", C.c_int),
        ("bias", C.c_void_p), ("act", C.c_int), ("Zout", C.c_void_p), ("ldz", C.c_int), ("Zin", C.c_void_p),
        ("dact", C.c_int), ("R", C.c_void_p), ("ldr", C.c_int), ("accumulate", C.c_int), ("dropout_p", C.c_float),
        ("dropout_seed", C.c_uint32), ("alpha", C.c_float), ("split_k", C.c_int),
        ("split_stride", C.c_longlong), ("a_rowsum", C.c_void_p), ("rowsum_k", C.c_int),
        ("a_fmt", C.c_int), ("b_fmt", C.c_int), ("a_scale_inv", C.c_void_p), ("b_scale_inv", C.c_void_p),
        ("rowstat", C.c_void_p), ("rowstat_ld", C.c_int), ("rowstat_nvalid", C.c_int),
        ("a_ln_stats", C.c_void_p), ("a_ln_colsum", C.c_void_p), ("a_ln_width", C.c_int), ("a_ln_eps", C.c_float),
        ("rowsum2", C.c_void_p), ("k_valid", C.c_int),
        ("c_q8_state", C.c_void_p), ("c_q8_amax", C.c_void_p), ("c_q8_fmt", C.c_int),
    ]


class Fp8Item(C.Structure):
    _fields_ = [("src", C.c_void_p), ("ld", C.c_int), ("rows", C.c_int), ("cols", C.c_int), ("rows_pad", C.c_int),
                ("q", C.c_void_p), ("ldq", C.c_int), ("qT", C.c_void_p), ("ldqT", C.c_int), ("state", C.c_void_p), ("amax_next", C.c_void_p),
                ("fmt", C.c_int)]


class Fp8Out(C.Structure):
    _fields_ = [("q", C.c_void_p), ("ldq", C.c_int), ("state", C.c_void_p), ("amax_next", C.c_void_p), ("fmt", C.c_int)]


class ColsumQ8Item(C.Structure):
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("scale_inv", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld", C.c_int), ("fmt", C.c_int)]


class ColsumItem(C.Structure):
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld", C.c_int)]


class BeamStepArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("K", C.c_int), ("max_len", C.c_int), ("V", C.c_int), ("cur_len", C.c_int),
        ("eos_token_id", C.c_int), ("pad_token_id", C.c_int), ("length_penalty", C.c_float), ("early_stopping", C.c_int),
        ("cand_val", C.c_void_p), ("cand_idx", C.c_void_p), ("running_seq", C.c_void_p), ("running_scores", C.c_void_p),
        ("seq", C.c_void_p), ("scores", C.c_void_p), ("finished", C.c_void_p), ("src_row", C.c_void_p),
        ("next_token", C.c_void_p), ("flags", C.c_void_p), ("gstate", C.c_void_p),
    ]


class GemmPlanInfo(C.Structure):
    _fields_ = [("tile", C.c_int), ("kgroups", C.c_int), ("blocks", C.c_int), ("grid", C.c_int), ("blocks_per_cu", C.c_int),
                ("phased", C.c_int), ("cu_budget", C.c_int), ("tile_m", C.c_int)]


class LnParamItem(C.Structure):
    _fields_ = [("partials", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("nblk", C.c_int), ("width", C.c_int), ("accumulate", C.c_int)]


class ImageItem(C.Structure):
    _fields_ = [("src", C.c_void_p), ("H", C.c_int), ("W", C.c_int), ("hwc", C.c_int)]


_i, _f, _p, _u32, _i64 = C.c_int, C.c_float, C.c_void_p, C.c_uint32, C.c_int64
_SIGS = {
    "mic_version": ([], C.c_int),
    "mic_last_error": ([], C.c_char_p),
    "mic_gemm": ([C.POINTER(GemmArgs), _p], C.c_int),
    "mic_gemm_grouped": ([C.POINTER(GemmArgs), _i, _p], C.c_int),
    "mic_set_cu_budget": ([_i], C.c_int),
    "mic_get_cu_budget": ([], C.c_int),
    "mic_gemm_plan": ([C.POINTER(GemmArgs), _i, C.POINTER(GemmPlanInfo)], C.c_int),
    "mic_comm_emulate": ([_p, _p, _i64, _f, _i, _p], C.c_int),
    "mic_ln_fold_weight": ([_i, _i, _i, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p], C.c_int),
    "mic_fp8_amax": ([C.POINTER(Fp8Item), _i, _p], C.c_int),
    "mic_fp8_quantize": ([C.POINTER(Fp8Item), _i, _p], C.c_int),
    "mic_fp8_amax_partials": ([], C.c_int),
    "mic_fp8_roll_amax": ([_p, _i, _p, _i, _p], C.c_int),
    "mic_sum_slabs": ([_i, _i, C.c_longlong, _i, _i, _p, _i, _p, _i, _p], C.c_int),
    "mic_layernorm_fwd": ([_i, _i, _i, _p, _p, _p, _f, _p, _p, _p, _f, _u32, _p], C.c_int),
    "mic_layernorm_bwd": ([_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _u32, _f, _u32, _p], C.c_int),
    "mic_layernorm_bwd_blocks": ([_i], C.c_int),
    "mic_layernorm_bwd_partials": ([_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _u32, _f, _u32, _p], C.c_int),
    "mic_ln_param_grads": ([C.POINTER(LnParamItem), _i, _p], C.c_int),
    "mic_layernorm_fwd_q8": ([_i, _i, _p, _p, _p, _f, _p, _p, _p, _f, _u32, C.POINTER(Fp8Out), _p], C.c_int),
    "mic_layernorm_bwd_partials_q8": ([_i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _u32, _f, _u32, C.POINTER(Fp8Out), _i, _p], C.c_int),
    "mic_attn_bwd_q8": ([_i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _p, _i, C.POINTER(Fp8Out), C.POINTER(Fp8Out), _p, _p], C.c_int),
    "mic_colsum_q8_grouped": ([C.POINTER(ColsumQ8Item), _i, _p], C.c_int),
    "mic_attn_fwd": ([_i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _p], C.c_int),
    "mic_attn_probs": ([_i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _p], C.c_int),
    "mic_attn_bwd": ([_i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p], C.c_int),
    "mic_attn_fwd_packed": ([_i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _p, _p], C.c_int),
    "mic_attn_bwd_packed": ([_i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p], C.c_int),
    "mic_attn_decode": ([_i, _i, _i, _i, _i, _p, _i, _p, _p, _i, _p, _i, _p, _i, _p], C.c_int),
    "mic_kv_append": ([_i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p], C.c_int),
    "mic_im2col": ([_i, _i, _i, _i, _p, _p, _i, _i, _p], C.c_int),
    "mic_vit_assemble": ([_i, _i, _i, _i, _p, _i, _p, _p, _p, _p], C.c_int),
    "mic_vit_assemble_bwd": ([_i, _i, _i, _i, _p, _p, _i, _p, _p, _p], C.c_int),
    "mic_embed_fwd": ([_i, _i, _i, _p, _p, _p, _p, _f, _p, _p], C.c_int),
    "mic_embed_bwd": ([_i, _i, _i, _p, _p, _p, _f, _p, _p, _p], C.c_int),
    "mic_embed_rows_add_det": ([_i, _i, _i, _i, _p, _p, _f, _p, _p, _p], C.c_int),
    "mic_embed_rows_add_det_ws": ([_i], C.c_int64),
    "mic_ce_rows": ([_i, _i, _i, _p, _i, _p, _p, _f, _p, _p, _p], C.c_int),
    "mic_ce_rows_tiles": ([_i, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p], C.c_int),
    "mic_row_topk_tiles": ([_i, _i, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p], C.c_int),
    "mic_ce_reduce": ([_i, _p, _p, _p, _p, _p], C.c_int),
    "mic_ce_bwd": ([_i, _i, _i, _i, _p, _i, _p, _p, _f, _p, _p, _f, _p], C.c_int),
    "mic_ce_bwd_t": ([_i, _i, _i, _p, _i, _p, _p, _f, _p, _p, _f, _p, _i, _i, _p, _p], C.c_int),
    "mic_ce_bwd_q8": ([_i, _i, _i, _p, _i, _p, _p, _f, _p, _p, _f, C.POINTER(Fp8Out), _p, _p, _p], C.c_int),
    "mic_head_label_terms": ([_i, _i, _p, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p], C.c_int),
    "mic_transpose_bf16": ([_i, _i, _i, _p, _i, _p, _i, _p], C.c_int),
    "mic_colsum": ([_i, _i, _i, _p, _i, _p, _i, _p], C.c_int),
    "mic_colsum_grouped": ([_i, C.POINTER(ColsumItem), _i, _p], C.c_int),
    "mic_dropout_mask": ([_p, _i64, _f, _u32, _p], C.c_int),
    "mic_cast": ([_i, _i, _p, _p, _i64, _p], C.c_int),
    "mic_zero": ([_p, _i64, _p], C.c_int),
    "mic_stream_create_cu_masked": ([_i, _i, C.POINTER(C.c_void_p)], C.c_int),
    "mic_stream_destroy": ([_p], C.c_int),
    "mic_copy_rows": ([_i, _i, _i, _p, _i, _p, _p, _i, _p, _p], C.c_int),
    "mic_cast2d": ([_i, _i, _i, _i, _p, _i, _p, _i, _p], C.c_int),
    "mic_adamw": ([_i64, _p, _p, _p, _p, _p, _p, C.c_double, C.c_double, C.c_double, C.c_double, _f, _p], C.c_int),
    "mic_adamw_rows": ([_i64, _i, _p, _i, _p, _p, _p, _p, _p, _p, C.c_double, C.c_double, C.c_double, C.c_double, _f, _p], C.c_int),
    "mic_row_flags": ([_p, _i, _p, _i, _p], C.c_int),
    "mic_row_lse_topk": ([_i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p], C.c_int),
    "mic_beam_step": ([C.POINTER(BeamStepArgs), _p], C.c_int),
    "mic_greedy_step": ([_i, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p], C.c_int),
    "mic_sample_rows": ([_i, _i, _i, _p, _i, _u32, _u32, _f, _i, _i, _i, _p, _p, _p, _p], C.c_int),
    "mic_warp_thresholds": ([_i, _i, _i, _p, _i, _f, _i, _i, _i, _f, _p, _p, _p], C.c_int),
    "mic_image_transform": ([C.POINTER(ImageItem), _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _p, _i, _p], C.c_int),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib() -> C.CDLL:
    """Load libmic_hip.so (once).  Raises MicError when it has not been built — there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MicError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  The HIP path has no fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(l, name)  # AttributeError if the symbol is not exported
            fn.argtypes, fn.restype = args, res
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise MicError(f"{what} failed (rc={rc}): {lib().mic_last_error().decode()}")
