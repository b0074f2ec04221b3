"""ctypes binding of libtavhip.so (include/tavhip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  `lib()` raises if the shared object has not
been built (run `python -c "import __graft_entry__ as g; g.build()"` or `make -C multi-modal-emotion_amd/csrc`).
Every wrapper turns a non-zero return code into RuntimeError carrying tav_error_string().
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TAV_LIB") or os.path.join(_HERE, "libtavhip.so")      # TAV_LIB: developer knob, A/B of two builds (tools/ab_build.sh)

TAV_F32, TAV_BF16, TAV_FP8 = 0, 1, 2
ABI_VERSION = 6

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmNTArgs(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("C_pre", vp), ("bias", vp), ("gelu_in", vp), ("resid", vp),
                ("M", i64), ("N", i64), ("K", i64),
                ("lda", i64), ("ldb", i64), ("ldc", i64), ("ld_pre", i64), ("ld_gelu_in", i64), ("ld_resid", i64),
                ("nzb", i32), ("nzg", i32),
                ("a_zb", i64), ("a_zg", i64), ("b_zb", i64), ("b_zg", i64), ("c_zb", i64), ("c_zg", i64), ("bias_zg", i64),
                ("in_dtype", i32), ("out_dtype", i32), ("act", i32), ("accumulate", i32), ("alpha", f32),
                ("a_dequant", vp), ("b_dequant", vp), ("gelu_zb", i64), ("tile_m_hint", i32)]


class GemmTNArgs(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("slabs", vp), ("out", vp), ("dbias", vp), ("bias_partials", vp),
                ("N1", i64), ("N2", i64), ("lda", i64), ("ldb", i64), ("rows_per_batch", i64), ("nbatch", i64),
                ("a_zb", i64), ("b_zb", i64), ("chunk_rows", i32), ("nsplit", i32), ("perm_inner", i32), ("perm_outer", i32),
                ("dtype", i32), ("accumulate", i32), ("scale", f32)]


class GemmTNProblem(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("out", vp), ("dbias", vp), ("N1", i64), ("N2", i64), ("lda", i64), ("ldb", i64)]


class AttnArgs(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("o", vp), ("key_mask", vp), ("lse", vp), ("corr", vp), ("o_soft", vp),
                ("dout", vp), ("dq", vp), ("dk", vp), ("dv", vp), ("delta", vp),
                ("B", i64), ("S", i64), ("nheads", i64),
                ("ld_q", i64), ("ld_k", i64), ("ld_v", i64), ("ld_o", i64), ("ld_do", i64), ("ld_dq", i64), ("ld_dk", i64), ("ld_dv", i64),
                ("dtype", i32), ("mask_mode", i32), ("scale", f32), ("q_prescaled", i32)]


class LnArgs(C.Structure):
    _fields_ = [("x", vp), ("x_dtype", i32), ("gamma", vp), ("beta", vp), ("y_f32", vp), ("y_lp", vp), ("lp_dtype", i32),
                ("mean", vp), ("rstd", vp),
                ("dy", vp), ("dy_dtype", i32), ("dx_add", vp), ("dx_f32", vp), ("dx_lp", vp),
                ("dgamma", vp), ("dbeta", vp), ("partials", vp), ("accumulate_params", i32),
                ("rows", i64), ("W", i64), ("ld_x", i64), ("ld_y", i64), ("ld_dy", i64), ("ld_dx", i64), ("eps", f32), ("act", i32),
                ("defer_param_reduce", i32)]


class LnReduceItem(C.Structure):
    _fields_ = [("partials", vp), ("dgamma", vp), ("dbeta", vp), ("nblocks", i32), ("W", i32), ("accumulate", i32), ("pad_", i32)]


LN_REDUCE_MAX = 64


class TextEmbedArgs(C.Structure):
    _fields_ = [("ids", vp), ("word", vp), ("pos", vp), ("type", vp), ("gamma", vp), ("beta", vp), ("pre", vp), ("pos_ids", vp),
                ("y_f32", vp), ("y_lp", vp), ("lp_dtype", i32), ("mean", vp), ("rstd", vp),
                ("B", i64), ("S", i64), ("W", i64), ("vocab", i64), ("max_pos", i64), ("pad_id", i32), ("eps", f32)]


_SIGS = {
    "tav_version": (C.c_int, []),
    "tav_error_string": (C.c_char_p, [C.c_int]),
    "tav_gemm_nt": (C.c_int, [C.POINTER(GemmNTArgs), vp]),
    "tav_gemm_nt_schedule": (C.c_int, [C.POINTER(GemmNTArgs), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "tav_gemm_tn_splits": (C.c_int, [i64, i64, i64, i64, C.POINTER(i32), C.POINTER(i32)]),
    "tav_gemm_tn": (C.c_int, [C.POINTER(GemmTNArgs), vp]),
    "tav_gemm_tn_grouped": (C.c_int, [C.POINTER(GemmTNProblem), i32, i64, i32, vp]),
    "tav_gemm_tn_grouped_ws_bytes": (C.c_int64, [C.POINTER(GemmTNProblem), i32, i64, i32, i32]),
    "tav_gemm_tn_grouped_ws": (C.c_int, [C.POINTER(GemmTNProblem), i32, i64, i32, vp, i64, i32, vp]),
    "tav_colsum": (C.c_int, [vp, i32, i64, i64, i64, vp, i32, vp, i32, vp]),
    "tav_fp8_amax_partials": (C.c_int, [i64, i64]),
    "tav_fp8_amax": (C.c_int, [vp, i32, i64, i64, i64, vp, vp, vp]),
    "tav_fp8_quantize": (C.c_int, [vp, i32, i64, i64, i64, vp, vp, i64, vp, i64, i64, vp]),
    "tav_fp8_quantize_delayed": (C.c_int, [vp, i32, i64, i64, i64, vp, vp, i64, vp, i64, i64, vp]),
    "tav_fp8_roll_states": (C.c_int, [vp, i64, vp]),
    "tav_splitk_reduce": (C.c_int, [vp, vp, i32, i64, i32, vp]),
    "tav_comm_rccl_version": (C.c_int, [C.POINTER(i32)]),
    "tav_comm_unique_id": (C.c_int, [vp]),
    "tav_comm_init_rank": (C.c_int, [C.POINTER(vp), i32, vp, i32]),
    "tav_comm_destroy": (C.c_int, [vp]),
    "tav_allreduce_bucket": (C.c_int, [vp, i64, i32, vp, vp]),
    "tav_attn_fwd": (C.c_int, [C.POINTER(AttnArgs), vp]),
    "tav_attn_bwd": (C.c_int, [C.POINTER(AttnArgs), vp]),
    "tav_attn_probs": (C.c_int, [C.POINTER(AttnArgs), vp, vp, i64, vp]),
    "tav_head_scale": (C.c_int, [vp, vp, vp, i32, vp, i64, f32, i64, i64, i64, i64, i64, i64, vp]),
    "tav_ln_fwd": (C.c_int, [C.POINTER(LnArgs), vp]),
    "tav_ln_bwd": (C.c_int, [C.POINTER(LnArgs), vp]),
    "tav_ln_bwd_partials": (C.c_int, [i64]),
    "tav_ln_param_reduce_multi": (C.c_int, [C.POINTER(LnReduceItem), i32, vp]),
    "tav_cast_weight": (C.c_int, [vp, i64, i64, vp, i64, vp, i64, i32, vp]),
    "tav_cast_weights_multi": (C.c_int, [vp, i32, i32, vp]),
    "tav_cast_conv_weight": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, i64, i32, vp]),
    "tav_zero_pad_rows": (C.c_int, [vp, i32, i64, i64, i64, i64, vp]),
    "tav_cast2d": (C.c_int, [vp, i32, i64, vp, i32, i64, i64, i64, vp]),
    "tav_transpose2d": (C.c_int, [vp, vp, i32, i64, i64, i64, vp]),
    "tav_add_f32": (C.c_int, [vp, vp, vp, vp, i32, i64, vp]),
    "tav_fill_f32": (C.c_int, [vp, f32, i64, vp]),
    "tav_embed_add_fwd": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, vp]),
    "tav_embed_add_bwd": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, i32, vp]),
    "tav_embed_add_bwd_parts": (C.c_int, [i64]),
    "tav_text_embed_fwd": (C.c_int, [C.POINTER(TextEmbedArgs), vp]),
    "tav_scatter_add_rows": (C.c_int, [vp, vp, vp, i64, i64, i64, vp]),
    "tav_gather_rows": (C.c_int, [vp, vp, vp, i64, i64, i64, vp]),
    "tav_patchify": (C.c_int, [vp, vp, vp, i32, i64, i64, i64, i64, i64, vp]),
    "tav_mask_to_index": (C.c_int, [vp, i32, vp, vp, i64, i64, i64, vp]),
    "tav_mean_pool_fwd": (C.c_int, [vp, vp, i64, i64, i64, vp]),
    "tav_mean_pool_bwd": (C.c_int, [vp, vp, vp, i32, i64, i64, i64, vp]),
    "tav_head_fwd": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, vp]),
    "tav_head_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, vp]),
    "tav_tanh_fwd": (C.c_int, [vp, vp, i64, vp]),
    "tav_tanh_bwd": (C.c_int, [vp, vp, vp, i64, vp]),
    "tav_cross_entropy": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, f32, vp]),
    "tav_dropout_fwd": (C.c_int, [vp, vp, vp, i64, f32, C.c_uint64, C.c_uint64, vp]),
    "tav_dropout_bwd": (C.c_int, [vp, vp, vp, i64, f32, vp]),
    "tav_conv0_fwd": (C.c_int, [vp, vp, vp, vp, i32, i64, i64, i64, i64, i64, i64, vp]),
    "tav_conv0_bwd_w": (C.c_int, [vp, vp, i32, vp, vp, vp, i64, i64, i64, i64, i64, i64, i32, vp]),
    "tav_conv0_bwd_partials": (C.c_int, [i64, i64, i64, i64]),
    "tav_gn_workspace_floats": (C.c_int, [i64, i64]),
    "tav_gn_gelu_fwd": (C.c_int, [vp, vp, i32, vp, vp, vp, vp, i64, i64, i64, f32, vp]),
    "tav_gn_gelu_bwd": (C.c_int, [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, vp]),
    "tav_gelu_fwd": (C.c_int, [vp, vp, i32, i64, vp]),
    "tav_gelu_bwd": (C.c_int, [vp, vp, vp, i32, i64, vp]),
    "tav_col2im_1d": (C.c_int, [vp, vp, vp, i32, i64, i64, i64, i64, i64, i64, vp]),
    "tav_group_pad": (C.c_int, [vp, i32, vp, i32, i64, i64, i64, i64, i64, i64, vp]),
    "tav_weight_norm_partials": (C.c_int, [i64, i64]),
    "tav_weight_norm_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i64, i64, i64, vp]),
    "tav_weight_norm_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, vp]),
    "tav_sumsq_partials": (C.c_int, [i32]),
    "tav_sumsq_multi": (C.c_int, [vp, vp, i32, vp, vp, vp]),
    "tav_clip_coef": (C.c_int, [vp, f32, vp, vp, vp]),
    "tav_adamw_multi": (C.c_int, [vp, vp, vp, vp, vp, i32, vp, vp, f32, f32, f32, f32, vp, vp, vp]),
    "tav_optim_chunk_elems": (C.c_int, []),
    "tav_sumsq_chunked": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, vp]),
    "tav_sum_partials": (C.c_int, [vp, i64, vp, vp]),
    "tav_adamw_chunked": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, f32, f32, f32, f32, vp, vp, vp]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; fail loudly when the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the TAV hot path has no fallback. Build it with "
                "`make -C multi-modal-emotion_amd/csrc` (hipcc --offload-arch=gfx950).")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)      # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        if h.tav_version() != ABI_VERSION:
            raise RuntimeError(f"libtavhip ABI {h.tav_version()} != binding {ABI_VERSION}; rebuild")
        _lib = h
    return _lib


def declared_symbols():
    return sorted(_SIGS)


def check(code, what=""):
    if code != 0:
        msg = lib().tav_error_string(code).decode()
        raise RuntimeError(f"libtavhip {what} failed: {msg} (code {code})")


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else t.data_ptr()


def dt(t_or_dtype):
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if d == torch.float32:
        return TAV_F32
    if d == torch.bfloat16:
        return TAV_BF16
    if d == torch.float8_e4m3fn:
        return TAV_FP8
    raise TypeError(f"libtavhip supports float32, bfloat16 and (GEMM operands) float8_e4m3fn tensors, got {d}")


def stream():
    return torch.cuda.current_stream().cuda_stream
