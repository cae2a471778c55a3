"""ctypes binding of libovqa_hip.so (the C ABI in include/ovqa_hip.h).

There is deliberately no fallback: if the library is missing or a kernel
reports an error, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libovqa_hip.so")

OVQA_F32, OVQA_BF16 = 0, 1
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL = 0, 1, 2
ABI_VERSION = 10

c_i64, c_int, c_f32, c_vp = C.c_int64, C.c_int, C.c_float, C.c_void_p


class WgradProblem(C.Structure):
    """ovqa_wgrad_problem (include/ovqa_hip.h)."""
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p),
                ("lddy", C.c_int64), ("ldx", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("accumulate", C.c_int32)]


class AdamTarget(C.Structure):
    """ovqa_adam_target (include/ovqa_hip.h): where the Adam epilogue of the grouped weight-gradient launch finds a problem's
    master weights, moments and bf16 shadows (param = NULL: plain gradient store)."""
    _fields_ = [("param", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("shadow", C.c_void_p),
                ("transposed", C.c_void_p), ("ld_transposed", C.c_int64)]


class AdamConsts(C.Structure):
    """ovqa_adam_consts (include/ovqa_hip.h)."""
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("grad_scale", C.c_float), ("lr_scale_ptr", C.c_void_p),
                ("step_ptr", C.c_void_p)]


class TransposeProblem(C.Structure):
    """ovqa_transpose_problem (include/ovqa_hip.h)."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("ld_src", C.c_int64), ("ld_dst", C.c_int64),
                ("rows", C.c_int32), ("cols", C.c_int32)]


class AdamTile(C.Structure):
    """ovqa_adam_tile (include/ovqa_hip.h)."""
    _fields_ = [("off", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32), ("r0", C.c_int32), ("c0", C.c_int32),
                ("reserved", C.c_int32 * 2)]


class ReduceProblem(C.Structure):
    """ovqa_reduce_problem (include/ovqa_hip.h)."""
    _fields_ = [("partial", C.c_void_p), ("out0", C.c_void_p), ("out1", C.c_void_p),
                ("blocks", C.c_int32), ("D", C.c_int32), ("accumulate", C.c_int32), ("reserved_", C.c_int32)]


class LnRef(C.Structure):
    """ovqa_ln_ref (include/ovqa_hip.h): a LayerNorm to recompute in the fp32 residual epilogue."""
    _fields_ = [("mean", C.c_void_p), ("rstd", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p)]


class GatherProblem(C.Structure):
    """ovqa_gather_problem (include/ovqa_hip.h)."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_bytes", C.c_int64),
                ("src_stride_bytes", C.c_int64), ("dst_stride_bytes", C.c_int64)]


class Dropout(C.Structure):
    _fields_ = [("p", C.c_float), ("seed", C.c_uint32), ("site", C.c_uint32), ("step", C.c_void_p)]


_DP = C.POINTER(Dropout)

# name -> argtypes (restype is int unless listed in _RESTYPE)
SIGNATURES = {
    "ovqa_abi_version": [],
    "ovqa_last_error": [],
    "ovqa_last_dispatch": [],
    "ovqa_workspace_bytes": [],
    "ovqa_stream_priority_range": [c_vp, c_vp],
    "ovqa_stream_create": [c_vp, c_int, c_vp, c_int],
    "ovqa_stream_destroy": [c_vp],
    "ovqa_linear_fwd": [c_int, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp,
                        c_i64, c_i64, c_i64, _DP, c_vp],
    "ovqa_linear_fwd_res32": [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, C.POINTER(LnRef), c_vp, c_i64, c_i64, c_i64, c_i64,
                              _DP, c_vp],
    "ovqa_linear_bwd_data": [c_int, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, _DP, c_vp],
    "ovqa_grouped_linear_bwd_weight": [c_int, c_vp, c_vp, c_i64, c_int, c_vp],
    "ovqa_grouped_linear_bwd_weight_adam": [c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp],
    "ovqa_bias_grad": [c_int, c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_vp],
    "ovqa_linear_bwd_data_wt": [c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64,
                                _DP, c_vp],
    "ovqa_grouped_transpose": [c_vp, c_int, c_int, c_vp],
    "ovqa_linear_bwd_weight": [c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp, c_vp],
    "ovqa_layernorm_fwd": [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_f32,
                           c_vp],
    "ovqa_layernorm_bwd": [c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                           c_i64, c_i64, c_int, _DP, c_vp, c_vp],
    "ovqa_layernorm_bwd_blocks": [c_i64, c_i64],
    "ovqa_grouped_partial_reduce": [c_vp, c_int, c_int, c_int, c_vp],
    "ovqa_attention_fwd": [c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64,
                           c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, _DP, c_vp],
    "ovqa_attention_fwd_prefix_lm": [c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64,
                                     c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_f32, c_vp],
    "ovqa_launch_timing_begin": [c_int],
    "ovqa_launch_timing_count": [],
    "ovqa_launch_timing_end": [c_vp, c_int],
    "ovqa_attention_qkv_fwd": [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp,
                               c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_vp],
    "ovqa_attention_q_fwd": [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64,
                             c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_vp],
    "ovqa_attention_decode": [c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64,
                              c_i64, c_i64, c_i64, c_i64, c_f32, c_vp],
    "ovqa_topk_rows": [c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp],
    "ovqa_linear_fwd_split3": [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64,
                               c_vp],
    "ovqa_decode_embed": [c_int, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_f32, c_vp, c_i64, c_i64,
                          c_vp, c_vp, c_i64, c_i64, c_vp],
    "ovqa_beam_candidates": [c_int, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp],
    "ovqa_beam_commit": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64,
                         c_i64, c_i64, c_vp],
    "ovqa_attention_bwd": [c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp,
                           c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp,
                           c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, _DP, c_vp],
    "ovqa_attention_bwd_do": [c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64,
                              c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp,
                              c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_vp],
    "ovqa_pointer_score": [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_f32, c_vp],
    "ovqa_batched_gemm": [c_int, c_int, c_int, c_int, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64,
                          c_i64, c_i64, c_i64, c_i64, c_f32, c_vp],
    "ovqa_adam_step": [c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp],
    "ovqa_adam_step_tiled": [c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_i64, c_f32, c_vp, c_f32, c_f32,
                             c_f32, c_f32, c_f32, c_vp, c_vp],
    "ovqa_increment_step": [c_vp, c_vp],
    "ovqa_increment_steps": [c_vp, c_vp, c_vp],
    "ovqa_begin_step": [c_vp, c_vp, c_vp, c_int, c_vp, c_vp],
    "ovqa_cast": [c_int, c_int, c_vp, c_vp, c_i64, c_vp],
    "ovqa_gelu_bwd": [c_int, c_vp, c_vp, c_vp, c_i64, _DP, c_vp],
    "ovqa_row_padding_mask": [c_int, c_vp, c_vp, c_i64, c_i64, c_f32, c_vp],
    "ovqa_dropout_keep_mask": [_DP, c_vp, c_i64, c_vp],
    "ovqa_grouped_row_gather": [c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp],
    "ovqa_sq_loss_fwd_bwd": [c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp],
    "ovqa_embed_gather": [c_int, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_vp, c_i64, c_vp],
    "ovqa_decoder_inputs": [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp],
    "ovqa_embed_scatter": [c_int, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_i64, c_int, c_vp],
    "ovqa_dropout_apply": [c_int, c_vp, c_vp, c_i64, _DP, c_vp],
    "ovqa_pool_fwd": [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, _DP, c_vp],
    "ovqa_pool_bwd": [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, _DP, c_vp],
    "ovqa_log_softmax_fwd": [c_int, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp],
    "ovqa_log_softmax_bwd": [c_int, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp],
    "ovqa_nll_loss": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp],
    "ovqa_lstm_saved_bytes": [c_i64, c_i64, c_i64],
    "ovqa_lstm_scratch_bytes": [c_i64, c_i64, c_i64],
    "ovqa_lstm_persistent_max_batch": [],
    "ovqa_lstm_status": [c_vp, c_vp],
    "ovqa_lstm_fwd": [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp],
    "ovqa_lstm_bwd": [c_int, c_vp, c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp],
}
_RESTYPE = {"ovqa_last_error": C.c_char_p, "ovqa_last_dispatch": C.c_char_p, "ovqa_workspace_bytes": C.c_int64,
            "ovqa_lstm_saved_bytes": C.c_int64, "ovqa_lstm_scratch_bytes": C.c_int64,
            "ovqa_lstm_persistent_max_batch": C.c_int64}

_lock = threading.Lock()
_lib = None


def load(path: str | None = None):
    """Load the shared library and declare every prototype.  Raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    with _lock:
        if _lib is not None and path is None:
            return _lib
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise RuntimeError(
                f"HIP kernel library not found at {p}; build it with "
                "`python -m openvivqa_amd.build` (needs hipcc, targets gfx950). There is no CPU fallback.")
        lib = C.CDLL(p)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = _RESTYPE.get(name, C.c_int)
        got = lib.ovqa_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"libovqa_hip.so ABI version {got} != expected {ABI_VERSION}; rebuild")
        if path is None:
            _lib = lib
        return lib


def last_dispatch() -> str:
    """Kernel family ("mfma" | "simple" | "") the last C-ABI call of this thread ran (ovqa_last_dispatch)."""
    v = load().ovqa_last_dispatch()
    return v.decode() if v else ""


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ovqa_last_error()
        raise RuntimeError(f"ovqa kernel error {rc} in {what}: {msg.decode() if msg else '?'}")
