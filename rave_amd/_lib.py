"""ctypes binding of librave_hip.so (the C ABI declared in include/rave_hip.h).

The product path has NO fallback: if the shared library is missing or fails to load, importing
this module raises, and every op raises on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: librave_hip.so binds to the libamdhip64 torch loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAVE_HIP_LIB: diagnostics only (tools/x6_variants.sh loads timing-ablation builds of the same ABI)
LIB_PATH = os.environ.get("RAVE_HIP_LIB") or os.path.join(_HERE, "librave_hip.so")

ACT_NONE, ACT_LEAKY, ACT_SNAKE = 0, 1, 2


class ConvDesc(C.Structure):
    """Mirror of ``rh_conv1d_desc`` (include/rave_hip.h)."""
    _fields_ = [
        ("batch", C.c_int32), ("c_in", C.c_int32), ("c_out", C.c_int32),
        ("l_in", C.c_int32), ("l_out", C.c_int32), ("kernel", C.c_int32),
        ("stride", C.c_int32), ("dilation", C.c_int32), ("pad_left", C.c_int32),
        ("transposed", C.c_int32), ("groups", C.c_int32), ("inner", C.c_int32),
        ("in_valid", C.c_int32), ("act", C.c_int32), ("act_slope", C.c_float),
        ("out_act", C.c_int32), ("out_slope", C.c_float),
    ]


class AdamItem(C.Structure):
    """Mirror of ``rh_adam_item`` (include/rave_hip.h)."""
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64), ("step", C.c_void_p)]


class WnBwdItem(C.Structure):
    """Mirror of ``rh_wn_bwd_item`` (include/rave_hip.h)."""
    _fields_ = [("dw", C.c_void_p), ("v", C.c_void_p), ("g", C.c_void_p), ("norms", C.c_void_p), ("dv", C.c_void_p),
                ("dg", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int64)]


class ReduceItem(C.Structure):
    """Mirror of ``rh_reduce_item`` (include/rave_hip.h)."""
    _fields_ = [("part", C.c_void_p), ("out", C.c_void_p), ("n", C.c_int64), ("Z", C.c_int32), ("reserved", C.c_int32)]


class LossItem(C.Structure):
    """Mirror of ``rh_loss_item`` (include/rave_hip.h)."""
    _fields_ = [("value", C.c_void_p), ("w1_dev", C.c_void_p), ("w1", C.c_float), ("w2", C.c_float)]


class FmItem(C.Structure):
    """Mirror of ``rh_fm_item`` (include/rave_hip.h)."""
    _fields_ = [("f", C.c_void_p), ("df", C.c_void_p), ("half", C.c_int64), ("w", C.c_float)]


class Conv2dDesc(C.Structure):
    """Mirror of ``rh_conv2d_desc`` (include/rave_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "batch", "c_in", "c_out", "h_in", "w_in", "h_out", "w_out", "kh", "kw", "sh", "sw", "dh", "dw", "ph", "pw",
        "act")] + [("act_slope", C.c_float)]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"rave_amd: {LIB_PATH} not found. Build it with `python -m rave_amd.build` "
            "(or __graft_entry__.build()). There is no CPU / PyTorch fallback for the HIP hot path.")
    lib = C.CDLL(LIB_PATH)
    P, I32, I64, F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    D = C.POINTER(ConvDesc)
    D2 = C.POINTER(Conv2dDesc)
    sig = {
        "rh_version": ([], C.c_int),
        "rh_last_error": ([], C.c_char_p),
        "rh_weight_norm_fwd_f32": ([P, P, I64, I64, P, P, P], C.c_int),
        "rh_weight_norm_bwd_f32": ([P, P, P, P, I64, I64, P, P, P], C.c_int),
        "rh_weight_norm_bwd_batched_f32": ([C.POINTER(WnBwdItem), I32, P], C.c_int),
        "rh_defer_reduce": ([C.POINTER(ReduceItem)], C.c_int),
        "rh_reduce_partials_batched_f32": ([C.POINTER(ReduceItem), I32, P], C.c_int),
        "rh_conv1d_packed_floats": ([D, C.c_int], I64),
        "rh_conv1d_pack_f32": ([D, P, P, P, P], C.c_int),
        "rh_conv1d_pack_wn_f32": ([D, P, P, P, P, P, P, P], C.c_int),
        "rh_prep_item_bytes": ([], I64),
        "rh_prep_array_bytes": ([I32], I64),
        "rh_prep_fill_item": ([D, P, P, P, P, P, P, P], C.c_int),
        "rh_prep_link": ([P, I32, C.POINTER(I64), C.POINTER(I64)], C.c_int),
        "rh_prep_run_f32": ([P, I32, I64, I64, P], C.c_int),
        "rh_conv1d_fwd_f32": ([D, P, P, P, P, P, P, P, I64, P], C.c_int),
        "rh_conv1d_bwd_data_f32": ([D, P, P, P, P, P, P, P, I64, P], C.c_int),
        "rh_conv1d_kernel_family": ([D, C.c_int, C.c_int, C.c_int], C.c_int),
        "rh_conv1d_bwd_weight_kernel_family": ([D], C.c_int),
        "rh_residual_unit_fused": ([D, D], C.c_int),
        "rh_residual_unit_fwd_f32": ([D, D, P, P, P, P, P, P], C.c_int),
        "rh_conv1d_plan_info": ([D, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)], C.c_int),
        "rh_conv1d_fwd_workspace_bytes": ([D], I64),
        "rh_conv1d_bwd_data_workspace_bytes": ([D], I64),
        "rh_conv1d_workspace_bytes": ([D], I64),
        "rh_conv1d_bwd_weight_f32": ([D, P, P, P, P, P, P, I64, P], C.c_int),
        "rh_conv1d_bwd_weight_wn_f32": ([D, P, P, P, P, P, P, P, P, P, P, P, I64, P], C.c_int),
        "rh_conv1d_bwd_weight_wn_fused_launches": ([], I64),
        "rh_conv2d_packed_floats": ([D2, C.c_int], I64),
        "rh_conv2d_pack_f32": ([D2, P, P, P, P], C.c_int),
        "rh_conv2d_fwd_f32": ([D2, P, P, P, P, P], C.c_int),
        "rh_conv2d_bwd_data_f32": ([D2, P, P, P, P, P], C.c_int),
        "rh_conv2d_workspace_bytes": ([D2], I64),
        "rh_conv2d_plan_info": ([D2, I32, C.POINTER(I64)], C.c_int),
        "rh_conv2d_bwd_weight_kernel_family": ([D2], C.c_int),
        "rh_conv2d_bwd_weight_f32": ([D2, P, P, P, P, P, P, I64, P], C.c_int),
        "rh_feed_batch_i16_f32": ([P, P, P, P, I32, I32, I32, P, P], C.c_int),
        "rh_vq_loss_partials": ([I64], I64),
        "rh_vq_assign_f32": ([P, P, I64, I32, I32, P, P, P, P, P], C.c_int),
        "rh_vq_assign_workspace_bytes": ([I64, I32, I32], I64),
        "rh_vq_assign_ws_f32": ([P, P, I64, I32, I32, P, P, P, P, P, I64, P], C.c_int),
        "rh_vq_ema_update_f32": ([P, P, I64, I32, I32, F, F, P, P, P, P], C.c_int),
        "rh_pqmf_analysis_fwd_f32": ([P, P, I32, I32, I32, I32, I32, I32, P, P], C.c_int),
        "rh_pqmf_analysis_bwd_f32": ([P, P, I32, I32, I32, I32, I32, I32, P, P], C.c_int),
        "rh_pqmf_synthesis_fwd_f32": ([P, P, I32, I32, I32, I32, I32, I32, P, P], C.c_int),
        "rh_pqmf_synthesis_bwd_f32": ([P, P, I32, I32, I32, I32, I32, I32, P, P], C.c_int),
        "rh_pqmf_fold_k1_f32": ([P, P, I32, I32, I32, I32, F, P, P], C.c_int),
        "rh_pqmf_fold_k2_f32": ([P, P, I32, I32, I32, I32, F, P, P], C.c_int),
        "rh_amp_tanh_fwd_f32": ([P, I32, I32, I32, P, P], C.c_int),
        "rh_amp_tanh_bwd_f32": ([P, P, I32, I32, I32, P, P], C.c_int),
        "rh_act_fwd_f32": ([P, P, I32, F, I32, I32, I32, P, P], C.c_int),
        "rh_act_bwd_f32": ([P, P, I32, F, I64, P, P], C.c_int),
        "rh_act_bwd_bias_workspace_bytes": ([I32], I64),
        "rh_act_bwd_bias_f32": ([P, P, I32, F, I32, I32, I64, P, P, P, I64, P], C.c_int),
        "rh_snake_bwd_workspace_bytes": ([I32, I32], I64),
        "rh_snake_bwd_f32": ([P, P, P, I32, I32, I32, P, P, P, I64, P], C.c_int),
        "rh_stft_frame_fwd_f32": ([P, P, I64, I32, I32, I32, I32, P, P], C.c_int),
        "rh_stft_frame_bwd_f32": ([P, P, I64, I32, I32, I32, I32, P, P], C.c_int),
        "rh_stft_frame_bwd_acc_f32": ([P, P, I64, I32, I32, I32, I32, P, I32, P], C.c_int),
        "rh_spectral_total_f32": ([P, P, I32, P, P], C.c_int),
        "rh_feature_matching_workspace_bytes": ([C.POINTER(FmItem), I32], I64),
        "rh_feature_matching_fwd_f32": ([C.POINTER(FmItem), I32, I32, P, I64, P, P, P], C.c_int),
        "rh_feature_matching_bwd_f32": ([C.POINTER(FmItem), I32, I32, P, P, P], C.c_int),
        "rh_reparam_workspace_bytes": ([], I64),
        "rh_reparam_fwd_f32": ([P, P, I32, I32, I32, P, P, P, I64, P], C.c_int),
        "rh_reparam_bwd_f32": ([P, P, P, P, I32, I32, I32, P, P], C.c_int),
        "rh_loss_combine_fwd_f32": ([C.POINTER(LossItem), I32, P, P, P], C.c_int),
        "rh_loss_combine_bwd_f32": ([C.POINTER(LossItem), I32, P, P, P], C.c_int),
        "rh_adam_step_f32": ([C.POINTER(AdamItem), I32, P, F, F, F, P, P], C.c_int),
        "rh_x6_uses_ranges": ([], C.c_int),
        "rh_x6_range_words": ([], C.c_int),
        "rh_x6_set_ranges": ([P, P, P, P], C.c_int),
        "rh_amax_f32": ([P, I64, P, P], C.c_int),
        "rh_set_kernel_events": ([P, P], C.c_int),
        "rh_kernel_events_used": ([], C.c_int),
        "rh_event_create": ([C.POINTER(C.c_void_p)], C.c_int),
        "rh_event_destroy": ([P], C.c_int),
        "rh_event_elapsed_ms": ([P, P, C.POINTER(C.c_float)], C.c_int),
        "rh_stft_loss_supported": ([I32, I32, I32, I64], C.c_int),
        "rh_stft_loss_plan_info": ([I32, I32, I64, C.POINTER(I64)], C.c_int),
        "rh_stft_loss_workspace_bytes": ([I32, I32, I64], I64),
        "rh_stft_loss_fwd_f32": ([P, P, P, P, I64, I32, I32, F, P, P, I64, P], C.c_int),
        "rh_stft_loss_finalize_all_f32": ([C.POINTER(C.c_void_p), C.POINTER(I64), I32, P, P, P, P], C.c_int),
        "rh_stft_loss_bwd_f32": ([P, P, P, P, I64, I32, I32, F, P, P, P, P, I32, P], C.c_int),
        "rh_spectral_distance_workspace_bytes": ([], I64),
        "rh_spectral_distance_fwd_f32": ([P, P, I64, F, P, P, I64, P], C.c_int),
        "rh_spectral_distance_bwd_f32": ([P, P, P, P, I64, F, P, P, I32, P], C.c_int),
        "rh_adain_stats_update_f32": ([P, I64, I32, P, P, P, P], C.c_int),
        "rh_adain_transfer_f32": ([P, I64, I32, P, P, P, P, P, P], C.c_int),
        "rh_avgpool2_fwd_f32": ([P, I64, I32, P, P], C.c_int),
        "rh_avgpool2_bwd_f32": ([P, I64, I32, P, P], C.c_int),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.argtypes = args
        fn.restype = res
    return lib


lib = _load()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.rh_last_error().decode(errors="replace")
        raise RuntimeError(f"librave_hip {what} failed (code {rc}): {msg}")


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream
