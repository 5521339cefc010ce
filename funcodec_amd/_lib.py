"""ctypes binding of include/funcodec_amd.h (the C ABI of the gfx950 engine)."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

FC_MAX_RATIOS = 8
FC_ABI_VERSION = 6


class FcArch(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("sample_rate", C.c_int32), ("audio_normalize", C.c_int32),
        ("n_filters", C.c_int32), ("dimension", C.c_int32), ("n_ratios", C.c_int32),
        ("ratios", C.c_int32 * FC_MAX_RATIOS),
        ("kernel_size", C.c_int32), ("last_kernel_size", C.c_int32), ("residual_kernel_size", C.c_int32),
        ("compress", C.c_int32), ("lstm_layers", C.c_int32), ("lstm_skip", C.c_int32),
        ("elu_alpha", C.c_float), ("gn_eps", C.c_float),
        ("codebook_size", C.c_int32), ("num_quantizers", C.c_int32),
        ("norm_type", C.c_int32), ("causal", C.c_int32), ("n_residual_layers", C.c_int32), ("dilation_base", C.c_int32),
        ("model_type", C.c_int32), ("input_channels", C.c_int32), ("n_fft", C.c_int32), ("stft_hop", C.c_int32),
        ("ratios_f", C.c_int32 * FC_MAX_RATIOS),
        ("enc_conv_group_ratio", C.c_int32), ("dec_conv_group_ratio", C.c_int32), ("dec_tr_conv_group_ratio", C.c_int32),
        ("codec_dim", C.c_int32), ("codec_range", C.c_float),
        ("q0_ds_ratio", C.c_int32),
    ]


class FcWork(C.Structure):
    _fields_ = [("total_flops", C.c_double), ("total_bytes", C.c_double), ("conv_flops", C.c_double),
                ("conv_bytes", C.c_double), ("lstm_flops", C.c_double), ("rvq_flops", C.c_double),
                ("conv_launches", C.c_int32), ("total_launches", C.c_int32)]


class FcProf(C.Structure):
    _fields_ = [("kernel", C.c_char * 64), ("total_ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double),
                ("launches", C.c_int32), ("reserved", C.c_int32)]


FC_PROF_CLASSES = 48


class FcLauraStack(C.Structure):
    _fields_ = [("idim", C.c_int32), ("d_model", C.c_int32), ("heads", C.c_int32), ("ff", C.c_int32), ("layers", C.c_int32),
                ("act", C.c_int32), ("embed_relu", C.c_int32), ("norm_style", C.c_int32)]


class FcLauraArch(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("input_size", C.c_int32), ("vocab_size", C.c_int32), ("codebook_size", C.c_int32),
                ("codebook_dim", C.c_int32), ("num_quantizers", C.c_int32), ("predict_nq", C.c_int32), ("pos_emb_split", C.c_int32),
                ("bidirectional_inputs", C.c_int32), ("max_positions", C.c_int32),
                ("text_encoder", FcLauraStack), ("codec_lm", FcLauraStack), ("codec_encoder", FcLauraStack)]


# every symbol include/funcodec_amd.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "fc_abi_version": (C.c_int, []),
    "fc_last_error": (C.c_char_p, []),
    "fc_engine_create": (C.c_int, [C.POINTER(FcArch), C.c_int, C.POINTER(_P)]),
    "fc_engine_destroy": (None, [_P]),
    "fc_engine_num_weights": (C.c_int, [_P]),
    "fc_engine_weight_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]),
    "fc_engine_set_weight": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int]),
    "fc_engine_finalize": (C.c_int, [_P]),
    "fc_engine_hop_length": (C.c_int, [_P]),
    "fc_engine_frames": (C.c_int, [_P, C.c_int]),
    "fc_engine_decoded_samples": (C.c_int, [_P, C.c_int]),
    "fc_engine_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int]),
    "fc_encode": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "fc_decode_emb": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_size_t, _P]),
    "fc_decode_codes": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_size_t, _P]),
    "fc_encode_decode": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "fc_rvq_encode": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, C.c_size_t, _P]),
    "fc_q0_source_frames": (C.c_int, [C.c_int, _P]),
    "fc_layer_forward": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_size_t, _P]),
    "fc_layer_out_len": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "fc_lstm_forward": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.c_int, _P, _P, C.c_size_t, _P]),
    "fc_resblock_forward": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.c_int, _P, _P, C.c_size_t, _P]),
    "fc_engine_work": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(FcWork)]),
    "fc_engine_profile": (C.c_int, [_P, C.c_int]),
    "fc_engine_profile_read": (C.c_int, [_P, C.POINTER(FcProf)]),
    "fc_debug_timeline": (C.c_int, [_P]),
    "fc_debug_conv_layout": (C.c_int, [C.c_int] * 7 + [_P, _P, C.c_size_t, _P, C.c_size_t]),
    "fc_debug_freq_features": (C.c_int, [_P, C.c_size_t, C.c_int]),
    "fc_codec_json_bound": (C.c_size_t, [C.c_int, C.c_int]),
    "fc_format_codec_json": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "fc_write_wav_pcm16": (C.c_int, [C.c_char_p, _P, C.c_int, C.c_int, C.c_int]),
    "fc_overlap_add": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "fc_engine_status": (C.c_int, [_P, C.POINTER(C.c_uint)]),
    # LauraTTS generation (ABI version 5)
    "fc_laura_create": (C.c_int, [C.POINTER(FcLauraArch), C.c_int, C.POINTER(_P)]),
    "fc_laura_destroy": (None, [_P]),
    "fc_laura_num_weights": (C.c_int, [_P]),
    "fc_laura_weight_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]),
    "fc_laura_set_weight": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int]),
    "fc_laura_finalize": (C.c_int, [_P]),
    "fc_laura_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "fc_laura_encode": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P, C.c_size_t, _P]),
    "fc_laura_lm_logprobs": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, C.c_int, _P, C.c_int, _P, C.c_size_t, _P]),
    "fc_laura_decode_codec": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                        C.c_uint64, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "fc_laura_codec_emb": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P, C.c_size_t, _P]),
    "fc_laura_linear": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_size_t, _P]),
    "fc_laura_debug_probe": (C.c_int, [_P, C.c_size_t, C.c_int, C.c_int, C.c_int]),
    "fc_laura_set_persistent_step": (C.c_int, [_P, C.c_int]),
    "fc_laura_persistent_step_fallbacks": (C.c_int, [_P]),
}

_lib = None


def lib_path() -> str:
    # FC_LIB: tuning aid only (e.g. a profiling build made with FC_TIMELINE=1 next to the product library)
    return os.environ.get("FC_LIB") or _build.LIB_PATH


def load():
    """Load libfuncodec_amd.so (built in-tree).  There is no fallback: if the HIP library cannot be
    loaded the product path raises."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so); it must be in the process BEFORE this
    # library is loaded so that both share ONE runtime -- two HIP runtimes in one process cannot both
    # see the device ("no HIP device visible").
    import torch  # noqa: F401
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -m funcodec_amd.build` (hipcc, gfx950). "
            "funcodec_amd has no CPU or PyTorch fallback path.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError = ABI drift, loudly
        fn.restype = res
        fn.argtypes = args
    if lib.fc_abi_version() != FC_ABI_VERSION:
        raise RuntimeError("libfuncodec_amd.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def last_error() -> str:
    return load().fc_last_error().decode("utf-8", "replace")
