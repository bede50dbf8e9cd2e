"""ctypes binding of libkoemorph_hip.so (include/koemorph.h).

There is NO fallback: if the library is missing or a call fails, a KoeMorphError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import build as _build

KM_ABI_VERSION = 2


class KoeMorphError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libkoemorph_hip: {msg} (status {code})")
        self.code = code


class KMMelConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("n_fft", C.c_int32), ("hop_length", C.c_int32),
                ("n_mels", C.c_int32), ("f_min", C.c_float), ("f_max", C.c_float),
                ("mel_scale", C.c_int32), ("slaney_norm", C.c_int32), ("pad_mode", C.c_int32),
                ("window_norm", C.c_int32), ("log_mode", C.c_int32), ("amin", C.c_float),
                ("top_db", C.c_float), ("db_add", C.c_float), ("db_scale", C.c_float),
                ("log_eps", C.c_float)]


class KMConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("d_model", C.c_int32), ("num_heads", C.c_int32),
                ("num_mel_channels", C.c_int32), ("mel_sequence_length", C.c_int32),
                ("mel_temporal_frames", C.c_int32), ("emotion_dim", C.c_int32),
                ("num_blendshapes", C.c_int32), ("temperature", C.c_float), ("mel", KMMelConfig)]


class KMLegacyConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("d_model", C.c_int32), ("num_heads", C.c_int32),
                ("decoder_hidden", C.c_int32), ("num_blendshapes", C.c_int32), ("mel", KMMelConfig)]


class KMKoeMorphConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "mel_dim", "emotion_dim", "d_model", "num_heads", "num_encoder_layers",
                                         "num_attention_layers", "decoder_hidden_dim", "decoder_layers", "decoder_activation",
                                         "causal", "window_size", "use_temporal_smoothing", "use_constraints", "num_blendshapes",
                                         "output_activation", "smoothing_method", "smoothing_window")]


class KMLossConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("perceptual_weight", C.c_float), ("temporal_weight", C.c_float), ("sparsity_weight", C.c_float),
                ("smoothness_weight", C.c_float), ("landmark_weight", C.c_float), ("velocity_weight", C.c_float),
                ("prev_pred_dev", C.c_void_p), ("prev_target_dev", C.c_void_p), ("landmark_w_dev", C.c_void_p),
                ("audio_energy_dev", C.c_void_p), ("ds_velocity_weight", C.c_float), ("ds_separation_weight", C.c_float),
                ("ds_prev_pred_dev", C.c_void_p)]


KM_MEL_SLANEY, KM_MEL_HTK = 0, 1
KM_PAD_CONSTANT, KM_PAD_REFLECT = 0, 1
KM_LOG_DB_MAX, KM_LOG_LN_EPS = 0, 1

KM_OK = 0
KM_ERR_INVALID_ARG, KM_ERR_UNSUPPORTED, KM_ERR_NOT_FINALIZED = -1, -2, -3
KM_ERR_WORKSPACE, KM_ERR_HIP, KM_ERR_NOT_READY = -4, -5, -6

_p = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32
_h = C.c_void_p

# name -> (restype, argtypes); lists EVERY symbol include/koemorph.h declares
# (tests/test_abi.py parses the header and compares).
SIGNATURES = {
    "km_abi_version": (C.c_int, []),
    "km_last_error": (C.c_char_p, []),
    "km_create": (C.c_int, [C.POINTER(KMConfig), C.POINTER(_h)]),
    "km_destroy": (C.c_int, [_h]),
    "km_load_param": (C.c_int, [_h, C.c_char_p, _p, C.POINTER(_i64), _i32]),
    "km_get_param": (C.c_int, [_h, C.c_char_p, _p, _i64]),
    "km_param_count": (C.c_int, [_h, C.POINTER(_i32), C.POINTER(_i32)]),
    "km_finalize": (C.c_int, [_h, _p]),
    "km_finalize_host": (C.c_int, [_h]),
    "km_reserve": (C.c_int, [_h, _i64, _i64]),
    "km_mel_batch": (C.c_int, [_h, _p, _i64, _i64, _p, _p, _p]),
    "km_mel_num_frames": (_i64, [_h, _i64]),
    "km_mel_extract": (C.c_int, [_h, C.POINTER(KMMelConfig), _p, _i64, _i64, _i64, _p, _p]),
    "km_core_forward": (C.c_int, [_h, _p, _i64, _i64, _p, _p, _p, _p, _p, _p]),
    "km_emotion_logit": (C.c_int, [_h, _p, _i64, _p, _p]),
    "km_core_forward_z": (C.c_int, [_h, _p, _i64, _i64, _p, _p, _p, _p, _p, _p]),
    "km_smooth": (C.c_int, [_h, _p, _p, _i64, _i32, _p]),
    "km_forward_audio": (C.c_int, [_h, _p, _i64, _i64, _p, _p, _p, _i32, _p]),
    "km_forward_audio_pipelined": (C.c_int, [_h, _p, _i64, _i64, _p, _p, _p, _i32, _p]),
    "km_pipeline_flush": (C.c_int, [_h, _p]),
    "km_sequence_num_outputs": (_i64, [_h, _i64, _i32]),
    "km_sequence_forward": (C.c_int, [_h, _p, _i64, _i64, _p, _i32, _i32, _p, _p]),
    "km_ema_scan": (C.c_int, [_h, _p, _i64, _i64, _p]),
    "km_train_init": (C.c_int, [_h, _i64, _p]),
    "km_train_num_params": (_i64, [_h]),
    "km_train_param_offset": (_i64, [_h, C.c_char_p]),
    "km_train_step": (C.c_int, [_h, _p, _i64, _i64, _p, _p, _p, C.c_float, C.c_float, _p, _p, _p, _p, _i32, _p]),
    "km_train_step_audio": (C.c_int, [_h, _p, _i64, _i64, _p, _p, C.c_float, C.c_float, _p, _p, _p, _p, _i32, _p]),
    "km_train_adamw": (C.c_int, [_h, _p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _i64, _p]),
    "km_train_get_params": (C.c_int, [_h, _p, _i64]),
    "km_train_set_params": (C.c_int, [_h, _p, _i64]),
    "km_train_sync": (C.c_int, [_h, _p]),
    "km_train_set_loss": (C.c_int, [_h, C.POINTER(KMLossConfig)]),
    "km_audio_energy": (C.c_int, [_p, _i64, _i64, _i64, _p, _p]),
    "km_train_set_dropout": (C.c_int, [_h, C.c_float, C.c_uint64, _i32]),
    "km_train_get_dropout_step": (C.c_int, [_h, C.POINTER(_i64)]),
    "km_train_set_dropout_step": (C.c_int, [_h, _i64]),
    "km_train_grad_split": (C.c_int, [_h, C.POINTER(_i64)]),
    "km_train_wait_early": (C.c_int, [_h, _p]),
    "km_train_get_dropout_masks": (C.c_int, [_h, _i64, _p, _p, _p, _p]),
    "km_train_set_dropout_masks": (C.c_int, [_h, _i64, _p, _p, _p, _p]),
    "km_train_get_optimizer_state": (C.c_int, [_h, _p, _p, _i64, _p]),
    "km_train_set_optimizer_state": (C.c_int, [_h, _p, _p, _i64, _p]),
    "km_resample_labels": (C.c_int, [_p, C.c_int64, C.c_int32, C.c_int64, _p, _p]),
    "km_gather_windows": (C.c_int, [_p, C.c_int64, _p, C.c_int64, C.c_int32, C.c_int64, _p, _p, C.c_int64, C.c_int32,
                                    C.c_int32, _p, _p, _p]),
    "km_format_frames": (C.c_int64, [_p, C.c_int64, C.c_int32, _p, C.c_int32, _p, C.c_int64, _p]),
    "km_legacy_create": (C.c_int, [C.POINTER(KMLegacyConfig), C.POINTER(_h)]),
    "km_legacy_forward": (C.c_int, [_h, _p, _i64, _i64, _p, _p]),
    "km_legacy_forward_mel": (C.c_int, [_h, _p, _i64, _i64, _p, _p]),
    "km_koemorph_create": (C.c_int, [C.POINTER(KMKoeMorphConfig), C.POINTER(_h)]),
    "km_koemorph_reserve": (C.c_int, [_h, _i64, _i64]),
    "km_koemorph_forward": (C.c_int, [_h, _p, _p, _i64, _i64, _p, _p, _p, _i32, _p, _p, _p, _p]),
    "km_stream_create": (C.c_int, [_h, _i64, C.c_double, C.c_double, C.POINTER(KMMelConfig)]),
    "km_stream_push": (C.c_int, [_h, _p, _i64, _p]),
    "km_stream_tick": (C.c_int, [_h, _p, _p, _p, _p]),
    "km_stream_reset": (C.c_int, [_h, _p]),
    "km_set_option": (C.c_int, [_h, C.c_char_p, _i64]),
    "km_egemaps_plan_create": (C.c_int, [C.POINTER(_p)]),
    "km_egemaps_plan_destroy": (C.c_int, [_p]),
    "km_egemaps_num_frames": (_i64, [_i64]),
    "km_egemaps_workspace_floats": (_i64, [_i64, _i64]),
    "km_egemaps_functionals": (C.c_int, [_p, _p, _i64, _i64, _i32, _p, _i64, _p, _p]),
    "km_egemaps_records": (C.c_int, [_p, _i64, _i64, _p, _p]),
    "km_linear": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p]),
    "km_enable_stage_timing": (C.c_int, [_h, _i32]),
    "km_stage_times": (C.c_int, [_h, C.POINTER(C.c_float)]),
    "km_debug_buffer": (C.c_int, [_h, C.c_char_p, _p, C.POINTER(_i64)]),
}

_lib: Optional[C.CDLL] = None


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load the shared library (building it in-tree first if the sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if os.environ.get("KM_LIBRARY"):          # A/B builds of the same ABI (tools/micro/mel_variants.sh): loaded as they are
        path, build_if_missing = os.environ["KM_LIBRARY"], False
    if build_if_missing and _build.is_stale():
        try:
            _build.build_library()
        except Exception as exc:
            # A library OLDER than its sources that cannot be rebuilt (a GPU box without hipcc or write access) is refused:
            # loading it would benchmark or test old kernels under new sources without anyone noticing.  KM_ALLOW_STALE=1
            # loads it anyway, with a warning; the ABI check below still catches a changed interface.
            if not os.path.exists(path) or os.environ.get("KM_ALLOW_STALE") != "1":
                raise KoeMorphError(KM_ERR_HIP, f"libkoemorph_hip.so is missing or older than its sources and the rebuild "
                                                f"failed ({exc}); set KM_ALLOW_STALE=1 to load the stale library") from exc
            import warnings
            warnings.warn(f"libkoemorph_hip.so is OLDER than its sources and the rebuild failed ({exc}); "
                          "loading the stale library (KM_ALLOW_STALE=1)", RuntimeWarning, stacklevel=2)
    if not os.path.exists(path):
        raise KoeMorphError(KM_ERR_HIP, f"{path} is missing: run `python -m koemorph_amd.build` "
                                        "(there is no CPU fallback)")
    # The library links the ROCm runtime by SONAME.  PyTorch ships its own copy of that runtime: if ours is loaded
    # first, the process ends up with two HIP runtimes and the one behind this library cannot open the device
    # ("no HIP device").  Importing torch first makes both resolve to the same, already loaded runtime.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.km_abi_version() != KM_ABI_VERSION:
        raise KoeMorphError(KM_ERR_INVALID_ARG, "ABI version mismatch between _lib.py and the shared library")
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != KM_OK:
        msg = load().km_last_error()
        raise KoeMorphError(code, msg.decode("utf-8", "replace") if msg else "unknown error")
