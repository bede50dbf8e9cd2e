"""Thin object wrapper over the C-ABI (include/koemorph.h).

torch is used here only as plumbing: device buffers (``tensor.data_ptr()``) and the current
HIP stream.  Every compute call goes through libkoemorph_hip.so; there is no eager / CPU
fallback -- a missing library or an unsupported configuration raises ``KoeMorphError``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Mapping, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import KMConfig, KMMelConfig, KoeMorphError, check


@dataclass
class MelConfig:
    """One log-mel front-end variant (see km_mel_config in include/koemorph.h)."""
    sample_rate: int = 16000
    n_fft: int = 1024
    hop_length: int = 533
    n_mels: int = 80
    f_min: float = 80.0
    f_max: float = 8000.0
    mel_scale: int = _lib.KM_MEL_SLANEY
    slaney_norm: int = 1
    pad_mode: int = _lib.KM_PAD_CONSTANT
    window_norm: int = 0
    log_mode: int = _lib.KM_LOG_DB_MAX
    amin: float = 1e-10
    top_db: float = 80.0
    db_add: float = 80.0
    db_scale: float = 1.0 / 80.0
    log_eps: float = 1e-8

    def to_c(self) -> KMMelConfig:
        return KMMelConfig(self.sample_rate, self.n_fft, self.hop_length, self.n_mels, self.f_min,
                           self.f_max, self.mel_scale, self.slaney_norm, self.pad_mode,
                           self.window_norm, self.log_mode, self.amin, self.top_db, self.db_add,
                           self.db_scale, self.log_eps)

    # the three front ends of the reference -------------------------------------------------
    @staticmethod
    def model_batch(sample_rate: int = 16000, target_fps: float = 30, n_fft: int = 1024) -> "MelConfig":
        """SimplifiedDualStreamModel.extract_mel_features
        (src/model/simplified_dual_stream_model.py:188-200): librosa defaults, (dB+80)/80."""
        return MelConfig(sample_rate=sample_rate, n_fft=n_fft, hop_length=int(sample_rate / target_fps),
                         f_min=80.0, f_max=8000.0)

    @staticmethod
    def sliding_window(sample_rate: int = 16000, n_fft: int = 512, hop_length: int = 532,
                       n_mels: int = 80, f_min: float = 80.0, f_max: Optional[float] = None,
                       pad_mode: str = "reflect") -> "MelConfig":
        """MelSlidingWindowExtractor (src/features/mel_sliding_window.py:280-295): dB, no affine."""
        return MelConfig(sample_rate=sample_rate, n_fft=n_fft, hop_length=hop_length, n_mels=n_mels,
                         f_min=f_min, f_max=float(f_max or sample_rate // 2),
                         pad_mode=_lib.KM_PAD_REFLECT if pad_mode == "reflect" else _lib.KM_PAD_CONSTANT,
                         db_add=0.0, db_scale=1.0)

    @staticmethod
    def torchaudio(sample_rate: int = 16000, target_fps: float = 30.0, n_fft: int = 512, n_mels: int = 80,
                   f_min: float = 80.0, f_max: Optional[float] = None, normalized: bool = True,
                   pad_mode: str = "reflect", eps: float = 1e-8) -> "MelConfig":
        """MelSpectrogramExtractor (src/features/stft.py:84-123): HTK, window-normalised, log(x+eps)."""
        return MelConfig(sample_rate=sample_rate, n_fft=n_fft, hop_length=int(sample_rate / target_fps),
                         n_mels=n_mels, f_min=f_min, f_max=float(f_max or sample_rate // 2),
                         mel_scale=_lib.KM_MEL_HTK, slaney_norm=0,
                         pad_mode=_lib.KM_PAD_REFLECT if pad_mode == "reflect" else _lib.KM_PAD_CONSTANT,
                         window_norm=1 if normalized else 0, log_mode=_lib.KM_LOG_LN_EPS, log_eps=eps)


def _torch():
    import torch
    return torch


def _ptr(t) -> int:
    return t.data_ptr() if t is not None else 0


def _stream_ptr(device) -> int:
    torch = _torch()
    return torch.cuda.current_stream(device).cuda_stream


class Engine:
    """One km_handle.  Not thread-safe (neither are the reference modules)."""

    def __init__(self, d_model: int = 256, num_heads: int = 8, num_mel_channels: int = 80,
                 mel_sequence_length: int = 256, mel_temporal_frames: int = 3, emotion_dim: int = 256,
                 num_blendshapes: int = 52, temperature: float = 1.0, mel: Optional[MelConfig] = None):
        self._lib = _lib.load()
        self.mel = mel or MelConfig()
        self.cfg = KMConfig(_lib.KM_ABI_VERSION, d_model, num_heads, num_mel_channels, mel_sequence_length,
                            mel_temporal_frames, emotion_dim, num_blendshapes, temperature, self.mel.to_c())
        self._h = C.c_void_p()
        check(self._lib.km_create(C.byref(self.cfg), C.byref(self._h)))
        self.d_model, self.num_heads, self.mel_sequence_length = d_model, num_heads, mel_sequence_length
        self.emotion_dim, self.num_blendshapes, self.n_mels = emotion_dim, num_blendshapes, num_mel_channels
        self._reserved: Tuple[int, int] = (0, 0)
        self.device = None
        self.fused = (d_model == 256 and num_heads == 8 and mel_sequence_length == 256 and num_mel_channels == 80)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.km_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters -------------------------------------------------------------------------
    def load_param(self, key: str, value) -> None:
        if hasattr(value, "detach"):
            value = value.detach().cpu().numpy()
        a = np.asarray(value, dtype=np.float32)
        ndim, dims = a.ndim, a.shape                     # ascontiguousarray would promote a 0-d scalar to 1-d
        a = np.ascontiguousarray(a)
        shape = (C.c_int64 * max(1, ndim))(*dims)
        check(self._lib.km_load_param(self._h, key.encode(), a.ctypes.data_as(C.c_void_p), shape, ndim))

    def load_state_dict(self, state: Mapping[str, object]) -> None:
        for k, v in state.items():
            self.load_param(k, v)

    def state_dict_shapes(self):
        """Keys and shapes of the model's parameters in state-dict order (DualStreamCrossAttention + smoothing_alpha)."""
        from . import synth
        shapes = synth.core_param_shapes(self.d_model, self.mel_sequence_length, 3, self.emotion_dim, self.num_blendshapes)
        out = {k: np.empty(v, np.float32) for k, v in shapes.items()}
        out["smoothing_alpha"] = np.empty((), np.float32)
        return out

    def get_param(self, key: str, shape) -> np.ndarray:
        out = np.empty(shape, np.float32)
        check(self._lib.km_get_param(self._h, key.encode(), out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def param_count(self) -> Tuple[int, int]:
        e, l = C.c_int32(), C.c_int32()
        check(self._lib.km_param_count(self._h, C.byref(e), C.byref(l)))
        return e.value, l.value

    def finalize_host(self) -> None:
        check(self._lib.km_finalize_host(self._h))

    def debug_buffer(self, name: str) -> np.ndarray:
        n = C.c_int64(0)
        check(self._lib.km_debug_buffer(self._h, name.encode(), None, C.byref(n)))
        out = np.empty(n.value, np.float32)
        check(self._lib.km_debug_buffer(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), C.byref(n)))
        return out

    def finalize(self, device=None) -> None:
        torch = _torch()
        if not torch.cuda.is_available():
            raise KoeMorphError(_lib.KM_ERR_HIP, "no GPU visible: libkoemorph_hip has no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        with torch.cuda.device(self.device):
            check(self._lib.km_finalize(self._h, _stream_ptr(self.device)))
        self._reserved = (0, 0)        # the generic-path workspace depends on the finalized configuration

    def reserve(self, max_windows: int, max_samples: int = 0) -> None:
        if max_windows <= self._reserved[0] and max_samples <= self._reserved[1]:
            return
        torch = _torch()
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)        # the old workspace may still be in use
            check(self._lib.km_reserve(self._h, max_windows, max_samples))
        self._reserved = (max(max_windows, self._reserved[0]), max(max_samples, self._reserved[1]))

    # ---- forward path -----------------------------------------------------------------------
    def _chk(self, t, name, ndim):
        torch = _torch()
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise ValueError(f"{name} must be a CUDA/HIP tensor")
        if t.dim() != ndim:
            raise ValueError(f"Expected {ndim}D {name}, got {t.dim()}D")
        if t.dtype != torch.float32:
            t = t.float()
        return t.contiguous()

    def mel_num_frames(self, L: int) -> int:
        return int(self._lib.km_mel_num_frames(self._h, L))

    def mel_batch(self, audio):
        """audio (B, L) -> (mel_long (B, F, n_mels), mel_short (B, 3, n_mels))."""
        torch = _torch()
        audio = self._chk(audio, "audio", 2)
        B, L = audio.shape
        F = self.mel_num_frames(L)
        self.reserve(B, L)
        long = torch.empty(B, F, self.n_mels, device=audio.device, dtype=torch.float32)
        short = torch.empty(B, 3, self.n_mels, device=audio.device, dtype=torch.float32)
        with torch.cuda.device(audio.device):
            check(self._lib.km_mel_batch(self._h, _ptr(audio), B, L, _ptr(long), _ptr(short),
                                         _stream_ptr(audio.device)))
        return long, short

    def mel_extract(self, cfg: MelConfig, audio, out_frames: int = 0):
        torch = _torch()
        audio = self._chk(audio, "audio", 2)
        B, L = audio.shape
        F = 1 + L // cfg.hop_length
        self.reserve(B, max(L, (F + 1) * self.mel.hop_length))
        nout = out_frames if out_frames > 0 else F
        out = torch.empty(B, nout, cfg.n_mels, device=audio.device, dtype=torch.float32)
        if nout == 0:
            return out
        ccfg = cfg.to_c()
        with torch.cuda.device(audio.device):
            check(self._lib.km_mel_extract(self._h, C.byref(ccfg), _ptr(audio), B, L, nout, _ptr(out),
                                           _stream_ptr(audio.device)))
        return out

    def core_forward(self, mel, mel_short, emotion, return_attention: bool = False) -> Dict[str, object]:
        torch = _torch()
        mel = self._chk(mel, "mel_features", 3)
        mel_short = self._chk(mel_short, "mel_temporal_features", 3)
        emotion = self._chk(emotion, "emotion_features", 2)
        B, T_in, C_ = mel.shape
        if C_ != self.n_mels or tuple(mel_short.shape) != (B, 3, self.n_mels) or tuple(emotion.shape) != (B, self.emotion_dim):
            raise ValueError(f"shape mismatch: mel {tuple(mel.shape)}, short {tuple(mel_short.shape)}, "
                             f"emotion {tuple(emotion.shape)}")
        self.reserve(B, 0)
        out = torch.empty(B, self.num_blendshapes, device=mel.device, dtype=torch.float32)
        raw = attn = None
        if return_attention:
            raw = torch.empty_like(out)
            attn = torch.empty(B, 28, self.n_mels, device=mel.device, dtype=torch.float32)
        with torch.cuda.device(mel.device):
            check(self._lib.km_core_forward(self._h, _ptr(mel), B, T_in, _ptr(mel_short), _ptr(emotion),
                                            _ptr(out), _ptr(raw), _ptr(attn), _stream_ptr(mel.device)))
        res = {"blendshapes": out}
        if return_attention:
            res["raw"] = raw
            res["mel_attention_weights"] = attn
        return res

    def emotion_logit(self, emotion):
        torch = _torch()
        emotion = self._chk(emotion, "emotion_features", 2)
        z = torch.empty(emotion.shape[0], device=emotion.device, dtype=torch.float32)
        with torch.cuda.device(emotion.device):
            check(self._lib.km_emotion_logit(self._h, _ptr(emotion), emotion.shape[0], _ptr(z),
                                             _stream_ptr(emotion.device)))
        return z

    def core_forward_z(self, mel, mel_short, z, out=None):
        """Mel stream + decoder as ONE kernel launch; tensors must already be contiguous fp32."""
        torch = _torch()
        B, T_in, _ = mel.shape
        if out is None:
            out = torch.empty(B, self.num_blendshapes, device=mel.device, dtype=torch.float32)
        check(self._lib.km_core_forward_z(self._h, _ptr(mel), B, T_in, _ptr(mel_short), _ptr(z), _ptr(out), 0, 0,
                                          _stream_ptr(mel.device)))
        return out

    def smooth(self, x, state, first: bool) -> None:
        torch = _torch()
        with torch.cuda.device(x.device):
            check(self._lib.km_smooth(self._h, _ptr(x), _ptr(state), x.shape[0], 1 if first else 0,
                                      _stream_ptr(x.device)))

    def forward_audio(self, audio, emotion, state=None, first: bool = True, out=None):
        torch = _torch()
        audio = self._chk(audio, "audio", 2)
        emotion = self._chk(emotion, "emotion_features", 2)
        B, L = audio.shape
        self.reserve(B, L)
        if out is None:
            out = torch.empty(B, self.num_blendshapes, device=audio.device, dtype=torch.float32)
        with torch.cuda.device(audio.device):
            check(self._lib.km_forward_audio(self._h, _ptr(audio), B, L, _ptr(emotion), _ptr(out), _ptr(state),
                                             1 if first else 0, _stream_ptr(audio.device)))
        return out

    def set_option(self, name: str, value: int) -> None:
        """Run-time switch of this handle (km_set_option): e.g. ("seq_per_window", 1), ("core_split", 3)."""
        check(self._lib.km_set_option(self._h, name.encode(), int(value)))

    def enable_stage_timing(self, enable: bool = True) -> None:
        check(self._lib.km_enable_stage_timing(self._h, 1 if enable else 0))

    def stage_times_ms(self):
        """(emotion, front end, core) milliseconds of the most recent forward_audio (HIP events)."""
        ms = (C.c_float * 3)()
        check(self._lib.km_stage_times(self._h, ms))
        return float(ms[0]), float(ms[1]), float(ms[2])

    def forward_audio_pipelined(self, audio, emotion, state=None, first: bool = True, out=None):
        """Throughput mode (km_forward_audio_pipelined): the result is complete after the next pipelined call or
        after pipeline_flush(); audio / emotion / out must stay alive and untouched until then."""
        torch = _torch()
        audio = self._chk(audio, "audio", 2)
        emotion = self._chk(emotion, "emotion_features", 2)
        B, L = audio.shape
        self.reserve(B, L)
        if out is None:
            out = torch.empty(B, self.num_blendshapes, device=audio.device, dtype=torch.float32)
        with torch.cuda.device(audio.device):
            check(self._lib.km_forward_audio_pipelined(self._h, _ptr(audio), B, L, _ptr(emotion), _ptr(out), _ptr(state),
                                                       1 if first else 0, _stream_ptr(audio.device)))
        return out

    def pipeline_flush(self) -> None:
        check(self._lib.km_pipeline_flush(self._h, _stream_ptr(self.device)))

    def sequence_forward(self, audio, emotion, stride_frames: int = 1, smooth: bool = True, max_tile: int = 2048):
        """audio (B, L) -> (B, N, 52): one frame per window position (km_sequence_forward)."""
        torch = _torch()
        audio = self._chk(audio, "audio", 2)
        emotion = self._chk(emotion, "emotion_features", 2)
        B, L = audio.shape
        N = self.sequence_num_outputs(L, stride_frames)
        W = self.mel_sequence_length * self.mel.hop_length
        self.reserve(max(B, min(B * N, max_tile)), W)
        out = torch.empty(B, N, self.num_blendshapes, device=audio.device, dtype=torch.float32)
        with torch.cuda.device(audio.device):
            check(self._lib.km_sequence_forward(self._h, _ptr(audio), B, L, _ptr(emotion), stride_frames,
                                                1 if smooth else 0, _ptr(out), _stream_ptr(audio.device)))
        return out

    def ema_scan(self, seq) -> None:
        """In-place temporal smoothing of (B, N, 52) along the frame axis (km_ema_scan): the last step of sequence_forward,
        exported for sequences whose frames were computed in chunks (parallel.sequence_apply)."""
        torch = _torch()
        seq = self._chk(seq, "seq", 3)
        with torch.cuda.device(seq.device):
            check(self._lib.km_ema_scan(self._h, _ptr(seq), seq.shape[0], seq.shape[1], _stream_ptr(seq.device)))

    def sequence_num_outputs(self, L: int, stride_frames: int = 1) -> int:
        return int(self._lib.km_sequence_num_outputs(self._h, L, stride_frames))
