"""Drop-in mirror of the reference's legacy ``SimplifiedKoeMorphModel``
(reference src/model/simplified_model.py:12-156) -- the single-stream variant used by src/train.py,
scripts/rt_simplified.py and scripts/test_model.py, and the only place where the north star's literal
"52 learnable queries cross-attending to ~256 mel frames" shape occurs.

Same constructor arguments, same parameter containers (state-dict keys ``audio_encoder.{0,3}.*``,
``attention.*``, ``decoder.{0,3,6}.*``, ``blendshape_queries``), ``forward(audio (B, T)) -> (B, 52)`` tensor,
``extract_mel_features``, ``get_num_parameters``, ``reset_temporal_state``.  Compute runs in
libkoemorph_hip.so (km_legacy_forward): HIP log-mel front end + exact-fp32 MFMA GEMM chain; eval-mode
arithmetic (the three Dropout(0.1) layers and the attention dropout are the identity).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .._lib import KMLegacyConfig, check
from ..engine import MelConfig, _ptr, _stream_ptr


class SimplifiedKoeMorphModel(nn.Module):
    def __init__(
        self,
        d_model: int = 256,
        d_query: int = 256,
        d_key: int = 256,
        d_value: int = 256,
        audio_encoder: dict = None,
        attention: dict = None,
        decoder: dict = None,
        smoothing: dict = None,
        num_blendshapes: int = 52,
        sample_rate: int = 16000,
        target_fps: int = 30,
    ):
        super().__init__()
        self.d_model = d_model
        self.num_blendshapes = num_blendshapes
        self.sample_rate = sample_rate
        self.target_fps = target_fps
        self.n_mels = 80
        self.hop_length = int(sample_rate // target_fps)
        self.n_fft = 1024
        self.num_heads = attention.get('num_heads', 8) if attention else 8
        self.decoder_hidden = decoder.get('hidden_dim', 128) if decoder else 128
        self.audio_encoder = nn.Sequential(
            nn.Linear(self.n_mels, d_model), nn.ReLU(), nn.Dropout(0.1),
            nn.Linear(d_model, d_model), nn.ReLU(), nn.Dropout(0.1),
        )
        self.attention = nn.MultiheadAttention(embed_dim=d_model, num_heads=self.num_heads,
                                               dropout=attention.get('dropout', 0.1) if attention else 0.1,
                                               batch_first=True)
        h = self.decoder_hidden
        self.decoder = nn.Sequential(
            nn.Linear(d_model, h), nn.ReLU(), nn.Dropout(0.1),
            nn.Linear(h, h), nn.ReLU(), nn.Dropout(0.1),
            nn.Linear(h, num_blendshapes), nn.Sigmoid(),
        )
        self.blendshape_queries = nn.Parameter(torch.randn(num_blendshapes, d_model) * 0.1)
        self._h: Optional[C.c_void_p] = None
        self._sig = None
        self._reserved = (0, 0)

    # ---- handle plumbing ------------------------------------------------------------------------
    def _handle(self):
        dev = self.blendshape_queries.device
        if dev.type != "cuda":
            raise RuntimeError("SimplifiedKoeMorphModel runs on the GPU only (there is no CPU fallback by design)")
        lib = _lib.load()
        sig = (str(dev),) + tuple((k, v.data_ptr(), v._version) for k, v in self.state_dict().items())
        if self._h is None:
            mel = MelConfig.model_batch(self.sample_rate, self.target_fps, self.n_fft)
            mel.hop_length = self.hop_length
            cfg = KMLegacyConfig(_lib.KM_ABI_VERSION, self.d_model, self.num_heads, self.decoder_hidden,
                                 self.num_blendshapes, mel.to_c())
            self._h = C.c_void_p()
            check(lib.km_legacy_create(C.byref(cfg), C.byref(self._h)))
        if self._sig != sig:
            for k, v in self.state_dict().items():
                a = np.ascontiguousarray(v.detach().cpu().numpy(), dtype=np.float32)
                shape = (C.c_int64 * max(1, a.ndim))(*a.shape)
                check(lib.km_load_param(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
            with torch.cuda.device(dev):
                check(lib.km_finalize(self._h, _stream_ptr(dev)))
            self._sig = sig
            self._reserved = (0, 0)
        return lib, self._h, dev

    def _reserve(self, lib, h, dev, B, L):
        if B > self._reserved[0] or L > self._reserved[1]:
            with torch.cuda.device(dev):
                torch.cuda.synchronize(dev)
                check(lib.km_reserve(h, max(B, self._reserved[0]), max(L, self._reserved[1])))
            self._reserved = (max(B, self._reserved[0]), max(L, self._reserved[1]))

    def __del__(self):  # pragma: no cover
        try:
            if self._h is not None and self._h.value:
                _lib.load().km_destroy(self._h)
        except Exception:
            pass

    # ---- reference API --------------------------------------------------------------------------
    def extract_mel_features(self, audio: torch.Tensor) -> torch.Tensor:
        """audio (B, T) -> (B, T_mel, 80) normalised log-mel (reference :79-112)."""
        lib, h, dev = self._handle()
        audio = audio.float().contiguous()
        B, L = audio.shape
        self._reserve(lib, h, dev, B, L)
        F = 1 + L // self.hop_length
        out = torch.empty(B, F, self.n_mels, device=dev)
        with torch.cuda.device(dev):
            check(lib.km_mel_batch(h, _ptr(audio), B, L, _ptr(out), 0, _stream_ptr(dev)))
        return out

    def forward(self, audio: torch.Tensor) -> torch.Tensor:
        """audio (B, T) -> blendshapes (B, 52)   (reference :114-149)."""
        if audio.dim() != 2:
            raise ValueError(f"Expected 2D input, got {audio.dim()}D")
        if self.training and torch.is_grad_enabled():
            raise RuntimeError("the HIP forward implements eval-mode arithmetic; call .eval() or torch.no_grad()")
        lib, h, dev = self._handle()
        audio = audio.float().contiguous()
        B, L = audio.shape
        self._reserve(lib, h, dev, B, L)
        out = torch.empty(B, self.num_blendshapes, device=dev)
        with torch.cuda.device(dev):
            check(lib.km_legacy_forward(h, _ptr(audio), B, L, _ptr(out), _stream_ptr(dev)))
        return out

    def forward_mel(self, mel_features: torch.Tensor) -> torch.Tensor:
        """Everything after ``extract_mel_features``: (B, T_mel, 80) -> (B, 52)."""
        lib, h, dev = self._handle()
        mel = mel_features.float().contiguous()
        B, T, _ = mel.shape
        self._reserve(lib, h, dev, B, max(self._reserved[1], T * self.hop_length))
        out = torch.empty(B, self.num_blendshapes, device=dev)
        with torch.cuda.device(dev):
            check(lib.km_legacy_forward_mel(h, _ptr(mel), B, T, _ptr(out), _stream_ptr(dev)))
        return out

    def get_num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def reset_temporal_state(self):
        """No-op, as in the reference (:155-156)."""
