"""Drop-in mirror of the reference's ``MelSpectrogramExtractor``
(reference src/features/stft.py:23-172), the front end scripts/rt.py instantiates.

Same constructor arguments, ``forward(waveform (B, L) | (L,)) -> (B, T, n_mels)``, ``hop_length``,
``get_output_length``, ``get_time_axis``.  The torchaudio pipeline (periodic Hann, reflect-centred STFT,
"window" normalisation, HTK filterbank, log(mel + eps), truncate / repeat-last-frame to
int(L / sr * fps) frames) runs in the HIP front end through km_mel_extract.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ..engine import Engine, MelConfig


class MelSpectrogramExtractor(nn.Module):
    def __init__(
        self,
        sample_rate: int = 16000,
        target_fps: float = 30.0,
        n_fft: int = 512,
        n_mels: int = 80,
        f_min: float = 80.0,
        f_max: Optional[float] = None,
        power: float = 2.0,
        normalized: bool = True,
        center: bool = True,
        pad_mode: str = "reflect",
        eps: float = 1e-8,
        engine: Optional[Engine] = None,
    ):
        super().__init__()
        self.sample_rate = sample_rate
        self.target_fps = target_fps
        self.n_fft = n_fft
        self.n_mels = n_mels
        self.f_min = f_min
        self.f_max = f_max or sample_rate // 2
        self.power = power
        self.eps = eps
        self.hop_length = int(sample_rate / target_fps)
        self.win_length = n_fft
        if self.hop_length <= 0:                                         # reference :78-81
            raise ValueError(f"Invalid hop_length {self.hop_length} for sr={sample_rate}, fps={target_fps}")
        if power != 2.0 or not center or pad_mode not in ("reflect", "constant"):
            raise ValueError("the HIP front end implements power=2.0, center=True, pad_mode reflect|constant")
        self.center = center
        self.cfg = MelConfig.torchaudio(sample_rate, target_fps, n_fft, n_mels, f_min, self.f_max, normalized,
                                        pad_mode, eps)
        self._engine = engine
        self.register_buffer("_anchor", torch.zeros(1), persistent=False)

    def _eng(self, device) -> Engine:
        if self._engine is None:
            # a front-end-only handle still carries a (zero) core state dict: the C-ABI has one handle type
            from .. import synth
            self._engine = Engine(mel=MelConfig.model_batch(self.sample_rate, self.target_fps))
            self._engine.load_state_dict(synth.make_core_params(0))
            self._engine.finalize(device)
        return self._engine

    def forward(self, waveform: torch.Tensor) -> torch.Tensor:
        if waveform.dim() == 1:
            waveform = waveform.unsqueeze(0)
        if waveform.dim() != 2:
            raise ValueError(f"Expected 1D or 2D input, got {waveform.dim()}D")          # reference :115-116
        if not waveform.is_cuda:
            raise RuntimeError("MelSpectrogramExtractor runs on the GPU only (no CPU fallback by design)")
        expected = int(waveform.shape[1] / self.sample_rate * self.target_fps)           # :130
        if expected == 0:
            return torch.empty(waveform.shape[0], 0, self.n_mels, device=waveform.device)
        return self._eng(waveform.device).mel_extract(self.cfg, waveform, out_frames=expected)

    def get_output_length(self, input_length: int) -> int:
        if self.center:
            input_length += 2 * (self.n_fft // 2)
        return (input_length - self.n_fft) // self.hop_length + 1

    def get_time_axis(self, seq_length: int) -> torch.Tensor:
        return torch.arange(seq_length, dtype=torch.float32) * self.hop_length / self.sample_rate
