"""Oracle: the whole-audio wrappers, composed from oracle.mel / oracle.core / oracle.smoothing.

TEST INFRASTRUCTURE (see oracle/__init__.py).
  * simplified_forward  follows SimplifiedDualStreamModel.forward
    (/root/reference/src/model/simplified_dual_stream_model.py:370-415)
  * sequential_forward  follows SequentialDualStreamModel.forward
    (/root/reference/src/model/sequential_dual_stream_model.py:63-167)

The 256-D emotion vector is an opaque INPUT here: in the reference it comes from openSMILE /
emotion2vec (third-party, out of scope); ``align_features`` is the identity
for the concatenated-eGeMAPS configuration (simplified_dual_stream_model.py:317-322).
"""

from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import core, mel, smoothing


class SimplifiedOracle:
    def __init__(self, params: Dict[str, np.ndarray], smoothing_alpha: float = 0.8,
                 num_heads: int = 8, mel_sequence_length: int = 256, sample_rate: int = 16000,
                 target_fps: int = 30, n_fft: int = 1024, precision: str = "ref"):
        self.params = params
        self.num_heads = num_heads
        self.mel_sequence_length = mel_sequence_length
        self.sample_rate = sample_rate
        self.hop = int(sample_rate / target_fps)        # simplified_dual_stream_model.py:54
        self.n_fft = n_fft                              # :55
        self.precision = precision
        self.smoother = smoothing.TemporalSmootherOracle(smoothing_alpha)

    def reset_temporal_state(self):
        self.smoother.reset()

    def extract_mel_features(self, audio: np.ndarray):
        return mel.mel_batch(audio, sample_rate=self.sample_rate, n_fft=self.n_fft, hop=self.hop,
                             precision=self.precision)

    def forward(self, audio: np.ndarray, emotion: np.ndarray, return_attention: bool = False,
                smooth: bool = True) -> Dict[str, np.ndarray]:
        long, short = self.extract_mel_features(audio)                      # :390
        out = core.core_forward_np(self.params, long, short, emotion, num_heads=self.num_heads,
                                   mel_sequence_length=self.mel_sequence_length,
                                   return_attention=return_attention)       # :405-410
        if smooth:
            out["blendshapes"] = self.smoother(out["blendshapes"])          # :413
        return out


class SequentialOracle(SimplifiedOracle):
    def __init__(self, *a, stride_frames: int = 1, **kw):
        super().__init__(*a, **kw)
        self.stride_frames = stride_frames
        self.window_frames = self.mel_sequence_length                       # :51
        self.window_samples = self.window_frames * self.hop                 # :54

    def forward(self, audio: np.ndarray, emotion: np.ndarray, return_attention: bool = False
                ) -> Dict[str, np.ndarray]:
        B, L = audio.shape
        num_frames = L // self.hop                                          # :84
        n_out = max(1, (num_frames - self.window_frames) // self.stride_frames + 1)   # :96
        self.reset_temporal_state()                                         # :99
        frames, attn = [], []
        for i in range(n_out):                                              # :101
            s = i * self.stride_frames * self.hop
            e = min((i * self.stride_frames + self.window_frames) * self.hop, L)
            if e - s < self.window_samples:                                 # :111-115 zero-pad tail
                win = np.zeros((B, self.window_samples), np.float32)
                win[:, :e - s] = audio[:, s:e]
            else:
                win = audio[:, s:e]
            o = SimplifiedOracle.forward(self, win, emotion, return_attention)
            frames.append(o["blendshapes"])
            if return_attention:
                attn.append(o["mel_attention_weights"])
        res = {"blendshapes": np.stack(frames, axis=1), "num_frames": n_out}   # :151-160
        if return_attention:
            res["mel_attention_weights"] = np.stack(attn, axis=1)
        return res
