"""Drop-in mirror of the reference's ``SequentialDualStreamModel``
(reference src/model/sequential_dual_stream_model.py:17-180).

``forward(audio (B, T)) -> {'blendshapes': (B, T_out, 52), 'num_frames', 'fps', ...}`` with one output
frame per window position (window = mel_sequence_length frames, stride = stride_frames).  The reference
walks the positions in a Python loop and recomputes the whole 257-frame mel on the CPU for each one
(:101-145); here every window of every clip is one workgroup of the same HIP kernels, addressed in place
inside the clip (km_sequence_forward), followed by an EMA scan along the frame axis.
"""
from __future__ import annotations

import logging
from typing import Dict, Optional

import torch

from .simplified_dual_stream_model import SimplifiedDualStreamModel

logger = logging.getLogger(__name__)


class SequentialDualStreamModel(SimplifiedDualStreamModel):
    def __init__(
        self,
        d_model: int = 256,
        num_heads: int = 8,
        num_blendshapes: int = 52,
        sample_rate: int = 16000,
        target_fps: int = 30,
        mel_sequence_length: int = 256,
        emotion_config: Optional[Dict] = None,
        device: str = "cuda",
        real_time_mode: bool = False,
        stride_frames: int = 1,
        emotion_provider=None,
        shard_across_ranks: bool = False,
    ):
        super().__init__(d_model=d_model, num_heads=num_heads, num_blendshapes=num_blendshapes,
                         sample_rate=sample_rate, target_fps=target_fps, mel_sequence_length=mel_sequence_length,
                         emotion_config=emotion_config, device=device, real_time_mode=real_time_mode,
                         emotion_provider=emotion_provider)
        self.stride_frames = stride_frames
        # not in the reference (single process): under torch.distributed the output frames of a clip are computed in contiguous
        # chunks, one per rank, and smoothed once over the gathered sequence (koemorph_amd.parallel.sequence_apply)
        self.shard_across_ranks = shard_across_ranks
        self.window_frames = mel_sequence_length                      # reference :51
        self.window_samples = self.window_frames * self.hop_length    # :54
        self.stride_samples = self.stride_frames * self.hop_length    # :55

    def forward(self, audio: torch.Tensor, return_attention: bool = False,
                emotion_features: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        if audio.dim() != 2:
            raise ValueError(f"Expected 2D input, got {audio.dim()}D")
        batch_size, audio_length = audio.shape
        emotion_metadata = {"backend_used": self.emotion_backend}
        if emotion_features is None:                                   # emotion features ONCE per clip (:88)
            emotion_features, emotion_metadata = self.extract_emotion_features(audio)
        eng = self.dual_stream_attention.engine()
        self.reset_temporal_state()                                    # :99
        self.dual_stream_attention.require_eval_mode()
        results: Dict[str, object] = {}
        if not return_attention and self.shard_across_ranks:
            from .. import parallel
            seq = parallel.sequence_apply(eng, audio, emotion_features, self.stride_frames, smooth=self.use_temporal_smoothing)
        elif not return_attention:
            seq = eng.sequence_forward(audio, emotion_features, self.stride_frames,
                                       smooth=self.use_temporal_smoothing)
        else:
            # attention maps are a visualisation aid: walk the positions like the reference does
            num_frames = audio_length // self.hop_length
            n_out = max(1, (num_frames - self.window_frames) // self.stride_frames + 1)
            frames, mel_att, emo_att = [], [], []
            for i in range(n_out):
                s = i * self.stride_samples
                e = min(s + self.window_samples, audio_length)
                win = audio[:, s:e]
                if e - s < self.window_samples:                        # zero-pad the last window (:111-115)
                    win = torch.nn.functional.pad(win, (0, self.window_samples - (e - s)))
                o = SimplifiedDualStreamModel.forward(self, win.contiguous(), True, emotion_features)
                frames.append(o['blendshapes'])
                mel_att.append(o['mel_attention_weights'])
                emo_att.append(o['emotion_attention_weights'])
            seq = torch.stack(frames, dim=1)
            results['mel_attention_weights'] = torch.stack(mel_att, dim=1)
            results['emotion_attention_weights'] = torch.stack(emo_att, dim=1)
        if self.use_temporal_smoothing and seq.shape[1] > 0:
            self.prev_blendshapes = seq[:, -1].clone()                 # the state the reference is left with
        results['blendshapes'] = seq
        results['num_frames'] = seq.shape[1]
        results['fps'] = self.target_fps
        results['emotion_backend'] = emotion_metadata.get("backend_used", "unknown")
        results['emotion_processing_time'] = emotion_metadata.get("processing_time", 0.0)
        return results

    def forward_single_frame(self, audio: torch.Tensor, frame_idx: int = None) -> Dict[str, torch.Tensor]:
        return SimplifiedDualStreamModel.forward(self, audio, return_attention=False)
