"""Drop-in mirror of the reference's ``DualStreamCrossAttention``
(reference src/model/dual_stream_attention.py:48-280).

Same constructor arguments, same parameter names / shapes / initialisation (the torch containers
``nn.Linear`` / ``nn.MultiheadAttention`` / ``nn.LayerNorm`` are instantiated exactly as the
reference does, so ``state_dict()`` / ``load_state_dict()`` round-trip reference checkpoints), same
``forward`` signature and return dictionary.  The containers only HOLD the weights: ``forward``
runs the hand-written HIP path in libkoemorph_hip.so through the C-ABI (koemorph_amd.engine).
There is no eager fallback.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from ..engine import Engine, MelConfig

# ARKit blendshape grouping (dual_stream_attention.py:14-45)
MOUTH_BLENDSHAPES = [
    'jawForward', 'jawLeft', 'jawRight', 'jawOpen',
    'mouthClose', 'mouthFunnel', 'mouthPucker', 'mouthLeft', 'mouthRight',
    'mouthSmileLeft', 'mouthSmileRight', 'mouthFrownLeft', 'mouthFrownRight',
    'mouthDimpleLeft', 'mouthDimpleRight', 'mouthStretchLeft', 'mouthStretchRight',
    'mouthRollLower', 'mouthRollUpper', 'mouthShrugLower', 'mouthShrugUpper',
    'mouthPressLeft', 'mouthPressRight', 'mouthLowerDownLeft', 'mouthLowerDownRight',
    'mouthUpperUpLeft', 'mouthUpperUpRight',
    'tongueOut',
]
ARKIT_BLENDSHAPES = [
    'eyeBlinkLeft', 'eyeLookDownLeft', 'eyeLookInLeft', 'eyeLookOutLeft', 'eyeLookUpLeft',
    'eyeSquintLeft', 'eyeWideLeft', 'eyeBlinkRight', 'eyeLookDownRight', 'eyeLookInRight',
    'eyeLookOutRight', 'eyeLookUpRight', 'eyeSquintRight', 'eyeWideRight', 'jawForward',
    'jawLeft', 'jawRight', 'jawOpen', 'mouthClose', 'mouthFunnel', 'mouthPucker',
    'mouthLeft', 'mouthRight', 'mouthSmileLeft', 'mouthSmileRight', 'mouthFrownLeft',
    'mouthFrownRight', 'mouthDimpleLeft', 'mouthDimpleRight', 'mouthStretchLeft',
    'mouthStretchRight', 'mouthRollLower', 'mouthRollUpper', 'mouthShrugLower',
    'mouthShrugUpper', 'mouthPressLeft', 'mouthPressRight', 'mouthLowerDownLeft',
    'mouthLowerDownRight', 'mouthUpperUpLeft', 'mouthUpperUpRight', 'browDownLeft',
    'browDownRight', 'browInnerUp', 'browOuterUpLeft', 'browOuterUpRight', 'cheekPuff',
    'cheekSquintLeft', 'cheekSquintRight', 'noseSneerLeft', 'noseSneerRight', 'tongueOut',
]
MOUTH_INDICES = [i for i, name in enumerate(ARKIT_BLENDSHAPES) if name in MOUTH_BLENDSHAPES]
EXPRESSION_INDICES = [i for i in range(52) if i not in MOUTH_INDICES]


class DualStreamCrossAttention(nn.Module):
    """Dual-stream cross-attention (mel stream -> 28 mouth rows, eGeMAPS stream -> 24 expression rows)."""

    def __init__(
        self,
        d_model: int = 256,
        num_heads: int = 8,
        num_mel_channels: int = 80,
        mel_sequence_length: int = 256,
        mel_temporal_frames: int = 3,
        emotion_dim: int = 256,
        emotion_sequence_length: int = 1,
        dropout: float = 0.1,
        num_blendshapes: int = 52,
        use_learnable_weights: bool = True,
        temperature: float = 1.0,
        mel_config: Optional[MelConfig] = None,
    ):
        super().__init__()
        self.d_model = d_model
        self.num_heads = num_heads
        self.num_mel_channels = num_mel_channels
        self.mel_sequence_length = mel_sequence_length
        self.mel_temporal_frames = mel_temporal_frames
        self.emotion_dim = emotion_dim
        self.emotion_sequence_length = emotion_sequence_length
        self.num_blendshapes = num_blendshapes
        self.temperature = temperature
        self.dropout = dropout

        # parameter containers, constructed as in the reference (:101-159) so names/shapes/init match
        self.total_mel_dim = num_mel_channels * (mel_sequence_length + mel_temporal_frames)
        self.mel_channel_encoder = nn.Linear(mel_sequence_length + mel_temporal_frames, d_model)
        self.mel_attention = nn.MultiheadAttention(embed_dim=d_model, num_heads=num_heads, dropout=dropout,
                                                   batch_first=True)
        self.emotion_encoder = nn.Linear(emotion_dim, d_model)
        self.emotion_attention = nn.MultiheadAttention(embed_dim=d_model, num_heads=num_heads, dropout=dropout,
                                                       batch_first=True)
        self.mouth_queries = nn.Parameter(torch.randn(len(MOUTH_INDICES), d_model) * 0.02)
        self.expression_queries = nn.Parameter(torch.randn(len(EXPRESSION_INDICES), d_model) * 0.02)
        if use_learnable_weights:
            self.mel_weights = nn.Parameter(torch.ones(num_blendshapes))
            self.emotion_weights = nn.Parameter(torch.ones(num_blendshapes))
            with torch.no_grad():
                self.mel_weights[MOUTH_INDICES] = 2.0
                self.mel_weights[EXPRESSION_INDICES] = 0.5
                self.emotion_weights[MOUTH_INDICES] = 0.5
                self.emotion_weights[EXPRESSION_INDICES] = 2.0
        else:
            mel_weights = torch.zeros(num_blendshapes)
            emotion_weights = torch.zeros(num_blendshapes)
            mel_weights[MOUTH_INDICES] = 1.0
            emotion_weights[EXPRESSION_INDICES] = 1.0
            self.register_buffer('mel_weights', mel_weights)
            self.register_buffer('emotion_weights', emotion_weights)
        self.mel_output_proj = nn.Linear(d_model, d_model)
        self.emotion_output_proj = nn.Linear(d_model, d_model)
        self.blendshape_decoder = nn.Sequential(
            nn.Linear(d_model, d_model // 2),
            nn.ReLU(),
            nn.Dropout(dropout),
            nn.Linear(d_model // 2, 1),
            nn.Sigmoid(),
        )
        self.mel_norm = nn.LayerNorm(d_model)
        self.emotion_norm = nn.LayerNorm(d_model)

        self._mel_config = mel_config
        self._engine: Optional[Engine] = None
        self._engine_sig = None
        self._extra_params: Dict[str, torch.Tensor] = {}     # e.g. smoothing_alpha from the wrapper model

    # ---- engine plumbing ------------------------------------------------------------------------
    def _signature(self):
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        sig = [(k, v.data_ptr(), v._version) for k, v in sd.items()]
        sig += [(k, v.data_ptr(), v._version) for k, v in self._extra_params.items()]
        return tuple(sig)

    def engine(self) -> Engine:
        """The km_handle bound to this module's weights; (re)built when a weight tensor changed."""
        dev = self.mouth_queries.device
        if dev.type != "cuda":
            raise RuntimeError("DualStreamCrossAttention runs on the GPU only: move the module with "
                               ".to('cuda') (there is no CPU fallback by design)")
        sig = (str(dev),) + self._signature()
        if self._engine is None or self._engine_sig != sig:
            if self._engine is None:
                self._engine = Engine(d_model=self.d_model, num_heads=self.num_heads,
                                      num_mel_channels=self.num_mel_channels,
                                      mel_sequence_length=self.mel_sequence_length,
                                      mel_temporal_frames=self.mel_temporal_frames, emotion_dim=self.emotion_dim,
                                      num_blendshapes=self.num_blendshapes, temperature=self.temperature,
                                      mel=self._mel_config)
            state = {k: v for k, v in self.state_dict().items()}
            self._engine.load_state_dict(state)
            for k, v in self._extra_params.items():
                self._engine.load_param(k, v)
            self._engine.finalize(dev)
            self._engine_sig = sig
        return self._engine

    def require_eval_mode(self) -> None:
        """The inference kernels implement eval-mode arithmetic (dropout is the identity, the projections are folded).
        A module in train() mode with autograd on would silently optimise a different objective, so it raises; the
        train-mode forward + backward (dropout included) is koemorph_amd.training.Trainer."""
        if self.training and torch.is_grad_enabled() and self.dropout > 0:
            raise RuntimeError("the HIP forward implements eval-mode arithmetic (dropout is the identity); call "
                               ".eval() or use torch.no_grad() -- training goes through koemorph_amd.training")

    # ---- reference API --------------------------------------------------------------------------
    def forward(
        self,
        mel_features: torch.Tensor,
        mel_temporal_features: torch.Tensor,
        emotion_features: torch.Tensor,
        return_attention: bool = False,
    ) -> Dict[str, torch.Tensor]:
        """mel_features (B, T, 80), mel_temporal_features (B, 3, 80), emotion_features (B, emotion_dim)
        -> {'blendshapes': (B, 52)[, 'mel_attention_weights' (B, 28, 80), 'emotion_attention_weights'
        (B, 24, 1), 'mel_blendshapes', 'emotion_blendshapes']}   (reference :162-280)."""
        self.require_eval_mode()
        eng = self.engine()
        out = eng.core_forward(mel_features, mel_temporal_features, emotion_features, return_attention)
        result = {'blendshapes': out['blendshapes']}
        if return_attention:
            raw = out['raw']
            mel_bs = torch.zeros_like(raw)
            emo_bs = torch.zeros_like(raw)
            mel_bs[:, MOUTH_INDICES] = raw[:, MOUTH_INDICES]                       # :257-262
            emo_bs[:, EXPRESSION_INDICES] = raw[:, EXPRESSION_INDICES]
            result['mel_attention_weights'] = out['mel_attention_weights']
            # softmax over a single key is identically 1 (:234-239)
            result['emotion_attention_weights'] = torch.ones(raw.shape[0], len(EXPRESSION_INDICES), 1,
                                                             device=raw.device, dtype=raw.dtype)
            result['mel_blendshapes'] = mel_bs
            result['emotion_blendshapes'] = emo_bs
        return result

    def get_frequency_bands(self) -> Dict[str, List[int]]:
        return {'low': list(range(0, 20)), 'mid_low': list(range(20, 40)),
                'mid_high': list(range(40, 60)), 'high': list(range(60, 80))}
