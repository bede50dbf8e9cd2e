"""Drop-in mirrors of the reference's whole-audio wrappers.

  SimplifiedDualStreamModel   reference src/model/simplified_dual_stream_model.py:23-522
  (SequentialDualStreamModel lives in sequential_dual_stream_model.py)

Same constructor arguments, ``forward(audio, return_attention=False) -> dict``,
``extract_mel_features``, ``apply_temporal_smoothing``, ``reset_temporal_state``,
``process_audio_frame_realtime`` and state-dict layout (``dual_stream_attention.*`` +
``smoothing_alpha``).  The mel front end, the attention core and the smoothing run as HIP kernels
through the C-ABI; there is no CPU path.

Emotion features.  In the reference the 256-D vector comes from openSMILE eGeMAPS / emotion2vec
(third-party CPU libraries that also fetch models; SURVEY.md section 2 marks them out of scope).
Here it is an INPUT: pass ``emotion_features=`` to ``forward`` or install an ``emotion_provider``
callable ``(audio (B, L) tensor) -> (B, emotion_dim) tensor``.  Without either, the model does
what the reference does when extraction fails (simplified_dual_stream_model.py:250-267): it logs
a warning and uses ``randn * 0.1`` dummy features.
"""
from __future__ import annotations

import logging
from typing import Any, Callable, Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from ..engine import MelConfig
from .dual_stream_attention import DualStreamCrossAttention

logger = logging.getLogger(__name__)

EmotionProvider = Callable[[torch.Tensor], torch.Tensor]


class SimplifiedDualStreamModel(nn.Module):
    def __init__(
        self,
        d_model: int = 256,
        num_heads: int = 8,
        num_blendshapes: int = 52,
        sample_rate: int = 16000,
        target_fps: int = 30,
        mel_sequence_length: int = 256,
        emotion_config: Optional[Dict] = None,
        mel_config: Optional[Dict] = None,
        device: str = "cuda",
        real_time_mode: bool = False,
        emotion_provider: Optional[EmotionProvider] = None,
    ):
        super().__init__()
        self.d_model = d_model
        self.num_blendshapes = num_blendshapes
        self.sample_rate = sample_rate
        self.target_fps = target_fps
        self.mel_sequence_length = mel_sequence_length
        self.device = device
        self.real_time_mode = real_time_mode

        self.n_mels = 80
        self.hop_length = int(sample_rate / target_fps)      # reference :54 (533 @30 fps, 266 @60 fps)
        self.n_fft = 1024                                    # :55

        # emotion dimension by backend, as the reference resolves it (:92-108)
        emotion_config = dict(emotion_config or {})
        backend = emotion_config.get("backend", "opensmile")
        self.emotion_backend = backend
        if "emotion_dim" in emotion_config:
            self.emotion_dim = int(emotion_config["emotion_dim"])
        elif backend == "emotion2vec":
            self.emotion_dim = 1024
        elif backend == "basic":
            self.emotion_dim = 9
        else:                                                # opensmile, concatenated 3x88 -> 256 (production)
            self.emotion_dim = 256
        self.emotion_provider = emotion_provider

        mel_config = dict(mel_config or {})
        self.mel_context_window = mel_config.get("context_window", 8.5)
        self.mel_update_interval = mel_config.get("update_interval", 0.0333)
        batch_mel = MelConfig.model_batch(sample_rate, target_fps, self.n_fft)

        self.dual_stream_attention = DualStreamCrossAttention(
            d_model=d_model, num_heads=num_heads, num_mel_channels=self.n_mels,
            mel_sequence_length=mel_sequence_length, mel_temporal_frames=3, emotion_dim=self.emotion_dim,
            dropout=0.1, num_blendshapes=num_blendshapes, use_learnable_weights=True, temperature=1.0,
            mel_config=batch_mel,
        )
        self.use_temporal_smoothing = True
        self.smoothing_alpha = nn.Parameter(torch.tensor(0.8))           # :163
        self.prev_blendshapes: Optional[torch.Tensor] = None             # :164
        self.dual_stream_attention._extra_params["smoothing_alpha"] = self.smoothing_alpha

        if self.real_time_mode:
            from ..features.mel_sliding_window import MelSlidingWindowExtractor
            self.mel_extractor = MelSlidingWindowExtractor(
                context_window=self.mel_context_window, update_interval=self.mel_update_interval,
                sample_rate=sample_rate, n_mels=self.n_mels, n_fft=mel_config.get("n_fft", 1024),
                hop_length=self.hop_length, f_min=mel_config.get("f_min", 80.0),
                f_max=mel_config.get("f_max", sample_rate // 2), device=device,
                engine_getter=lambda: self.dual_stream_attention.engine())
        else:
            self.mel_extractor = None

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        # nn.Module._apply may REPLACE parameters (e.g. .to('cuda')): keep the alias fresh
        self.dual_stream_attention._extra_params["smoothing_alpha"] = self.smoothing_alpha
        if self.prev_blendshapes is not None:
            self.prev_blendshapes = fn(self.prev_blendshapes)
        return r

    # ---- features -------------------------------------------------------------------------------
    def extract_mel_features(self, audio: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """audio (B, T) -> (long (B, T_mel, 80), short (B, 3, 80))   (reference :166-229)."""
        return self.dual_stream_attention.engine().mel_batch(audio)

    def extract_emotion_features(self, audio: torch.Tensor) -> Tuple[torch.Tensor, Dict]:
        if self.emotion_provider is not None:
            feats = self.emotion_provider(audio)
            feats = torch.as_tensor(feats, dtype=torch.float32, device=audio.device)
            if feats.ndim == 1:
                feats = feats.unsqueeze(0)
            return feats, {"backend_used": self.emotion_backend}
        logger.warning("Emotion extraction failed, using dummy features")      # reference :250-259
        dummy = torch.randn(audio.shape[0], self.emotion_dim, device=audio.device) * 0.1
        return dummy, {"backend_used": "dummy", "extraction_failed": True}

    def align_features(self, mel_features, emotion_features):
        return mel_features, emotion_features          # identity for concatenated eGeMAPS (:317-322)

    # ---- smoothing ------------------------------------------------------------------------------
    def apply_temporal_smoothing(self, blendshapes: torch.Tensor) -> torch.Tensor:
        """EMA with alpha = sigmoid(smoothing_alpha); first call / batch-size change passes through
        (reference :341-368)."""
        if not self.use_temporal_smoothing:
            return blendshapes
        eng = self.dual_stream_attention.engine()
        first = self.prev_blendshapes is None or self.prev_blendshapes.shape[0] != blendshapes.shape[0]
        if first:
            self.prev_blendshapes = torch.empty_like(blendshapes)
        out = blendshapes.contiguous().clone()
        eng.smooth(out, self.prev_blendshapes, first)
        return out

    def reset_temporal_state(self):
        self.prev_blendshapes = None

    # ---- forward --------------------------------------------------------------------------------
    def forward(self, audio: torch.Tensor, return_attention: bool = False,
                emotion_features: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """audio (B, T) -> {'blendshapes': (B, 52)[, attention extras]}   (reference :370-415)."""
        if audio.dim() != 2:
            raise ValueError(f"Expected 2D input, got {audio.dim()}D")
        self.dual_stream_attention.require_eval_mode()      # the fused path below never reaches that module's forward
        if emotion_features is None:
            emotion_features, _ = self.extract_emotion_features(audio)
        eng = self.dual_stream_attention.engine()
        if not return_attention:
            # fused path: emotion kernel + front end + core (+ EMA in the core epilogue)
            B = audio.shape[0]
            state, first = None, True
            if self.use_temporal_smoothing:
                first = self.prev_blendshapes is None or self.prev_blendshapes.shape[0] != B
                if first:
                    self.prev_blendshapes = torch.empty(B, self.num_blendshapes, device=audio.device)
                state = self.prev_blendshapes
            return {'blendshapes': eng.forward_audio(audio, emotion_features, state=state, first=first)}
        mel_features, mel_temporal = self.extract_mel_features(audio)
        output = self.dual_stream_attention(mel_features=mel_features, mel_temporal_features=mel_temporal,
                                            emotion_features=emotion_features, return_attention=True)
        output['blendshapes'] = self.apply_temporal_smoothing(output['blendshapes'])
        return output

    def get_model_info(self) -> Dict[str, Any]:
        return {
            'model_type': 'SimplifiedDualStreamModel',
            'd_model': self.d_model,
            'num_heads': self.dual_stream_attention.num_heads,
            'num_blendshapes': self.num_blendshapes,
            'emotion_backend': self.emotion_backend,
            'mel_sequence_length': self.mel_sequence_length,
            'n_mels': self.n_mels,
            'emotion_dim': self.emotion_dim,
            'total_parameters': sum(p.numel() for p in self.parameters() if p.requires_grad),
            'real_time_mode': self.real_time_mode,
        }

    # ---- real-time ------------------------------------------------------------------------------
    def process_audio_frame_realtime(self, audio_frame: np.ndarray, return_attention: bool = False,
                                     emotion_features: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """One ~hop-sized frame in, (52,) coefficients out, or None while the 8.5 s ring is filling
        (reference :452-498).  The reference omits the required ``mel_temporal_features`` argument at
        :491-495 and therefore raises; here the three short-term rows are the last three frames of the
        sliding-window mel, which is what its batch path feeds (:206-207)."""
        if not self.real_time_mode:
            raise RuntimeError("Model not in real-time mode. Use forward() for batch processing.")
        if self.mel_extractor is None:
            raise RuntimeError("Mel extractor not initialized for real-time mode.")
        mel = self.mel_extractor.process_audio_frame_device(audio_frame)
        if mel is None:
            return None
        dev = mel.device
        mel = mel.unsqueeze(0)                                               # (1, T, 80)
        if emotion_features is None:
            emotion_features, _ = self.extract_emotion_features(
                torch.from_numpy(np.asarray(audio_frame, np.float32)).unsqueeze(0).to(dev))
        short = mel[:, -3:, :].contiguous()
        out = self.dual_stream_attention(mel_features=mel, mel_temporal_features=short,
                                         emotion_features=emotion_features, return_attention=return_attention)
        return self.apply_temporal_smoothing(out['blendshapes']).squeeze(0)

    def reset_realtime_state(self):
        if self.real_time_mode and self.mel_extractor:
            self.mel_extractor.reset()
        self.reset_temporal_state()

    def get_realtime_stats(self) -> Dict[str, Any]:
        if not self.real_time_mode:
            return {"error": "Not in real-time mode"}
        return {"mel_stats": self.mel_extractor.get_stats() if self.mel_extractor else {}}
