"""Drop-in mirror of the reference's legacy multi-layer ``KoeMorphModel``
(reference src/model/gaussian_face.py:29-368) -- the class ``create_koemorph_model`` returns and
``scripts/rt.py:283-304`` loads.

Same constructor arguments, the same parameter containers (so a reference checkpoint loads with
``load_state_dict(strict=True)``, buffers of the smoother / constraints included), the same
``forward(mel_features, emotion_features, audio_mask, prev_blendshapes, apply_smoothing, apply_constraints,
return_attention) -> dict`` with ``blendshapes`` / ``raw_blendshapes`` / ``attention_weights``,
``inference_step``, ``reset_temporal_state``, ``get_num_parameters``, ``get_model_info``.  Compute runs in
libkoemorph_hip.so (``km_koemorph_forward``): exact-fp32 MFMA GEMMs + row kernels, eval-mode arithmetic (all
Dropout layers are the identity).  Differences, all loud:
  * ``d_query`` must equal ``d_model`` (the reference's own residual ``attn_out + attention_output`` raises otherwise,
    its default ``d_query=128`` included);
  * smoothing methods other than "exponential", ``output_activation`` other than "sigmoid" and decoder activations
    other than relu / gelu raise ``NotImplementedError``;
  * there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .._lib import KMKoeMorphConfig, check
from ..engine import _ptr, _stream_ptr


class _DualStreamEncoder(nn.Module):
    """Parameter container with the layout of the reference's DualStreamEncoder (dual_stream_attention.py:296-351)."""

    def __init__(self, mel_dim, emotion_dim, d_model, num_layers, dropout):
        super().__init__()
        self.mel_encoder = nn.Sequential(nn.Linear(mel_dim, d_model), nn.ReLU(), nn.Dropout(dropout), nn.LayerNorm(d_model))
        self.emotion_encoder = nn.Sequential(nn.Linear(emotion_dim, d_model), nn.ReLU(), nn.Dropout(dropout), nn.LayerNorm(d_model))
        if num_layers > 0:
            layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=8, dim_feedforward=d_model * 4, dropout=dropout,
                                               activation="gelu", batch_first=True)
            self.mel_transformer = nn.TransformerEncoder(layer, num_layers=num_layers)
            self.emotion_transformer = nn.TransformerEncoder(layer, num_layers=num_layers)


class _QueryEmbedding(nn.Module):
    def __init__(self, num_blendshapes, d_query, dropout):
        super().__init__()
        self.query_embeddings = nn.Parameter(torch.randn(num_blendshapes, d_query))
        self.conditioning_net = nn.Sequential(nn.Linear(num_blendshapes, d_query // 2), nn.ReLU(), nn.Dropout(dropout),
                                              nn.Linear(d_query // 2, d_query))
        nn.init.xavier_uniform_(self.query_embeddings)


class _CrossAttention(nn.Module):
    def __init__(self, d_query, d_model):
        super().__init__()
        self.q_proj = nn.Linear(d_query, d_model)
        self.k_proj = nn.Linear(d_model, d_model)
        self.v_proj = nn.Linear(d_model, d_model)
        self.out_proj = nn.Linear(d_model, d_model)
        for m in (self.q_proj, self.k_proj, self.v_proj, self.out_proj):
            nn.init.xavier_uniform_(m.weight)
            nn.init.zeros_(m.bias)


class _Decoder(nn.Module):
    def __init__(self, d_model, hidden_dim, num_blendshapes, num_layers):
        super().__init__()
        self.input_proj = nn.Linear(d_model, hidden_dim)
        self.hidden_layers = nn.ModuleList([nn.Linear(hidden_dim, hidden_dim) for _ in range(num_layers)])
        self.layer_norms = nn.ModuleList([nn.LayerNorm(hidden_dim) for _ in range(num_layers)])
        self.output_proj = nn.Linear(hidden_dim, num_blendshapes)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)


class _Smoother(nn.Module):
    """State-dict layout of TemporalSmoother(learnable=True) (decoder.py:212-236): ``alpha`` is a parameter only for the
    exponential method, ``gaussian_weights`` (ones / window) only for the gaussian one, median has no parameter."""

    def __init__(self, num_blendshapes, alpha, window_size=5, method="exponential"):
        super().__init__()
        self.window_size = window_size
        if method == "exponential":
            self.alpha = nn.Parameter(torch.tensor(float(alpha)))
        elif method == "gaussian":
            self.gaussian_weights = nn.Parameter(torch.ones(window_size) / window_size)
        self.register_buffer("prev_output", torch.zeros(1, num_blendshapes))
        self.register_buffer("history", torch.zeros(window_size, 1, num_blendshapes))
        self.register_buffer("history_ptr", torch.tensor(0, dtype=torch.long))


class _Constraints(nn.Module):
    def __init__(self, num_blendshapes):
        super().__init__()
        self.register_buffer("min_values_buf", torch.zeros(num_blendshapes))
        self.register_buffer("max_values_buf", torch.ones(num_blendshapes))
        self.register_buffer("prev_blendshapes", torch.zeros(1, num_blendshapes))


_DECODER_ACT = {"relu": 0, "gelu": 1, "swish": 2, "leaky_relu": 3}          # km_koemorph_config.decoder_activation
_OUTPUT_ACT = {"sigmoid": 0, "tanh": 1, "none": 2}
_SMOOTHING = {"exponential": 0, "gaussian": 1, "median": 2}


class KoeMorphModel(nn.Module):
    """Mirror of the reference's KoeMorphModel (src/model/gaussian_face.py:29-268) on km_koemorph_*.

    ``smoothing_method`` "gaussian" / "median": the reference's TemporalSmoother raises on the first call of either
    (decoder.py:339 assigns a Python int to the registered buffer ``history_ptr`` -> TypeError), so no reference output
    exists for them; what is built is what that code evidently means -- a ring of ``window_size`` = 5 past outputs, one slot
    overwritten per call, combined by the softmax of the learnable slot weights or by torch.median -- pinned to
    oracle/koemorph_model.py only (parity unpinned).  Everything else is pinned to the reference's own outputs."""

    def __init__(
        self,
        mel_dim: int = 80,
        emotion_dim: int = 256,
        d_model: int = 256,
        d_query: int = 128,
        d_key: int = 256,
        d_value: int = 256,
        num_heads: int = 8,
        num_encoder_layers: int = 2,
        num_attention_layers: int = 4,
        attention_dropout: float = 0.1,
        decoder_hidden_dim: int = 128,
        decoder_layers: int = 2,
        decoder_activation: str = "gelu",
        output_activation: str = "sigmoid",
        use_temporal_smoothing: bool = True,
        smoothing_method: str = "exponential",
        smoothing_alpha: float = 0.8,
        use_constraints: bool = True,
        causal: bool = True,
        window_size: Optional[int] = 30,
        num_blendshapes: int = 52,
        dropout: float = 0.1,
    ):
        super().__init__()
        if d_query != d_model:
            raise ValueError(f"d_query ({d_query}) must equal d_model ({d_model}): the reference adds the attention output "
                             "(d_model) to the queries (d_query) and fails otherwise (gaussian_face.py:230-231)")
        if d_model % num_heads != 0:                       # attention.py:68-71
            raise ValueError(f"d_model ({d_model}) must be divisible by num_heads ({num_heads})")
        if decoder_activation not in _DECODER_ACT:                       # decoder.py:68-77
            raise ValueError(f"Unknown activation: {decoder_activation}")
        if output_activation not in _OUTPUT_ACT:                         # decoder.py:162-169 (raised there at the first forward)
            raise ValueError(f"Unknown output activation: {output_activation}")
        if use_temporal_smoothing and smoothing_method not in _SMOOTHING:  # decoder.py:260-273 (likewise)
            raise ValueError(f"Unknown smoothing method: {smoothing_method}")
        self.mel_dim, self.emotion_dim, self.d_model, self.num_blendshapes = mel_dim, emotion_dim, d_model, num_blendshapes
        self.num_heads = num_heads
        self.use_temporal_smoothing, self.use_constraints = use_temporal_smoothing, use_constraints
        self.causal, self.window_size = causal, window_size
        self.decoder_activation, self.output_activation, self.smoothing_method = decoder_activation, output_activation, smoothing_method
        self.audio_encoder = _DualStreamEncoder(mel_dim, emotion_dim, d_model, num_encoder_layers, dropout)
        self.num_encoder_layers = num_encoder_layers
        self.query_embeddings = _QueryEmbedding(num_blendshapes, d_query, dropout)
        self.cross_attention_layers = nn.ModuleList([_CrossAttention(d_query, d_model) for _ in range(num_attention_layers)])
        self.attention_layer_norms = nn.ModuleList([nn.LayerNorm(d_model) for _ in range(num_attention_layers)])
        self.decoder = _Decoder(d_model, decoder_hidden_dim, num_blendshapes, decoder_layers)
        if use_temporal_smoothing:
            self.temporal_smoother = _Smoother(num_blendshapes, smoothing_alpha, method=smoothing_method)
        if use_constraints:
            self.constraints = _Constraints(num_blendshapes)
        self._h: Optional[C.c_void_p] = None
        self._sig = None
        self._reserved = (0, 0)
        self._smoother_state: Optional[torch.Tensor] = None

    # ---- handle plumbing ------------------------------------------------------------------------
    def _c_config(self) -> KMKoeMorphConfig:
        return KMKoeMorphConfig(_lib.KM_ABI_VERSION, self.mel_dim, self.emotion_dim, self.d_model, self.num_heads,
                                self.num_encoder_layers, len(self.cross_attention_layers), self.decoder.input_proj.out_features,
                                len(self.decoder.hidden_layers), _DECODER_ACT[self.decoder_activation],
                                1 if self.causal else 0, -1 if self.window_size is None else int(self.window_size),
                                1 if self.use_temporal_smoothing else 0, 1 if self.use_constraints else 0, self.num_blendshapes,
                                _OUTPUT_ACT[self.output_activation], _SMOOTHING[self.smoothing_method],
                                self.temporal_smoother.window_size if self.use_temporal_smoothing else 5)

    def _learnable(self):
        return {k: v for k, v in self.named_parameters()}

    def _handle(self):
        dev = self.query_embeddings.query_embeddings.device
        if dev.type != "cuda":
            raise RuntimeError("KoeMorphModel runs on the GPU only (there is no CPU fallback by design)")
        lib = _lib.load()
        params = self._learnable()
        sig = (str(dev),) + tuple((k, v.data_ptr(), v._version) for k, v in params.items())
        if self._h is None:
            cfg = self._c_config()
            self._h = C.c_void_p()
            check(lib.km_koemorph_create(C.byref(cfg), C.byref(self._h)))
        if self._sig != sig:
            for k, v in params.items():
                a = np.ascontiguousarray(v.detach().cpu().numpy(), dtype=np.float32)
                shape = (C.c_int64 * max(1, v.dim()))(*v.shape)          # v.dim() may be 0 (temporal_smoother.alpha)
                check(lib.km_load_param(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), shape, v.dim()))
            with torch.cuda.device(dev):
                check(lib.km_finalize(self._h, _stream_ptr(dev)))
            self._sig = sig
            self._reserved = (0, 0)
        return lib, self._h, dev

    def __del__(self):  # pragma: no cover
        try:
            if self._h is not None and self._h.value:
                _lib.load().km_destroy(self._h)
        except Exception:
            pass

    # ---- reference API --------------------------------------------------------------------------
    def forward(self, mel_features: torch.Tensor, emotion_features: torch.Tensor, audio_mask: Optional[torch.Tensor] = None,
                prev_blendshapes: Optional[torch.Tensor] = None, apply_smoothing: bool = True, apply_constraints: bool = True,
                return_attention: bool = False) -> Dict[str, torch.Tensor]:
        if self.training and torch.is_grad_enabled():
            raise RuntimeError("the HIP forward implements eval-mode arithmetic; call .eval() or torch.no_grad()")
        if mel_features.dim() != 3 or emotion_features.dim() != 3 or mel_features.shape[:2] != emotion_features.shape[:2]:
            raise ValueError(f"expected (B, T, mel_dim) and (B, T, emotion_dim), got {tuple(mel_features.shape)} and {tuple(emotion_features.shape)}")
        lib, h, dev = self._handle()
        mel = mel_features.float().contiguous()
        emo = emotion_features.float().contiguous()
        B, T, _ = mel.shape
        if B > self._reserved[0] or T > self._reserved[1]:
            with torch.cuda.device(dev):
                torch.cuda.synchronize(dev)
                check(lib.km_koemorph_reserve(h, max(B, self._reserved[0]), max(T, self._reserved[1])))
            self._reserved = (max(B, self._reserved[0]), max(T, self._reserved[1]))
        nb = self.num_blendshapes
        prev = None if prev_blendshapes is None else prev_blendshapes.float().contiguous()
        valid = None
        if audio_mask is not None:                        # (B, T) bool, True = valid frame
            if tuple(audio_mask.shape) != (B, T):
                raise ValueError(f"audio_mask must be (B, T) = {(B, T)}, got {tuple(audio_mask.shape)}")
            valid = audio_mask.to(device=dev, dtype=torch.uint8).contiguous()
        smooth = apply_smoothing and self.use_temporal_smoothing
        if smooth and self.smoothing_method != "exponential" and (self._smoother_state is None or self._smoother_state.shape[0] != B):
            # gaussian / median: every batch element's history ring (window, 52) + its slot pointer (include/koemorph.h).  A
            # new batch size continues from batch element 0's history, as decoder.py:333-337 expands history[:, :1, :]
            W = self.temporal_smoother.window_size
            first = torch.zeros(1, W * nb + 1, device=dev) if self._smoother_state is None else self._smoother_state[:1]
            self._smoother_state = first.expand(B, -1).contiguous()
        if smooth and self.smoothing_method == "exponential" and (self._smoother_state is None or self._smoother_state.shape[0] != B):
            # decoder.py:282-283: the (1, 52) state is expanded to the batch (zeros after a reset); a state that already
            # holds another batch size cannot be expanded -- torch raises there, and so does this mirror
            if self._smoother_state is not None and self._smoother_state.shape[0] != 1:
                raise RuntimeError(f"temporal smoother holds state for batch {self._smoother_state.shape[0]}, got batch {B}: "
                                   "call reset_temporal_state() between sequences (the reference's expand() fails the same way)")
            first = torch.zeros(1, nb, device=dev) if self._smoother_state is None else self._smoother_state
            self._smoother_state = first.expand(B, -1).contiguous()
        out = torch.empty(B, nb, device=dev)
        raw = torch.empty(B, nb, device=dev)
        L = len(self.cross_attention_layers)
        attn = torch.empty(L, B, self.num_heads, nb, T, device=dev) if return_attention and L else None
        with torch.cuda.device(dev):
            check(lib.km_koemorph_forward(h, _ptr(mel), _ptr(emo), B, T, _ptr(valid) if valid is not None else None,
                                          _ptr(prev) if prev is not None else None,
                                          _ptr(self._smoother_state) if smooth else None, 1 if apply_constraints else 0,
                                          _ptr(out), _ptr(raw), _ptr(attn) if attn is not None else None, _stream_ptr(dev)))
        output = {"blendshapes": out, "raw_blendshapes": raw}
        if attn is not None:
            output["attention_weights"] = [attn[i] for i in range(L)]
        return output

    def reset_temporal_state(self):
        """New sequence: the smoother starts from zeros again (gaussian_face.py:270-276)."""
        self._smoother_state = None

    def inference_step(self, mel_features, emotion_features, prev_blendshapes=None) -> torch.Tensor:
        with torch.no_grad():
            return self.forward(mel_features, emotion_features, prev_blendshapes=prev_blendshapes, apply_smoothing=True,
                                apply_constraints=True, return_attention=False)["blendshapes"]

    def get_num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def get_model_info(self) -> Dict[str, object]:
        return {"total_parameters": self.get_num_parameters(), "mel_dim": self.mel_dim, "emotion_dim": self.emotion_dim,
                "d_model": self.d_model, "num_blendshapes": self.num_blendshapes,
                "num_attention_layers": len(self.cross_attention_layers),
                "use_temporal_smoothing": self.use_temporal_smoothing, "use_constraints": self.use_constraints}


_CONFIG_DEFAULTS = dict(mel_dim=80, emotion_dim=256, d_model=256, d_query=128, d_key=256, d_value=256, num_heads=8,
                        num_encoder_layers=2, num_attention_layers=4, attention_dropout=0.1, decoder_hidden_dim=128,
                        decoder_layers=2, decoder_activation="gelu", output_activation="sigmoid", use_temporal_smoothing=True,
                        smoothing_method="exponential", smoothing_alpha=0.8, use_constraints=True, causal=True, window_size=30,
                        num_blendshapes=52, dropout=0.1)


def create_koemorph_model(config: dict) -> KoeMorphModel:
    """Model from a configuration mapping, with the reference's defaults for missing keys (gaussian_face.py:325-368)."""
    return KoeMorphModel(**{k: config.get(k, v) for k, v in _CONFIG_DEFAULTS.items()})
