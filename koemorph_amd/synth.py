"""Deterministic synthetic inputs and parameters for tests, fixtures and bench.

Everything here is generated from a counter-based integer hash (splitmix64), so the
same (seed, shape) gives bit-identical float32 arrays on any numpy version and on any
machine.  That lets the golden fixtures under ``tests/golden/`` store only the seeds
and the expected outputs instead of megabytes of weights.

Shapes and state-dict keys follow the reference's ``DualStreamCrossAttention``
(/root/reference/src/model/dual_stream_attention.py:57-159); the speech-like audio
generator follows the reference's own synthetic stream
(/root/reference/test_realtime_dual_stream.py:29-57: F0 120 +/- 30 Hz, formants
850/1300 Hz, 2.5 Hz envelope, gaussian noise, peak 0.7).
"""

from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n float64 values in (0, 1), a pure function of (seed, stream, index)."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = _splitmix64(np.asarray([seed], dtype=np.uint64) * np.uint64(0x100000001B3)
                          + np.uint64(stream) * np.uint64(0xD6E8FEB86659FD93))
        bits = _splitmix64(idx ^ key)
    # 53 random mantissa bits, never exactly 0
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def uniform(seed: int, shape, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    n = int(np.prod(shape))
    return (lo + (hi - lo) * uniform01(seed, n)).astype(np.float32).reshape(shape)


def normal(seed: int, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """Box-Muller on two independent uniform streams."""
    n = int(np.prod(shape))
    u1 = uniform01(seed, n, stream=1)
    u2 = uniform01(seed, n, stream=2)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * z).astype(np.float32).reshape(shape)


# ---------------------------------------------------------------------------
# ARKit grouping (values verified against the reference's computed lists,
# dual_stream_attention.py:14-45)
# ---------------------------------------------------------------------------
MOUTH_INDICES = list(range(14, 41)) + [51]
EXPRESSION_INDICES = list(range(0, 14)) + list(range(41, 51))


def core_param_shapes(d_model: int = 256, mel_sequence_length: int = 256,
                      mel_temporal_frames: int = 3, emotion_dim: int = 256,
                      num_blendshapes: int = 52) -> "OrderedDict[str, Tuple[int, ...]]":
    """State-dict keys and shapes of DualStreamCrossAttention, in its own order
    (dual_stream_attention.py:102-159)."""
    d, kt = d_model, mel_sequence_length + mel_temporal_frames
    return OrderedDict([
        ("mouth_queries", (len(MOUTH_INDICES), d)),
        ("expression_queries", (len(EXPRESSION_INDICES), d)),
        ("mel_weights", (num_blendshapes,)),
        ("emotion_weights", (num_blendshapes,)),
        ("mel_channel_encoder.weight", (d, kt)),
        ("mel_channel_encoder.bias", (d,)),
        ("mel_attention.in_proj_weight", (3 * d, d)),
        ("mel_attention.in_proj_bias", (3 * d,)),
        ("mel_attention.out_proj.weight", (d, d)),
        ("mel_attention.out_proj.bias", (d,)),
        ("emotion_encoder.weight", (d, emotion_dim)),
        ("emotion_encoder.bias", (d,)),
        ("emotion_attention.in_proj_weight", (3 * d, d)),
        ("emotion_attention.in_proj_bias", (3 * d,)),
        ("emotion_attention.out_proj.weight", (d, d)),
        ("emotion_attention.out_proj.bias", (d,)),
        ("mel_output_proj.weight", (d, d)),
        ("mel_output_proj.bias", (d,)),
        ("emotion_output_proj.weight", (d, d)),
        ("emotion_output_proj.bias", (d,)),
        ("blendshape_decoder.0.weight", (d // 2, d)),
        ("blendshape_decoder.0.bias", (d // 2,)),
        ("blendshape_decoder.3.weight", (1, d // 2)),
        ("blendshape_decoder.3.bias", (1,)),
        ("mel_norm.weight", (d,)),
        ("mel_norm.bias", (d,)),
        ("emotion_norm.weight", (d,)),
        ("emotion_norm.bias", (d,)),
    ])


def make_core_params(seed: int, d_model: int = 256, mel_sequence_length: int = 256,
                     emotion_dim: int = 256, style: str = "init",
                     num_blendshapes: int = 52) -> "OrderedDict[str, np.ndarray]":
    """Deterministic float32 state dict.

    style="init"    mirrors the scale of a freshly constructed module (uniform
                    +/-1/sqrt(fan_in) linears, xavier in_proj, 0.02-std queries, unit
                    LayerNorm, 2.0/0.5 stream weights).
    style="trained" exercises the ranges a trained checkpoint reaches: larger queries
                    (peaky softmax), non-trivial LayerNorm affine, random stream
                    weights, non-zero attention biases, larger decoder gain.
    """
    shapes = core_param_shapes(d_model, mel_sequence_length, 3, emotion_dim, num_blendshapes)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    trained = style == "trained"
    for i, (key, shp) in enumerate(shapes.items()):
        s = seed * 1000 + i
        if key in ("mouth_queries", "expression_queries"):
            v = normal(s, shp, std=0.6 if trained else 0.02)
        elif key == "mel_weights":
            v = np.ones(shp, np.float32)
            v[MOUTH_INDICES] = 2.0
            v[EXPRESSION_INDICES] = 0.5
            if trained:
                v = v + normal(s, shp, std=1.5)
        elif key == "emotion_weights":
            v = np.ones(shp, np.float32)
            v[MOUTH_INDICES] = 0.5
            v[EXPRESSION_INDICES] = 2.0
            if trained:
                v = v + normal(s, shp, std=1.5)
        elif key.endswith("_norm.weight"):
            v = np.ones(shp, np.float32)
            if trained:
                v = v + normal(s, shp, std=0.3)
        elif key.endswith("_norm.bias"):
            v = normal(s, shp, std=0.2) if trained else np.zeros(shp, np.float32)
        elif key.endswith("in_proj_weight"):
            bound = float(np.sqrt(6.0 / (shp[0] + shp[1])))  # xavier_uniform
            v = uniform(s, shp, -bound, bound) * (2.0 if trained else 1.0)
        elif key.endswith("in_proj_bias") or key.endswith("out_proj.bias"):
            v = normal(s, shp, std=0.1) if trained else np.zeros(shp, np.float32)
        elif key.endswith(".weight"):
            bound = 1.0 / float(np.sqrt(shp[-1]))
            v = uniform(s, shp, -bound, bound) * (2.5 if trained else 1.0)
        elif key.endswith(".bias"):
            fan_in = shapes[key[:-5] + ".weight"][-1]
            bound = 1.0 / float(np.sqrt(fan_in))
            v = uniform(s, shp, -bound, bound)
        else:  # pragma: no cover
            raise KeyError(key)
        out[key] = np.ascontiguousarray(v.astype(np.float32))
    return out


def make_core_inputs(seed: int, batch: int, t_in: int = 257, n_mels: int = 80,
                     emotion_dim: int = 256, style: str = "mel01"):
    """(mel (B,t_in,80), mel_short (B,3,80), emotion (B,emotion_dim)) float32.

    style="mel01"  values in [0,1] like the (dB+80)/80 normalised mel of
                   simplified_dual_stream_model.py:199-200;
    style="randn"  unit gaussians.
    """
    if style == "mel01":
        mel = uniform(seed * 7 + 1, (batch, t_in, n_mels), 0.0, 1.0)
        short = uniform(seed * 7 + 2, (batch, 3, n_mels), 0.0, 1.0)
    else:
        mel = normal(seed * 7 + 1, (batch, t_in, n_mels))
        short = normal(seed * 7 + 2, (batch, 3, n_mels))
    emo = normal(seed * 7 + 3, (batch, emotion_dim))
    return mel, short, emo


def make_audio(seed: int, batch: int, length: int, style: str = "speech",
               sample_rate: int = 16000) -> np.ndarray:
    """(B, L) float32 mono audio in [-1, 1].

    style="speech": voiced-speech-like signal after test_realtime_dual_stream.py:29-57
                    (per-window random phase / F0 so the windows differ);
    style="uniform": white uniform noise * 0.5.
    """
    if style == "uniform":
        return uniform(seed * 13 + 5, (batch, length), -0.5, 0.5)
    t = np.arange(length, dtype=np.float64) / sample_rate
    out = np.empty((batch, length), np.float32)
    par = uniform01(seed * 13 + 7, batch * 4).reshape(batch, 4)
    noise = normal(seed * 13 + 11, (batch, length), std=0.03).astype(np.float64)
    for b in range(batch):
        f0 = 120.0 + 30.0 * np.sin(2 * np.pi * 0.5 * t + 2 * np.pi * par[b, 0]) + 20.0 * (par[b, 1] - 0.5)
        phase = 2 * np.pi * np.cumsum(f0) / sample_rate
        voiced = np.sin(phase) + 0.5 * np.sin(2 * phase) + 0.25 * np.sin(3 * phase)
        form = 0.3 * np.sin(2 * np.pi * 850.0 * t + par[b, 2]) + 0.2 * np.sin(2 * np.pi * 1300.0 * t + par[b, 3])
        env = 0.5 * (1.0 + np.sin(2 * np.pi * 2.5 * t + 2 * np.pi * par[b, 1]))
        sig = (voiced + form) * env + noise[b]
        sig = sig / np.max(np.abs(sig)) * 0.7
        out[b] = sig.astype(np.float32)
    return out


def params_checksum(params: Dict[str, np.ndarray]) -> float:
    """Order-independent checksum used by the fixtures to detect generator drift."""
    acc = 0.0
    for k in sorted(params):
        a = params[k].astype(np.float64).ravel()
        acc += float(np.sum(a * (1.0 + (np.arange(a.size) % 7)))) + float(np.sum(np.abs(a)))
    return acc
