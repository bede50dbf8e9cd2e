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


# ---- legacy multi-layer KoeMorphModel (reference src/model/gaussian_face.py:29-173): configuration, state-dict layout
# and seeded parameters for tests, fixtures and tools/bench_koemorph.py -------------------------------------------------
from dataclasses import asdict, dataclass   # noqa: E402
from typing import Optional                 # noqa: E402


@dataclass
class KoeMorphConfig:
    mel_dim: int = 80
    emotion_dim: int = 256
    d_model: int = 256
    num_heads: int = 8
    num_encoder_layers: int = 2
    num_attention_layers: int = 4
    decoder_hidden_dim: int = 128
    decoder_layers: int = 2
    decoder_activation: str = "gelu"
    causal: bool = True
    window_size: Optional[int] = 30
    use_temporal_smoothing: bool = True
    use_constraints: bool = True
    num_blendshapes: int = 52
    output_activation: str = "sigmoid"        # sigmoid | tanh | none           (decoder.py:162-167)
    smoothing_method: str = "exponential"     # exponential | gaussian | median (decoder.py:260-331)
    smoothing_window: int = 5                 # TemporalSmoother's default; KoeMorphModel never passes another

    def to_dict(self):
        return asdict(self)




def koemorph_param_shapes(c: KoeMorphConfig):
    """State-dict keys and shapes of the learnable tensors (buffers of the smoother / constraints excluded)."""
    d, dq, hid, nb = c.d_model, c.d_model, c.decoder_hidden_dim, c.num_blendshapes
    s = []
    for stream, dim in (("mel", c.mel_dim), ("emotion", c.emotion_dim)):
        p = f"audio_encoder.{stream}_encoder."
        s += [(p + "0.weight", (d, dim)), (p + "0.bias", (d,)), (p + "3.weight", (d,)), (p + "3.bias", (d,))]
    for stream in ("mel", "emotion"):
        for i in range(c.num_encoder_layers):
            p = f"audio_encoder.{stream}_transformer.layers.{i}."
            s += [(p + "self_attn.in_proj_weight", (3 * d, d)), (p + "self_attn.in_proj_bias", (3 * d,)),
                  (p + "self_attn.out_proj.weight", (d, d)), (p + "self_attn.out_proj.bias", (d,)),
                  (p + "linear1.weight", (4 * d, d)), (p + "linear1.bias", (4 * d,)),
                  (p + "linear2.weight", (d, 4 * d)), (p + "linear2.bias", (d,)),
                  (p + "norm1.weight", (d,)), (p + "norm1.bias", (d,)), (p + "norm2.weight", (d,)), (p + "norm2.bias", (d,))]
    s += [("query_embeddings.query_embeddings", (nb, dq)),
          ("query_embeddings.conditioning_net.0.weight", (dq // 2, nb)), ("query_embeddings.conditioning_net.0.bias", (dq // 2,)),
          ("query_embeddings.conditioning_net.3.weight", (dq, dq // 2)), ("query_embeddings.conditioning_net.3.bias", (dq,))]
    for i in range(c.num_attention_layers):
        p = f"cross_attention_layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s += [(p + n + ".weight", (d, d)), (p + n + ".bias", (d,))]
    for i in range(c.num_attention_layers):
        s += [(f"attention_layer_norms.{i}.weight", (d,)), (f"attention_layer_norms.{i}.bias", (d,))]
    s += [("decoder.input_proj.weight", (hid, d)), ("decoder.input_proj.bias", (hid,))]
    for i in range(c.decoder_layers):
        s += [(f"decoder.hidden_layers.{i}.weight", (hid, hid)), (f"decoder.hidden_layers.{i}.bias", (hid,))]
    for i in range(c.decoder_layers):
        s += [(f"decoder.layer_norms.{i}.weight", (hid,)), (f"decoder.layer_norms.{i}.bias", (hid,))]
    s += [("decoder.output_proj.weight", (nb, hid)), ("decoder.output_proj.bias", (nb,))]
    if c.use_temporal_smoothing and c.smoothing_method == "exponential":
        s += [("temporal_smoother.alpha", ())]
    if c.use_temporal_smoothing and c.smoothing_method == "gaussian":
        s += [("temporal_smoother.gaussian_weights", (c.smoothing_window,))]
    return s


def make_koemorph_params(seed: int, c: KoeMorphConfig, scale: float = 1.0) -> Dict[str, np.ndarray]:
    """Seeded 'trained-like' parameters: uniform weights of +-1.5/sqrt(fan_in), LayerNorm gains around 1, small biases."""
    out = {}
    for i, (k, shp) in enumerate(koemorph_param_shapes(c)):
        sd = seed * 1000 + i
        if k == "temporal_smoother.alpha":
            out[k] = np.float32(0.8 + 0.1 * float(normal(sd, (1,))[0])).reshape(())
        elif k == "temporal_smoother.gaussian_weights":
            out[k] = normal(sd, shp, std=0.7)                  # softmaxed by the smoother: clearly non-uniform slot weights
        elif k == "query_embeddings.query_embeddings":
            out[k] = normal(sd, shp, std=0.5 * scale)
        elif (".norm" in k or "layer_norms" in k or "_encoder.3." in k) and k.endswith("weight"):
            out[k] = (1.0 + 0.1 * normal(sd, shp)).astype(np.float32)
        elif k.endswith("weight"):
            b = scale * 1.5 / float(np.sqrt(shp[-1]))
            out[k] = uniform(sd, shp, -b, b)
        else:
            out[k] = normal(sd, shp, std=0.05 * scale)
    return out




def make_av_features(seed: int, B: int, T: int = 40, D: int = 80) -> np.ndarray:
    """audio_features (B, T, D) for the audio-visual term of the reference's PerceptualBlendshapeLoss
    (src/model/losses.py:340-378): mel-like rows whose overall level differs from window to window."""
    return (uniform(seed * 3 + 7, (B, T, D), 0.0, 1.0) * uniform(seed * 3 + 8, (B, 1, 1), 0.2, 2.0)).astype(np.float32)


def make_vowel(seed: int, f0: float, seconds: float, formants=((700.0, 90.0), (1200.0, 110.0), (2600.0, 160.0)),
               jitter: float = 0.0, noise: float = 0.002, sample_rate: int = 16000, vibrato: float = 0.0) -> np.ndarray:
    """A synthetic vowel with KNOWN parameters for the eGeMAPS tests: an impulse train of fundamental f0 (period jittered by
    `jitter` relative standard deviation, optional slow vibrato of relative depth `vibrato`) through two-pole resonators at the
    given (frequency, bandwidth) pairs, plus white noise; peak normalised.  Pure numpy, reproducible from the seed."""
    n = int(seconds * sample_rate)
    x = np.zeros(n, np.float64)
    g = normal(seed, (4 * int(seconds * f0) + 16,)).astype(np.float64)
    t, i = 0.0, 0
    while t < n - 1:
        x[int(t)] = 1.0
        f = f0 * (1.0 + vibrato * np.sin(2.0 * np.pi * 5.0 * t / sample_rate))
        t += sample_rate / f * (1.0 + jitter * g[i % len(g)])
        i += 1
    for fc, bw in formants:
        r, th = np.exp(-np.pi * bw / sample_rate), 2.0 * np.pi * fc / sample_rate
        a1, a2 = -2.0 * r * np.cos(th), r * r
        y = np.zeros(n)
        y1 = y2 = 0.0
        for k in range(n):
            v = x[k] - a1 * y1 - a2 * y2
            y[k] = v
            y2, y1 = y1, v
        x = y
    x = x + noise * np.abs(x).max() * normal(seed + 1, (n,)).astype(np.float64)
    return (x / np.abs(x).max()).astype(np.float32)
