"""Oracle: the legacy KoeMorphModel.forward (eval mode) restated with explicit torch-CPU math.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, in /root/reference/src/model:
  gaussian_face.py:175-268      KoeMorphModel.forward (encode both streams, average, 4 x cross-attention +
                                residual LayerNorm, decoder, smoother, constraints)
  dual_stream_attention.py:296-390  DualStreamEncoder (Linear/ReLU/LayerNorm front + nn.TransformerEncoder:
                                post-norm layers, nhead 8, dim_feedforward 4 d, exact-erf GELU)
  attention.py:101-246          MultiHeadCrossAttention (q/k/v/out projections, scale (hd * temperature)^-0.5,
                                causal = triu(diagonal=1) over (52, T), window mask around int(i * T / 52))
  attention.py:481-514          BlendshapeQueryEmbedding (learned rows + conditioning MLP of the previous frame)
  decoder.py:108-177            BlendshapeDecoder (input_proj, residual hidden layers with LayerNorm, the diagonal
                                of output_proj, sigmoid, 0.9 / 0.1 mix with the previous frame)
  decoder.py:278-340            TemporalSmoother, learnable=True as the model builds it: exponential (alpha = sigmoid(param),
                                y = alpha * prev + (1 - alpha) * x, state starts at zero), gaussian (softmax of the learnable
                                weights over the history ring's slots) and median (torch.median over the slots)
  decoder.py:434-466            BlendshapeConstraints: clamp to [0, 1], pairs (25, 26) and (20, 21) divided by their
                                sum + 1e-8
The reference module is importable in the build container: oracle/gen_golden.py runs it on seeded parameters
(make_koemorph_params below) and tests/test_oracle_koemorph.py pins this restatement to those outputs.
A query row whose keys are all masked is NaN in the reference (softmax over -inf only); it is NaN here too.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F


# configuration + seeded parameters live on the product side (data generators, shared with tools/bench_koemorph.py)
from koemorph_amd.synth import KoeMorphConfig, koemorph_param_shapes as param_shapes, make_koemorph_params  # noqa: E402,F401

ENCODER_HEADS = 8          # nn.TransformerEncoderLayer(nhead=8) is hard-wired (dual_stream_attention.py:338)
EXCLUSION_PAIRS = ((25, 26), (20, 21))     # decoder.py:384-387


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _act(x, name):
    """BlendshapeDecoder's activation (decoder.py:68-75): nn.ReLU / nn.GELU / nn.SiLU / nn.LeakyReLU(0.1)."""
    if name == "gelu":
        return _gelu(x)
    if name == "swish":
        return x * torch.sigmoid(x)
    if name == "leaky_relu":
        return torch.where(x >= 0, x, 0.1 * x)
    return torch.relu(x)


def smoother_state_shape(c: KoeMorphConfig, B: int):
    """The smoother's state as km_koemorph_forward keeps it: (B, 52) for the exponential method; for gaussian / median every
    batch element's history ring (window, 52) followed by its slot pointer: (B, window * 52 + 1).  Zeros = reset."""
    return (B, c.num_blendshapes) if c.smoothing_method == "exponential" else (B, c.smoothing_window * c.num_blendshapes + 1)


def _ln(x, P, prefix, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * P[prefix + "weight"] + P[prefix + "bias"]


def _heads(x, H):
    B, N, d = x.shape
    return x.view(B, N, H, d // H).transpose(1, 2)          # (B, H, N, hd)


def _encoder_layer(x, P, p, H, key_valid=None):
    """nn.TransformerEncoderLayer, norm_first=False, activation gelu, eval mode; key_valid (B, T) True = attend
    (src_key_padding_mask = ~key_valid, dual_stream_attention.py:371-374)."""
    d = x.shape[-1]
    qkv = F.linear(x, P[p + "self_attn.in_proj_weight"], P[p + "self_attn.in_proj_bias"])
    q, k, v = (_heads(t, H) for t in qkv.split(d, dim=-1))
    s = q @ k.transpose(-2, -1) / math.sqrt(d // H)
    if key_valid is not None:
        s = s.masked_fill(~key_valid[:, None, None, :], float("-inf"))
    a = torch.softmax(s, dim=-1) @ v
    a = a.transpose(1, 2).reshape(x.shape)
    x = _ln(x + F.linear(a, P[p + "self_attn.out_proj.weight"], P[p + "self_attn.out_proj.bias"]), P, p + "norm1.")
    ff = F.linear(_gelu(F.linear(x, P[p + "linear1.weight"], P[p + "linear1.bias"])), P[p + "linear2.weight"], P[p + "linear2.bias"])
    return _ln(x + ff, P, p + "norm2.")


def encode_stream(x, P, stream, num_layers, key_valid=None):
    p = f"audio_encoder.{stream}_encoder."
    x = _ln(torch.relu(F.linear(x, P[p + "0.weight"], P[p + "0.bias"])), P, p + "3.")
    for i in range(num_layers):
        x = _encoder_layer(x, P, f"audio_encoder.{stream}_transformer.layers.{i}.", ENCODER_HEADS, key_valid)
    return x


def attention_mask(nq: int, T: int, causal: bool, window_size: Optional[int]) -> np.ndarray:
    """True = masked (attention.py:208-246)."""
    m = np.zeros((nq, T), dtype=bool)
    j = np.arange(T)
    for i in range(nq):
        if causal:
            m[i] |= j > i
        if window_size is not None:
            kp = int(i * T / nq)
            lo, hi = max(0, kp - window_size // 2), min(T, kp + window_size // 2 + 1)
            m[i] |= ~((j >= lo) & (j < hi))
    return m


def koemorph_forward(params: Dict[str, np.ndarray], c: KoeMorphConfig, mel, emotion, prev_blendshapes=None,
                     smoother_state=None, apply_smoothing=True, apply_constraints=True, dtype=torch.float64, audio_mask=None):
    """Returns dict(blendshapes, raw_blendshapes, attention_weights [L x (B, H, 52, T)], smoother_state).
    audio_mask (B, T) bool, True = valid frame (gaussian_face.py:180,204,224): padded frames are masked as keys in the
    encoder's self-attention and in every cross-attention layer.  What the encoder leaves AT padded positions never
    reaches the output (torch's eval fast path zeroes them, the plain path does not), so it is not pinned."""
    P = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in params.items()}
    mel = torch.as_tensor(np.asarray(mel)).to(dtype)
    emo = torch.as_tensor(np.asarray(emotion)).to(dtype)
    B, T, _ = mel.shape
    d, H, nb = c.d_model, c.num_heads, c.num_blendshapes
    valid = None if audio_mask is None else torch.as_tensor(np.asarray(audio_mask)).bool()
    enc = (encode_stream(mel, P, "mel", c.num_encoder_layers, valid) + encode_stream(emo, P, "emotion", c.num_encoder_layers, valid)) / 2
    x = P["query_embeddings.query_embeddings"].unsqueeze(0).repeat(B, 1, 1)
    prev = None if prev_blendshapes is None else torch.as_tensor(np.asarray(prev_blendshapes)).to(dtype)
    if prev is not None:
        h = torch.relu(F.linear(prev, P["query_embeddings.conditioning_net.0.weight"], P["query_embeddings.conditioning_net.0.bias"]))
        x = x + F.linear(h, P["query_embeddings.conditioning_net.3.weight"], P["query_embeddings.conditioning_net.3.bias"]).unsqueeze(1)
    mask = torch.from_numpy(attention_mask(nb, T, c.causal, c.window_size))
    scale = float(d // H) ** -0.5
    attn = []
    for i in range(c.num_attention_layers):
        p = f"cross_attention_layers.{i}."
        q = _heads(F.linear(x, P[p + "q_proj.weight"], P[p + "q_proj.bias"]), H)
        k = _heads(F.linear(enc, P[p + "k_proj.weight"], P[p + "k_proj.bias"]), H)
        v = _heads(F.linear(enc, P[p + "v_proj.weight"], P[p + "v_proj.bias"]), H)
        s = (q @ k.transpose(-2, -1)) * scale
        s = s.masked_fill(mask, float("-inf"))
        if valid is not None:
            s = s.masked_fill(~valid[:, None, None, :], float("-inf"))
        w = torch.softmax(s, dim=-1)
        attn.append(w)
        o = (w @ v).transpose(1, 2).reshape(B, nb, d)
        o = F.linear(o, P[p + "out_proj.weight"], P[p + "out_proj.bias"])
        x = _ln(o + x, P, f"attention_layer_norms.{i}.")
    h = _act(F.linear(x, P["decoder.input_proj.weight"], P["decoder.input_proj.bias"]), c.decoder_activation)
    for i in range(c.decoder_layers):
        r = h
        h = F.linear(h, P[f"decoder.hidden_layers.{i}.weight"], P[f"decoder.hidden_layers.{i}.bias"])
        h = _act(_ln(h, P, f"decoder.layer_norms.{i}."), c.decoder_activation) + r
    z = (h * P["decoder.output_proj.weight"].unsqueeze(0)).sum(-1) + P["decoder.output_proj.bias"]     # diagonal of (B, 52, 52)
    raw = torch.sigmoid(z) if c.output_activation == "sigmoid" else (torch.tanh(z) if c.output_activation == "tanh" else z)   # decoder.py:162-167
    if prev is not None:
        raw = 0.9 * raw + 0.1 * prev
    y = raw
    state = None
    if apply_smoothing and c.use_temporal_smoothing:
        st = torch.zeros(smoother_state_shape(c, B), dtype=dtype) if smoother_state is None else torch.as_tensor(np.asarray(smoother_state)).to(dtype)
        if c.smoothing_method == "exponential":                       # decoder.py:278-292
            alpha = torch.sigmoid(P["temporal_smoother.alpha"])
            y = alpha * st + (1 - alpha) * y
            state = y.clone()
        else:                                                         # decoder.py:294-340: ring of `window` slots, one written per call
            W = c.smoothing_window
            state = st.clone()
            ring = state[:, :W * nb].view(B, W, nb)
            ptr = int(round(float(state[0, W * nb])))
            ring[:, ptr] = y
            state[:, W * nb] = float((ptr + 1) % W)
            if c.smoothing_method == "gaussian":                      # the (learnable) weights go with the SLOT, not the age (:307-317)
                w = torch.softmax(P["temporal_smoother.gaussian_weights"], dim=0)
                y = (w.view(1, W, 1) * ring).sum(dim=1)
            else:                                                     # torch.median over the slots (:319-331)
                y = torch.median(ring, dim=1)[0]
    if apply_constraints and c.use_constraints:
        y = y.clamp(0.0, 1.0).clone()
        for a, b in EXCLUSION_PAIRS:
            comb = y[:, a] + y[:, b]
            ya, yb = y[:, a] / (comb + 1e-8), y[:, b] / (comb + 1e-8)
            y[:, a], y[:, b] = ya, yb
    f32 = lambda t: t.to(torch.float32).numpy()
    return {"blendshapes": f32(y), "raw_blendshapes": f32(raw), "attention_weights": [f32(w) for w in attn],
            "smoother_state": None if state is None else f32(state)}
