"""Oracle: the legacy SimplifiedKoeMorphModel.forward restated with explicit torch-CPU math.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference/src/model/simplified_model.py:
mel front end :79-112 (same librosa call as the dual-stream model -> oracle.mel.mel_batch_window),
audio_encoder :44-51, nn.MultiheadAttention :54-59/:136-141, decoder :63-72, mean over the query rows :147.
The module cannot be imported here (it imports librosa at module scope), so the restatement is pinned in
tests/test_oracle_legacy.py against the SAME torch containers (nn.Sequential / nn.MultiheadAttention)
the reference instantiates, driven with the oracle's mel; the mel itself is parity-unpinned (oracle/mel.py).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import mel
from .core import mha_forward


def legacy_forward_mel(params: Dict[str, np.ndarray], mel_features, num_heads: int = 8, dtype=torch.float32):
    P = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in params.items()}
    x = torch.as_tensor(mel_features).to(dtype)
    B = x.shape[0]
    e = torch.relu(F.linear(x, P["audio_encoder.0.weight"], P["audio_encoder.0.bias"]))     # Dropout = identity
    e = torch.relu(F.linear(e, P["audio_encoder.3.weight"], P["audio_encoder.3.bias"]))
    q = P["blendshape_queries"].unsqueeze(0).repeat(B, 1, 1)
    a, _ = mha_forward(q, e, P["attention.in_proj_weight"], P["attention.in_proj_bias"],
                       P["attention.out_proj.weight"], P["attention.out_proj.bias"], num_heads, False)
    h = torch.relu(F.linear(a, P["decoder.0.weight"], P["decoder.0.bias"]))
    h = torch.relu(F.linear(h, P["decoder.3.weight"], P["decoder.3.bias"]))
    y = torch.sigmoid(F.linear(h, P["decoder.6.weight"], P["decoder.6.bias"]))               # (B, 52, 52)
    return y.mean(dim=1).numpy()


def legacy_forward(params, audio: np.ndarray, num_heads: int = 8, sample_rate: int = 16000, target_fps: int = 30):
    hop = int(sample_rate // target_fps)
    long, _ = mel.mel_batch(audio, sample_rate=sample_rate, n_fft=1024, hop=hop)
    return legacy_forward_mel(params, long, num_heads)


def make_legacy_params(seed: int, d_model: int = 256, hidden: int = 128, nb: int = 52, scale: float = 1.0):
    from koemorph_amd import synth
    shapes = [("audio_encoder.0.weight", (d_model, 80)), ("audio_encoder.0.bias", (d_model,)),
              ("audio_encoder.3.weight", (d_model, d_model)), ("audio_encoder.3.bias", (d_model,)),
              ("attention.in_proj_weight", (3 * d_model, d_model)), ("attention.in_proj_bias", (3 * d_model,)),
              ("attention.out_proj.weight", (d_model, d_model)), ("attention.out_proj.bias", (d_model,)),
              ("decoder.0.weight", (hidden, d_model)), ("decoder.0.bias", (hidden,)),
              ("decoder.3.weight", (hidden, hidden)), ("decoder.3.bias", (hidden,)),
              ("decoder.6.weight", (nb, hidden)), ("decoder.6.bias", (nb,)),
              ("blendshape_queries", (nb, d_model))]
    out = {}
    for i, (k, shp) in enumerate(shapes):
        if k == "blendshape_queries":
            out[k] = synth.normal(seed * 100 + i, shp, std=0.5 * scale)
        elif k.endswith("weight"):
            b = scale * 2.0 / float(np.sqrt(shp[-1]))
            out[k] = synth.uniform(seed * 100 + i, shp, -b, b)
        else:
            out[k] = synth.normal(seed * 100 + i, shp, std=0.1 * scale)
    return out
