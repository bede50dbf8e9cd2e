#!/usr/bin/env python3
"""Error study for DESIGN section 7 (1b): what the 52 outputs lose if every weight product of the mel stream is assembled
from bf16 pieces (fp32 accumulation) instead of IEEE fp32 multiplies.  CPU only, oracle arithmetic; not collected by pytest.
  terms 1: plain bf16 operands          terms 3: a = hi + lo (16 mantissa bits), hi*hi + hi*lo + lo*hi
  terms 6: a = hi + mid + lo (24 bits), the six products of weight >= 2^-16
Reference = the float64 oracle; the float32 oracle is printed as the noise floor of the present path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from koemorph_amd import synth
from oracle import core


def pieces(x, n):
    out, r = [], x
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        r = r - p
    return out


def split_linear(terms):
    def lin(x, w, b=None):
        if x.dtype != torch.float32 or w.shape[-1] < 64:        # small contractions stay fp32 (they would on the GPU too)
            return F_linear(x, w, b)
        n = {1: 1, 3: 2, 6: 3}[terms]
        xs, ws = pieces(x, n), pieces(w, n)
        pairs = {1: [(0, 0)], 3: [(0, 0), (0, 1), (1, 0)], 6: [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]}[terms]
        y = sum(F_linear(xs[i], ws[j]) for i, j in pairs)
        return y if b is None else y + b
    return lin


F_linear = F.linear
cases = []
for style, seed in (("init", 11), ("trained", 12)):
    p = synth.make_core_params(seed, 256, 256, 256, style)
    cases.append((style, p))
    q = {k: v.copy() for k, v in p.items()}
    q["mel_weights"] = q["mel_weights"].copy(); q["mel_weights"][20] += 25.0        # stream-weight softmax concentrated on one coefficient
    q["emotion_weights"] = q["emotion_weights"].copy(); q["emotion_weights"][20] += 25.0
    cases.append((style + ", concentrated stream weights", q))
mel, short, emo = synth.make_core_inputs(7, 4, 257, style="mel01")
for name, p in cases:
    ref = core.core_forward(p, mel, short, emo, dtype=torch.float64)["blendshapes"].numpy()
    row = {"fp32": float(np.abs(core.core_forward(p, mel, short, emo)["blendshapes"].numpy() - ref).max())}
    for terms in (1, 3, 6):
        F.linear = split_linear(terms)
        try:
            got = core.core_forward(p, mel, short, emo)["blendshapes"].numpy()
        finally:
            F.linear = F_linear
        row[f"bf16 x{terms}"] = float(np.abs(got - ref).max())
    print(f"{name:45s} max|out| {np.abs(ref).max():.3f}  " + "  ".join(f"{k} {v:.2e}" for k, v in row.items()))
