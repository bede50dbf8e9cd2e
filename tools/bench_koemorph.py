#!/usr/bin/env python3
"""Legacy KoeMorphModel (km_koemorph_forward) on ONE GPU at the reference's default depth (d_model 256, 2 + 2 encoder layers,
4 cross-attention layers, decoder 128 x 2): windows/s for a batch of T-frame windows and for the T = 1 per-tick form that
scripts/rt.py feeds.  The CPU figure beside it comes from tests/perf_koemorph_cpu.py (the oracle lives under tests' side of
the fence: nothing outside tests/, smoke() and bench.py's cpu_baseline leg imports it)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from koemorph_amd import synth
from koemorph_amd.model import KoeMorphModel


def flops(c, T):
    d, hid, nb, L = c.d_model, c.decoder_hidden_dim, c.num_blendshapes, c.num_attention_layers
    enc = 2 * T * (c.mel_dim + c.emotion_dim) * d + 2 * c.num_encoder_layers * (24 * T * d * d + 4 * T * T * d)
    cross = 2 * T * d * 2 * L * d + L * (4 * nb * d * d + 4 * nb * T * d)
    dec = 2 * nb * d * hid + c.decoder_layers * 2 * nb * hid * hid + 2 * nb * hid
    return enc + cross + dec


cfg = synth.KoeMorphConfig()
params = synth.make_koemorph_params(5, cfg)
m = KoeMorphModel(d_query=cfg.d_model)
sd = m.state_dict(); sd.update({k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}); m.load_state_dict(sd)
m = m.cuda().eval()
CASES = ((int(os.environ.get("B", 256)), 30), (1024, 1))
if os.environ.get("ONLY"):
    CASES = (CASES[int(os.environ["ONLY"])],)
for B, T in CASES:
    m.reset_temporal_state()                 # a smoother state of another batch size raises, like the reference's expand()
    mel = torch.from_numpy(synth.normal(1, (B, T, 80))).cuda()
    emo = torch.from_numpy(synth.normal(2, (B, T, 256))).cuda()
    prev = torch.from_numpy(synth.uniform(3, (B, 52), 0, 1)).cuda()
    with torch.no_grad():
        for _ in range(int(os.environ.get("WARM", 3))): m(mel, emo, prev_blendshapes=prev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = int(os.environ.get("ITERS", 20))
        for _ in range(n): m(mel, emo, prev_blendshapes=prev)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(json.dumps({"workload": f"KoeMorphModel d256, {B} windows x {T} frames", "ms_per_forward": round(dt * 1e3, 3),
                      "windows_per_s": round(B / dt, 1), "mflop_per_window": round(flops(cfg, T) / 1e6, 1),
                      "algorithmic_tflops": round(flops(cfg, T) * B / dt / 1e12, 2)}))
