#!/usr/bin/env python3
"""Time the front-end kernel alone (C2 shape) -- used for kernel A/B experiments."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from koemorph_amd import synth
from koemorph_amd.engine import Engine
eng = Engine(); eng.load_state_dict(synth.make_core_params(0)); eng.finalize()
B = 256
audio = torch.from_numpy(synth.make_audio(1, B, 136448, "uniform")).cuda()
emo = torch.from_numpy(synth.normal(2, (B, 256))).cuda()
eng.reserve(B, 136448)
out = torch.empty(B, 52, device="cuda")
eng.enable_stage_timing(True)
import statistics
rec = [[], [], []]
for i in range(110):
    eng.forward_audio(audio, emo, out=out)
    if i >= 10:
        for k, t in enumerate(eng.stage_times_ms()): rec[k].append(t * 1e3)
med = [round(statistics.median(r), 1) for r in rec]
mn = [round(min(r), 1) for r in rec]
print(json.dumps({"tag": os.environ.get("KM_EXTRA_FLAGS", ""), "emotion_us": med[0], "mel_us": med[1], "core_us": med[2],
                  "min_us": mn}))
