#!/usr/bin/env python3
"""Legacy SimplifiedKoeMorphModel (row a12, km_legacy_forward): windows/s from audio resident in HBM."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from koemorph_amd import synth
from koemorph_amd.model import SimplifiedKoeMorphModel

B = int(os.environ.get("B", 256))
m = SimplifiedKoeMorphModel().cuda().eval()
audio = torch.from_numpy(synth.make_audio(1, B, 136448, "uniform")).cuda()
with torch.no_grad():
    for _ in range(int(os.environ.get("WARM", 50))): m(audio)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = int(os.environ.get("ITERS", 100))
    for _ in range(n): out = m(audio)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(json.dumps({"workload": f"SimplifiedKoeMorphModel, {B} windows x 136448 samples", "ms_per_forward": round(dt * 1e3, 3), "windows_per_s": round(B / dt, 1)}))
