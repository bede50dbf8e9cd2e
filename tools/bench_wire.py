#!/usr/bin/env python3
"""Host-side output encoding: km_format_frames (one C call per tick) vs the reference's per-frame json.dumps, at the C5
shape (1024 streams x 52 coefficients per tick)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from koemorph_amd.wire import format_frames_raw

S = 1024
frames = (np.random.default_rng(0).uniform(0, 1, (S, 52)) * 0.02).astype(np.float32)
ts = np.full(S, 1728000000.123456)
def per_frame():
    return [json.dumps({"timestamp": float(t), "blendshapes": r.tolist()}).encode() for r, t in zip(frames, ts)]
def batched():
    return format_frames_raw(frames, ts)
for f in (per_frame, batched): f()
res = {}
for name, f in (("json_dumps_per_frame", per_frame), ("km_format_frames", batched)):
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 1.0:
        f(); n += 1
    dt = (time.perf_counter() - t0) / n
    res[name] = {"ms_per_tick": round(dt * 1e3, 3), "frames_per_s": round(S / dt)}
raw, off = batched()
res["identical"] = [raw[off[i]:off[i + 1]] for i in range(S)] == per_frame()
res["needed_frames_per_s"] = 1024 * 30
print(json.dumps(res))
