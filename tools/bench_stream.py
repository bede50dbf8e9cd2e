#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU: 128 concurrent speaker streams, per-tick decode under a hipGraph.
Prints ticks/s and p50/p99 tick latency (host wall clock around replay + the 26 KB result readback)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from koemorph_amd import synth
from koemorph_amd.engine import Engine
from koemorph_amd.streaming import StreamEngine

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=128)
ap.add_argument("--ticks", type=int, default=1000)
args = ap.parse_args()
eng = Engine(); eng.load_state_dict(synth.make_core_params(0)); eng.finalize()
se = StreamEngine(eng, args.streams)
S = args.streams
frames = torch.from_numpy(synth.make_audio(1, S, 533 * 8, "uniform")).cuda()
emo = torch.from_numpy(synth.normal(2, (S, 256))).cuda()
for t in range(258):                                   # fill the rings (eager)
    se.push(frames[:, (t % 8) * 533:(t % 8 + 1) * 533]); se.tick(emo)
host_out = torch.empty(S, 52, pin_memory=True)
se.capture(533, host_out=host_out)
lat = []
torch.cuda.synchronize()
t_all = time.perf_counter()
for t in range(args.ticks):
    t0 = time.perf_counter()
    out, ready = se.replay(frames[:, (t % 8) * 533:(t % 8 + 1) * 533], emo)
    torch.cuda.synchronize()
    lat.append(time.perf_counter() - t0)
t_all = time.perf_counter() - t_all
lat = np.array(lat) * 1e3
print(json.dumps({"workload": f"C5: {S} streams/GPU, one 533-sample frame per stream per tick, hipGraph replay + D2H of {S}x52 floats",
                  "ticks_per_s": round(args.ticks / t_all, 1), "frames_per_s": round(args.ticks * S / t_all, 1),
                  "tick_latency_ms_p50": round(float(np.percentile(lat, 50)), 4),
                  "tick_latency_ms_p99": round(float(np.percentile(lat, 99)), 4),
                  "realtime_budget_ms": 33.3, "all_ready": bool(ready.cpu().all())}))
