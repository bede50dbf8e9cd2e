#!/usr/bin/env python3
"""CPU figure quoted beside tools/bench_koemorph.py: the oracle's KoeMorphModel forward (float32 torch, 16 threads = the
box's CPU share for one GPU) on a bounded sample of the same two workloads.  Not collected by pytest (no test_ prefix);
run as `python tests/perf_koemorph_cpu.py`."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from koemorph_amd import synth
from oracle import koemorph_model as okm

cfg = synth.KoeMorphConfig()
params = synth.make_koemorph_params(5, cfg)
cores = min(16, os.cpu_count() or 1)
torch.set_num_threads(cores)
for T in (30, 1):
    nb = 16
    mel, emo, prev = synth.normal(1, (nb, T, 80)), synth.normal(2, (nb, T, 256)), synth.uniform(3, (nb, 52), 0, 1)
    okm.koemorph_forward(params, cfg, mel, emo, prev_blendshapes=prev, dtype=torch.float32)       # warm-up
    t0 = time.perf_counter()
    for _ in range(3): okm.koemorph_forward(params, cfg, mel, emo, prev_blendshapes=prev, dtype=torch.float32)
    cpu = (time.perf_counter() - t0) / (3 * nb)
    print(json.dumps({"workload": f"KoeMorphModel d256, CPU oracle, windows of {T} frames", "windows_per_s": round(1 / cpu, 1),
                      "cores": cores, "sample": f"3 x {nb} windows, float32 torch"}))
