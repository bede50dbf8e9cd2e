"""eGeMAPSv02 front end throughput: 20 s windows per second on one GPU (synthetic voiced audio), with a per-kernel split
when run under rocprofv3.   python tools/bench_egemaps.py [windows=64] [reps=20]
Per-kernel roofs (HBM fraction of the algorithmic bytes, executed vector work against the vector pipe, LDS busy, wait share):
tools/prof_egemaps.sh -> profiles/r04_egemaps_roofline.txt."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from koemorph_amd import synth
from koemorph_amd.features.opensmile_extractor import EGeMAPSEngine

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng = EGeMAPSEngine()
base = [synth.make_vowel(100 + i, 110.0 + 10.0 * i, 20.0, jitter=0.01) for i in range(8)]
x = np.stack([base[i % 8] for i in range(nw)]).astype(np.float32)
xd = torch.from_numpy(x).cuda()
for _ in range(3):
    eng.functionals(xd)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    eng.functionals(xd)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{nw} windows of 20 s: {dt * 1e3:.2f} ms per call => {nw / dt:.0f} windows/s ({nw * 20 / dt:.0f} x real time)")
