#!/usr/bin/env python3
"""Experiment: the legacy model's 256 windows as two half batches on two streams (two model instances = two workspaces):
do the kernels of one half fill the tails of the other's?"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from koemorph_amd import synth
from koemorph_amd.model import SimplifiedKoeMorphModel

B = 256
NS = int(os.environ.get("NS", 2))
torch.manual_seed(0)
ms = [SimplifiedKoeMorphModel().cuda().eval() for _ in range(NS)]
for m in ms[1:]:
    m.load_state_dict(ms[0].state_dict())
audio = torch.from_numpy(synth.make_audio(1, B, 136448, "uniform")).cuda()
parts = [audio[i * B // NS:(i + 1) * B // NS].contiguous() for i in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]

def step():
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    outs = []
    for m, p, s in zip(ms, parts, streams):
        with torch.cuda.stream(s):
            outs.append(m(p))
    for s in streams:
        cur.wait_stream(s)
    return outs

with torch.no_grad():
    for _ in range(50): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 200
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    ref = ms[0](audio)
    got = torch.cat(step(), 0)
print(json.dumps({"streams": NS, "ms_per_256_windows": round(dt * 1e3, 3), "max_abs_diff_vs_one_call": float((ref - got).abs().max())}))
