#!/usr/bin/env python3
"""C2 step in stream order against km_forward_audio_pipelined (front end of call i on an internal stream beside the core of call
i - 1, double-buffered workspace), both on rotating inputs."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from koemorph_amd import synth
from koemorph_amd.engine import Engine

B, L = 256, 136448
eng = Engine(); eng.load_state_dict(synth.make_core_params(0)); eng.finalize("cuda:0"); eng.reserve(B, L)
audio = torch.from_numpy(synth.make_audio(1, B, L, "uniform")).cuda()
bufs = [audio] + [torch.roll(audio, shifts=i, dims=0).contiguous() for i in range(1, 4)]
emo = torch.from_numpy(synth.normal(2, (B, 256))).cuda()
out = torch.empty(B, 52, device="cuda"); state = torch.zeros(B, 52, device="cuda")
outs = [torch.empty(B, 52, device="cuda") for _ in range(4)]

def run(pipelined, n):
    for i in range(n):
        if pipelined:
            eng.forward_audio_pipelined(bufs[i & 3], emo, state=state, first=False, out=outs[i & 3])
        else:
            eng.forward_audio(bufs[i & 3], emo, state=state, first=False, out=outs[i & 3])
    if pipelined:
        eng.pipeline_flush()
res = {}
for mode in (False, True, False, True):
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end: run(mode, 20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(mode, 400)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 400
    res.setdefault("pipelined" if mode else "stream_order", []).append(round(dt * 1e3, 4))
print(json.dumps({"ms_per_256_windows": res}))
