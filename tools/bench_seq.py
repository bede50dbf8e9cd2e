#!/usr/bin/env python3
"""Sequence mode (SequentialDualStreamModel.forward): output frames/s of km_sequence_forward at stride 1,
shared-frame path vs per-window STFT (option seq_per_window)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from koemorph_amd import synth
from koemorph_amd.engine import Engine

clips, seconds, stride = int(os.environ.get("CLIPS", 4)), float(os.environ.get("SECONDS", 20)), int(os.environ.get("STRIDE", 1))
L = int(seconds * 16000)
eng = Engine(); eng.load_state_dict(synth.make_core_params(0)); eng.finalize()
audio = torch.from_numpy(synth.make_audio(1, clips, L, "uniform")).cuda()
emo = torch.from_numpy(synth.normal(2, (clips, 256))).cuda()
res = {}
for mode in ("shared", "per_window"):
    eng.set_option("seq_per_window", 1 if mode == "per_window" else 0)
    out = eng.sequence_forward(audio, emo, stride, True, max_tile=256)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = eng.sequence_forward(audio, emo, stride, True, max_tile=256)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    res[mode] = {"ms_per_call": round(dt * 1e3, 3), "frames_per_s": round(out.shape[0] * out.shape[1] / dt, 1)}
    keep = out.clone() if mode == "shared" else keep
res["windows"] = int(out.shape[0] * out.shape[1]); res["clips"] = clips; res["seconds"] = seconds; res["stride"] = stride
res["bit_identical"] = bool(torch.equal(keep, out))
print(json.dumps(res))
