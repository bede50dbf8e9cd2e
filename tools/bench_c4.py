#!/usr/bin/env python3
"""BASELINE config 4 on ONE GPU: d_model 512, window 512, 60 fps (hop 266), H in {8, 16}: frames/s of km_forward_audio
through the shape-generic path (staged log-mel + GEMM chain)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from koemorph_amd import synth
from koemorph_amd.engine import Engine, MelConfig

B = int(os.environ.get("B", 64))
for H in (8, 16):
    eng = Engine(d_model=512, num_heads=H, mel_sequence_length=512, mel=MelConfig.model_batch(target_fps=60))
    eng.load_state_dict(synth.make_core_params(0, 512, 512, 256, "init"))
    eng.finalize()
    L = 512 * 266
    eng.reserve(B, L)
    audio = torch.from_numpy(synth.make_audio(1, B, L, "uniform")).cuda()
    emo = torch.from_numpy(synth.normal(2, (B, 256))).cuda()
    out = torch.empty(B, 52, device="cuda")
    for _ in range(5): eng.forward_audio(audio, emo, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    for _ in range(n): eng.forward_audio(audio, emo, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(json.dumps({"workload": f"C4: d_model 512, window 512, H={H}, {B} windows x {L} samples, hop 266", "ms_per_step": round(dt * 1e3, 3),
                      "frames_per_s": round(B / dt, 1), "algorithmic_tflops": round(190e6 * B / dt / 1e12, 2)}))
