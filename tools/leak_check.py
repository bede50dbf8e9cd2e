#!/usr/bin/env python3
"""Create / finalize / reserve / run / destroy handles in a loop and watch the device's free memory: nothing may leak."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from koemorph_amd import synth
from koemorph_amd.engine import Engine
from koemorph_amd.model import KoeMorphModel, SimplifiedKoeMorphModel
from koemorph_amd.training import Trainer


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2 ** 20


def cycle(i):
    eng = Engine(); eng.load_state_dict(synth.make_core_params(i, style="init")); eng.finalize(); eng.reserve(32, 136448)
    audio = torch.from_numpy(synth.make_audio(1, 32, 136448, "uniform")).cuda()
    emo = torch.from_numpy(synth.normal(2, (32, 256))).cuda()
    eng.forward_audio(audio, emo)
    tr = Trainer(eng, max_windows=8, use_smoothing=False)
    tr.step(audio[:8], emo[:8], torch.rand(8, 52, device="cuda"))
    g = Engine(d_model=64, num_heads=4, mel_sequence_length=32); g.load_state_dict(synth.make_core_params(i, 64, 32, 256, "init")); g.finalize()
    g.reserve(4, 32 * 533); g.forward_audio(audio[:4, :32 * 533].contiguous(), emo[:4])
    m = KoeMorphModel(d_model=64, d_query=64, num_heads=4, num_encoder_layers=1, num_attention_layers=1, decoder_hidden_dim=32, emotion_dim=8).cuda().eval()
    with torch.no_grad():
        m(torch.randn(2, 12, 80, device="cuda"), torch.randn(2, 12, 8, device="cuda"))
    mf = KoeMorphModel(d_query=256).cuda().eval()       # default width: the two fused kernels and their weight blobs
    with torch.no_grad():
        mf(torch.randn(3, 30, 80, device="cuda"), torch.randn(3, 30, 256, device="cuda"))
    del tr, eng, g, m, mf, audio, emo
    gc.collect(); torch.cuda.empty_cache()


cycle(0)
base = free_mb()
n = int(os.environ.get("REPS", 30))
for i in range(1, n + 1):
    cycle(i)
    if i % 10 == 0:
        print(f"after {i} cycles: free {free_mb():.1f} MiB (start {base:.1f})", flush=True)
leak = base - free_mb()
print(f"leak over {n} cycles: {leak:.1f} MiB")
sys.exit(1 if leak > 64 else 0)
