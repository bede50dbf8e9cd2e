#!/usr/bin/env python3
"""Determinism soak: the same batch through km_forward_audio (and the stream tick, the sequence path, the train step)
many times; every repetition must reproduce the first result bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from koemorph_amd import synth
from koemorph_amd.engine import Engine
from koemorph_amd.training import Trainer

n = int(os.environ.get("REPS", 300))
eng = Engine(); eng.load_state_dict(synth.make_core_params(0, style="trained")); eng.finalize()
B = 256
audio = torch.from_numpy(synth.make_audio(1, B, 136448, "uniform")).cuda()
emo = torch.from_numpy(synth.normal(2, (B, 256))).cuda()
eng.reserve(B, 136448)
first = eng.forward_audio(audio, emo).clone()
bad = sum(int(not torch.equal(eng.forward_audio(audio, emo), first)) for _ in range(n))
print(f"forward_audio: {n} repetitions, {bad} differ")
seq_a = torch.from_numpy(synth.make_audio(3, 2, 136448 + 533 * 40)).cuda()
s0 = eng.sequence_forward(seq_a, emo[:2], 1, True, max_tile=16).clone()
bad_s = sum(int(not torch.equal(eng.sequence_forward(seq_a, emo[:2], 1, True, max_tile=16), s0)) for _ in range(50))
print(f"sequence_forward: 50 repetitions, {bad_s} differ")
tr = Trainer(eng, max_windows=8, use_smoothing=False)
tgt = torch.from_numpy(synth.uniform(4, (8, 52), 0, 1)).cuda()
tr.forward_backward(audio[:8], emo[:8], tgt); g0 = tr.flat_grad.clone(); l0 = float(tr.loss.item())
bad_t = 0
for _ in range(100):
    tr.forward_backward(audio[:8], emo[:8], tgt)
    bad_t += int(not torch.equal(tr.flat_grad, g0) or float(tr.loss.item()) != l0)
print(f"train forward_backward: 100 repetitions, {bad_t} differ")
# round 4: LDS-DMA tiles / attention blocks with dropout (the same Philox step every time), 8 and 64 windows, and 1000 optimiser steps
import ctypes
for Bt in (8, 64):
    tr2 = Trainer(eng, max_windows=Bt, use_smoothing=False, dropout=0.1, seed=7)
    tgt2 = torch.from_numpy(synth.uniform(5, (Bt, 52), 0, 1)).cuda()
    def once():
        assert tr2._lib.km_train_set_dropout_step(tr2._h, 0) == 0
        tr2.forward_backward(audio[:Bt], emo[:Bt], tgt2)
        return tr2.flat_grad.clone(), float(tr2.loss.item())
    g0, l0 = once()
    bad_d = 0
    for _ in range(100):
        g, l = once()
        bad_d += int(not torch.equal(g, g0) or l != l0)
    print(f"train forward_backward with dropout, {Bt} windows: 100 repetitions, {bad_d} differ")
    bad_t += bad_d
    if Bt == 8:
        for _ in range(1000):
            tr2.step(audio[:8], emo[:8], tgt2)
        fin = float(tr2.loss.item())
        print(f"1000 optimiser steps at 8 windows: final loss {fin:.6f} (finite: {fin == fin and abs(fin) < 1e9})")
        bad_t += int(not (fin == fin and abs(fin) < 1e9))
core_bad = bad or bad_s or bad_t

# the fused legacy models: determinism of the multi-wave reductions (LayerNorm partials, online softmax)
import numpy as np
from koemorph_amd.model import KoeMorphModel, SimplifiedKoeMorphModel
cfg = synth.KoeMorphConfig()
m = KoeMorphModel(d_query=cfg.d_model)
sd = m.state_dict(); sd.update({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_koemorph_params(5, cfg).items()}); m.load_state_dict(sd)
m = m.cuda().eval()
mel = torch.from_numpy(synth.normal(1, (64, 30, 80))).cuda(); emf = torch.from_numpy(synth.normal(2, (64, 30, 256))).cuda()
with torch.no_grad():
    m.reset_temporal_state(); k0 = m(mel, emf, apply_smoothing=False)["blendshapes"].clone()
    bad_k = sum(int(not torch.equal(m(mel, emf, apply_smoothing=False)["blendshapes"], k0)) for _ in range(100))
print(f"KoeMorphModel (fused kernels): 100 repetitions, {bad_k} differ")
lg = SimplifiedKoeMorphModel().cuda().eval()
with torch.no_grad():
    a0 = lg(audio[:32]).clone()
    bad_l = sum(int(not torch.equal(lg(audio[:32]), a0)) for _ in range(100))
print(f"SimplifiedKoeMorphModel (fused kernels): 100 repetitions, {bad_l} differ")
sys.exit(1 if (core_bad or bad_k or bad_l) else 0)
