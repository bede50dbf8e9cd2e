#!/usr/bin/env python3
"""Latency of the legacy KoeMorphModel forward at small batches (one real-time stream: B = 1), fused kernels vs the
launch-per-step chain (option kmm_no_fuse) on the same handle."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from koemorph_amd import _lib, synth
from koemorph_amd.model import KoeMorphModel

cfg = synth.KoeMorphConfig()
params = synth.make_koemorph_params(5, cfg)
m = KoeMorphModel(d_query=cfg.d_model)
sd = m.state_dict(); sd.update({k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}); m.load_state_dict(sd)
m = m.cuda().eval()
res = {}
for B, T in ((1, 1), (1, 30), (8, 30), (32, 30)):
    mel = torch.from_numpy(synth.normal(1, (B, T, 80))).cuda()
    emo = torch.from_numpy(synth.normal(2, (B, T, 256))).cuda()
    prev = torch.from_numpy(synth.uniform(3, (B, 52), 0, 1)).cuda()
    for mode in ("fused", "chain"):
        m.reset_temporal_state()
        with torch.no_grad():
            m(mel, emo, prev_blendshapes=prev)
            lib, h, _ = m._handle()
            _lib.check(lib.km_set_option(h, b"kmm_no_fuse", 0 if mode == "fused" else 1))
            for _ in range(50): m(mel, emo, prev_blendshapes=prev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200): m(mel, emo, prev_blendshapes=prev)
            torch.cuda.synchronize()
        res[f"B{B}_T{T}_{mode}_ms"] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
    _lib.check(lib.km_set_option(h, b"kmm_no_fuse", 0))
print(json.dumps(res))
