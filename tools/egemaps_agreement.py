"""Observed agreement between the GPU eGeMAPS descriptors and oracle/egemaps.py on the discrete decisions (voicing, formant
validity, pitch marks) over several speech-like signals: the numbers the thresholds of tests/test_gpu_egemaps.py are set from.
   python tools/egemaps_agreement.py  [seeds=11,12,13,14,15,16,17,18]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from koemorph_amd.features.opensmile_extractor import EGeMAPSEngine
from oracle import egemaps as eg
import test_gpu_egemaps as T

seeds = [int(s) for s in (sys.argv[1].split(",") if len(sys.argv) > 1 else "11,12,13,14,15,16,17,18".split(","))]
engine = EGeMAPSEngine()
R = T.R
rows = []
for seed in seeds:
    x = T.speechlike(seed)
    engine.functionals(torch.from_numpy(x[None]).cuda(), normalize=True)
    rec = engine.records()[0]
    d = eg.llds(eg.normalise(x))
    gv, ov = rec[:, R["f0"]] > 0, d["f0"] > 0
    both = gv & ov
    voicing = float((gv == ov).mean())
    fvalid = [float(((both & (rec[:, R["F"] + i] > 0) & (d["F"][:, i] > 0)).sum()) / max(both.sum(), 1)) for i in range(3)]
    jit = float((both & (np.abs(rec[:, R["jit"]] - d["jitterLocal"]) < 1e-4)).sum() / max(both.sum(), 1))
    rows.append((seed, len(gv), int(both.sum()), voicing, fvalid, jit))
    print(f"seed {seed}: frames {len(gv)} voiced-in-both {both.sum()}  voicing agreement {voicing:.4f}  formant valid in both "
          f"{fvalid[0]:.4f} {fvalid[1]:.4f} {fvalid[2]:.4f}  jitter agrees {jit:.4f}")
print("min voicing", min(r[3] for r in rows), "min formant", min(min(r[4]) for r in rows), "min jitter", min(r[5] for r in rows))
