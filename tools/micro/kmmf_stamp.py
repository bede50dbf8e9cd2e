"""Where a wave of kmmf_encoder_kernel spends its cycles (256 windows x 30 frames).  Needs a library built with -DKM_KMMF_STAMP:
  bash tools/micro/lib_variant.sh kmmf_stamp km_kmmf.hip -DKM_KMMF_STAMP
  KM_LIBRARY=tools/micro/bin/libkm_kmmf_stamp.so python tools/micro/kmmf_stamp.py"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from koemorph_amd import _lib, synth           # noqa: E402
from koemorph_amd.model import KoeMorphModel   # noqa: E402

# part, MFMAs a wave issues in it per pass through the kernel (32 rows, 4 waves: per layer unless noted)
PARTS = [("staging + barrier (once)", 0), ("input projection GEMM (once)", None), ("LayerNorm 0 (once)", 0),
         ("Q K V projection, 2 heads", 2 * 48 * 16), ("scores + softmax + P V, 2 heads", 2 * (32 + 32)), ("barrier after attention", 0),
         ("out_proj GEMM", 32 * 16), ("bias + residual + LayerNorm 1 + barrier", 0), ("linear1 GEMM (4 chunks)", 4 * 32 * 16),
         ("bias + GELU + LDS stores + barriers", 0), ("linear2 GEMM (4 chunks)", 4 * 32 * 16),
         ("bias + residual + LayerNorm 2 + barrier", 0), ("output rows (once)", 0)]


def main():
    B, T = 256, 30
    cfg = synth.KoeMorphConfig()
    params = synth.make_koemorph_params(5, cfg)
    m = KoeMorphModel(d_query=cfg.d_model)
    sd = m.state_dict(); sd.update({k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}); m.load_state_dict(sd)
    m = m.cuda().eval()
    mel = torch.from_numpy(synth.normal(1, (B, T, 80))).cuda()
    emo = torch.from_numpy(synth.normal(2, (B, T, 256))).cuda()
    prev = torch.from_numpy(synth.uniform(3, (B, 52), 0, 1)).cuda()
    with torch.no_grad():
        t0 = time.time()
        while time.time() - t0 < 0.4:
            for _ in range(20):
                m(mel, emo, prev_blendshapes=prev)
            torch.cuda.synchronize()
    lib = _lib.load()
    n = 512 * 4 * 16
    buf = (C.c_ulonglong * n)()
    lib.km_debug_kmmf_stamps.restype = C.c_int
    assert lib.km_debug_kmmf_stamps(buf, n) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(512, 4, 16).astype(np.float64)
    mean = st.mean(axis=(0, 1))
    total = mean[:13].sum()
    print(f"cycles per wave (s_memtime units), mean over 512 workgroups x 4 waves; total {total:.0f}")
    in_dim_mfma = (5 + 16) / 2 * 32          # mel 5 k blocks, emotion 16: mean over the two streams
    for i, (name, mf) in enumerate(PARTS):
        mf = in_dim_mfma if mf is None else mf
        layers = 1 if "(once)" in name else 2
        print(f"{i:2d} {name:45s} {mean[i]:10.0f}  {100 * mean[i] / total:5.1f} %   mfma {mf * layers:6.0f}  cycles/mfma {mean[i] / (mf * layers) if mf else 0:6.1f}")
    print("spread over waves (total cycles): min %.0f max %.0f" % (st[:, :, :13].sum(axis=2).min(), st[:, :, :13].sum(axis=2).max()))
    n2 = 2048 * 4 * 16 + 256 * 8 * 16
    buf2 = (C.c_ulonglong * n2)()
    assert lib.km_debug_kmmf_stamps(buf2, n2) == 0
    sd = np.frombuffer(buf2, dtype=np.uint64)[2048 * 4 * 16:].reshape(256, 8, 16).astype(np.float64)
    mean = sd.mean(axis=(0, 1))
    total = mean[:8].sum()
    print(f"decode kernel: cycles per wave, mean over 256 workgroups x 8 waves; total {total:.0f}")
    DPARTS = [("conditioning net + queries + averaged keys", 0), ("Q K V projection (4 layers)", 4 * 64 * 16), ("scores + softmax + P V", 4 * 128),
              ("barrier after attention", 0), ("out_proj GEMM", 4 * 32 * 16), ("bias + residual + LayerNorm + barrier", 0),
              ("decoder MLP (input + 2 hidden layers)", 16 * 16 + 2 * 16 * 8), ("output projection + tail", 0)]
    for i, (name, mf) in enumerate(DPARTS):
        print(f"{i:2d} {name:45s} {mean[i]:10.0f}  {100 * mean[i] / total:5.1f} %   mfma {mf:6.0f}  cycles/mfma {mean[i] / mf if mf else 0:6.1f}")


main()
