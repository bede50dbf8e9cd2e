"""Where a wave of mel_power_rp_kernel spends its cycles at the C2 shape.  Needs a library built with -DKM_MEL_STAMP
(tools/micro/mel_xchg.sh with EXTRA=-DKM_MEL_STAMP):  KM_LIBRARY=tools/micro/bin/libkm_xchg0.so python tools/micro/mel_stamp.py"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from koemorph_amd import _lib, synth           # noqa: E402
from koemorph_amd.engine import Engine         # noqa: E402

PARTS = ["loop top / sample wait", "window + pass 1", "exchange 1 (+ next loads issued)", "pass 2", "exchange 2", "pass 3",
         "partner permutation", "post-processing + power stores", "(frame loop exit)", "barrier before mel stage", "mel stage",
         "barrier after mel stage"]


def main():
    B, L = 256, 136448
    eng = Engine()
    eng.load_state_dict(synth.make_core_params(0, style="init"))
    eng.finalize("cuda:0")
    eng.reserve(B, L)
    audio = torch.from_numpy(synth.make_audio(100, B, L, style="uniform")).cuda()
    emo = torch.from_numpy(synth.normal(200, (B, 256))).cuda()
    state = torch.zeros(B, 52, device="cuda"); out = torch.empty(B, 52, device="cuda")
    eng.forward_audio(audio, emo, state=state, first=True, out=out)
    t0 = time.time()
    while time.time() - t0 < 0.4:                    # steady-state clock
        for _ in range(50):
            eng.forward_audio(audio, emo, state=state, out=out)
        torch.cuda.synchronize()
    lib = _lib.load()
    n = 512 * 8 * 24
    buf = (C.c_ulonglong * n)()
    lib.km_debug_mel_stamps.restype = C.c_int
    rc = lib.km_debug_mel_stamps(buf, n)
    assert rc == 0, rc
    st = np.frombuffer(buf, dtype=np.uint64).reshape(512, 8, 24).astype(np.float64)
    total = st[:, :, 12]
    print(f"waves: {st.shape[0] * 8}; cycles per wave (first to last stamp): mean {total.mean():.0f}  min {total.min():.0f}  max {total.max():.0f}")
    for even in (0, 1):
        sel = st[even::2]
        tot = sel[:, :, 12].mean()
        print(f"-- workgroups with blockIdx.x = {even} ({'9 chunks' if even == 0 else '8 chunks + emotion rider'}): {tot:.0f} cycles per wave")
        for i, name in enumerate(PARTS):
            v = sel[:, :, i].mean()
            print(f"   {name:34s} {v:9.0f} cycles  {100 * v / tot:5.1f} %")
    ent, first, last, rt0, rt1, end = (st[:, :, i] for i in (13, 14, 15, 16, 17, 18))
    # the shader-clock counter is per XCD (not comparable across workgroups); the 100 MHz real-time clock is global
    span_rt = rt1.max() - rt0.min()
    ghz = np.median((end - ent) / np.maximum(rt1 - rt0, 1)) / 10.0
    print(f"kernel span (real-time clock): {span_rt / 100:.2f} us; shader-clock counter runs at {ghz:.3f} GHz")
    for role in (0, 1):
        sel = slice(role, None, 2)
        print(f"   role {role}: starts {np.mean(rt0[sel] - rt0.min()) / 100:6.2f} us after the first wave (max {np.max(rt0[sel] - rt0.min()) / 100:.2f}); "
              f"prologue {np.mean(first[sel] - ent[sel]) / ghz / 1e3:6.2f} us; loop {np.mean(last[sel] - first[sel]) / ghz / 1e3:6.2f} us; "
              f"ends {np.mean(rt1[sel] - rt0.min()) / 100:6.2f} us (max {np.max(rt1[sel] - rt0.min()) / 100:.2f}, min {np.min(rt1[sel] - rt0.min()) / 100:.2f})")
    frames = 257 * 256 / (512 * 8)
    print(f"frames per wave: {frames:.2f}")


if __name__ == "__main__":
    main()
