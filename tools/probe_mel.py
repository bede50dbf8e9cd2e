import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from koemorph_amd import synth
from koemorph_amd.engine import Engine
log = open("gpurun_out/probe.log", "a")
def say(m): log.write(m + "\n"); log.flush(); print(m, flush=True)
eng = Engine(); eng.load_state_dict(synth.make_core_params(0)); eng.finalize()
say("finalized")
for B, L in ((1, 700), (1, 5000), (2, 136448), (256, 136448)):
    audio = torch.from_numpy(synth.make_audio(1, B, L, "uniform")).cuda()
    emo = torch.from_numpy(synth.normal(2, (B, 256))).cuda()
    eng.reserve(B, L)
    say(f"launch B={B} L={L}")
    lo, sh = eng.mel_batch(audio)
    torch.cuda.synchronize()
    say(f"mel_batch ok {tuple(lo.shape)} finite={bool(torch.isfinite(lo).all())}")
    out = eng.forward_audio(audio, emo)
    torch.cuda.synchronize()
    say(f"forward_audio ok {float(out.sum()):.6f}")
