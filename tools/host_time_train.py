import time, torch, numpy as np
from koemorph_amd import synth
from koemorph_amd.engine import Engine
from koemorph_amd.training import Trainer
B, L = 8, 136448
eng = Engine(); eng.load_state_dict(synth.make_core_params(0)); eng.finalize("cuda:0")
tr = Trainer(eng, max_windows=B, dropout=0.1)
audio = torch.from_numpy(synth.make_audio(10, B, L, "uniform")).cuda(); emo = torch.from_numpy(synth.normal(20, (B, 256))).cuda(); target = torch.from_numpy(synth.uniform(30, (B, 52), 0, 1)).cuda()
for _ in range(200): tr.step(audio, emo, target)
torch.cuda.synchronize()
# host enqueue time: few steps so that the queue never fills
ts = []
for rep in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): tr.step(audio, emo, target)
    ts.append((time.perf_counter() - t0) / 8)
    torch.cuda.synchronize()
print("host enqueue time per step: %.1f us (min %.1f)" % (np.median(ts) * 1e6, min(ts) * 1e6))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): tr.step(audio, emo, target)
torch.cuda.synchronize(); print("steady step: %.1f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
