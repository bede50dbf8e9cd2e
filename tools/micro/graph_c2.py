"""C2 step eager vs hipGraph replay (torch.cuda.graph around Engine.forward_audio): what the inter-kernel gaps cost.
   python tools/micro/graph_c2.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from koemorph_amd import synth
from koemorph_amd.engine import Engine

dev = torch.device("cuda:0")
B, L = 256, 136448
eng = Engine(); eng.load_state_dict(synth.make_core_params(0, style="init")); eng.finalize(dev); eng.reserve(B, L)
audio = torch.from_numpy(synth.make_audio(100, B, L, style="uniform")).to(dev)
emo = torch.from_numpy(synth.normal(200, (B, 256))).to(dev)
state = torch.zeros(B, 52, device=dev); out = torch.empty(B, 52, device=dev)
def step(): eng.forward_audio(audio, emo, state=state, first=False, out=out)
eng.forward_audio(audio, emo, state=state, first=True, out=out)
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("eager  %.1f us/step" % timeit(step))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    step()
print("graph  %.1f us/step" % timeit(g.replay))
g4 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g4, stream=s):
    for _ in range(4): step()
print("graph x4 %.1f us/step" % (timeit(g4.replay, 50) / 4))
print("eager  %.1f us/step" % timeit(step))
