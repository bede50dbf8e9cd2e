#!/usr/bin/env python3
"""BASELINE config 3: train_sequential-style dense stride-1 step, window 256, 8 windows per GPU, data parallel.

    python tools/bench_train.py                                   # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_train.py

Per step and per rank: log-mel front end of 8 x 136448 samples -> forward -> MSE loss -> backward -> ONE all-reduce
of the flat fp32 gradient bucket over RCCL/xGMI -> global-norm clip -> AdamW.  Prints one JSON line on rank 0."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from koemorph_amd import parallel, synth
from koemorph_amd.engine import Engine
from koemorph_amd.training import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--graph", action="store_true", help="replay forward+backward from a hipGraph")
args = ap.parse_args()
rank, world, local = parallel.init_from_env()
torch.cuda.set_device(local)
dev = torch.device(f"cuda:{local}")
eng = Engine(); eng.load_state_dict(synth.make_core_params(0)); eng.finalize(dev)
tr = Trainer(eng, max_windows=args.batch)
B = args.batch
audio = torch.from_numpy(synth.make_audio(10 + rank, B, 136448, "uniform")).to(dev)
emo = torch.from_numpy(synth.normal(20 + rank, (B, 256))).to(dev)
target = torch.from_numpy(synth.uniform(30 + rank, (B, 52), 0, 1)).to(dev)
for _ in range(args.warmup):
    tr.step(audio, emo, target)
if args.graph:
    tr.capture(B, 136448)
    step = lambda: tr.step_graph(audio, emo, target)
else:
    step = lambda: tr.step(audio, emo, target)
torch.cuda.synchronize(dev)
if world > 1:
    dist.barrier(); torch.cuda.synchronize(dev)
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
torch.cuda.synchronize(dev)
if world > 1:
    dist.barrier(); torch.cuda.synchronize(dev)
dt = time.perf_counter() - t0
if world > 1:
    t = torch.tensor([dt], device=dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
if rank == 0:
    print(json.dumps({"workload": f"C3: train step, {B} windows/GPU x 136448 samples, window 256, d_model 256, AdamW, "
                      f"flat {tr.n_params}-float gradient all-reduce", "n_gpus": world, "graph": bool(args.graph), "steps": args.steps,
                      "ms_per_step": round(dt / args.steps * 1e3, 4), "windows_per_s": round(B * world * args.steps / dt, 1),
                      "final_loss": float(tr.loss.item())}))
if world > 1:
    dist.destroy_process_group()
