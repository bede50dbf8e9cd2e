"""N>1 path on the CPU: world_size-2 gloo processes exercise the sharding, gather and gradient all-reduce
logic that the GPU ranks use over RCCL (there is no data-path collective to test for inference)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from koemorph_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    audio = torch.randn(7, 64, generator=g)                     # 7 windows: uneven split 4 + 3
    emo = torch.randn(7, 5, generator=g)
    calls = []

    def fn(a, e):                                               # stand-in for the per-rank HIP forward
        calls.append(a.shape[0])
        return torch.stack([a.sum(1), e.sum(1), a[:, 0]], dim=1)

    out = parallel.sharded_apply(fn, audio, emo)
    want = fn(audio, emo)
    ok_gather = torch.equal(out, want) and calls[0] == (4 if rank == 0 else 3)
    local = parallel.sharded_apply(fn, audio, emo, gather=False)
    lo, hi = parallel.shard_range(7, rank, world)
    ok_local = torch.equal(local, want[lo:hi])
    grad = torch.full((837738,), float(rank + 1))               # the flat gradient bucket of the d=256 model
    parallel.allreduce_gradients(grad)
    ok_grad = bool(torch.all(grad == 1.5))
    # unequal shares (3 + 2 of 5 windows): each rank's local-mean gradient weighted by its share -> the global mean
    grad = torch.full((1024,), float(rank + 1))
    parallel.allreduce_gradients(grad, weight=(3 if rank == 0 else 2) / 5)
    ok_grad = ok_grad and bool(torch.allclose(grad, torch.full((1024,), 0.6 * 1 + 0.4 * 2)))
    # a rank without any window contributes weight 0 and the other rank's gradient passes through unchanged
    grad = torch.full((16,), 7.0 if rank == 0 else 123.0)
    parallel.allreduce_gradients(grad, weight=1.0 if rank == 0 else 0.0)
    ok_grad = ok_grad and bool(torch.all(grad == 7.0))
    q.put((rank, ok_gather, ok_local, ok_grad))
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_gather_and_gradient_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True, True, True), (1, True, True, True)]


def test_shard_ranges_and_stream_owner():
    for n in (0, 1, 7, 256, 1024, 1025):
        for world in (1, 2, 3, 8):
            rs = [parallel.shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [h - l for l, h in rs]
            assert max(sizes) - min(sizes) <= 1
    # BASELINE config 5: 1024 streams sharded 128 per GPU
    assert [parallel.stream_owner(s, 1024, 8) for s in (0, 127, 128, 1023)] == [0, 0, 1, 7]
    with pytest.raises(ValueError):
        parallel.stream_owner(1024, 1024, 8)


@pytest.mark.parametrize("n", [2, 3])
def test_bench_starts_its_own_ranks(n):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (the driver's command line) must start N ranks
    itself.  The `rendezvous` workload touches neither the GPU nor the library, so the launcher, the gloo rendezvous,
    the all-reduce-of-ones rank count, the max-over-ranks clock and the single rank-0 line are testable on the CPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--workload", "rendezvous", "--steps", "4",
                          "--warmup", "1"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["rccl_ranks"] == n and d["steps"] == 4
    assert d["ms_per_step"] >= 10.0 * n * 0.9                 # the clock is the SLOWEST rank's (rank r sleeps 10 (r+1) ms per step)
    col = d["collective"]                                     # the gradient bucket's all-reduce in both modes, N ranks, same run
    assert col["ranks"] == n and col["floats"] == 837744 and col["bytes"] == 3350976 and col["default"] == "ring"
    for mode in ("ring", "direct"):
        assert col[mode]["allreduce_ms"] > 0 and col[mode]["sum_check"] is True


def _direct_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_from_env("gloo")
    g = torch.Generator().manual_seed(100 + rank)
    n = 837744 + 3                                             # not a multiple of the world size or of 4
    mine = torch.randn(n, generator=g)
    ref = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    out = parallel.allreduce_sum_direct(mine.clone())
    os.environ["KM_ALLREDUCE"] = "direct"
    via = parallel.allreduce_gradients(mine.clone(), weight=1.0)
    q.put((rank, bool(torch.allclose(out, ref, atol=1e-5)), bool(torch.equal(out, via)), out[:64].clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_direct_allreduce_world3():
    """all-to-all + local sum in rank order + all-gather (the xGMI-shaped form of the gradient all-reduce): equals the
    plain sum, and every rank ends with bit-identical values."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_direct_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] for r in res)
    assert torch.equal(res[0][3], res[1][3]) and torch.equal(res[0][3], res[2][3])


# ---- sequence mode across ranks: chunks of a clip's output frames + one window of halo -----------------
class _FakeMel:
    hop_length = 10


class _FakeSeqEngine:
    """Host-side stand-in for koemorph_amd.engine.Engine: a window's "blendshapes" are sums over its (zero-padded) samples, the
    EMA is the real recurrence -- enough to pin chunk ranges, halos, tail padding, gather order and the one smoothing pass."""
    mel = _FakeMel()
    mel_sequence_length = 4          # window = 40 samples
    num_blendshapes = 3
    alpha = 0.7

    def sequence_num_outputs(self, L, stride):
        return max(1, (L // 10 - 4) // stride + 1)

    def sequence_forward(self, audio, emo, stride, smooth=True):
        B, L = audio.shape
        N = self.sequence_num_outputs(L, stride)
        out = torch.zeros(B, N, 3)
        for i in range(N):
            w = torch.zeros(B, 40)
            seg = audio[:, i * stride * 10: i * stride * 10 + 40]
            w[:, : seg.shape[1]] = seg
            out[:, i] = torch.stack([w.sum(1), (w * torch.arange(40.0)).sum(1), w[:, 0] + emo.sum(1)], dim=1)
        if smooth:
            self.ema_scan(out)
        return out

    def ema_scan(self, seq):
        for n in range(1, seq.shape[1]):
            seq[:, n] = self.alpha * seq[:, n] + (1 - self.alpha) * seq[:, n - 1]


def _seq_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_from_env("gloo")
    eng = _FakeSeqEngine()
    g = torch.Generator().manual_seed(1)
    ok = True
    for L, stride in ((40 + 10 * 9, 1), (40 + 10 * 9 + 7, 2), (40 + 10 * 2, 1), (25, 1), (40 + 10 * 30 + 3, 3)):
        audio = torch.randn(2, L, generator=g)
        emo = torch.randn(2, 5, generator=g)
        want = eng.sequence_forward(audio, emo, stride, smooth=True)
        got = parallel.sequence_apply(eng, audio, emo, stride, smooth=True)
        ok = ok and torch.equal(got, want)
        got = parallel.sequence_apply(eng, audio, emo, stride, smooth=False)
        ok = ok and torch.equal(got, eng.sequence_forward(audio, emo, stride, smooth=False))
    # the chunks tile the output frames, and a chunk's sample range is its own span + one window of halo, cut at the clip end
    N = eng.sequence_num_outputs(137, 2)
    spans = [parallel.sequence_chunk(137, 10, 4, 2, N, r, world) for r in range(world)]
    ok = ok and spans[0][0] == 0 and spans[-1][1] == N and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    for lo, hi, s0, s1 in spans:
        if hi > lo:
            ok = ok and s0 == lo * 20 and s1 == min(137, s0 + (hi - lo - 1) * 20 + 40)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sequence_mode_chunks_over_ranks(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seq_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]
