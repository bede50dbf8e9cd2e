"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The hot path shards embarrassingly: every window is independent and the only
cross-frame state (EMA, ring buffers) is per stream.  So
  * batch inference / sequence mode: a global batch of windows is split into contiguous per-rank ranges,
    weights are replicated (1 MB), there is NO data-path collective; gathering the (B, 52) results is
    optional and 208 B per frame;
  * streaming: stream s lives on rank ``stream_owner(s)`` for its whole life (ring, cached mel, EMA state
    never leave a GPU);
  * training: the only collective is ONE all-reduce of the flat fp32 gradient bucket per step
    (``allreduce_gradients``), 837 738 floats = 3.35 MB at d=256 -- latency-bound on xGMI, so it is issued
    as a single bucket rather than per-tensor.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank); a no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this host driver
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def collectives_active() -> bool:
    """A process group with more than one rank -- or with ONE rank and KM_COLLECTIVES_AT_WORLD_1=1: the bring-up switch that
    sends the step's collectives through the backend anyway, so that a 1-GPU box runs RCCL's all-reduce / all-gather on the
    step's own tensors and streams (tests/test_gpu_parallel.py; a sum over one rank leaves every bit as it was)."""
    if not dist.is_initialized():
        return False
    return dist.get_world_size() > 1 or os.environ.get("KM_COLLECTIVES_AT_WORLD_1", "0") == "1"


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of n items for `rank`: sizes differ by at most one, earlier ranks get the extras."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def stream_owner(stream_id: int, n_streams: int, world: int) -> int:
    """Rank that owns a speaker stream (contiguous blocks: 1024 streams on 8 GPUs -> 128 per GPU)."""
    for r in range(world):
        lo, hi = shard_range(n_streams, r, world)
        if lo <= stream_id < hi:
            return r
    raise ValueError(f"stream {stream_id} out of range for {n_streams} streams")


def sharded_apply(fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], audio: torch.Tensor,
                  emotion: torch.Tensor, gather: bool = True) -> torch.Tensor:
    """Run ``fn(audio_shard, emotion_shard) -> (b, 52)`` on this rank's contiguous share of a global batch that
    every rank holds (or can index), and optionally all-gather the rows back in global order."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return fn(audio, emotion)
    rank, world = dist.get_rank(), dist.get_world_size()
    n = audio.shape[0]
    lo, hi = shard_range(n, rank, world)
    local = fn(audio[lo:hi], emotion[lo:hi])
    if not gather:
        return local
    sizes = [shard_range(n, r, world) for r in range(world)]
    maxb = max(h - l for l, h in sizes)
    pad = torch.zeros((maxb,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: hi - lo] = local
    parts: List[torch.Tensor] = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: h - l] for p, (l, h) in zip(parts, sizes)], dim=0)


def sequence_chunk(L: int, hop: int, window_frames: int, stride_frames: int, n_out: int, rank: int, world: int):
    """This rank's contiguous chunk [lo, hi) of a clip's ``n_out`` output frames and the sample range [s0, s1) of the clip it
    needs: window i starts at sample i * stride * hop and spans window_frames * hop samples, so a chunk of n windows reads
    (n - 1) * stride * hop + window_frames * hop samples -- its own span plus a halo of one window.
    The last chunk ends with the clip (its windows are zero-padded beyond it exactly as on one GPU)."""
    lo, hi = shard_range(n_out, rank, world)
    if hi <= lo:
        return lo, hi, 0, 0
    s0 = lo * stride_frames * hop
    s1 = min(L, s0 + (hi - lo - 1) * stride_frames * hop + window_frames * hop)
    return lo, hi, s0, s1


def sequence_apply(engine, audio: torch.Tensor, emotion: torch.Tensor, stride_frames: int = 1, smooth: bool = True) -> torch.Tensor:
    """``SequentialDualStreamModel.forward`` of ONE batch of clips over all ranks (reference
    src/model/sequential_dual_stream_model.py:84-151): every rank holds the clips, computes a contiguous chunk of each clip's
    output frames WITHOUT smoothing (``engine.sequence_forward`` on the chunk's samples + one window of halo), the chunks are
    all-gathered in rank order (208 bytes per frame: the only exchange, and it is not on the kernels' path) and the EMA --
    a first-order recurrence along the frame axis -- runs once over the gathered (B, N, 52).  Every rank returns the full
    sequence; it is bit-identical to the single-rank result because every window is computed by the same kernels from the same
    samples whichever sub-clip it is addressed in."""
    if not collectives_active():
        return engine.sequence_forward(audio, emotion, stride_frames, smooth=smooth)
    rank, world = dist.get_rank(), dist.get_world_size()
    B, L = audio.shape
    hop, T = int(engine.mel.hop_length), int(engine.mel_sequence_length)
    N = int(engine.sequence_num_outputs(L, stride_frames))
    nb = int(getattr(engine, "num_blendshapes", 52))
    lo, hi, s0, s1 = sequence_chunk(L, hop, T, stride_frames, N, rank, world)
    sizes = [shard_range(N, r, world) for r in range(world)]
    maxn = max(h - l for l, h in sizes)
    pad = torch.zeros((B, maxn, nb), dtype=torch.float32, device=audio.device)
    if hi > lo:
        part = engine.sequence_forward(audio[:, s0:s1].contiguous(), emotion, stride_frames, smooth=False)
        if part.shape[1] != hi - lo:
            raise RuntimeError(f"sequence chunk [{lo}, {hi}) of {N} frames came back with {part.shape[1]} frames")
        pad[:, : hi - lo] = part
    parts: List[torch.Tensor] = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    seq = torch.cat([p[:, : h - l] for p, (l, h) in zip(parts, sizes)], dim=1).contiguous()
    if smooth and seq.shape[1] > 0:
        engine.ema_scan(seq)
    return seq


class DirectAllReduce:
    """Sum of a flat fp32 tensor over the ranks as ONE exchange over every link at once (opt-in: KM_ALLREDUCE=direct).

    xGMI on an MI355X node is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce of this 3.35 MB bucket is
    2 (N - 1) hops over one link at a time.  Direct form: all-to-all (rank j receives everyone's copy of shard j, one
    shard per link), a local sum of the N copies in rank order, all-gather of the summed shards (again one shard per
    link): two hops regardless of N, every link busy, and every rank ends with bit-identical values.  The three staging
    buffers are allocated once per (size, world) -- nothing is allocated in a training step.

    Status: correct under gloo (tests/test_parallel_gloo.py); NOT the default -- RCCL's own small-message all-reduce is
    what the training step uses (``dist.all_reduce``) until a hardware A/B on an 8-GPU node says otherwise; bench.py
    --gpus N reports both in its ``collective`` object."""

    def __init__(self, numel: int, world: int, device, dtype=torch.float32):
        shard = (numel + world - 1) // world
        self.shard = (shard + 3) // 4 * 4
        self.numel, self.world = numel, world
        self.send = torch.zeros(world * self.shard, dtype=dtype, device=device)      # the tail beyond numel stays zero
        self.recv = torch.empty_like(self.send)
        self.mine = torch.empty(self.shard, dtype=dtype, device=device)

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        n, shard = self.numel, self.shard
        assert t.numel() == n and dist.get_world_size() == self.world
        self.send[:n].copy_(t.reshape(-1))
        dist.all_to_all_single(self.recv, self.send)
        self.mine.copy_(self.recv[:shard])
        for j in range(1, self.world):                  # fixed order
            self.mine.add_(self.recv[j * shard:(j + 1) * shard])
        dist.all_gather_into_tensor(self.recv, self.mine)
        t.reshape(-1).copy_(self.recv[:n])
        return t


_direct_cache = {}


def allreduce_sum_direct(t: torch.Tensor) -> torch.Tensor:
    """``DirectAllReduce`` with buffers cached per (size, world, device): in-place sum of ``t`` over the ranks."""
    key = (t.numel(), dist.get_world_size(), str(t.device), t.dtype)
    ar = _direct_cache.get(key)
    if ar is None:
        ar = _direct_cache[key] = DirectAllReduce(t.numel(), dist.get_world_size(), t.device, t.dtype)
    return ar(t)


def allreduce_mode() -> str:
    """``ring`` (default) = one ``dist.all_reduce`` per bucket piece: the collective the north star names, on RCCL's own
    algorithm choice; ``direct`` (KM_ALLREDUCE=direct, world > 2) = DirectAllReduce."""
    return os.environ.get("KM_ALLREDUCE", "ring")


def _sum_over_ranks(t: torch.Tensor) -> None:
    if allreduce_mode() == "direct" and dist.get_world_size() > 2:
        allreduce_sum_direct(t)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def allreduce_gradients(flat_grad: torch.Tensor, average: bool = True, weight: Optional[float] = None) -> torch.Tensor:
    """The training step's single collective: sum of the flat gradient bucket over the ranks, in place.

    Every rank's loss kernel normalises by its LOCAL window count, so the bucket holds the gradient of a per-rank
    mean.  ``weight`` = n_local / n_global turns the sum into the gradient of the GLOBAL-batch mean whatever the
    shard sizes are (3/3/2 windows, the short last batch of a clip, a rank with no window at all: weight 0);
    without it the ranks are assumed to hold equal shares and the sum is divided by the world size."""
    if weight is not None:
        flat_grad.mul_(float(weight))
    if collectives_active():
        _sum_over_ranks(flat_grad)
        if average and weight is None:
            flat_grad.div_(dist.get_world_size())
    return flat_grad
