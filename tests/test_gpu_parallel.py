"""Data-parallel training step on a real GPU: two processes (one share of the batch each, gloo all-reduce of the flat
gradient bucket -- on a node it is RCCL; the code path is the same `parallel.allreduce_gradients`) must take the same
optimisation step as one process on the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(B):
    from koemorph_amd import synth
    from koemorph_amd.engine import Engine
    from koemorph_amd.training import Trainer
    eng = Engine(); eng.load_state_dict(synth.make_core_params(3, style="trained")); eng.finalize("cuda:0")
    tr = Trainer(eng, max_windows=B, lr=1e-3, use_smoothing=False, l1_weight=0.1)
    audio = torch.from_numpy(synth.make_audio(5, 8, 136448)).cuda()
    emo = torch.from_numpy(synth.normal(6, (8, 256))).cuda()
    target = torch.from_numpy(synth.uniform(7, (8, 52), 0, 1)).cuda()
    return synth, eng, tr, audio, emo, target


def _worker(rank, world, port, q, n_global=8):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from koemorph_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    synth, eng, tr, audio, emo, target = _setup(4)
    lo, hi = parallel.shard_range(n_global, rank, world)
    losses = []
    for _ in range(3):
        if hi == lo:                                   # fewer windows than ranks (SequentialTrainer.train_epoch)
            tr.flat_grad.zero_()
            tr.optimizer_step(weight=0.0)
            losses.append(0.0)
            continue
        losses.append(float(tr.step(audio[lo:hi], emo[lo:hi], target[lo:hi], global_batch=n_global).item()))
    shapes = {k: v.shape for k, v in synth.make_core_params(3, style="trained").items()}
    q.put((rank, losses, {k: v for k, v in tr.params(shapes).items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    synth, eng, tr, audio, emo, target = _setup(8)
    ref_losses = [float(tr.step(audio, emo, target).item()) for _ in range(3)]
    shapes = {k: v.shape for k, v in synth.make_core_params(3, style="trained").items()}
    ref = tr.params(shapes)
    # both ranks hold identical weights after every step (same reduced gradient, same update)
    for k in ref:
        assert np.array_equal(res[0][2][k], res[1][2][k]), k
    # the mean of the two half-batch losses is the full-batch loss; the weights follow the single-process trajectory
    for s in range(3):
        assert abs(0.5 * (res[0][1][s] + res[1][1][s]) - ref_losses[s]) < 2e-6 * max(1.0, abs(ref_losses[s]))
    worst = max(float(np.abs(res[0][2][k] - ref[k]).max()) for k in ref)
    assert worst < 2e-5, worst          # 3 AdamW steps at lr 1e-3 move weights by ~3e-3; summation order differs between the runs


@pytest.mark.parametrize("n_global", [5, 1])
def test_two_rank_training_with_unequal_shares(n_global):
    """3 + 2 windows, and 1 + 0 windows: the reduced gradient must be the GLOBAL-batch mean (each rank's local-mean
    gradient weighted by n_local / n_global), not a mean of per-rank means."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, n_global)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    synth, eng, tr, audio, emo, target = _setup(8)
    for _ in range(3):
        tr.step(audio[:n_global], emo[:n_global], target[:n_global])
    shapes = {k: v.shape for k, v in synth.make_core_params(3, style="trained").items()}
    ref = tr.params(shapes)
    for k in ref:
        assert np.array_equal(res[0][2][k], res[1][2][k]), k
    worst = max(float(np.abs(res[0][2][k] - ref[k]).max()) for k in ref)
    assert worst < 2e-5, worst


def _worker_graph(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from koemorph_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    synth, eng, tr, audio, emo, target = _setup(4)
    lo, hi = parallel.shard_range(8, rank, world)
    tr.step(audio[lo:hi], emo[lo:hi], target[lo:hi], global_batch=8)        # one eager step, then the replayed ones
    tr.capture(hi - lo, audio.shape[1])
    assert tr._last_was_step is False                                        # no side-stream overlap behind a capture
    for _ in range(2):
        tr.step_graph(audio[lo:hi], emo[lo:hi], target[lo:hi], weight=(hi - lo) / 8.0)
    shapes = {k: v.shape for k, v in synth.make_core_params(3, style="trained").items()}
    q.put((rank, tr.params(shapes)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_graph_replay_matches_single_process():
    """capture() + step_graph() on two ranks: the gradient exchange behind a replayed step has to see the replay's
    gradients (the step's "early" event is recorded on a capturing stream, so nothing may wait on it): the weights follow
    the single-process eager trajectory."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_graph, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    synth, eng, tr, audio, emo, target = _setup(8)
    for _ in range(3):
        tr.step(audio, emo, target)
    shapes = {k: v.shape for k, v in synth.make_core_params(3, style="trained").items()}
    ref = tr.params(shapes)
    for k in ref:
        assert np.array_equal(res[0][1][k], res[1][1][k]), k
    worst = max(float(np.abs(res[0][1][k] - ref[k]).max()) for k in ref)
    assert worst < 2e-5, worst


def test_bench_two_rank_rehearsal():
    """bench.py's N > 1 path exactly as the driver invokes it -- plain `python bench.py --gpus 2`, NO torchrun: bench.py
    starts the two ranks itself (rendezvous, per-rank inputs, barrier + max-over-ranks clock, whole-job value, rank-0
    JSON).  The two ranks share the one card here, so the collectives run over gloo (KM_BENCH_BACKEND); the driver's
    runs use RCCL, one rank per GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["KM_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--cpu-seconds", "0"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 10 and d["warmup"] == 3 and d["scaling"] == "weak"
    B = d["config"]["windows_per_gpu"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 2 * B) < 0.01 * 2 * B      # whole-job frames per step = 2 ranks x B
    assert d["roofline"]["frac"] > 0 and d["vs_baseline"] is None
    col = d["collective"]                                    # the training step's gradient exchange, measured in the same run
    assert col["ranks"] == 2 and col["bytes"] == 4 * col["floats"] and col["floats"] > 837000
    for mode in ("ring", "direct"):
        assert col[mode]["ms_per_step"] > 0 and col[mode]["allreduce_ms"] > 0
    assert col["default"] == "ring"


def _seq_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    from koemorph_amd import synth
    from koemorph_amd.model import SequentialDualStreamModel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    params = synth.make_core_params(5, style="trained")
    res = {}
    for stride, L in ((1, 136448 + 533 * 9 + 100), (3, 136448 + 533 * 20)):
        audio = torch.from_numpy(synth.make_audio(70 + stride, 2, L)).cuda()
        emo = torch.from_numpy(synth.normal(80 + stride, (2, 256))).cuda()
        m = SequentialDualStreamModel(stride_frames=stride, shard_across_ranks=True).cuda().eval()
        sd = {"dual_stream_attention." + k: torch.from_numpy(v) for k, v in params.items()}
        sd["smoothing_alpha"] = torch.tensor(0.8)
        m.load_state_dict(sd)
        sharded = m(audio, emotion_features=emo)["blendshapes"]
        m.shard_across_ranks = False
        single = m(audio, emotion_features=emo)["blendshapes"]
        res[stride] = (tuple(sharded.shape), bool(torch.equal(sharded, single)))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sequence_mode_is_bit_identical_to_one_rank():
    """SURVEY.md section 8e, sequence mode: each rank computes a contiguous chunk of every clip's output frames (its samples
    + one window of halo, no smoothing), the chunks are gathered and the EMA runs once -- bit-identical to km_sequence_forward on
    one rank (10 / 7 output frames over two ranks, stride 1 and 3, a ragged clip end)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seq_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, res in out:
        assert res[1] == ((2, 10, 52), True), (rank, res)
        assert res[3] == ((2, 7, 52), True), (rank, res)


def _worker_rccl(q, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      KM_COLLECTIVES_AT_WORLD_1="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from koemorph_amd import parallel
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    assert parallel.collectives_active() and dist.get_backend() == "nccl"
    synth, eng, tr, audio, emo, target = _setup(8)
    losses = [float(tr.step(audio, emo, target).item()) for _ in range(3)]          # two-piece all-reduce, first piece on the side stream
    shapes = {k: v.shape for k, v in synth.make_core_params(3, style="trained").items()}
    params = tr.params(shapes)
    # the opt-in exchange (all_to_all_single + all_gather_into_tensor) on a bucket-sized device tensor
    t = torch.arange(tr.n_params, dtype=torch.float32, device="cuda")
    parallel.allreduce_sum_direct(t)
    direct_ok = bool(torch.equal(t, torch.arange(tr.n_params, dtype=torch.float32, device="cuda")))
    # sequence mode: chunk + all_gather + one EMA scan
    clip = torch.from_numpy(synth.make_audio(9, 1, 136448 + 40 * 533)).cuda()
    seq = parallel.sequence_apply(eng, clip, emo[:1], stride_frames=4).cpu().numpy()
    q.put((losses, params, direct_ok, seq))
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_rccl_group_runs_the_steps_collectives_on_the_gpu():
    """RCCL itself (backend "nccl"), on the one GPU of this box: a process group of ONE rank with KM_COLLECTIVES_AT_WORLD_1=1
    sends the training step's two-piece gradient all-reduce (first piece on the side stream behind km_train_wait_early), the
    direct all-to-all form and sequence mode's all_gather through the library on the step's own device tensors.  A sum over one
    rank changes nothing: losses, weights and frames must be the bits of a run without a process group."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl, args=(q, _free_port()))
    p.start()
    losses, params, direct_ok, seq = q.get(timeout=300)
    p.join(60)
    assert p.exitcode == 0
    assert direct_ok
    synth, eng, tr, audio, emo, target = _setup(8)
    ref_losses = [float(tr.step(audio, emo, target).item()) for _ in range(3)]
    assert losses == ref_losses
    ref = tr.params({k: v.shape for k, v in synth.make_core_params(3, style="trained").items()})
    for k in ref:
        assert np.array_equal(params[k], ref[k]), k
    clip = torch.from_numpy(synth.make_audio(9, 1, 136448 + 40 * 533)).cuda()
    ref_seq = eng.sequence_forward(clip, emo[:1], 4).cpu().numpy()
    assert np.array_equal(seq, ref_seq)
