"""Data-parallel training step of the dual-stream model on the GPU.

Mirrors the step of the reference's ``SequentialTrainer.train_epoch`` (src/train_sequential.py:130-181) and its
optimizer / scheduler set-up (:73-86): ``AdamW(lr, weight_decay, betas=(0.9, 0.999))``,
``CosineAnnealingWarmRestarts(T_0=10, T_mult=2, eta_min=1e-6)`` stepped per epoch (:209),
``clip_grad_norm_(params, 1.0)`` (:175-179).  The reference is single-process; the data-parallel part is new
construction: one process per GPU, every rank runs the HIP forward/backward on its own windows, ONE all-reduce
of the flat fp32 gradient bucket over RCCL (koemorph_amd.parallel.allreduce_gradients), then the fused AdamW.

Loss: the reference's intended ``MultiTaskLoss`` does not exist in its repository (:24-29 import error); the
nearest real criterion is ``KoeMorphLoss`` (src/model/losses.py:29) whose frame-local terms are implemented here:
``mse_weight * MSE + l1_weight * L1`` (:112-121).  Targets are one (B, 52) frame per window (the reference's
(B, 256, 52) targets against a (B, 52) prediction are shape-inconsistent).
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import parallel
from ._lib import check
from .engine import Engine, _ptr, _stream_ptr


def cosine_warm_restarts_lr(epoch: int, base_lr: float, T_0: int = 10, T_mult: int = 2, eta_min: float = 1e-6) -> float:
    """torch.optim.lr_scheduler.CosineAnnealingWarmRestarts, evaluated at an integer epoch."""
    if T_mult == 1:
        T_cur, T_i = epoch % T_0, T_0
    else:
        n = int(math.log(epoch / T_0 * (T_mult - 1) + 1, T_mult)) if epoch >= T_0 else 0
        T_cur = epoch - T_0 * (T_mult ** n - 1) // (T_mult - 1)
        T_i = T_0 * T_mult ** n
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * T_cur / T_i)) / 2


class Trainer:
    def __init__(self, engine: Engine, max_windows: int = 8, lr: float = 1e-4, weight_decay: float = 1e-5,
                 betas=(0.9, 0.999), eps: float = 1e-8, grad_clip: float = 1.0, mse_weight: float = 1.0,
                 l1_weight: float = 0.0, use_smoothing: bool = True, dropout: float = 0.0, seed: int = 0):
        """``dropout``: training-mode dropout probability of the two attention modules and the decoder (the reference
        trains at 0.1 under model.train(), src/train_sequential.py:118); 0 = eval-mode arithmetic."""
        if engine.device is None:
            raise RuntimeError("finalize the Engine first")
        self.engine, self.device = engine, engine.device
        self._lib, self._h = engine._lib, engine._h
        self.base_lr = self.lr = lr
        self.weight_decay, self.betas, self.eps, self.grad_clip = weight_decay, betas, eps, grad_clip
        self.mse_weight, self.l1_weight, self.use_smoothing = mse_weight, l1_weight, use_smoothing
        self.max_windows = max_windows
        with torch.cuda.device(self.device):
            check(self._lib.km_train_init(self._h, max_windows, _stream_ptr(self.device)))
        self.dropout = float(dropout)
        if self.dropout > 0:
            self.set_dropout(self.dropout, seed)
        self.n_params = int(self._lib.km_train_num_params(self._h))
        self.flat_grad = torch.zeros(self.n_params, device=self.device)
        self.loss = torch.zeros(1, device=self.device)
        self.out = torch.zeros(max_windows, 52, device=self.device)
        self.ema_state = torch.zeros(max_windows, 52, device=self.device)
        self._ema_batch: Optional[int] = None
        self._side = None                 # side stream of the overlapped gradient all-reduce
        self._last_was_step = False       # the bucket was just produced by km_train_step* on this stream (its early event is fresh)
        self.step_count = 0
        self.epoch = 0

    # ---- introspection --------------------------------------------------------------------------
    def offset(self, key: str) -> int:
        off = int(self._lib.km_train_param_offset(self._h, key.encode()))
        if off < 0:
            raise KeyError(key)
        return off

    def grads(self, shapes: Dict[str, tuple]) -> Dict[str, np.ndarray]:
        flat = self.flat_grad.cpu().numpy()
        return {k: flat[self.offset(k):self.offset(k) + int(np.prod(s, dtype=np.int64))].reshape(s).copy()
                for k, s in shapes.items()}

    def params(self, shapes: Dict[str, tuple]) -> Dict[str, np.ndarray]:
        flat = np.empty(self.n_params, np.float32)
        check(self._lib.km_train_get_params(self._h, flat.ctypes.data, self.n_params))
        return {k: flat[self.offset(k):self.offset(k) + int(np.prod(s, dtype=np.int64))].reshape(s).copy()
                for k, s in shapes.items()}

    def load_params(self, state: Dict[str, np.ndarray]) -> None:
        """Overwrite the fp32 master weights (e.g. from a checkpoint); moments and step counters are kept."""
        flat = np.empty(self.n_params, np.float32)
        check(self._lib.km_train_get_params(self._h, flat.ctypes.data, self.n_params))
        for k, v in state.items():
            a = np.asarray(v, np.float32).ravel()
            off = self.offset(k)
            flat[off:off + a.size] = a
        check(self._lib.km_train_set_params(self._h, flat.ctypes.data, self.n_params))

    def _param_table(self):
        """(key, offset, size) of every trainable tensor in the flat bucket (whose order is an implementation detail of
        the library: tensors the backward pass finishes last come last)."""
        shapes = {k: tuple(v.shape) for k, v in self.engine.state_dict_shapes().items()}
        return [(k, self.offset(k), int(np.prod(s, dtype=np.int64)) if len(s) else 1) for k, s in shapes.items()]

    def optimizer_state(self) -> Dict[str, object]:
        """AdamW moments PER STATE-DICT KEY + step counters (host copies) for a checkpoint.  Keyed, not flat: the order of
        the tensors inside the library's bucket may change between versions, a checkpoint must not depend on it."""
        m, v = np.empty(self.n_params, np.float32), np.empty(self.n_params, np.float32)
        steps = np.zeros(2, np.int32)
        check(self._lib.km_train_get_optimizer_state(self._h, m.ctypes.data, v.ctypes.data, self.n_params, steps.ctypes.data))
        import ctypes
        drop_step = ctypes.c_int64(0)
        check(self._lib.km_train_get_dropout_step(self._h, ctypes.byref(drop_step)))
        tab = self._param_table()
        return {"layout": "per-key-v1",
                "exp_avg": {k: torch.from_numpy(m[o:o + n].copy()) for k, o, n in tab},
                "exp_avg_sq": {k: torch.from_numpy(v[o:o + n].copy()) for k, o, n in tab},
                "steps": torch.from_numpy(steps), "step_count": self.step_count, "epoch": self.epoch,
                "ema_state": self.ema_state.cpu(), "ema_batch": -1 if self._ema_batch is None else int(self._ema_batch),
                "dropout_step": int(drop_step.value)}

    def load_optimizer_state(self, st: Dict[str, object]) -> None:
        if st.get("layout") != "per-key-v1":
            raise ValueError("optimizer state without a per-key layout (written before the bucket was reordered): the flat "
                             "moments cannot be assigned to tensors safely -- resume from the model weights only")
        m, v = np.zeros(self.n_params, np.float32), np.zeros(self.n_params, np.float32)
        for k, o, n in self._param_table():
            a, b = st["exp_avg"][k].numpy().ravel(), st["exp_avg_sq"][k].numpy().ravel()
            if a.size != n or b.size != n:
                raise ValueError(f"optimizer state of {k}: {a.size} values, the model has {n}")
            m[o:o + n] = a; v[o:o + n] = b
        steps = np.ascontiguousarray(st["steps"].numpy(), np.int32)
        check(self._lib.km_train_set_optimizer_state(self._h, m.ctypes.data, v.ctypes.data, self.n_params, steps.ctypes.data))
        self.step_count = int(st["step_count"]); self.epoch = int(st["epoch"])
        self.lr = cosine_warm_restarts_lr(self.epoch, self.base_lr)
        self.ema_state.copy_(st["ema_state"].to(self.device))
        self._ema_batch = None if int(st["ema_batch"]) < 0 else int(st["ema_batch"])
        if "dropout_step" in st:
            check(self._lib.km_train_set_dropout_step(self._h, int(st["dropout_step"])))

    def restart_optimizer(self, epoch: int = 0) -> None:
        """Fresh AdamW moments and step counters with the schedule positioned at ``epoch``: what a resume falls back to when
        the checkpoint's optimizer state cannot be assigned to tensors (a foreign ``torch.optim`` dict, a pre-per-key layout)."""
        z = np.zeros(self.n_params, np.float32)
        steps = np.zeros(2, np.int32)
        check(self._lib.km_train_set_optimizer_state(self._h, z.ctypes.data, z.ctypes.data, self.n_params, steps.ctypes.data))
        self.step_count = 0
        self.epoch = int(epoch)
        self.lr = cosine_warm_restarts_lr(self.epoch, self.base_lr)
        self.ema_state.zero_()
        self._ema_batch = None

    def reset_temporal_state(self):
        self._ema_batch = None

    # ---- training-mode dropout -------------------------------------------------------------------
    def set_dropout(self, p: float, seed: int = 0, external_masks: bool = False) -> None:
        """km_train_set_dropout: Philox masks per step (keyed by ``seed`` and a device-side step counter), or the masks
        last given to ``set_dropout_masks`` when ``external_masks``."""
        check(self._lib.km_train_set_dropout(self._h, float(p), int(seed) & (2 ** 64 - 1), 1 if external_masks else 0))
        self.dropout = float(p)

    def _mask_shapes(self, B: int):
        e = self.engine
        return (B, e.num_heads, 28, e.n_mels), (B, e.num_heads, 24, 1), (B, 52, e.d_model // 2)

    def dropout_masks(self, B: int) -> Dict[str, np.ndarray]:
        """Keep masks of the most recent step as {"mel", "emo", "dec"} boolean arrays (oracle.core.core_forward layout)."""
        shapes = self._mask_shapes(B)
        bufs = [np.empty(s, np.uint8) for s in shapes]
        with torch.cuda.device(self.device):
            check(self._lib.km_train_get_dropout_masks(self._h, B, bufs[0].ctypes.data, bufs[1].ctypes.data, bufs[2].ctypes.data,
                                                       _stream_ptr(self.device)))
        return {k: b.astype(bool) for k, b in zip(("mel", "emo", "dec"), bufs)}

    def set_dropout_masks(self, masks: Dict[str, np.ndarray]) -> None:
        B = int(masks["mel"].shape[0])
        bufs = [np.ascontiguousarray(np.asarray(masks[k]).reshape(s), np.uint8) for k, s in zip(("mel", "emo", "dec"), self._mask_shapes(B))]
        with torch.cuda.device(self.device):
            check(self._lib.km_train_set_dropout_masks(self._h, B, bufs[0].ctypes.data, bufs[1].ctypes.data, bufs[2].ctypes.data,
                                                       _stream_ptr(self.device)))

    def set_loss_terms(self, perceptual_weight: float = 0.0, temporal_weight: float = 0.0, sparsity_weight: float = 0.0,
                       smoothness_weight: float = 0.0, landmark_weight: float = 0.0, velocity_weight: float = 0.0,
                       prev_pred: Optional[torch.Tensor] = None, prev_target: Optional[torch.Tensor] = None,
                       landmark_weights: Optional[torch.Tensor] = None, audio_features: Optional[torch.Tensor] = None,
                       ds_velocity_weight: float = 0.0, ds_separation_weight: float = 0.0,
                       ds_prev_pred: Optional[torch.Tensor] = None) -> None:
        """The remaining terms of the reference's KoeMorphLoss (src/model/losses.py:29-178), added to mse/l1.
        prev_pred / prev_target (B,52) and landmark_weights (136,52) are device tensors the trainer keeps alive;
        call with no arguments to switch the extra terms off.
        ``ds_*``: the velocity and stream-separation terms of ``DualStreamLoss`` (src/train_dual_stream.py:434-516; its
        L1 / L2 terms are ``l1_weight`` / ``mse_weight`` of the trainer: 1.0 / 0.1 there); ``ds_prev_pred`` (B,52) = the
        previous step's predictions (a constant)."""
        from ._lib import KM_ABI_VERSION, KMLossConfig
        keep = []
        def dev(t, shape_tail):
            if t is None:
                return None
            t = t.to(self.device, torch.float32).contiguous()
            assert tuple(t.shape[-len(shape_tail):]) == shape_tail, (t.shape, shape_tail)
            keep.append(t)
            return _ptr(t)
        energy = None
        if audio_features is not None:     # audio-visual term of the perceptual loss (losses.py:340-378): per-window energy
            af = audio_features.to(self.device, torch.float32).contiguous()
            if af.dim() == 2:
                af = af.unsqueeze(1)
            energy = torch.empty(af.shape[0], device=self.device)
            with torch.cuda.device(self.device):
                check(self._lib.km_audio_energy(_ptr(af), af.shape[0], af.shape[1], af.shape[2], _ptr(energy), _stream_ptr(self.device)))
            keep.append(energy)
        cfg = KMLossConfig(KM_ABI_VERSION, perceptual_weight, temporal_weight, sparsity_weight, smoothness_weight, landmark_weight,
                           velocity_weight, dev(prev_pred, (52,)), dev(prev_target, (52,)), dev(landmark_weights, (136, 52)),
                           None if energy is None else _ptr(energy), ds_velocity_weight, ds_separation_weight,
                           dev(ds_prev_pred, (52,)))
        self._loss_tensors = keep
        import ctypes
        check(self._lib.km_train_set_loss(self._h, ctypes.byref(cfg)))

    # ---- one optimisation step ------------------------------------------------------------------
    def _ema_args(self, B):
        if not self.use_smoothing:
            return 0, 1
        first = self._ema_batch != B          # first call or batch-size change (simplified_dual_stream_model.py:357-359)
        self._ema_batch = B
        return _ptr(self.ema_state), (1 if first else 0)

    def forward_backward_mel(self, mel, mel_short, emotion, target):
        B, T_in, _ = mel.shape
        st, first = self._ema_args(B)
        check(self._lib.km_train_step(self._h, _ptr(mel.contiguous()), B, T_in, _ptr(mel_short.contiguous()),
                                      _ptr(emotion.contiguous()), _ptr(target.contiguous()), self.mse_weight,
                                      self.l1_weight, _ptr(self.flat_grad), _ptr(self.loss), _ptr(self.out), st, first,
                                      _stream_ptr(self.device)))
        self._last_was_step = True
        return self.loss

    def forward_backward(self, audio, emotion, target):
        B, L = audio.shape
        self.engine.reserve(B, L)
        st, first = self._ema_args(B)
        check(self._lib.km_train_step_audio(self._h, _ptr(audio.contiguous()), B, L, _ptr(emotion.contiguous()),
                                            _ptr(target.contiguous()), self.mse_weight, self.l1_weight,
                                            _ptr(self.flat_grad), _ptr(self.loss), _ptr(self.out), st, first,
                                            _stream_ptr(self.device)))
        self._last_was_step = True
        return self.loss

    def _allreduce_two_piece(self, weight: Optional[float], overlap: bool) -> None:
        """The step's one gradient exchange, issued as two pieces: floats [0, E) of the bucket (83 %: everything but the
        tensors the backward pass finishes last, km_train_grad_split) are reduced on a side stream that waits only for
        phase P11 of the program (km_train_wait_early), i.e. while the launch stream still computes the LayerNorm / channel
        encoder gradients; the rest follows on the launch stream when the step is done.  Every rank issues the same two
        collectives whether or not it ran a step (a rank without windows contributes zeros): only the placement differs."""
        import ctypes
        early = ctypes.c_int64(0)
        check(self._lib.km_train_grad_split(self._h, ctypes.byref(early)))
        E = int(early.value)
        # KM_ALLREDUCE_PIECES=1: the exchange as ONE collective behind the step (on a latency-bound 3.35 MB bucket the second
        # piece's full latency sits on the critical path; bench.py --gpus N times both forms in its `collective` object)
        if E <= 0 or E >= self.n_params or os.environ.get("KM_ALLREDUCE_PIECES", "2") == "1":
            parallel.allreduce_gradients(self.flat_grad, weight=weight)
            return
        if not overlap or os.environ.get("KM_ALLREDUCE_OVERLAP", "1") == "0":
            parallel.allreduce_gradients(self.flat_grad[:E], weight=weight)
            parallel.allreduce_gradients(self.flat_grad[E:], weight=weight)
            return
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        check(self._lib.km_train_wait_early(self._h, self._side.cuda_stream))
        with torch.cuda.stream(self._side):
            parallel.allreduce_gradients(self.flat_grad[:E], weight=weight)
        parallel.allreduce_gradients(self.flat_grad[E:], weight=weight)
        main.wait_stream(self._side)

    def optimizer_step(self, weight: Optional[float] = None):
        """All-reduce + clip + AdamW.  ``weight`` = this rank's windows / windows of the global batch when the ranks'
        shares differ (see parallel.allreduce_gradients); None = equal shares."""
        if parallel.collectives_active():
            self._allreduce_two_piece(weight, overlap=self._last_was_step)  # the ONE exchange of the training step
        elif weight is not None:
            self.flat_grad.mul_(float(weight))
        self._last_was_step = False
        self.step_count += 1
        check(self._lib.km_train_adamw(self._h, _ptr(self.flat_grad), self.lr, self.betas[0], self.betas[1], self.eps,
                                       self.weight_decay, self.grad_clip if self.grad_clip else 0.0, self.step_count,
                                       _stream_ptr(self.device)))

    def step(self, audio, emotion, target, global_batch: Optional[int] = None) -> torch.Tensor:
        """forward + loss + backward + all-reduce + clip + AdamW on this rank's windows; returns the loss (device).
        ``global_batch`` = windows of the whole batch over all ranks (None: every rank holds the same number)."""
        self.forward_backward(audio, emotion, target)
        self.optimizer_step(None if global_batch is None else audio.shape[0] / float(global_batch))
        return self.loss

    # ---- hipGraph replay of the ~60 launches of forward + backward -----------------------------------
    def capture(self, B: int, L: int) -> None:
        """Record ``forward_backward`` on static input buffers into a hipGraph.  Call after at least one eager
        step with the same batch size (so the in-forward EMA is past its first-call branch)."""
        dev = self.device
        self._g_audio = torch.zeros(B, L, device=dev)
        self._g_emo = torch.zeros(B, self.engine.emotion_dim, device=dev)
        self._g_target = torch.zeros(B, 52, device=dev)
        self.engine.reserve(B, L)
        if self.use_smoothing and self._ema_batch != B:
            raise RuntimeError("run one eager step with this batch size before capture()")
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.forward_backward(self._g_audio, self._g_emo, self._g_target)
        self._graph = g
        # The "early" event of the step was recorded on a CAPTURING stream: it is a graph node, not an event a side stream
        # can wait on.  Replayed steps therefore reduce the bucket on the launch stream (two pieces, no overlap).
        self._last_was_step = False

    def step_graph(self, audio, emotion, target, weight: Optional[float] = None) -> torch.Tensor:
        self._g_audio.copy_(audio, non_blocking=True)
        self._g_emo.copy_(emotion, non_blocking=True)
        self._g_target.copy_(target, non_blocking=True)
        self._graph.replay()
        self._last_was_step = False      # see capture(): no side-stream overlap behind a replay
        self.optimizer_step(weight)
        return self.loss

    def end_epoch(self):
        """scheduler.step() of the reference (:209)."""
        self.epoch += 1
        self.lr = cosine_warm_restarts_lr(self.epoch, self.base_lr)

    def sync_inference_weights(self):
        """Fold + pack the trained master weights so the inference kernels see them (km_train_sync)."""
        with torch.cuda.device(self.device):
            check(self._lib.km_train_sync(self._h, _stream_ptr(self.device)))
