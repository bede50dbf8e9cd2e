"""Sequential (time-series aware) training -- mirror of the reference's ``src/train_sequential.py``.

``SequentialTrainer`` keeps the reference's shape: ``train_epoch()`` walks the windows of each clip in time order,
resets the model's temporal (EMA) state whenever the file changes (:136-155), runs forward, loss, backward, global-norm
clipping and AdamW (:157-181), steps the CosineAnnealingWarmRestarts schedule once per epoch (:209); ``validate()``,
``save_checkpoint()`` / ``load_checkpoint()`` with the reference's checkpoint keys (:303-339).  What differs is where the
work happens: batches come from the device-resident ``koemorph_amd.data.SequentialKoeMorphDataset`` and the whole step is
the HIP training step behind the C-ABI (``koemorph_amd.training.Trainer``).  Data parallel: one process per GPU
(``torchrun``), every rank takes a contiguous share of each batch and the flat gradient is all-reduced over RCCL.

The reference's ``MultiTaskLoss`` does not exist in its ``src/model/losses.py``; the loss here is ``KoeMorphLoss``
(mse + optional terms, src/model/losses.py:29-178) against the label of each window's last frame.  The 256-D emotion
vector is an input of the model: ``emotion_provider(audio) -> (B, 256)``; without one the reference's own failure
fallback is used (``randn * 0.1``, simplified_dual_stream_model.py:250-267), seeded per window for reproducibility.
No hydra / TensorBoard dependency: plain argparse, metrics to the log.
"""
from __future__ import annotations

import argparse
import logging
import time
from pathlib import Path
from typing import Callable, Dict, Optional

import numpy as np
import torch

from .. import parallel, synth
from ..data import SequentialKoeMorphDataset
from ..engine import Engine
from ..training import Trainer

logger = logging.getLogger(__name__)


class SequentialTrainer:
    def __init__(self, engine: Engine, train_data: SequentialKoeMorphDataset, val_data: Optional[SequentialKoeMorphDataset] = None,
                 device: str = "cuda", learning_rate: float = 1e-4, weight_decay: float = 1e-5, gradient_clip: float = 1.0,
                 mse_weight: float = 1.0, l1_weight: float = 0.0, extra_loss_terms: Optional[Dict[str, float]] = None,
                 emotion_provider: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, dropout: float = 0.1, seed: int = 0):
        """``dropout``: the reference trains under ``model.train()`` (src/train_sequential.py:118) on a model built with
        dropout 0.1 (simplified_dual_stream_model.py:155): the attention weights of both streams and the decoder's hidden
        layer are dropped per step; ``validate()`` runs the eval-mode inference kernels.  Every rank draws its own masks
        (generator seed = ``seed`` + rank); the generator's step counter is part of the checkpoint."""
        self.engine, self.device = engine, torch.device(device)
        self.train_data, self.val_data = train_data, val_data
        self.emotion_provider = emotion_provider
        self.rank, self.world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) \
            if torch.distributed.is_initialized() else (0, 1)
        self.trainer = Trainer(engine, max_windows=train_data.batch_size, lr=learning_rate, weight_decay=weight_decay,
                               grad_clip=gradient_clip, mse_weight=mse_weight, l1_weight=l1_weight, use_smoothing=True,
                               dropout=dropout, seed=seed + self.rank)
        if extra_loss_terms:
            self.trainer.set_loss_terms(**extra_loss_terms)
        self.epoch = 0
        self.global_step = 0
        self.best_val_loss = float("inf")
        self.current_file_idx = None
        self._shapes = {k: tuple(v.shape) for k, v in engine.state_dict_shapes().items()}

    # ---- helpers ----------------------------------------------------------------------------------------
    def _emotion(self, batch) -> torch.Tensor:
        if self.emotion_provider is not None:
            return self.emotion_provider(batch["audio"]).to(self.device, torch.float32)
        rows = [0.1 * synth.normal(1000003 * int(f) + int(w), (256,)) for f, w in zip(batch["file_indices"], batch["window_indices"])]
        return torch.from_numpy(np.stack(rows)).to(self.device)

    def _my_share(self, batch):
        """This rank's contiguous share of the batch (windows shard embarrassingly; only the gradient is reduced)."""
        B = batch["audio"].shape[0]
        lo, hi = parallel.shard_range(B, self.rank, self.world)
        return {k: (v[lo:hi] if isinstance(v, (torch.Tensor, list)) else v) for k, v in batch.items()}, hi - lo

    # ---- reference API ----------------------------------------------------------------------------------
    def train_epoch(self) -> Dict[str, float]:
        total, n = 0.0, 0
        t0 = time.time()
        for batch in self.train_data:
            file_idx = int(batch["file_indices"][0])
            if self.current_file_idx != file_idx:            # new clip: the EMA state must not leak across files
                self.current_file_idx = file_idx
                self.trainer.reset_temporal_state()
            share, nb = self._my_share(batch)
            B_global = batch["audio"].shape[0]
            if nb == 0:                                      # fewer windows than ranks: weight 0 in the global mean
                self.trainer.flat_grad.zero_()
                self.trainer.optimizer_step(weight=0.0)
                continue
            # every rank's gradient is weighted by its share of the GLOBAL batch (shares differ by one window when the
            # batch does not divide, and the last batch of a clip is short): the sum is the full-batch gradient
            loss = self.trainer.step(share["audio"], self._emotion(share), share["target"], global_batch=B_global)
            total += float(loss.item())
            n += 1
            self.global_step += 1
        self.trainer.end_epoch()                             # CosineAnnealingWarmRestarts(T_0=10, T_mult=2, eta_min=1e-6)
        self.epoch += 1
        return {"total": total / max(n, 1), "batches": n, "lr": self.trainer.lr, "seconds": time.time() - t0}

    def validate(self) -> Dict[str, float]:
        if self.val_data is None:
            return {}
        self.trainer.sync_inference_weights()
        total, n = 0.0, 0
        state = None
        current = None
        with torch.no_grad():
            for batch in self.val_data:
                file_idx = int(batch["file_indices"][0])
                B = batch["audio"].shape[0]
                first = current != file_idx or state is None or state.shape[0] != B
                if first:
                    current, state = file_idx, torch.zeros(B, 52, device=self.device)
                pred = self.engine.forward_audio(batch["audio"], self._emotion(batch), state=state, first=first)
                total += float(torch.nn.functional.mse_loss(pred, batch["target"]).item())
                n += 1
        return {"total": total / max(n, 1), "batches": n}

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """The reference model's state dict (keys of SimplifiedDualStreamModel: dual_stream_attention.* + smoothing_alpha)."""
        p = self.trainer.params(self._shapes)
        out = {}
        for k, v in p.items():
            out[k if k == "smoothing_alpha" else "dual_stream_attention." + k] = torch.from_numpy(np.asarray(v))
        return out

    def save_checkpoint(self, path, is_best: bool = False):
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        ckpt = {"epoch": self.epoch, "global_step": self.global_step, "model_state_dict": self.state_dict(),
                "best_val_loss": self.best_val_loss, "optimizer_state_dict": self.trainer.optimizer_state(),
                "current_file_idx": -1 if self.current_file_idx is None else int(self.current_file_idx),
                "model_config": {"d_model": self.engine.d_model, "num_heads": self.engine.num_heads,
                                 "mel_sequence_length": self.engine.mel_sequence_length}}
        if self.rank == 0:
            torch.save(ckpt, path)
            if is_best:
                torch.save(ckpt, path.parent / "best_model.pth")

    def load_checkpoint(self, path):
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        sd = {k.replace("dual_stream_attention.", "", 1): v.numpy() for k, v in ckpt["model_state_dict"].items()}
        self.trainer.load_params(sd)
        self.epoch = int(ckpt["epoch"]); self.global_step = int(ckpt["global_step"])
        self.best_val_loss = float(ckpt.get("best_val_loss", float("inf")))
        try:
            self.trainer.load_optimizer_state(ckpt["optimizer_state_dict"])
        except (ValueError, KeyError, TypeError, AttributeError) as exc:
            # a checkpoint of the reference's own trainer (optimizer_state_dict = a torch.optim dict keyed by parameter index,
            # src/train_sequential.py:303-339) or of a build that predates the per-key layout: the weights above are what
            # matters, the moments restart -- the same thing torch users do with load_state_dict(strict=False) on a new optimizer
            import warnings
            warnings.warn(f"optimizer state of {path} not loaded ({exc}); resuming from the model weights with fresh AdamW moments",
                          RuntimeWarning, stacklevel=2)
            self.trainer.restart_optimizer(self.epoch)
        cf = int(ckpt.get("current_file_idx", -1))
        self.current_file_idx = None if cf < 0 else cf


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Sequential training of the dual-stream KoeMorph model on MI355X")
    p.add_argument("--data_dir", required=True, help="directory with *.wav + *.jsonl pairs")
    p.add_argument("--val_dir", help="validation directory (optional)")
    p.add_argument("--epochs", type=int, default=10)
    p.add_argument("--batch_size", type=int, default=8, help="windows per step over ALL ranks")
    p.add_argument("--window_frames", type=int, default=256)
    p.add_argument("--stride_frames", type=int, default=1)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--weight_decay", type=float, default=1e-5)
    p.add_argument("--gradient_clip", type=float, default=1.0)
    p.add_argument("--l1_weight", type=float, default=0.0)
    p.add_argument("--dropout", type=float, default=0.1, help="train-mode dropout probability (the reference's model: 0.1; 0 = eval-mode arithmetic)")
    p.add_argument("--seed", type=int, default=0, help="dropout generator seed (rank r uses seed + r)")
    p.add_argument("--checkpoint_dir", default="checkpoints")
    p.add_argument("--resume", help="checkpoint to resume from")
    p.add_argument("--max_files", type=int)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    rank, world, local = parallel.init_from_env()
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    eng = Engine(mel_sequence_length=args.window_frames)
    eng.load_state_dict(synth.make_core_params(0, T=args.window_frames, style="init"))
    eng.finalize(device)
    kw = dict(window_frames=args.window_frames, stride_frames=args.stride_frames, shuffle_files=False, loop_dataset=False,
              batch_size=args.batch_size, device=device, max_files=args.max_files)
    train = SequentialKoeMorphDataset(args.data_dir, **kw)
    val = SequentialKoeMorphDataset(args.val_dir, **kw) if args.val_dir else None
    st = SequentialTrainer(eng, train, val, device=device, learning_rate=args.learning_rate, weight_decay=args.weight_decay,
                           gradient_clip=args.gradient_clip, l1_weight=args.l1_weight, dropout=args.dropout, seed=args.seed)
    if args.resume:
        st.load_checkpoint(args.resume)
    for _ in range(st.epoch, args.epochs):
        m = st.train_epoch()
        v = st.validate()
        is_best = bool(v) and v["total"] < st.best_val_loss
        if is_best:
            st.best_val_loss = v["total"]
        if rank == 0:
            logger.info(f"epoch {st.epoch}: train {m['total']:.6f} ({m['batches']} steps, {m['seconds']:.1f} s, lr {m['lr']:.2e})"
                        + (f", val {v['total']:.6f}" if v else ""))
        st.save_checkpoint(Path(args.checkpoint_dir) / f"checkpoint_epoch_{st.epoch}.pth", is_best=is_best)


if __name__ == "__main__":
    main()
