"""Device-resident mirror of the reference's ``AdaptiveSequentialDataset`` (src/data/adaptive_sequential_dataset.py:21-343).

Same constructor arguments, stride schedule (dense / sparse / progressive / mixed, :111-126), ``set_epoch`` (:128-132),
alignment rule (:224-229), window order and per-window keys (:156-209: ``audio``, ``blendshapes``, ``file_indices``,
``window_indices``, ``start_frames``, ``file_names``, ``is_dense``).  The stride schedule is host logic; the windows
themselves are gathered in HBM by start frame from clips uploaded once (km_gather_windows), exactly like
``SequentialKoeMorphDataset``.  Differences from the reference that follow from that: batches are assembled here
(``batch_size`` windows of ONE clip per batch; the reference leaves batching to a DataLoader, :328-336), and each batch
also carries ``target`` (B, 52), the label row of every window's last frame.  Labels are NOT resampled (the reference's
adaptive loader reads the JSONL rows as they are, :144-154).
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Dict, Iterator, List, Optional, Tuple, Union

import numpy as np
import torch

from .._lib import check, load
from .sequential_dataset import _Clip, _load_wav, load_jsonl_labels

logger = logging.getLogger(__name__)

STRIDE_MODES = ("dense", "sparse", "progressive", "mixed")


def calculate_stride(stride_mode: str, initial_stride: int, final_stride: int, epoch: int, max_epochs: int) -> int:
    """_calculate_stride (:111-126)."""
    if stride_mode == "dense":
        return 1
    if stride_mode == "sparse":
        return initial_stride
    if stride_mode == "progressive":
        progress = min(1.0, epoch / max(1, max_epochs - 1))             # linear decrease over the epochs
        stride = int(initial_stride - progress * (initial_stride - final_stride))
        return max(final_stride, stride)
    if stride_mode == "mixed":
        return initial_stride                                            # the dense part is drawn per clip
    raise ValueError(f"Unknown stride mode: {stride_mode}")


def window_plan(n_frames: int, window_frames: int, stride_mode: str, current_stride: int, initial_stride: int,
                dense_sampling_ratio: float = 0.1) -> List[Tuple[int, int, bool]]:
    """(window index, start frame, is_dense) of every window of one clip in the reference's order
    (_dense_windows :156-180, _sparse_windows :182-209, the mixed branch of _process_file_pair :240-278; the dense
    indices of mixed mode come from ``np.random.choice`` on numpy's global generator, as in the reference)."""
    plan: List[Tuple[int, int, bool]] = []
    if n_frames < window_frames:
        return plan

    def sparse(stride):
        out = []
        for i in range((n_frames - window_frames) // stride + 1):
            if i * stride + window_frames > n_frames:
                break
            out.append((i, i * stride, False))
        return out

    if stride_mode == "dense":
        plan = [(i, i, True) for i in range(n_frames - window_frames + 1)]
    elif stride_mode in ("sparse", "progressive"):
        plan = sparse(current_stride)
    elif stride_mode == "mixed":
        dense_samples = int((n_frames - window_frames) * dense_sampling_ratio)
        dense_indices = np.random.choice(n_frames - window_frames, size=dense_samples, replace=False) \
            if n_frames > window_frames else np.zeros(0, np.int64)
        plan = [(int(i), int(i), True) for i in sorted(dense_indices)] + sparse(initial_stride)
    else:
        raise ValueError(f"Unknown stride mode: {stride_mode}")
    return plan


class AdaptiveSequentialDataset:
    def __init__(self, data_dir: Union[str, Path], window_frames: int = 256, stride_mode: str = "progressive",
                 initial_stride: int = 32, final_stride: int = 1, epoch: int = 0, max_epochs: int = 100,
                 sample_rate: int = 16000, target_fps: int = 30, dense_sampling_ratio: float = 0.1,
                 shuffle_files: bool = True, loop_dataset: bool = True, max_files: Optional[int] = None,
                 batch_size: int = 4, device: Union[str, torch.device] = "cuda"):
        self.data_dir = Path(data_dir)
        self.window_frames, self.stride_mode = window_frames, stride_mode
        self.initial_stride, self.final_stride = initial_stride, final_stride
        self.epoch, self.max_epochs = epoch, max_epochs
        self.sample_rate, self.target_fps = sample_rate, target_fps
        self.dense_sampling_ratio = dense_sampling_ratio
        self.shuffle_files, self.loop_dataset = shuffle_files, loop_dataset
        self.hop_length = int(sample_rate / target_fps)
        self.window_samples = window_frames * self.hop_length
        self.batch_size = batch_size
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("AdaptiveSequentialDataset keeps clips in GPU memory: device must be a HIP device")
        self.file_pairs = self._find_file_pairs()
        if max_files:
            self.file_pairs = self.file_pairs[:max_files]
        if len(self.file_pairs) == 0:
            raise ValueError(f"No valid audio/JSONL pairs found in {data_dir}")
        self.current_stride = self._calculate_stride()
        self._lib = load()
        self._clips: Dict[int, _Clip] = {}

    def _find_file_pairs(self) -> List[Tuple[Path, Path]]:
        pairs = []
        for audio_path in self.data_dir.glob("**/*.wav"):
            jsonl_path = audio_path.with_suffix(".jsonl")
            if jsonl_path.exists():
                pairs.append((audio_path, jsonl_path))
        return sorted(pairs)

    def _calculate_stride(self) -> int:
        return calculate_stride(self.stride_mode, self.initial_stride, self.final_stride, self.epoch, self.max_epochs)

    def set_epoch(self, epoch: int):
        self.epoch = epoch
        self.current_stride = self._calculate_stride()
        logger.info(f"Epoch {epoch}: stride updated to {self.current_stride}")

    # ---- one clip on the device (reference _process_file_pair :217-229) -----------------------------------
    def clip(self, file_idx: int) -> _Clip:
        if file_idx not in self._clips:
            audio_path, jsonl_path = self.file_pairs[file_idx]
            audio = torch.from_numpy(_load_wav(audio_path, self.sample_rate)).to(self.device)
            rows, _ = load_jsonl_labels(jsonl_path)
            labels = torch.from_numpy(rows).to(self.device)
            expected_frames = audio.shape[0] // self.hop_length
            if abs(labels.shape[0] - expected_frames) > 1:
                num_frames = min(labels.shape[0], expected_frames)
                audio = audio[:num_frames * self.hop_length].contiguous()
                labels = labels[:num_frames].contiguous()
            self._clips[file_idx] = _Clip(audio, labels, 0, audio_path.stem)
        return self._clips[file_idx]

    def plan(self, file_idx: int) -> List[Tuple[int, int, bool]]:
        """Windows of one clip under the current mode / stride; a window is dropped when its audio slice would be short
        (:175, :201, :262 -- labels may run one frame past the audio under the |diff| <= 1 alignment rule)."""
        c = self.clip(file_idx)
        plan = window_plan(int(c.labels.shape[0]), self.window_frames, self.stride_mode, self.current_stride,
                           self.initial_stride, self.dense_sampling_ratio)
        n_audio = int(c.audio.shape[0])
        return [(i, s, d) for i, s, d in plan if (s + self.window_frames) * self.hop_length <= n_audio]

    def gather(self, file_idx: int, windows: List[Tuple[int, int, bool]]) -> Dict[str, object]:
        c = self.clip(file_idx)
        B = len(windows)
        starts = torch.tensor([s for _, s, _ in windows], dtype=torch.int32).to(self.device)
        audio = torch.empty(B, self.window_samples, device=self.device)
        bs = torch.empty(B, self.window_frames, c.labels.shape[1], device=self.device)
        target = torch.empty(B, c.labels.shape[1], device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.km_gather_windows(c.audio.data_ptr(), c.audio.shape[0], starts.data_ptr(), B, self.hop_length,
                                              self.window_samples, audio.data_ptr(), c.labels.data_ptr(), c.labels.shape[0],
                                              self.window_frames, c.labels.shape[1], bs.data_ptr(), target.data_ptr(),
                                              torch.cuda.current_stream(self.device).cuda_stream))
        return {"audio": audio, "blendshapes": bs, "target": target,
                "file_indices": torch.full((B,), file_idx, dtype=torch.int64),
                "window_indices": torch.tensor([i for i, _, _ in windows], dtype=torch.int64),
                "start_frames": torch.tensor([s for _, s, _ in windows], dtype=torch.int64),
                "file_names": [c.name] * B,
                "is_dense": torch.tensor([d for _, _, d in windows], dtype=torch.bool)}

    def __iter__(self) -> Iterator[Dict[str, object]]:
        while True:
            order = list(range(len(self.file_pairs)))
            if self.shuffle_files:
                order = torch.randperm(len(order)).tolist()
            for fi in order:
                plan = self.plan(fi)
                for w0 in range(0, len(plan), self.batch_size):
                    yield self.gather(fi, plan[w0:w0 + self.batch_size])
            if not self.loop_dataset:
                break

    def estimate_epoch_size(self) -> int:
        """:296-317 (the reference's rough estimate from an assumed 300-frame clip)."""
        avg_file_frames = 300
        if self.stride_mode == "dense":
            return len(self.file_pairs) * max(0, avg_file_frames - self.window_frames + 1)
        if self.stride_mode in ("sparse", "progressive"):
            return len(self.file_pairs) * max(0, (avg_file_frames - self.window_frames) // self.current_stride + 1)
        if self.stride_mode == "mixed":
            dense_windows = int((avg_file_frames - self.window_frames) * self.dense_sampling_ratio)
            sparse_windows = (avg_file_frames - self.window_frames) // self.initial_stride + 1
            return len(self.file_pairs) * (dense_windows + sparse_windows)
        return 0


def create_adaptive_dataloader(data_dir: Union[str, Path], batch_size: int = 4, stride_mode: str = "progressive",
                               epoch: int = 0, max_epochs: int = 100, num_workers: int = 2, **kwargs) -> AdaptiveSequentialDataset:
    """:320-343.  The device-resident dataset batches by itself, so it IS the loader (num_workers has no meaning when the
    clips live in HBM and is accepted for signature compatibility)."""
    return AdaptiveSequentialDataset(data_dir=data_dir, stride_mode=stride_mode, epoch=epoch, max_epochs=max_epochs,
                                     batch_size=batch_size, **kwargs)
