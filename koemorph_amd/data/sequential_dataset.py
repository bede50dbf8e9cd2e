"""Device-resident mirror of the reference's ``SequentialKoeMorphDataset`` (src/data/sequential_dataset.py:22-243).

Same constructor arguments, file discovery, frame-rate detection, alignment rule and window order; the difference is
where the data lives.  The reference keeps clips on the host and, per window, slices 136 448 samples + 256 label rows and
ships them to the GPU (at stride 1 every sample crosses PCIe 256 times).  Here each clip and its labels are uploaded
ONCE, labels are resampled on the device (km_resample_labels: numpy's linspace + interp in float64, bit-identical), and
batches of windows are gathered in HBM by start frame (km_gather_windows).  Batches carry the reference's keys
(:199-206): audio (B, window_samples), blendshapes (B, window_frames, 52), file_indices, window_indices, start_frames,
file_names -- plus ``target`` (B, 52), the label row of each window's last frame, which is what a (B, 52) prediction is
trained against (the reference's (B, 256, 52) target is shape-inconsistent with its own model output).
"""
from __future__ import annotations

import json
import logging
from pathlib import Path
from typing import Dict, Iterator, List, Optional, Tuple, Union

import numpy as np
import torch

from .._lib import check, load

logger = logging.getLogger(__name__)


def load_jsonl_labels(jsonl_path: Union[str, Path]) -> Tuple[np.ndarray, List[float]]:
    """(frames (F, 52) float32, timestamps) from the reference's label format (src/data/io.py:119-131)."""
    rows, ts = [], []
    with open(jsonl_path, "r") as f:
        for line in f:
            rec = json.loads(line.strip())
            rows.append(rec["blendshapes"])
            if "timestamp" in rec:
                ts.append(rec["timestamp"])
    return np.array(rows, dtype=np.float32), ts


def detect_source_fps(timestamps) -> float:
    """Frame rate of a label track (sequential_dataset.py:121-133): 1 / mean timestamp delta, snapped to 30 or 60."""
    source_fps = 30.0
    if len(timestamps) > 1:
        avg_delta = np.mean(np.diff(timestamps))
        if avg_delta > 0:
            source_fps = 1.0 / avg_delta
            if abs(source_fps - 30) < 2:
                source_fps = 30.0
            elif abs(source_fps - 60) < 2:
                source_fps = 60.0
    return float(source_fps)


def _load_wav(path: Path, sample_rate: int) -> np.ndarray:
    from scipy.io import wavfile
    sr, data = wavfile.read(str(path))
    if data.ndim > 1:
        data = data.mean(axis=1)
    if np.issubdtype(data.dtype, np.integer):
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    data = data.astype(np.float32)
    if sr != sample_rate:          # the reference calls librosa.resample here (:100-101); polyphase is the stand-in
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr), int(sample_rate))
        data = resample_poly(data, sample_rate // g, sr // g).astype(np.float32)
    return data


class _Clip:
    """One audio/label pair resident on the device."""

    def __init__(self, audio: torch.Tensor, labels: torch.Tensor, num_windows: int, name: str):
        self.audio, self.labels, self.num_windows, self.name = audio, labels, num_windows, name


class SequentialKoeMorphDataset:
    def __init__(self, data_dir: Union[str, Path], window_frames: int = 256, stride_frames: int = 1,
                 sample_rate: int = 16000, target_fps: int = 30, shuffle_files: bool = True, loop_dataset: bool = True,
                 max_files: Optional[int] = None, batch_size: int = 8, device: Union[str, torch.device] = "cuda"):
        self.data_dir = Path(data_dir)
        self.window_frames, self.stride_frames = window_frames, stride_frames
        self.sample_rate, self.target_fps = sample_rate, target_fps
        self.shuffle_files, self.loop_dataset = shuffle_files, loop_dataset
        self.hop_length = int(sample_rate / target_fps)
        self.window_samples = window_frames * self.hop_length
        self.stride_samples = stride_frames * self.hop_length
        self.batch_size = batch_size
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("SequentialKoeMorphDataset keeps clips in GPU memory: device must be a HIP device")
        self.file_pairs = self._find_file_pairs()
        if max_files:
            self.file_pairs = self.file_pairs[:max_files]
        if len(self.file_pairs) == 0:
            raise ValueError(f"No valid audio/JSONL pairs found in {data_dir}")
        self._lib = load()
        self._clips: Dict[int, _Clip] = {}

    def _find_file_pairs(self) -> List[Tuple[Path, Path]]:
        pairs = []
        for audio_path in self.data_dir.glob("**/*.wav"):
            jsonl_path = audio_path.with_suffix(".jsonl")
            if jsonl_path.exists():
                pairs.append((audio_path, jsonl_path))
        return sorted(pairs)

    # ---- device-side preparation of one clip (reference _process_file_pair :156-178) ----------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def resample_labels(self, labels: torch.Tensor, source_fps: float) -> torch.Tensor:
        """_resample_blendshapes (:136-154) on the device."""
        if abs(source_fps - self.target_fps) < 0.1:
            return labels
        ratio = self.target_fps / source_fps
        source_len = labels.shape[0]
        target_len = int(source_len * ratio)
        out = torch.empty(target_len, labels.shape[1], device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            check(self._lib.km_resample_labels(labels.data_ptr(), source_len, labels.shape[1], target_len, out.data_ptr(),
                                               self._stream()))
        return out

    def clip(self, file_idx: int) -> _Clip:
        if file_idx not in self._clips:
            audio_path, jsonl_path = self.file_pairs[file_idx]
            audio = torch.from_numpy(_load_wav(audio_path, self.sample_rate)).to(self.device)
            rows, ts = load_jsonl_labels(jsonl_path)
            source_fps = detect_source_fps(ts)
            labels = torch.from_numpy(rows).to(self.device)
            if abs(source_fps - self.target_fps) > 0.1:
                labels = self.resample_labels(labels, source_fps)
            expected_frames = audio.shape[0] // self.hop_length
            if abs(labels.shape[0] - expected_frames) > 1:          # :170-178: use the minimum to ensure alignment
                num_frames = min(labels.shape[0], expected_frames)
                audio = audio[:num_frames * self.hop_length].contiguous()
                labels = labels[:num_frames].contiguous()
            num_windows = (labels.shape[0] - self.window_frames) // self.stride_frames + 1
            self._clips[file_idx] = _Clip(audio, labels, max(0, num_windows), audio_path.stem)
        return self._clips[file_idx]

    def gather(self, file_idx: int, window_indices) -> Dict[str, object]:
        """One batch of windows of one clip, assembled on the device."""
        c = self.clip(file_idx)
        wi = torch.as_tensor(window_indices, dtype=torch.int32)
        starts = (wi * self.stride_frames).to(self.device)
        B = int(wi.numel())
        audio = torch.empty(B, self.window_samples, device=self.device)
        bs = torch.empty(B, self.window_frames, c.labels.shape[1], device=self.device)
        target = torch.empty(B, c.labels.shape[1], device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.km_gather_windows(c.audio.data_ptr(), c.audio.shape[0], starts.data_ptr(), B, self.hop_length,
                                              self.window_samples, audio.data_ptr(), c.labels.data_ptr(), c.labels.shape[0],
                                              self.window_frames, c.labels.shape[1], bs.data_ptr(), target.data_ptr(),
                                              self._stream()))
        return {"audio": audio, "blendshapes": bs, "target": target,
                "file_indices": torch.full((B,), file_idx, dtype=torch.int64), "window_indices": wi.to(torch.int64),
                "start_frames": (wi * self.stride_frames).to(torch.int64), "file_names": [c.name] * B}

    # ---- iteration: the reference's order (files, then windows in time order), already batched --------------
    def __iter__(self) -> Iterator[Dict[str, object]]:
        while True:
            order = list(range(len(self.file_pairs)))
            if self.shuffle_files:
                order = torch.randperm(len(order)).tolist()
            for fi in order:
                n = self.clip(fi).num_windows
                # a window is valid only when both slices are full (:191); with aligned lengths that is every i < n
                for w0 in range(0, n, self.batch_size):
                    yield self.gather(fi, list(range(w0, min(n, w0 + self.batch_size))))
            if not self.loop_dataset:
                break

    def get_num_windows(self) -> int:
        return sum(self.clip(i).num_windows for i in range(len(self.file_pairs)))
