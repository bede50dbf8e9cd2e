"""Drop-in mirrors of the reference's real-time mel front end
(reference src/features/mel_sliding_window.py):

  MelAudioBuffer               :21-154   overwrite-oldest circular buffer of 8.5 s of audio (host)
  MelSlidingWindowExtractor    :157-420  per-tick log-mel of the whole buffer
  create_mel_extractor         :423-440

The buffer is host-side index logic, identical to the reference including its quirks (the buffer
derives its OWN hop, int(sr / (1/update_interval)) = 532 for the default 0.0333 s, and pads / truncates
every accepted frame to it, :47-50,:80-93).  The log-mel of the window (librosa melspectrogram with
pad_mode reflect + power_to_db(ref=max), no affine, truncated to int(context/update) frames, :280-307)
runs in the HIP front end through km_mel_extract.  For many concurrent streams use the device-resident
ring buffers of koemorph_amd.streaming instead (one H2D copy of ~hop samples per stream per tick).
"""
from __future__ import annotations

import logging
import threading
import time
from collections import deque
from typing import Any, Callable, Dict, Optional

import numpy as np

from ..engine import Engine, MelConfig

logger = logging.getLogger(__name__)


class MelAudioBuffer:
    def __init__(self, context_window: float = 8.5, sample_rate: int = 16000, update_interval: float = 0.0333):
        self.context_window = context_window
        self.sample_rate = sample_rate
        self.update_interval = update_interval
        self.buffer_size = int(context_window * sample_rate)             # 136000
        target_fps = 1.0 / update_interval
        self.hop_length = int(sample_rate / target_fps)                  # 532 for 0.0333 (reference :47-50)
        self.audio_buffer = np.zeros(self.buffer_size, dtype=np.float32)
        self.write_ptr = 0
        self.is_full = False
        self._lock = threading.Lock()
        self.total_frames_added = 0
        self.buffer_overruns = 0

    def add_audio_frame(self, audio_frame: np.ndarray) -> bool:
        frame_len = len(audio_frame)
        if abs(frame_len - self.hop_length) > 1:                          # +/-1 sample tolerance (:80-82)
            logger.warning(f"Frame size mismatch: expected ~{self.hop_length}, got {frame_len}")
            return False
        if frame_len < self.hop_length:
            audio_frame = np.pad(audio_frame, (0, self.hop_length - frame_len), mode='constant')
        elif frame_len > self.hop_length:
            audio_frame = audio_frame[:self.hop_length]
        with self._lock:
            end_ptr = (self.write_ptr + self.hop_length) % self.buffer_size
            if end_ptr > self.write_ptr:
                self.audio_buffer[self.write_ptr:end_ptr] = audio_frame
            else:
                first = self.buffer_size - self.write_ptr
                self.audio_buffer[self.write_ptr:] = audio_frame[:first]
                self.audio_buffer[:end_ptr] = audio_frame[first:]
            self.write_ptr = end_ptr
            self.total_frames_added += 1
            if not self.is_full and self.total_frames_added * self.hop_length >= self.buffer_size:
                self.is_full = True
        return True

    def get_current_audio(self) -> Optional[np.ndarray]:
        with self._lock:
            if not self.is_full:
                return None
            if self.write_ptr == 0:
                return self.audio_buffer.copy()
            return np.concatenate([self.audio_buffer[self.write_ptr:], self.audio_buffer[:self.write_ptr]])

    def get_stats(self) -> Dict[str, Any]:
        with self._lock:
            return {
                "context_window": self.context_window,
                "buffer_size": self.buffer_size,
                "hop_length": self.hop_length,
                "total_frames_added": self.total_frames_added,
                "buffer_overruns": self.buffer_overruns,
                "is_full": self.is_full,
                "write_ptr": self.write_ptr,
                "buffer_utilization": (self.total_frames_added * self.hop_length / self.buffer_size
                                       if self.total_frames_added > 0 else 0.0),
            }


class MelSlidingWindowExtractor:
    def __init__(
        self,
        context_window: float = 8.5,
        update_interval: float = 0.0333,
        sample_rate: int = 16000,
        n_mels: int = 80,
        n_fft: int = 512,
        hop_length: Optional[int] = None,
        win_length: Optional[int] = None,
        f_min: float = 80.0,
        f_max: Optional[float] = None,
        power: float = 2.0,
        center: bool = True,
        pad_mode: str = "reflect",
        device: str = "cuda",
        engine_getter: Optional[Callable[[], Engine]] = None,
        clock: Callable[[], float] = time.time,
    ):
        if power != 2.0 or not center or (win_length not in (None, n_fft)):
            raise ValueError("the HIP front end implements power=2.0, center=True, win_length=n_fft")
        self.context_window = context_window
        self.update_interval = update_interval
        self.sample_rate = sample_rate
        self.n_mels = n_mels
        self.n_fft = n_fft
        self.f_min = f_min
        self.f_max = f_max or sample_rate // 2
        self.power = power
        self.center = center
        self.pad_mode = pad_mode
        self.device = device
        target_fps = 1.0 / update_interval
        self.hop_length = hop_length or int(sample_rate / target_fps)     # reference :213-214
        self.win_length = win_length or n_fft
        self.audio_buffer = MelAudioBuffer(context_window, sample_rate, update_interval)
        self.cfg = MelConfig.sliding_window(sample_rate, n_fft, self.hop_length, n_mels, f_min, self.f_max, pad_mode)
        self._engine_getter = engine_getter
        self._engine: Optional[Engine] = None
        self._clock = clock
        self.current_features: Optional[np.ndarray] = None
        self._current_device = None
        self.last_update_time = 0
        self.features_ready = False
        self.extraction_times = deque(maxlen=100)
        self.total_extractions = 0
        self.failed_extractions = 0
        expected_frames = int(context_window / update_interval)
        self.feature_shape = (expected_frames, n_mels)

    def _eng(self) -> Engine:
        if self._engine_getter is not None:
            return self._engine_getter()
        if self._engine is None:
            from .. import synth
            self._engine = Engine()
            self._engine.load_state_dict(synth.make_core_params(0))
            self._engine.finalize(self.device if self.device != "cpu" else None)
        return self._engine

    def _extract(self, audio_window: np.ndarray, out_frames: int):
        import torch
        eng = self._eng()
        x = torch.from_numpy(np.ascontiguousarray(audio_window, dtype=np.float32)).unsqueeze(0).to(eng.device)
        return eng.mel_extract(self.cfg, x, out_frames=out_frames)[0]

    def process_audio_frame_device(self, audio_frame: np.ndarray):
        """Like process_audio_frame but returns the (T, n_mels) features as a device tensor."""
        if not self.audio_buffer.add_audio_frame(audio_frame):
            return None
        now = self._clock()
        if now - self.last_update_time < self.update_interval * 0.3:       # wall-clock gate (:267-269)
            return self._current_device
        audio_window = self.audio_buffer.get_current_audio()
        if audio_window is None:
            return None
        t0 = time.time()
        expected = int(self.context_window / self.update_interval)         # 255 (:300-307)
        feats = self._extract(audio_window, expected)
        self._current_device = feats
        self.current_features = None
        self.last_update_time = now
        self.features_ready = True
        self.extraction_times.append(time.time() - t0)
        self.total_extractions += 1
        return feats

    def process_audio_frame(self, audio_frame: np.ndarray) -> Optional[np.ndarray]:
        """Reference signature (:252-324): numpy (T, n_mels) float32 in [-80, 0] dB, or None."""
        feats = self.process_audio_frame_device(audio_frame)
        if feats is None:
            return None
        if self.current_features is None:
            self.current_features = feats.cpu().numpy().astype(np.float32)
        return self.current_features

    def process_audio_batch(self, audio: np.ndarray) -> np.ndarray:
        """Whole-clip extraction, no truncation (:326-365)."""
        return self._extract(np.asarray(audio), 0).cpu().numpy().astype(np.float32)

    def get_current_features(self) -> Optional[np.ndarray]:
        if not self.features_ready:
            return None
        if self.current_features is None and self._current_device is not None:
            self.current_features = self._current_device.cpu().numpy().astype(np.float32)
        return self.current_features

    def reset(self):
        self.audio_buffer = MelAudioBuffer(self.context_window, self.sample_rate, self.update_interval)
        self.current_features = None
        self._current_device = None
        self.last_update_time = 0
        self.features_ready = False

    def get_stats(self) -> Dict[str, Any]:
        ext = {
            "total_extractions": self.total_extractions,
            "failed_extractions": self.failed_extractions,
            "success_rate": (self.total_extractions - self.failed_extractions) / max(1, self.total_extractions),
            "features_ready": self.features_ready,
        }
        if self.extraction_times:
            ext.update({"avg_extraction_time": float(np.mean(self.extraction_times)),
                        "max_extraction_time": float(np.max(self.extraction_times)),
                        "min_extraction_time": float(np.min(self.extraction_times))})
        return {"context_window": self.context_window, "update_interval": self.update_interval,
                "feature_shape": self.feature_shape, "buffer_stats": self.audio_buffer.get_stats(),
                "extraction_stats": ext}

    @property
    def feature_dim(self) -> int:
        return self.n_mels


def create_mel_extractor(context_window: float = 8.5, update_interval: float = 0.0333, sample_rate: int = 16000,
                         n_mels: int = 80, **kwargs) -> MelSlidingWindowExtractor:
    return MelSlidingWindowExtractor(context_window=context_window, update_interval=update_interval,
                                     sample_rate=sample_rate, n_mels=n_mels, **kwargs)
