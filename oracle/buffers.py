"""Oracle: the reference's two audio ring buffers, restated (index logic only).

TEST INFRASTRUCTURE (see oracle/__init__.py).
  * RingBufferOracle     follows /root/reference/scripts/rt.py:48-99
  * MelAudioBufferOracle follows /root/reference/src/features/mel_sliding_window.py:28-140

Written as straightforward modular-index loops rather than the reference's split slices,
so that it is an independent check of the wrap-around handling.
"""

from __future__ import annotations

from typing import Optional

import numpy as np


class RingBufferOracle:
    """FIFO of float32 samples; write drops what does not fit (rt.py:61-64), read(n)
    returns None when fewer than n samples are available (:81-82) and consumes."""

    def __init__(self, size: int):
        self.size = size
        self.buffer = np.zeros(size, dtype=np.float32)
        self.write_ptr = 0
        self.read_ptr = 0
        self.available = 0

    def write(self, data: np.ndarray) -> None:
        data = np.asarray(data).astype(np.float32)
        n = min(len(data), self.size - self.available)
        for i in range(n):
            self.buffer[(self.write_ptr + i) % self.size] = data[i]
        if n:
            self.write_ptr = (self.write_ptr + n) % self.size
            self.available = min(self.available + n, self.size)

    def read(self, size: int) -> Optional[np.ndarray]:
        if self.available < size:
            return None
        out = np.array([self.buffer[(self.read_ptr + i) % self.size] for i in range(size)],
                       dtype=np.float32)
        self.read_ptr = (self.read_ptr + size) % self.size
        self.available -= size
        return out


class MelAudioBufferOracle:
    """Overwrite-oldest circular buffer of int(context_window*sr) samples.

    Quirk kept from the reference: the buffer derives its own hop as
    int(sr / (1/update_interval)); with the default update_interval 0.0333 that is
    int(16000/30.03) = 532, not 533 (mel_sliding_window.py:47-50).  Frames within +/-1 of
    that hop are accepted and padded/truncated to it (:80-93); is_full flips once
    total_frames_added*hop >= buffer_size (:112)."""

    def __init__(self, context_window: float = 8.5, sample_rate: int = 16000,
                 update_interval: float = 0.0333):
        self.buffer_size = int(context_window * sample_rate)
        target_fps = 1.0 / update_interval
        self.hop_length = int(sample_rate / target_fps)
        self.audio_buffer = np.zeros(self.buffer_size, dtype=np.float32)
        self.write_ptr = 0
        self.is_full = False
        self.total_frames_added = 0

    def add_audio_frame(self, frame: np.ndarray) -> bool:
        n = len(frame)
        if abs(n - self.hop_length) > 1:
            return False
        frame = np.asarray(frame, dtype=np.float32)
        if n < self.hop_length:
            frame = np.concatenate([frame, np.zeros(self.hop_length - n, np.float32)])
        elif n > self.hop_length:
            frame = frame[:self.hop_length]
        for i in range(self.hop_length):
            self.audio_buffer[(self.write_ptr + i) % self.buffer_size] = frame[i]
        self.write_ptr = (self.write_ptr + self.hop_length) % self.buffer_size
        self.total_frames_added += 1
        if not self.is_full and self.total_frames_added * self.hop_length >= self.buffer_size:
            self.is_full = True
        return True

    def get_current_audio(self) -> Optional[np.ndarray]:
        if not self.is_full:
            return None
        idx = (self.write_ptr + np.arange(self.buffer_size)) % self.buffer_size
        return self.audio_buffer[idx].copy()


class AudioBufferOracle:
    """Restatement of AudioBuffer (/root/reference/src/features/opensmile_extractor.py:29-154) with explicit modular
    indexing: a ring of max_samples; ``is_full`` once the write position has wrapped (:83-91); ``get_window``: zeros (and an
    underrun count) from an empty buffer, ALL samples so far while fewer than the window have arrived (:111-116), the OLDEST
    `window` samples while the ring has not wrapped yet (:128-130 -- not the newest: kept as written there), the newest
    `window` samples once it has (:119-127)."""

    def __init__(self, max_duration: float, sample_rate: int = 16000):
        self.n = int(max_duration * sample_rate)
        self.sample_rate = sample_rate
        self.ring = np.zeros(self.n, np.float32)
        self.w = 0
        self.full = False
        self.underruns = 0
        self.total = 0

    def append(self, x: np.ndarray) -> None:
        x = np.asarray(x, np.float32)
        if self.w + len(x) > self.n:
            self.full = True
        for i, v in enumerate(x):
            self.ring[(self.w + i) % self.n] = v
        self.w += len(x)
        if self.w >= self.n:
            self.full = True
            self.w %= self.n
        self.total += len(x)

    def get_window(self, duration=None) -> np.ndarray:
        want = self.n if duration is None else min(int(duration * self.sample_rate), self.n)
        if not self.full:
            if self.w == 0:
                if want > 0:
                    self.underruns += 1
                    return np.zeros(want, np.float32)
            return self.ring[:min(self.w, want)].copy()
        return np.array([self.ring[(self.w - want + i) % self.n] for i in range(want)], np.float32)
