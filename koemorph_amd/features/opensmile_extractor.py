"""MI355X mirror of the reference's long-context emotion feature extractor (src/features/opensmile_extractor.py).

Same classes, constructor arguments, methods and state machine as the reference (``AudioBuffer`` :29-154,
``OpenSMILEeGeMAPSExtractor`` :157-665, ``create_opensmile_extractor`` :668-697); what differs is who computes the 88
eGeMAPSv02 functionals: the reference hands each 20 s window to the third-party ``opensmile`` package on a CPU core
(:227-235, :439), here the window goes to ``km_egemaps_functionals`` (koemorph_amd/csrc/km_egemaps.hip) -- and
``extract_batch`` does it for any number of windows at once, which is what 1 024 concurrent speaker streams need.

PARITY UNPINNED for the feature values (openSMILE is neither vendored nor pinned by the reference and is not installed here;
see oracle/egemaps.py for the published definitions that are implemented).  The buffering / update / 3-window concatenation
logic is the reference's own and is tested against a restatement (oracle/buffers.py).
"""
from __future__ import annotations

import ctypes as C
import logging
import threading
import time
from collections import deque
from typing import Dict, List, Optional

import numpy as np

from .. import _lib
from .._lib import check

logger = logging.getLogger(__name__)

EGEMAPS_DIM = 88


class AudioBuffer:
    """Circular buffer of mono samples with 'most recent window' reads (reference :29-154)."""

    def __init__(self, max_duration: float, sample_rate: int = 16000, dtype: np.dtype = np.float32):
        self.max_duration = max_duration
        self.sample_rate = sample_rate
        self.max_samples = int(max_duration * sample_rate)
        self.dtype = dtype
        self.buffer = np.zeros(self.max_samples, dtype=dtype)
        self.write_pos = 0
        self.is_full = False
        self.lock = threading.Lock()
        self.total_samples_written = 0
        self.buffer_underruns = 0

    def append(self, audio_data: np.ndarray) -> None:
        if audio_data.ndim != 1:
            raise ValueError("Audio data must be 1D array")
        audio_data = audio_data.astype(self.dtype)
        with self.lock:
            n = len(audio_data)
            room = self.max_samples - self.write_pos
            if n <= room:
                self.buffer[self.write_pos:self.write_pos + n] = audio_data
                self.write_pos += n
            else:                                       # wrap around (a chunk longer than the buffer is the caller's problem, as in :80-86)
                self.buffer[self.write_pos:] = audio_data[:room]
                self.buffer[:n - room] = audio_data[room:]
                self.write_pos = n - room
                self.is_full = True
            if self.write_pos >= self.max_samples:
                self.write_pos = 0
                self.is_full = True
            self.total_samples_written += n

    def get_window(self, duration: Optional[float] = None) -> np.ndarray:
        if duration is None:
            duration = self.max_duration
        want = min(int(duration * self.sample_rate), self.max_samples)
        with self.lock:
            if not self.is_full and self.write_pos < want:
                if self.write_pos == 0:
                    self.buffer_underruns += 1
                    return np.zeros(want, dtype=self.dtype)
                return self.buffer[:self.write_pos]
            if self.is_full:
                if self.write_pos >= want:
                    return self.buffer[self.write_pos - want:self.write_pos].copy()
                return np.concatenate([self.buffer[self.max_samples - (want - self.write_pos):], self.buffer[:self.write_pos]])
            return self.buffer[:min(self.write_pos, want)].copy()

    def get_stats(self) -> Dict[str, int]:
        with self.lock:
            return {"total_samples_written": self.total_samples_written, "buffer_underruns": self.buffer_underruns,
                    "current_fill": self.write_pos if not self.is_full else self.max_samples, "is_full": self.is_full,
                    "max_samples": self.max_samples}

    def reset(self):
        with self.lock:
            self.buffer.fill(0)
            self.write_pos = 0
            self.is_full = False
            self.total_samples_written = 0
            self.buffer_underruns = 0


class EGeMAPSEngine:
    """Thin owner of a km_egemaps plan + workspace (one per device)."""

    def __init__(self, device="cuda"):
        import torch
        self._torch = torch
        if not torch.cuda.is_available():
            raise _lib.KoeMorphError(_lib.KM_ERR_HIP, "no GPU visible: the eGeMAPS extractor has no CPU fallback")
        self.device = torch.device(device if device not in (None, "cpu", "auto") else "cuda")
        self._lib = _lib.load()
        self._plan = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self._lib.km_egemaps_plan_create(C.byref(self._plan)))
        self._work = None
        self._work_for = (0, 0)

    def close(self):
        if getattr(self, "_plan", None) is not None and self._plan.value:
            self._lib.km_egemaps_plan_destroy(self._plan)
            self._plan = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def num_frames(self, L: int) -> int:
        return int(self._lib.km_egemaps_num_frames(L))

    def functionals(self, audio, normalize: bool = True):
        """audio (B, L) fp32 on the device -> (B, 88) eGeMAPSv02 functionals."""
        torch = self._torch
        audio = audio.to(self.device, torch.float32).contiguous()
        if audio.dim() != 2:
            raise ValueError(f"Expected 2D audio (B, L), got {audio.dim()}D")
        B, L = audio.shape
        need = int(self._lib.km_egemaps_workspace_floats(B, L))
        if self._work is None or self._work.numel() < need:
            self._work = torch.empty(need, device=self.device)
        self._work_for = (B, L)
        out = torch.empty(B, EGEMAPS_DIM, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.km_egemaps_functionals(self._plan, audio.data_ptr(), B, L, 1 if normalize else 0, self._work.data_ptr(),
                                                   self._work.numel(), out.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream))
        return out

    def records(self) -> np.ndarray:
        """Per-frame low-level descriptors of the most recent call, (B, frames, 36) -- see km_egemaps.hip (tests)."""
        B, L = self._work_for
        rec = np.empty((B, self.num_frames(L), 36), np.float32)
        with self._torch.cuda.device(self.device):
            check(self._lib.km_egemaps_records(self._work.data_ptr(), B, L, rec.ctypes.data,
                                               self._torch.cuda.current_stream(self.device).cuda_stream))
        return rec


class OpenSMILEeGeMAPSExtractor:
    """eGeMAPS extractor with a sliding window (reference :157-665); the features come from the GPU."""

    def __init__(self, sample_rate: int = 16000, context_window: float = 20.0, update_interval: float = 0.3,
                 feature_set: str = "eGeMAPSv02", feature_level: str = "Functionals", enable_caching: bool = True,
                 cache_dir: Optional[str] = None, device: str = "cpu", temporal_history_frames: int = 30,
                 use_concatenation: bool = False, clock=time.time):
        if sample_rate != 16000:
            raise ValueError("the GPU eGeMAPS extractor is built for 16 kHz audio")
        if context_window < 1.0:
            raise ValueError("Context window must be at least 1.0 seconds")
        if update_interval < 0.1:
            raise ValueError("Update interval must be at least 0.1 seconds")
        if update_interval > context_window:
            raise ValueError("Update interval cannot be larger than context window")
        if feature_set != "eGeMAPSv02":
            raise ValueError(f"Unsupported feature set: {feature_set}")       # the reference also offers GeMAPS (:214-215)
        if feature_level != "Functionals":
            raise ValueError(f"Unsupported feature level: {feature_level}")
        self.sample_rate, self.context_window, self.update_interval = sample_rate, context_window, update_interval
        self.enable_caching = enable_caching
        self.device = device
        self.temporal_history_frames, self.use_concatenation = temporal_history_frames, use_concatenation
        self._clock = clock
        self.engine = EGeMAPSEngine(device)
        self.feature_dim = EGEMAPS_DIM
        self.audio_buffer = AudioBuffer(max_duration=context_window + 2.0, sample_rate=sample_rate)
        self.last_update_time = 0.0
        self.current_features: Optional[np.ndarray] = None
        self.total_updates = 0
        self.failed_extractions = 0
        self.feature_history = deque(maxlen=temporal_history_frames)
        self.temporal_features_ready = False
        if use_concatenation:
            self.window_intervals = [0.0, 0.3, 0.6]
            self.window_features = {i: None for i in self.window_intervals}
            self.last_window_updates = {i: 0.0 for i in self.window_intervals}
            self.compression_layer = None
            self.concatenated_features_ready = False
        self.extraction_times = deque(maxlen=100)
        self.feature_cache = {} if enable_caching else None

    # ---- the part openSMILE did -------------------------------------------------------------------------------------
    def extract_batch(self, audio, normalize: bool = True):
        """(B, L) windows (numpy or tensor) -> (B, 88) tensor on the device: every window in one call."""
        import torch
        if isinstance(audio, np.ndarray):
            audio = torch.from_numpy(np.ascontiguousarray(audio, np.float32))
        return self.engine.functionals(audio, normalize)

    def _extract_features_from_audio(self, audio: np.ndarray) -> Optional[np.ndarray]:
        """One window -> (88,) float32 (reference :427-454: float32, peak normalisation, NaN / Inf -> 0)."""
        try:
            audio = np.asarray(audio, np.float32)
            if self.engine.num_frames(len(audio)) < 1:
                logger.warning("Window shorter than one analysis frame")
                return None
            feats = self.extract_batch(audio[None, :], normalize=True)[0].cpu().numpy()
            if np.any(np.isnan(feats)) or np.any(np.isinf(feats)):
                logger.warning("Invalid features detected (NaN/Inf)")
                feats = np.nan_to_num(feats, nan=0.0, posinf=0.0, neginf=0.0)
            return feats.astype(np.float32)
        except Exception as e:
            logger.warning(f"eGeMAPS feature extraction failed: {e}")
            return None

    # ---- reference state machine ------------------------------------------------------------------------------------
    def process_audio_frame(self, audio_frame: np.ndarray, force_update: bool = False) -> Optional[np.ndarray]:
        now = self._clock()
        self.audio_buffer.append(audio_frame)
        if force_update or now - self.last_update_time >= self.update_interval or self.current_features is None:
            return self._extract_features(now)
        return self.current_features

    def process_audio_batch(self, audio_batch: np.ndarray, frame_length: Optional[int] = None) -> np.ndarray:
        single = audio_batch.ndim == 1
        if single:
            audio_batch = audio_batch[None, :]
        batch_size, seq_len = audio_batch.shape
        if frame_length is None:
            frame_length = int(self.sample_rate * self.update_interval)
        batch_features = []
        for b in range(batch_size):
            audio = audio_batch[b]
            sample_features = []
            self.audio_buffer.reset()
            for start in range(0, seq_len, frame_length):
                f = self.process_audio_frame(audio[start:min(start + frame_length, seq_len)], force_update=True)
                if f is not None:
                    sample_features.append(f)
            if sample_features:
                batch_features.append(np.stack(sample_features))
            else:
                batch_features.append(self._extract_features_from_audio(audio)[None, :])
        result = np.stack(batch_features) if batch_features else np.zeros((batch_size, 1, self.feature_dim))
        return result[0] if single else result

    def _extract_features(self, current_time: float) -> Optional[np.ndarray]:
        t0 = time.time()
        try:
            window = self.audio_buffer.get_window(self.context_window)
            if len(window) < int(self.sample_rate * 0.5):
                return self.current_features
            feats = self._extract_features_from_audio(window)
            if feats is None:
                self.failed_extractions += 1
                return self.current_features
            self.current_features = feats
            self.last_update_time = current_time
            self.total_updates += 1
            self.feature_history.append(feats.copy())
            if self.use_concatenation:
                self._update_window_features(current_time, feats)
            if len(self.feature_history) >= self.temporal_history_frames:
                self.temporal_features_ready = True
            self.extraction_times.append(time.time() - t0)
            return feats
        except Exception as e:
            logger.warning(f"Feature extraction failed: {e}")
            self.failed_extractions += 1
            return self.current_features

    def _update_window_features(self, current_time: float, current_features: np.ndarray):
        """Reference :456-502.  As written there, the 'time since start' it tests is always zero (the current window's
        timestamp has just been set to `current_time`), so the 300 ms and 600 ms slots are filled ONCE, with the first
        features, and only the current slot follows the audio.  Kept: it is the behaviour the model was trained with."""
        self.window_features[0.0] = current_features.copy()
        self.last_window_updates[0.0] = current_time
        for interval in (0.3, 0.6):
            if self.window_features[interval] is None:
                self.window_features[interval] = current_features.copy()
                self.last_window_updates[interval] = current_time
        if all(self.window_features[i] is not None for i in self.window_intervals):
            self.concatenated_features_ready = True

    def get_temporal_features(self) -> Optional[np.ndarray]:
        if len(self.feature_history) == 0:
            return None
        hist = list(self.feature_history)
        pad = self.temporal_history_frames - len(hist)
        if pad > 0:
            hist = [np.zeros(self.feature_dim, dtype=np.float32)] * pad + hist
        return np.stack(hist)

    def get_concatenated_features(self) -> Optional[np.ndarray]:
        """3 windows x 88 -> 264 -> Linear(264, 256) (reference :559-592; the layer is created on first use with torch's
        default initialisation and is not trained there either).  The product runs on the GPU (km_linear)."""
        if not self.use_concatenation:
            logger.warning("get_concatenated_features() called but use_concatenation=False")
            return None
        if not self.concatenated_features_ready:
            return None
        import torch
        parts = [self.window_features[i] if self.window_features[i] is not None else np.zeros(self.feature_dim, np.float32)
                 for i in self.window_intervals]
        concatenated = np.concatenate(parts)
        if self.compression_layer is None:
            self.compression_layer = torch.nn.Linear(264, 256)
        dev = self.engine.device
        layer = self.compression_layer
        x = torch.from_numpy(concatenated.astype(np.float32)).to(dev).unsqueeze(0).contiguous()
        w = layer.weight.detach().to(dev, torch.float32).contiguous()
        b = layer.bias.detach().to(dev, torch.float32).contiguous()
        out = torch.empty(1, 256, device=dev)
        with torch.cuda.device(dev):
            check(self.engine._lib.km_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), 1, 264, 256, out.data_ptr(),
                                             torch.cuda.current_stream(dev).cuda_stream))
        return out[0].cpu().numpy().astype(np.float32)

    def emotion_features_batch(self, audio):
        """(B, L) windows -> (B, 256) emotion vectors, the END STATE of the reference's production path
        (EmotionExtractor._extract_opensmile, src/features/emotion_extractor.py:435-470, with use_concatenation): per
        sample it resets the audio buffer, feeds the window in update-interval frames with force_update (an openSMILE run
        on the growing window at EVERY frame), then concatenates the three window slots and compresses 264 -> 256.  Only
        the last of those runs reaches the output; and the 300 / 600 ms slots are filled once per extractor lifetime, by
        the first extraction it ever makes (see _update_window_features).  So: one batched extraction of the B full
        windows, one more of the first sample's first two frames if the slots are still empty, one batched Linear."""
        import torch
        if not self.use_concatenation:
            raise RuntimeError("emotion_features_batch needs use_concatenation=True (the reference's production configuration)")
        if isinstance(audio, np.ndarray):
            audio = torch.from_numpy(np.ascontiguousarray(audio, np.float32))
        audio = audio.to(self.engine.device, torch.float32)
        B, L = audio.shape
        frame = int(self.sample_rate * self.update_interval)
        ctx = int(self.context_window * self.sample_rate)
        if self.window_features[0.3] is None:
            # the first frame count at which the buffer holds >= 0.5 s (reference _extract_features :366-368)
            k = 1
            while min(k * frame, L) < int(self.sample_rate * 0.5) and k * frame < L:
                k += 1
            first = self.extract_batch(audio[:1, :min(k * frame, L)][:, -ctx:])[0].cpu().numpy()
            self._update_window_features(self._clock(), first)
        current = self.extract_batch(audio[:, -ctx:])                                    # (B, 88)
        self.window_features[0.0] = current[-1].cpu().numpy()
        self.current_features = self.window_features[0.0]
        dev = self.engine.device
        past = torch.from_numpy(np.concatenate([self.window_features[0.3], self.window_features[0.6]])).to(dev)
        cat = torch.cat([current, past.unsqueeze(0).expand(B, -1)], dim=1).contiguous()  # (B, 264)
        if self.compression_layer is None:
            self.compression_layer = torch.nn.Linear(264, 256)
        w = self.compression_layer.weight.detach().to(dev, torch.float32).contiguous()
        b = self.compression_layer.bias.detach().to(dev, torch.float32).contiguous()
        out = torch.empty(B, 256, device=dev)
        with torch.cuda.device(dev):
            check(self.engine._lib.km_linear(cat.data_ptr(), w.data_ptr(), b.data_ptr(), B, 264, 256, out.data_ptr(),
                                             torch.cuda.current_stream(dev).cuda_stream))
        return out

    def get_feature_names(self) -> List[str]:
        from ..egemaps_names import FEATURE_NAMES
        return list(FEATURE_NAMES)

    def get_stats(self) -> Dict[str, object]:
        return {"total_updates": self.total_updates, "failed_extractions": self.failed_extractions,
                "success_rate": self.total_updates / max(self.total_updates + self.failed_extractions, 1),
                "avg_extraction_time": np.mean(self.extraction_times) if self.extraction_times else 0.0,
                "context_window": self.context_window, "update_interval": self.update_interval, "feature_dim": self.feature_dim,
                "temporal_history_frames": self.temporal_history_frames, "temporal_features_ready": self.temporal_features_ready,
                "history_length": len(self.feature_history), "use_concatenation": self.use_concatenation,
                "concatenated_features_ready": getattr(self, "concatenated_features_ready", False),
                "buffer_stats": self.audio_buffer.get_stats(), "current_features_available": self.current_features is not None}

    def reset(self):
        self.audio_buffer.reset()
        self.current_features = None
        self.last_update_time = 0.0
        self.total_updates = 0
        self.failed_extractions = 0
        self.extraction_times.clear()
        self.feature_history.clear()
        self.temporal_features_ready = False
        if self.use_concatenation:
            self.window_features = {i: None for i in self.window_intervals}
            self.last_window_updates = {i: 0.0 for i in self.window_intervals}
            self.concatenated_features_ready = False

    def set_context_window(self, duration: float):
        if duration < 1.0:
            raise ValueError("Context window must be at least 1.0 seconds")
        self.context_window = duration
        self.audio_buffer = AudioBuffer(max_duration=duration + 2.0, sample_rate=self.sample_rate)

    def set_update_interval(self, interval: float):
        if interval < 0.1:
            raise ValueError("Update interval must be at least 0.1 seconds")
        if interval > self.context_window:
            raise ValueError("Update interval cannot be larger than context window")
        self.update_interval = interval


def create_opensmile_extractor(config: Dict) -> OpenSMILEeGeMAPSExtractor:
    """Reference :668-697."""
    return OpenSMILEeGeMAPSExtractor(
        sample_rate=config.get("sample_rate", 16000), context_window=config.get("context_window", 20.0),
        update_interval=config.get("update_interval", 0.3), feature_set=config.get("feature_set", "eGeMAPSv02"),
        feature_level=config.get("feature_level", "Functionals"), enable_caching=config.get("enable_caching", True),
        cache_dir=config.get("cache_dir"), device=config.get("device", "cpu"),
        temporal_history_frames=config.get("temporal_history_frames", 30), use_concatenation=config.get("use_concatenation", False))
