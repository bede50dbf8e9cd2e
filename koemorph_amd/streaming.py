"""Many concurrent speaker streams with device-resident state (BASELINE config 5: 1024 streams, 128 per GPU).

Each stream owns an 8.5 s audio ring (``MelAudioBuffer`` semantics, reference
src/features/mel_sliding_window.py:28-140), and an EMA state; both live in the km_handle on the GPU and never
move.  A tick is ``push`` (one ~hop-sized frame per stream, the only host->device traffic: n_streams x 533
floats) followed by ``tick`` (emotion kernel, sliding-window front end over every full ring, fused core with
per-stream EMA).  Neither allocates nor synchronises, so ``capture()`` records one tick into a hipGraph and
``replay()`` re-launches it with a single API call per 33 ms frame.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check
from .engine import Engine, MelConfig, _ptr, _stream_ptr


class StreamEngine:
    def __init__(self, engine: Engine, n_streams: int, context_window: float = 8.5, update_interval: float = 0.0333,
                 mel: Optional[MelConfig] = None):
        if engine.device is None:
            raise _lib.KoeMorphError(_lib.KM_ERR_NOT_FINALIZED, "finalize the Engine before creating streams")
        self.engine = engine
        self.n_streams = n_streams
        self.mel = mel or MelConfig.sliding_window(n_fft=1024, hop_length=engine.mel.hop_length)
        self._lib = engine._lib
        cfg = self.mel.to_c()
        with torch.cuda.device(engine.device):
            check(self._lib.km_stream_create(engine._h, n_streams, context_window, update_interval, C.byref(cfg)))
        self.ring_hop = int(self.mel.sample_rate / (1.0 / update_interval))
        dev = engine.device
        self.out = torch.zeros(n_streams, engine.num_blendshapes, device=dev)
        self.ready = torch.zeros(n_streams, dtype=torch.uint8, device=dev)
        self._graph = None
        self._g_samples = self._g_emotion = None

    def push(self, samples: torch.Tensor) -> None:
        """samples (n_streams, n) fp32 on the device, n within +/-1 of the ring hop (532)."""
        if samples.dim() != 2 or samples.shape[0] != self.n_streams:
            raise ValueError(f"expected ({self.n_streams}, n) samples, got {tuple(samples.shape)}")
        samples = samples.contiguous()
        check(self._lib.km_stream_push(self.engine._h, _ptr(samples), samples.shape[1], _stream_ptr(samples.device)))

    def tick(self, emotion: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """emotion (n_streams, emotion_dim) -> (out (n_streams, 52), ready (n_streams) uint8)."""
        emotion = emotion.contiguous()
        check(self._lib.km_stream_tick(self.engine._h, _ptr(emotion), _ptr(self.out), _ptr(self.ready),
                                       _stream_ptr(emotion.device)))
        return self.out, self.ready

    def reset(self) -> None:
        check(self._lib.km_stream_reset(self.engine._h, _stream_ptr(self.engine.device)))
        self.out.zero_()
        self.ready.zero_()

    # ---- hipGraph replay ------------------------------------------------------------------------
    def capture(self, n_per_stream: int = 533, host_out: Optional[torch.Tensor] = None) -> None:
        """Record push + tick on static input buffers into a hipGraph (torch.cuda.CUDAGraph drives
        hipStreamBeginCapture on the current stream; the kernels are launched by libkoemorph_hip).  ``host_out``: a pinned
        (n_streams, 52) host tensor -- the tick's result readback becomes the graph's last node instead of a call per tick."""
        dev = self.engine.device
        if host_out is not None and (not host_out.is_pinned() or tuple(host_out.shape) != tuple(self.out.shape)):
            raise ValueError("host_out must be a pinned host tensor of the shape of the result")
        self._g_samples = torch.zeros(self.n_streams, n_per_stream, device=dev)
        self._g_emotion = torch.zeros(self.n_streams, self.engine.emotion_dim, device=dev)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.push(self._g_samples)
            self.tick(self._g_emotion)
            if host_out is not None:
                host_out.copy_(self.out, non_blocking=True)
        self._graph = g
        self._g_host_out = host_out

    def replay(self, samples: torch.Tensor, emotion: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._graph is None:
            raise RuntimeError("capture() first")
        self._g_samples.copy_(samples, non_blocking=True)
        self._g_emotion.copy_(emotion, non_blocking=True)
        self._graph.replay()
        return self.out, self.ready
