"""Oracle: SimplifiedDualStreamModel.apply_temporal_smoothing, restated in numpy.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
/root/reference/src/model/simplified_dual_stream_model.py:341-368 (state at :163-164,
reset at :417-419).  The wrapper model itself cannot be imported here (librosa), so this
is restated from the source text; it is a three-line recurrence.
"""

from __future__ import annotations

from typing import Optional

import numpy as np


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, dtype=np.float64)))


class TemporalSmootherOracle:
    """alpha = sigmoid(smoothing_alpha) (param init 0.8 -> 0.68997...), first call or a
    batch-size change stores the input and returns it unchanged (:357-359); otherwise
    y = alpha*x + (1-alpha)*prev and prev = y (:362-366)."""

    def __init__(self, smoothing_alpha: float = 0.8, dtype=np.float32):
        self.smoothing_alpha = smoothing_alpha
        self.prev: Optional[np.ndarray] = None
        self.dtype = dtype

    def reset(self):                                    # :417-419
        self.prev = None

    def __call__(self, x: np.ndarray) -> np.ndarray:
        x = np.asarray(x, dtype=self.dtype)
        if self.prev is None or self.prev.shape[0] != x.shape[0]:
            self.prev = x.copy()
            return x
        # torch computes sigmoid in float32 on the float32 parameter
        alpha = self.dtype(1.0) / (self.dtype(1.0) + np.exp(-self.dtype(self.smoothing_alpha)))
        y = (alpha * x + (self.dtype(1.0) - alpha) * self.prev).astype(self.dtype)
        self.prev = y.copy()
        return y


def smooth_sequence(frames: np.ndarray, smoothing_alpha: float = 0.8, dtype=np.float32) -> np.ndarray:
    """Apply the recurrence along axis 0 of (N, B, 52) starting from a reset state."""
    sm = TemporalSmootherOracle(smoothing_alpha, dtype)
    return np.stack([sm(f) for f in frames])
