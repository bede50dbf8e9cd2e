"""Wire encoding of blendshape frames: byte-identical to the reference's per-frame
``json.dumps({"timestamp": t, "blendshapes": row.tolist()})`` (scripts/rt.py:209-231, src/data/io.py:119-131), done for a
whole tick of streams by one call into the C library (km_format_frames, koemorph_amd/csrc/km_wire.cpp)."""
from __future__ import annotations

import ctypes
from typing import List, Sequence, Union

import numpy as np

from ._lib import KoeMorphError, load


def format_frames_raw(frames: np.ndarray, timestamps: Union[float, Sequence[float], np.ndarray], newline: bool = False):
    """(buffer: bytes, offsets: int64[n+1]) -- frame f is buffer[offsets[f]:offsets[f+1]]."""
    frames = np.ascontiguousarray(frames, dtype=np.float32)
    if frames.ndim == 1:
        frames = frames[None, :]
    n, k = frames.shape
    ts = np.ascontiguousarray(np.broadcast_to(np.asarray(timestamps, dtype=np.float64), (n,)))
    cap = n * (64 + 34 * k) + 64
    buf = ctypes.create_string_buffer(cap)
    off = np.empty(n + 1, np.int64)
    lib = load()
    rc = lib.km_format_frames(frames.ctypes.data, n, k, ts.ctypes.data, 1 if newline else 0, buf, cap, off.ctypes.data)
    if rc < 0:
        raise KoeMorphError(int(rc), "km_format_frames failed")
    return buf.raw[:rc], off


def format_frames(frames: np.ndarray, timestamps, newline: bool = False) -> List[bytes]:
    """One UTF-8 JSON text per frame, exactly what json.dumps(...).encode() gives in the reference."""
    raw, off = format_frames_raw(frames, timestamps, newline)
    return [raw[off[i]:off[i + 1]] for i in range(len(off) - 1)]
