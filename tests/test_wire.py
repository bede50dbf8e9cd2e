"""Wire format (SURVEY 8b 'Output wire format', 8f-3): km_format_frames must reproduce, byte for byte, the
json.dumps({"timestamp": t, "blendshapes": row.tolist()}) of scripts/rt.py:209-231 -- checked against CPython's own
json module (the library the reference calls), including the layout switch points of float repr."""
import json
import math

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from koemorph_amd.wire import format_frames, format_frames_raw


def ref(frames, ts, newline=False):
    frames = np.asarray(frames, np.float32)
    return [(json.dumps({"timestamp": float(t), "blendshapes": row.tolist()}) + ("\n" if newline else "")).encode("utf-8")
            for row, t in zip(frames, ts)]


def test_typical_tick_matches_json_dumps():
    rng = np.random.default_rng(0)
    frames = rng.uniform(0, 1, (128, 52)).astype(np.float32) * 0.02
    ts = 1728000000.0 + np.arange(128) / 30.0
    assert format_frames(frames, ts) == ref(frames, ts)
    assert format_frames(frames, ts, newline=True) == ref(frames, ts, newline=True)
    raw, off = format_frames_raw(frames, ts, newline=True)
    assert raw.decode().count("\n") == 128 and off[0] == 0 and off[-1] == len(raw)
    assert json.loads(raw.decode().splitlines()[5])["blendshapes"] == frames[5].tolist()      # and it parses back exactly


def test_repr_layout_switch_points_and_specials():
    vals = [0.0, -0.0, 1.0, -1.0, 0.5, 0.1, 1e-4, 9.999e-5, 1e-5, 1.5e-7, 123456.789, 1e15, 9.9999998e15, 1e16, 1.2e16,
            3.4028235e38, -3.4028235e38, 1.17549435e-38, 1e-45, 7e-45, 16777216.0, 0.009999999776482582, 100.0, 1e22,
            float("inf"), -float("inf"), float("nan")]
    frames = np.asarray(vals, np.float32)[None, :]
    ts = [0.0]
    got = format_frames(frames, ts)[0]
    assert got == ref(frames, ts)[0]
    for t in (0.0, 1.5, 1728000000.123456, 1e-7, 1e16, 1e17, 2.5e-5, float(np.float64(1e15) + 0.3)):
        assert format_frames(frames[:, :3], [t]) == ref(frames[:, :3], [t])


@settings(max_examples=300, deadline=None)
@given(st.lists(st.floats(width=32, allow_nan=True, allow_infinity=True), min_size=1, max_size=60),
       st.floats(min_value=0.0, max_value=4e9, allow_nan=False))
def test_any_float32_row_matches_json_dumps(row, t):
    frames = np.asarray(row, np.float32)[None, :]
    assert format_frames(frames, [t]) == ref(frames, [t])


def test_empty_and_one_dimensional_inputs():
    assert format_frames(np.zeros((0, 52), np.float32), []) == []
    one = format_frames(np.full(52, 0.25, np.float32), 3.0)
    assert one == ref(np.full((1, 52), 0.25, np.float32), [3.0])


def test_capacity_and_argument_errors():
    import ctypes
    from koemorph_amd._lib import load
    lib = load()
    frames = np.linspace(0, 1, 104, dtype=np.float32).reshape(2, 52)
    ts = np.array([1.0, 2.0])
    small = ctypes.create_string_buffer(100)
    rc = lib.km_format_frames(frames.ctypes.data, 2, 52, ts.ctypes.data, 0, small, 100, None)
    assert rc < -100                                   # -(bytes that certainly suffice)
    big = ctypes.create_string_buffer(-rc)
    n = lib.km_format_frames(frames.ctypes.data, 2, 52, ts.ctypes.data, 0, big, -rc, None)
    assert 0 < n <= -rc and big.raw[:n] == b"".join(ref(frames, ts))
    assert lib.km_format_frames(None, 2, 52, ts.ctypes.data, 0, big, -rc, None) == -1        # KM_ERR_INVALID_ARG
    assert lib.km_format_frames(frames.ctypes.data, 2, 52, None, 0, big, -rc, None) == -1
    assert lib.km_format_frames(frames.ctypes.data, 0, 52, ts.ctypes.data, 0, big, -rc, None) == 0
