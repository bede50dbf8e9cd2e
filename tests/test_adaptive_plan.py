"""Stride schedule and window order of the AdaptiveSequentialDataset mirror (host logic, no GPU) against the restated
reference in oracle/dataset.py (src/data/adaptive_sequential_dataset.py:111-126, 156-283)."""
import numpy as np
import pytest

from koemorph_amd.data import calculate_stride, window_plan
from oracle import dataset as od


@pytest.mark.parametrize("mode", ["dense", "sparse", "progressive", "mixed"])
def test_stride_schedule(mode):
    for initial, final, max_epochs in [(32, 1, 100), (16, 2, 10), (8, 8, 1), (5, 1, 2)]:
        for epoch in list(range(0, max_epochs + 3)):
            assert calculate_stride(mode, initial, final, epoch, max_epochs) == od.adaptive_stride(mode, initial, final, epoch, max_epochs)
    # the documented end points of the progressive schedule (:118-121)
    assert calculate_stride("progressive", 32, 1, 0, 100) == 32 and calculate_stride("progressive", 32, 1, 99, 100) == 1
    assert calculate_stride("progressive", 32, 1, 500, 100) == 1
    with pytest.raises(ValueError):
        calculate_stride("nope", 32, 1, 0, 100)


@pytest.mark.parametrize("mode,stride", [("dense", 1), ("sparse", 32), ("sparse", 7), ("progressive", 13), ("mixed", 32), ("mixed", 5)])
@pytest.mark.parametrize("n_frames,audio_extra", [(300, 0), (256, 0), (255, 0), (257, 1), (611, -1), (420, 2000)])
def test_window_plan_matches_reference_order(mode, stride, n_frames, audio_extra):
    hop, W = 7, 256                                  # a small hop keeps the arrays small; the logic is hop-independent
    labels = np.arange(n_frames * 3, dtype=np.float32).reshape(n_frames, 3)
    audio = np.arange(n_frames * hop + audio_extra, dtype=np.float32)
    np.random.seed(1234)
    ref = list(od.adaptive_windows(audio, labels, mode, stride, stride, 0.1, W, hop))
    # the mirror applies the alignment rule when it loads the clip, then plans on the aligned lengths
    a, l = audio, labels
    expected = len(a) // hop
    if abs(len(l) - expected) > 1:
        n = min(len(l), expected)
        a, l = a[:n * hop], l[:n]
    np.random.seed(1234)
    plan = [(i, s, d) for i, s, d in window_plan(len(l), W, mode, stride, stride, 0.1) if (s + W) * hop <= len(a)]
    assert [(i, s, d) for i, s, d, _, _ in ref] == plan
    for (i, s, d, aw, bw) in ref:
        assert np.array_equal(aw, a[s * hop:(s + W) * hop]) and np.array_equal(bw, l[s:s + W])
    if mode == "dense" and n_frames >= W and audio_extra >= 0:
        assert len(plan) == len(l) - W + 1
