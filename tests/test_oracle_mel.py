"""oracle/mel.py is PARITY UNPINNED by the reference (librosa / torchaudio absent, no mel
fixtures in the reference).  These tests cross-check its building blocks against the
independent pure-numpy implementation in transformers.audio_utils, and check the
frame-count quirks SURVEY.md section 7 lists."""
import numpy as np
import pytest

from koemorph_amd import synth
from oracle import mel

au = pytest.importorskip("transformers.audio_utils")


def test_slaney_filterbank_matches_transformers():
    fb = mel.mel_filterbank_librosa(16000, 1024, 80, 80.0, 8000.0)           # (80, 513)
    ref = au.mel_filter_bank(num_frequency_bins=513, num_mel_filters=80, min_frequency=80.0,
                             max_frequency=8000.0, sampling_rate=16000, norm="slaney",
                             mel_scale="slaney").T
    np.testing.assert_allclose(fb, ref, atol=1e-7, rtol=1e-5)
    fb512 = mel.mel_filterbank_librosa(16000, 512, 80, 80.0, 8000.0)
    ref512 = au.mel_filter_bank(257, 80, 80.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").T
    np.testing.assert_allclose(fb512, ref512, atol=1e-7, rtol=1e-5)


def test_htk_filterbank_matches_transformers():
    fb = mel.mel_filterbank_torchaudio(257, 80.0, 8000.0, 80, 16000)           # (257, 80)
    ref = au.mel_filter_bank(257, 80, 80.0, 8000.0, 16000, norm=None, mel_scale="htk")
    np.testing.assert_allclose(fb, ref, atol=1e-6, rtol=1e-5)


def test_power_to_db_matches_transformers():
    S = np.abs(synth.normal(3, (257, 80))).astype(np.float32) ** 2 + 1e-12
    mine = mel.power_to_db(S)
    ref = au.power_to_db(S, reference=float(S.max()), min_value=1e-10, db_range=80.0)
    np.testing.assert_allclose(mine, ref, atol=2e-4)


def test_stft_power_matches_transformers_spectrogram():
    y = synth.make_audio(5, 1, 16000)[0]
    P = mel.stft_power(y, 1024, 533, center=True, pad_mode="constant", precision="f64")
    ref = au.spectrogram(y.astype(np.float64), window=mel.hann_periodic(1024), frame_length=1024,
                         hop_length=533, fft_length=1024, power=2.0, center=True,
                         pad_mode="constant", onesided=True).T
    assert P.shape == ref.shape
    np.testing.assert_allclose(P, ref, rtol=1e-6, atol=1e-9 * ref.max())


def test_frame_count_quirks():
    # 136448-sample window -> 257 frames; the last 3 are taken BEFORE the core truncates to 256
    y = synth.make_audio(1, 1, 136448)[0]
    long, short = mel.mel_batch_window(y)
    assert long.shape == (257, 80) and short.shape == (3, 80)
    assert np.array_equal(short, long[254:257])
    assert long.max() == pytest.approx(1.0) and long.min() >= 0.0
    # real-time ring: 136000 samples, hop 532 -> 256 frames -> truncated to int(8.5/0.0333)=255
    y2 = synth.make_audio(2, 1, 136000)[0]
    db = mel.mel_sliding_window(y2)
    assert db.shape == (255, 80) and db.max() == 0.0 and db.min() >= -80.0
    # torchaudio path: 136000 samples -> 256 raw frames -> int(8.5*30) = 255 kept
    lm = mel.mel_torchaudio(y2)
    assert lm.shape == (1, 255, 80)
    # 60 fps hop is int(16000/60) = 266
    assert mel.num_frames(512 * 266, 1024, 266) == 513


def test_ref_precision_close_to_f64():
    y = synth.make_audio(7, 1, 136448)[0]
    a, _ = mel.mel_batch_window(y, precision="ref")
    b, _ = mel.mel_batch_window(y, precision="f64")
    assert np.max(np.abs(a - b)) < 5e-5
