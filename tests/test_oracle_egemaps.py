"""oracle/egemaps.py (PARITY UNPINNED against openSMILE, see its header) on signals with KNOWN answers, and the host logic
of the extractor mirror (AudioBuffer) against a straightforward restatement."""
import numpy as np
import pytest

from koemorph_amd import synth
from koemorph_amd.egemaps_names import FEATURE_NAMES
from oracle import egemaps as eg


def test_feature_names_and_count():
    assert FEATURE_NAMES == eg.FEATURE_NAMES and len(FEATURE_NAMES) == 88
    assert FEATURE_NAMES[0] == "F0semitoneFrom27.5Hz_sma3nz_amean" and FEATURE_NAMES[10] == "loudness_sma3_amean"
    assert FEATURE_NAMES[30] == "jitterLocal_sma3nz_amean" and FEATURE_NAMES[87] == "equivalentSoundLevel_dBp"


@pytest.mark.parametrize("f0", [110.0, 150.0, 220.0])
def test_known_pitch_formants_and_harmonicity(f0):
    x = synth.make_vowel(3, f0, 1.2)
    d = eg.llds(x)
    v = d["f0"] > 0
    assert v.mean() > 0.95
    assert abs(np.median(d["f0"][v]) - f0) < 0.01 * f0                                  # pitch within 1 %
    F = np.median(d["F"][v], axis=0)
    assert abs(F[0] - 700) < 80 and abs(F[1] - 1200) < 80                                 # resonances of the synthetic tract
    if f0 <= 150:                      # at 220 Hz the harmonics are too sparse for an order-11 LPC to hold the weak third resonance
        assert abs(F[2] - 2600) < 100
    assert np.median(d["HNRdBACF"][v]) > 10.0                                            # a clean harmonic source
    assert np.median(d["jitterLocal"][v]) < 0.012                                        # only the integer-sample period grid
    f = eg.functionals(x)
    assert abs(f[0] - 12 * np.log2(f0 / 27.5)) < 0.2 and f[1] < 0.01                     # semitone mean, tiny variation
    assert f[82] <= 1.0 / 1.0 and f[83] > 1.0                                            # one long voiced segment


def test_jitter_is_seen_and_noise_is_unvoiced():
    clean = eg.llds(synth.make_vowel(5, 140.0, 1.0))
    rough = eg.llds(synth.make_vowel(5, 140.0, 1.0, jitter=0.03))
    vc, vr = clean["f0"] > 0, rough["f0"] > 0
    assert vr.sum() > 10
    assert np.median(rough["jitterLocal"][vr]) > 1.5 * np.median(clean["jitterLocal"][vc])
    assert np.median(rough["HNRdBACF"][vr]) < np.median(clean["HNRdBACF"][vc])
    noise = 0.3 * synth.normal(9, (16000,))
    dn = eg.llds(noise)
    assert (dn["f0"] > 0).mean() < 0.05
    f = eg.functionals(noise)
    assert f[0] == 0 and f[30] == 0 and f[82] < 1.0                                       # no voiced frames: the voiced-only features are zero
    # louder audio -> more loudness and a higher equivalent sound level
    quiet = eg.functionals(0.1 * synth.make_vowel(5, 140.0, 1.0))
    loud = eg.functionals(synth.make_vowel(5, 140.0, 1.0))
    assert loud[10] > quiet[10] and abs((loud[87] - quiet[87]) - 20.0) < 0.1


def test_alpha_ratio_and_slopes_follow_the_spectrum():
    rng_lo = synth.make_vowel(7, 120.0, 0.8, formants=((400.0, 80.0),))       # energy low in the spectrum
    rng_hi = synth.make_vowel(7, 120.0, 0.8, formants=((3000.0, 200.0),))     # ... and high
    a = eg.llds(rng_lo); b = eg.llds(rng_hi)
    assert np.median(a["alphaRatio"]) > np.median(b["alphaRatio"]) + 10.0
    assert np.median(a["hammarbergIndex"]) > np.median(b["hammarbergIndex"]) + 10.0


def test_functionals_helpers():
    v = np.array([0, 1, 3, 2, 0, 4], float)
    r_mean, r_std, f_mean, f_std = eg.slopes(v)
    # rising parts: 0->3 over 2 frames, 0->4 over 1 frame; falling: 3->0 over 2 frames
    assert np.isclose(r_mean, np.mean([3 / 0.02, 4 / 0.01])) and np.isclose(f_mean, -3 / 0.02) and f_std == 0
    assert eg.percentile(np.array([1.0, 2.0, 3.0, 4.0, 5.0]), 0.2) == pytest.approx(1.8)
    assert list(eg.segments(np.array([1, 1, 0, 1, 0, 0, 1, 1, 1], bool))) == [2, 1, 3]
    assert np.allclose(eg.sma3(np.array([0.0, 3, 0, 0, 6, 9]), True), [0, 3, 0, 0, 7.5, 7.5])
    assert np.allclose(eg.sma3(np.array([3.0, 6, 9]), False), [4.5, 6, 7.5])


def test_audio_buffer_mirror_matches_restated_semantics():
    """AudioBuffer of the extractor mirror against oracle/buffers.py AudioBufferOracle (restated from
    src/features/opensmile_extractor.py:29-154) on random traffic, wrap-around included."""
    from koemorph_amd.features.opensmile_extractor import AudioBuffer
    from oracle.buffers import AudioBufferOracle
    rng = np.random.RandomState(0)
    buf, orc = AudioBuffer(max_duration=0.05, sample_rate=16000), AudioBufferOracle(0.05, 16000)      # 800 samples
    for dur in (0.01, None):
        assert np.array_equal(buf.get_window(dur), orc.get_window(dur))
    assert buf.get_stats()["buffer_underruns"] == orc.underruns == 2
    for step in range(80):
        chunk = rng.randn(rng.randint(1, 300)).astype(np.float32)
        buf.append(chunk); orc.append(chunk)
        assert buf.is_full == orc.full and buf.write_pos == orc.w
        for dur in (0.01, 0.03, 0.05, None):
            assert np.array_equal(buf.get_window(dur), orc.get_window(dur)), (step, dur)
    assert buf.get_stats()["total_samples_written"] == orc.total
    with pytest.raises(ValueError):
        buf.append(np.zeros((2, 2), np.float32))
    buf.reset()
    assert buf.get_stats()["total_samples_written"] == 0 and not buf.is_full
