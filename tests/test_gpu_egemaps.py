"""GPU eGeMAPSv02 front end (km_egemaps_*) against oracle/egemaps.py -- PARITY UNPINNED against openSMILE
itself (third-party, absent, see the oracle's header) -- and against known answers on synthetic vowels.

Tolerances: the oracle runs in float64 with numpy's FFT and np.roots, the kernels in float32 with an LDS radix-2 FFT and a
Durand-Kerner root finder.  Continuous descriptors agree to 1e-3 relative or better; descriptors behind a discrete decision
(which candidate, where a pitch mark lands) are compared on the frames where the decisions coincide, and the agreement rate
itself is asserted.

Agreement thresholds (round 4): tools/egemaps_agreement.py measures, over eight seeds of the speech-like test signal (195
frames, 135 voiced in both), voicing decisions 100 % equal, all three formants valid in both on 100 % of the voiced frames
and pitch marks (jitter) equal on 100 % of them.  The assertions below allow ONE OR TWO frames of 195 / 135 to differ (0.99 /
0.97) -- a frame whose normalised autocorrelation sits within float32 rounding of the 0.55 voicing cutoff, or whose two best
pitch-mark lags tie, may legitimately fall on either side -- instead of the 2 % / 10 % of rounds 2 - 3."""
import numpy as np
import pytest
import torch

from koemorph_amd import synth
from koemorph_amd.features.opensmile_extractor import EGeMAPSEngine, OpenSMILEeGeMAPSExtractor
from oracle import egemaps as eg

pytestmark = pytest.mark.gpu

R = dict(loud=0, alpha=1, hamm=2, sl0=3, sl1=4, flux=5, mfcc=6, rms=10, cf=11, cs=14, voi=17, F=18, BW=21, f0=24, jit=25, shim=26,
         hnr=27, h1h2=28, h1a3=29, famp=30)


@pytest.fixture(scope="module")
def engine():
    return EGeMAPSEngine("cuda")


def speechlike(seed, seconds=2.0):
    """voiced - silence - noise - voiced with vibrato: exercises voiced / unvoiced functionals and segment statistics."""
    a = synth.make_vowel(seed, 130.0, seconds * 0.35, vibrato=0.03)
    b = np.zeros(int(seconds * 0.1 * 16000), np.float32)
    c = (0.2 * synth.normal(seed + 2, (int(seconds * 0.2 * 16000),))).astype(np.float32)
    d = 0.6 * synth.make_vowel(seed + 3, 190.0, seconds * 0.35, formants=((500.0, 80.0), (1500.0, 120.0), (2500.0, 150.0)), vibrato=0.02)
    return np.concatenate([a, b, c, d]).astype(np.float32)


def test_low_level_descriptors_match_oracle(engine):
    x = speechlike(11)
    feats = engine.functionals(torch.from_numpy(x[None]).cuda(), normalize=True)
    rec = engine.records()[0]
    d = eg.llds(eg.normalise(x))
    nf = len(d["f0"])
    assert rec.shape == (nf, 36) and feats.shape == (1, 88)
    live = d["rms"] > 1e-4                                                     # digital silence: log quantities sit on their floors
    for name, col in (("loudness", R["loud"]), ("alphaRatio", R["alpha"]), ("hammarbergIndex", R["hamm"]), ("slope0-500", R["sl0"]),
                      ("slope500-1500", R["sl1"]), ("spectralFlux", R["flux"])):
        np.testing.assert_allclose(rec[live, col], d[name][live], rtol=2e-3, atol=2e-3 * np.abs(d[name][live]).max(), err_msg=name)
    np.testing.assert_allclose(rec[live, R["mfcc"]:R["mfcc"] + 4], d["mfcc"][live], rtol=2e-3, atol=2e-2)
    np.testing.assert_allclose(rec[:, R["rms"]], d["rms"], rtol=1e-4, atol=1e-7)
    # pitch: same voicing decision on (almost) every frame, same F0 where both are voiced
    gv, ov = rec[:, R["f0"]] > 0, d["f0"] > 0
    assert (gv == ov).mean() >= 0.99 and ov.sum() > 50
    both = gv & ov
    np.testing.assert_allclose(rec[both, R["f0"]], d["f0"][both], rtol=2e-3)
    np.testing.assert_allclose(rec[both, R["voi"]], d["voicing"][both], atol=5e-3)
    # formants (voiced frames): the same roots
    for i in range(3):
        ok = both & (rec[:, R["F"] + i] > 0) & (d["F"][:, i] > 0)
        assert ok.sum() >= 0.97 * both.sum()
        np.testing.assert_allclose(rec[ok, R["F"] + i], d["F"][ok, i], rtol=5e-3, atol=2.0)
        assert np.median(np.abs(rec[ok, R["BW"] + i] - d["BW"][ok, i])) < 2.0
    np.testing.assert_allclose(rec[both, R["hnr"]], d["HNRdBACF"][both], atol=0.2)
    np.testing.assert_allclose(rec[both, R["h1h2"]], d["H1-H2"][both], atol=0.05)
    # jitter / shimmer depend on where the pitch marks land: compare where they agree, and require that to be the rule
    same = both & (np.abs(rec[:, R["jit"]] - d["jitterLocal"]) < 1e-4)
    assert same.sum() >= 0.97 * both.sum()
    np.testing.assert_allclose(rec[same, R["shim"]], d["shimmerLocaldB"][same], atol=2e-3)


def test_functionals_match_oracle_and_known_answers(engine):
    for seed, f0 in ((21, 120.0), (22, 200.0)):
        x = synth.make_vowel(seed, f0, 1.5, vibrato=0.02)
        got = engine.functionals(torch.from_numpy(x[None]).cuda())[0].cpu().numpy()
        want = eg.functionals(eg.normalise(x))
        assert np.isfinite(got).all()
        # known answers straight from the GPU: mean pitch in semitones above 27.5 Hz, formants of the synthetic tract
        assert abs(got[0] - 12 * np.log2(f0 / 27.5)) < 0.3
        assert abs(got[40] - 700) < 90 and abs(got[46] - 1200) < 90 and (f0 > 150 or abs(got[52] - 2600) < 120)
        assert got[34] > 8.0 and got[30] < 0.02                                   # harmonic, steady
        scale = np.maximum(np.abs(want), 1e-3)
        rel = np.abs(got - want) / scale
        tight = [i for i in range(88) if not any(k in eg.FEATURE_NAMES[i] for k in ("Slope", "stddevNorm", "jitter", "shimmer"))]
        assert rel[tight].max() < 2e-2, [(eg.FEATURE_NAMES[i], got[i], want[i]) for i in tight if rel[i] >= 2e-2]
        assert np.median(rel) < 2e-3


def test_mixed_signal_functionals(engine):
    x = speechlike(31, 3.0)
    got = engine.functionals(torch.from_numpy(x[None]).cuda())[0].cpu().numpy()
    want = eg.functionals(eg.normalise(x))
    # segment statistics and unvoiced-frame means are exact counts / plain means
    for i in (81, 82, 83, 84, 85, 86):
        assert abs(got[i] - want[i]) <= 0.05 * max(abs(want[i]), 0.05), eg.FEATURE_NAMES[i]
    assert abs(got[87] - want[87]) < 0.01                                       # equivalent sound level (dB)
    for i in range(76, 81):
        assert abs(got[i] - want[i]) <= 2e-2 * max(abs(want[i]), 1.0), eg.FEATURE_NAMES[i]
    assert got[82] >= 1.0 / 3.0 and got[85] > 0.05                              # two voiced stretches, a pause between them


def test_batching_is_exact_and_normalisation_is_peak(engine):
    xs = np.stack([synth.make_vowel(40 + i, 100.0 + 30 * i, 1.0) * (0.2 + 0.2 * i) for i in range(5)])
    t = torch.from_numpy(xs).cuda()
    batch = engine.functionals(t)
    for i in range(5):
        assert torch.equal(engine.functionals(t[i:i + 1])[0], batch[i])          # a window's result does not depend on its neighbours
    scaled = engine.functionals(3.0 * t)
    # peak normalisation (opensmile_extractor.py:431-433): x and 3 x give the same features up to the rounding of 1 / max
    robust = [0, 2, 3, 4, 10, 12, 13, 14, 20, 22, 24, 26, 28, 34, 40, 46, 58, 60, 62, 64, 87]     # means / percentiles, no decision-sensitive ones
    assert float(((scaled - batch).abs() / (batch.abs() + 1e-2))[:, robust].max()) < 2e-2
    raw = engine.functionals(t, normalize=False)
    assert float((raw[:4, 87] - batch[:4, 87]).abs().min()) > 1.0                # ... which the level features see when it is off
    assert float((raw[4, 87] - batch[4, 87]).abs()) < 1e-3                       # (the fifth window already peaks at 1)
    with pytest.raises(Exception):
        engine.functionals(torch.zeros(1, 500).cuda())                           # shorter than one 60 ms frame
    with pytest.raises(Exception):
        engine.functionals(torch.zeros(1, 16000 * 21).cuda())                    # more than 2048 frames


def test_extractor_mirror_state_machine():
    clock = [100.0]
    ex = OpenSMILEeGeMAPSExtractor(context_window=2.0, update_interval=0.3, use_concatenation=True, temporal_history_frames=4,
                                   clock=lambda: clock[0])
    assert ex.feature_dim == 88 and len(ex.get_feature_names()) == 88
    x = speechlike(51, 3.0)
    outs = []
    for k in range(0, len(x) - 1600, 1600):                                     # 100 ms chunks
        clock[0] += 0.125                                                        # exactly representable: an update every third chunk
        f = ex.process_audio_frame(x[k:k + 1600])
        outs.append(None if f is None else f.copy())
    assert outs[0] is None or outs[0].shape == (88,)                            # 0.1 s of audio: below the 0.5 s minimum -> no features yet
    first = next(i for i, f in enumerate(outs) if f is not None)
    assert first == 4                                                           # 0.5 s of audio
    # between updates (every 0.3 s) the cached vector is returned
    changed = [i for i in range(first + 1, len(outs)) if not np.array_equal(outs[i], outs[i - 1])]
    assert all(b - a == 3 for a, b in zip(changed, changed[1:]))
    # the last vector is the functionals of the last 2 s of audio
    n_in = (len(outs)) * 1600
    last_update = changed[-1]
    win = x[:(last_update + 1) * 1600][-32000:]
    np.testing.assert_allclose(outs[last_update], eg.functionals(eg.normalise(win)), rtol=5e-2, atol=5e-2)
    th = ex.get_temporal_features()
    assert th.shape == (4, 88) and np.array_equal(th[-1], ex.current_features)
    # 3-window concatenation + Linear(264, 256): slots 0.3 / 0.6 hold the FIRST features (reference quirk), slot 0 the current ones
    cc = ex.get_concatenated_features()
    cat = np.concatenate([ex.window_features[0.0], ex.window_features[0.3], ex.window_features[0.6]])
    assert np.array_equal(ex.window_features[0.3], outs[first]) and np.array_equal(ex.window_features[0.0], ex.current_features)
    want = torch.nn.functional.linear(torch.from_numpy(cat)[None], ex.compression_layer.weight, ex.compression_layer.bias)[0].detach().numpy()
    np.testing.assert_allclose(cc, want, rtol=1e-4, atol=5e-4)      # fp32 products of features up to a few hundred
    ex.reset()
    assert ex.current_features is None and ex.get_temporal_features() is None and ex.get_concatenated_features() is None
    with pytest.raises(ValueError):
        OpenSMILEeGeMAPSExtractor(update_interval=0.05)
    (n_in)


def test_batched_emotion_vectors_equal_the_reference_procedure():
    """emotion_features_batch (two batched extractions + one Linear) against the reference's per-sample procedure driven
    through the mirror's state machine: process_audio_batch (an extraction per 0.3 s frame on the growing window) followed by
    get_concatenated_features (emotion_extractor.py:443-447)."""
    xs = np.stack([speechlike(61 + i, 1.5)[:24000] for i in range(3)])
    slow = OpenSMILEeGeMAPSExtractor(context_window=20.0, update_interval=0.3, use_concatenation=True)
    torch.manual_seed(5)
    slow.compression_layer = torch.nn.Linear(264, 256)
    want = []
    for x in xs:
        slow.process_audio_batch(x)
        want.append(slow.get_concatenated_features())
    fast = OpenSMILEeGeMAPSExtractor(context_window=20.0, update_interval=0.3, use_concatenation=True)
    fast.compression_layer = slow.compression_layer
    got = fast.emotion_features_batch(xs).cpu().numpy()
    assert got.shape == (3, 256)
    np.testing.assert_allclose(got, np.stack(want), rtol=1e-4, atol=2e-3)
    assert np.array_equal(fast.window_features[0.3], slow.window_features[0.3])       # both hold the first extraction ever made
    # usable as the model's emotion provider: (B, L) audio -> (B, 256)
    from koemorph_amd.model import SimplifiedDualStreamModel
    m = SimplifiedDualStreamModel(emotion_provider=fast.emotion_features_batch).cuda().eval()
    with torch.no_grad():
        out = m(torch.from_numpy(synth.make_audio(7, 2, 136448)).cuda())["blendshapes"]
    assert out.shape == (2, 52) and bool(torch.isfinite(out).all())
