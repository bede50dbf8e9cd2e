"""HIP KoeMorphModel (km_koemorph_*) against golden vectors from the reference's own module and against the oracle."""
import ctypes as C
import glob
import os

import numpy as np
import pytest
import torch

from koemorph_amd import _lib
from koemorph_amd.model import KoeMorphModel, create_koemorph_model
from oracle import koemorph_model as okm
from test_oracle_koemorph import GOLDEN, assert_same, case_frames, case_mask, load_case

pytestmark = pytest.mark.gpu
TOL = 1e-4          # north-star tolerance on the 52 coefficients; observed <= 3e-6


def build(cfg: okm.KoeMorphConfig, params):
    m = KoeMorphModel(mel_dim=cfg.mel_dim, emotion_dim=cfg.emotion_dim, d_model=cfg.d_model, d_query=cfg.d_model,
                      num_heads=cfg.num_heads, num_encoder_layers=cfg.num_encoder_layers,
                      num_attention_layers=cfg.num_attention_layers, decoder_hidden_dim=cfg.decoder_hidden_dim,
                      decoder_layers=cfg.decoder_layers, decoder_activation=cfg.decoder_activation,
                      output_activation=cfg.output_activation, smoothing_method=cfg.smoothing_method,
                      use_temporal_smoothing=cfg.use_temporal_smoothing, use_constraints=cfg.use_constraints, causal=cfg.causal,
                      window_size=cfg.window_size)
    sd = m.state_dict()
    sd.update({k: torch.from_numpy(np.asarray(v)) for k, v in params.items()})
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval()


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_two_frames_match_reference_golden(path):
    z, cfg, params, (mel1, emo1), (mel2, emo2) = load_case(path)
    m = build(cfg, params)
    am = case_mask(z, mel1.shape[1])
    am = None if am is None else torch.from_numpy(am).cuda()
    with torch.no_grad():
        o1 = m(torch.from_numpy(mel1).cuda(), torch.from_numpy(emo1).cuda(), audio_mask=am, return_attention=True)
        o2 = m(torch.from_numpy(mel2).cuda(), torch.from_numpy(emo2).cuda(), audio_mask=am, prev_blendshapes=o1["blendshapes"],
               return_attention=True)
    for tag, o in (("f1", o1), ("f2", o2)):
        assert_same(o["blendshapes"].cpu().numpy(), z[tag + "/blendshapes"], TOL)
        assert_same(o["raw_blendshapes"].cpu().numpy(), z[tag + "/raw_blendshapes"], TOL)
        assert len(o["attention_weights"]) == cfg.num_attention_layers
        for li, w in enumerate(o["attention_weights"]):
            assert_same(w.cpu().numpy()[:, :, ::13, :], z[f"{tag}/attn{li}"], TOL)


@pytest.mark.parametrize("path", [p for p in GOLDEN if "swish" in p or "leaky" in p or "none" in p],
                         ids=lambda p: os.path.basename(p)[:-4])
def test_activation_variants_match_reference_golden(path):
    """decoder_activation swish / leaky_relu, output_activation tanh / none: three chained frames of the reference module."""
    z, cfg, params, _, _ = load_case(path)
    m = build(cfg, params)
    prev = None
    for i, (mel, emo) in enumerate(case_frames(path)):
        with torch.no_grad():
            o = m(torch.from_numpy(mel).cuda(), torch.from_numpy(emo).cuda(), prev_blendshapes=prev)
        prev = o["blendshapes"]
        assert_same(o["blendshapes"].cpu().numpy(), z[f"f{i + 1}/blendshapes"], TOL)
        assert_same(o["raw_blendshapes"].cpu().numpy(), z[f"f{i + 1}/raw_blendshapes"], TOL)


@pytest.mark.parametrize("method,T", [("gaussian", 20), ("median", 20), ("median", 256)])
def test_windowed_smoothers_match_oracle(method, T):
    """TemporalSmoother gaussian / median (decoder.py:294-340) over eight chained frames (the 5-slot ring wraps), and a batch
    size change in between (the history of batch element 0 carries on, :333-337).  PARITY UNPINNED: the reference raises on
    the first call of these methods (decoder.py:339), the oracle restates what the code means.  T = 256 with the default
    mask: rows >= 5 are NaN, and a median over a window that holds a NaN is NaN (torch.median)."""
    from koemorph_amd import synth
    cfg = okm.KoeMorphConfig(d_model=64, num_heads=4, num_encoder_layers=1, num_attention_layers=1, decoder_hidden_dim=32,
                             decoder_layers=1, emotion_dim=24, smoothing_method=method)
    params = okm.make_koemorph_params(91, cfg)
    m = build(cfg, params)
    state, prev_o, prev_g = None, None, None
    for i in range(8):
        B = 3 if i < 5 else 2
        mel, emo = synth.normal(500 + i, (B, T, 80)), synth.normal(600 + i, (B, T, 24))
        if state is not None and state.shape[0] != B:
            state = np.repeat(state[:1], B, axis=0)
            if prev_o is not None:
                prev_o, prev_g = prev_o[:B], prev_g[:B]
        want = okm.koemorph_forward(params, cfg, mel, emo, prev_blendshapes=prev_o, smoother_state=state)
        with torch.no_grad():
            got = m(torch.from_numpy(mel).cuda(), torch.from_numpy(emo).cuda(), prev_blendshapes=prev_g)
        state = want["smoother_state"]
        if T != 256:                                          # (a NaN fed back through prev_blendshapes makes every row NaN)
            prev_o, prev_g = want["blendshapes"], got["blendshapes"]
        assert_same(got["blendshapes"].cpu().numpy(), want["blendshapes"], TOL)
        assert_same(got["raw_blendshapes"].cpu().numpy(), want["raw_blendshapes"], TOL)
        if T == 256:
            assert np.isnan(want["blendshapes"][:, 5:]).all() and not np.isnan(want["blendshapes"][:, :5]).any()
    m.reset_temporal_state()                                  # a new sequence starts from an empty ring
    mel, emo = synth.normal(500, (3, T, 80)), synth.normal(600, (3, T, 24))
    with torch.no_grad():
        again = m(torch.from_numpy(mel).cuda(), torch.from_numpy(emo).cuda())
    assert_same(again["blendshapes"].cpu().numpy(), okm.koemorph_forward(params, cfg, mel, emo)["blendshapes"], TOL)


def test_options_against_oracle():
    cfg = okm.KoeMorphConfig(d_model=64, num_heads=4, num_encoder_layers=1, num_attention_layers=2, decoder_hidden_dim=32,
                             decoder_layers=2, emotion_dim=24, window_size=9)
    params = okm.make_koemorph_params(71, cfg)
    m = build(cfg, params)
    from koemorph_amd import synth
    mel, emo = synth.normal(1, (5, 40, 80)), synth.normal(2, (5, 40, 24))
    prev = synth.uniform(3, (5, 52), 0, 1)
    with torch.no_grad():
        a = m(torch.from_numpy(mel).cuda(), torch.from_numpy(emo).cuda(), prev_blendshapes=torch.from_numpy(prev).cuda(),
              apply_smoothing=False, apply_constraints=False)
        b = m(torch.from_numpy(mel).cuda(), torch.from_numpy(emo).cuda(), apply_smoothing=True, apply_constraints=True)
    wa = okm.koemorph_forward(params, cfg, mel, emo, prev_blendshapes=prev, apply_smoothing=False, apply_constraints=False)
    wb = okm.koemorph_forward(params, cfg, mel, emo)
    np.testing.assert_allclose(a["blendshapes"].cpu().numpy(), wa["blendshapes"], atol=TOL)
    np.testing.assert_array_equal(a["blendshapes"].cpu().numpy(), a["raw_blendshapes"].cpu().numpy())
    np.testing.assert_allclose(b["blendshapes"].cpu().numpy(), wb["blendshapes"], atol=TOL)
    # attention rows sum to one (reference tests/model/test_attention.py:53-56), outputs in [0, 1]
    with torch.no_grad():
        o = m(torch.from_numpy(mel).cuda(), torch.from_numpy(emo).cuda(), return_attention=True)
    for w in o["attention_weights"]:
        np.testing.assert_allclose(w.sum(-1).cpu().numpy(), 1.0, atol=1e-5)
    bs = o["blendshapes"].cpu().numpy()
    assert (bs >= 0).all() and (bs <= 1).all()
    # a new sequence starts the smoother from zero again
    m.reset_temporal_state()
    with torch.no_grad():
        c = m(torch.from_numpy(mel).cuda(), torch.from_numpy(emo).cuda())
    np.testing.assert_array_equal(c["blendshapes"].cpu().numpy(), b["blendshapes"].cpu().numpy())


def test_inference_step_chain_and_factory():
    m = create_koemorph_model({"d_model": 64, "d_query": 64, "num_heads": 8, "num_encoder_layers": 1, "num_attention_layers": 1,
                               "decoder_hidden_dim": 32, "emotion_dim": 16}).cuda().eval()
    assert m.get_model_info()["num_attention_layers"] == 1 and m.get_num_parameters() > 0
    prev = None
    for t in range(3):                                    # T = 1 frames, as scripts/rt.py feeds them
        mel = torch.randn(1, 1, 80, device="cuda")
        emo = torch.randn(1, 1, 16, device="cuda")
        prev = m.inference_step(mel, emo, prev)
        assert prev.shape == (1, 52) and torch.isfinite(prev).all()


def test_error_paths():
    with pytest.raises(ValueError):
        KoeMorphModel()                                   # d_query 128 != d_model 256: the reference's default does not run
    with pytest.raises(ValueError):
        KoeMorphModel(d_model=100, d_query=100, num_heads=8)
    for bad in (dict(smoothing_method="kalman"), dict(decoder_activation="mish"), dict(output_activation="softmax")):
        with pytest.raises(ValueError, match="Unknown"):  # the reference's messages (decoder.py:76-77, :168-169, :272-273)
            KoeMorphModel(d_query=256, **bad)
    m = KoeMorphModel(d_model=64, d_query=64, num_heads=4, num_encoder_layers=0, num_attention_layers=1, decoder_hidden_dim=32,
                      emotion_dim=8).cuda().eval()
    with pytest.raises(ValueError):
        m(torch.zeros(1, 4, 80).cuda(), torch.zeros(1, 4, 8).cuda(), audio_mask=torch.ones(1, 5, dtype=torch.bool).cuda())
    with pytest.raises(ValueError):
        m(torch.zeros(1, 4, 80).cuda(), torch.zeros(1, 5, 8).cuda())
    with torch.no_grad():
        assert m(torch.zeros(2, 4, 80).cuda(), torch.zeros(2, 4, 8).cuda())["blendshapes"].shape == (2, 52)
        with pytest.raises(RuntimeError, match="reset_temporal_state"):       # smoother state of batch 2, now batch 3
            m(torch.zeros(3, 4, 80).cuda(), torch.zeros(3, 4, 8).cuda())
        m.reset_temporal_state()
        assert m(torch.zeros(3, 4, 80).cuda(), torch.zeros(3, 4, 8).cuda())["blendshapes"].shape == (3, 52)
        m.reset_temporal_state()
        m(torch.zeros(2, 4, 80).cuda(), torch.zeros(2, 4, 8).cuda())
    lib = _lib.load()
    h = m._h
    with pytest.raises(_lib.KoeMorphError):               # workspace was reserved for 2 x 4 frames
        _lib.check(lib.km_koemorph_forward(h, C.c_void_p(8), C.c_void_p(8), 64, 64, None, None, None, 0, C.c_void_p(8), None, None, None))
    with pytest.raises(_lib.KoeMorphError):
        _lib.check(lib.km_reserve(h, 1, 16000))           # audio workspace entry point on a KoeMorphModel handle


@pytest.mark.parametrize("T", [1, 3, 10, 70, 130])
def test_key_axis_lengths_against_oracle(T):
    """Every lane grouping of the masked softmax (1 / 4 / 16 / 64 lanes per row) and the > 128-key row softmax of the encoder."""
    from koemorph_amd import synth
    cfg = okm.KoeMorphConfig(d_model=64, num_heads=4, num_encoder_layers=1, num_attention_layers=2, decoder_hidden_dim=32,
                             decoder_layers=1, emotion_dim=24, window_size=None if T == 130 else 30, causal=T != 130)
    params = okm.make_koemorph_params(80 + T, cfg)
    m = build(cfg, params)
    mel, emo = synth.normal(T, (7, T, 80)), synth.normal(T + 1, (7, T, 24))
    with torch.no_grad():
        o = m(torch.from_numpy(mel).cuda(), torch.from_numpy(emo).cuda(), return_attention=True)
    w = okm.koemorph_forward(params, cfg, mel, emo)
    assert_same(o["blendshapes"].cpu().numpy(), w["blendshapes"], TOL)
    for a, b in zip(o["attention_weights"], w["attention_weights"]):
        assert_same(a.cpu().numpy(), b, TOL)


def test_padding_is_invisible():
    """A padded batch gives each item the result of its own unpadded forward (reference tests/model/test_koemorph_model.py:77-97
    only checks the shape): frames behind the mask never reach the output, whatever they hold."""
    from koemorph_amd import synth
    cfg = okm.KoeMorphConfig(d_model=64, num_heads=4, num_encoder_layers=2, num_attention_layers=2, decoder_hidden_dim=32,
                             decoder_layers=1, emotion_dim=24, causal=False, window_size=None)
    params = okm.make_koemorph_params(91, cfg)
    m = build(cfg, params)
    T, lens = 24, [24, 17, 5, 1]
    mel, emo = synth.normal(5, (4, T, 80)), synth.normal(6, (4, T, 24))
    mask = torch.arange(T)[None, :] < torch.tensor(lens)[:, None]
    junk = mel.copy(); junk[~mask.numpy()] = 1e3                                  # garbage in the padded frames
    with torch.no_grad():
        padded = m(torch.from_numpy(junk).cuda(), torch.from_numpy(emo).cuda(), audio_mask=mask.cuda(), apply_smoothing=False)["blendshapes"].cpu().numpy()
        for i, n in enumerate(lens):
            alone = m(torch.from_numpy(mel[i:i + 1, :n]).cuda(), torch.from_numpy(emo[i:i + 1, :n]).cuda(), apply_smoothing=False)["blendshapes"].cpu().numpy()
            np.testing.assert_allclose(padded[i:i + 1], alone, atol=2e-6)


@pytest.mark.parametrize("B,T,masked,causal,window,act,prev", [
    (2, 30, False, True, 30, "gelu", True),       # the reference's defaults (the d256 golden's shape)
    (5, 30, True, True, 30, "gelu", True),        # odd batch: the last encoder workgroup holds one window; key padding mask
    (3, 32, True, False, None, "swish", False),   # full tiles, no causal / local-window mask, no conditioning
    (4, 17, False, True, 8, "leaky_relu", True),  # narrow window: the last queries see few keys
    (7, 1, False, True, 30, "relu", True),        # the per-tick form scripts/rt.py feeds
    (3, 9, True, False, 4, "gelu", False),        # slots of 16 rows: one window per row tile
    (70, 3, True, True, 30, "gelu", True),        # slots of 4 rows: 16 windows per encoder workgroup, 5 workgroups per stream
    (9, 16, True, False, None, "relu", True),
])
def test_fused_kernels_match_the_chain_and_the_oracle(B, T, masked, causal, window, act, prev):
    """The default width runs two fused kernels (km_kmmf.hip); option kmm_no_fuse runs the launch-per-step chain on the same
    handle.  Both against the oracle (pinned by the reference goldens), two chained frames, attention weights included."""
    from koemorph_amd import synth
    cfg = okm.KoeMorphConfig(causal=causal, window_size=window, decoder_activation=act)
    params = okm.make_koemorph_params(17 + B, cfg)
    m = build(cfg, params)
    lib, h, _ = m._handle()
    am = None
    if masked:                                        # ragged tails, and leading padding on element 0
        am = np.ones((B, T), dtype=bool)
        for b in range(B):
            if b % 3:
                am[b, T - (b % 3):] = False
        am[0, : T // 3] = False
    res = {}
    for mode in ("fused", "chain"):
        _lib.check(lib.km_set_option(h, b"kmm_no_fuse", 0 if mode == "fused" else 1))
        m.reset_temporal_state()
        outs, pb = [], None
        for f in range(2):
            mel, emo = synth.normal(300 + f, (B, T, 80)), synth.normal(400 + f, (B, T, 256))
            with torch.no_grad():
                o = m(torch.from_numpy(mel).cuda(), torch.from_numpy(emo).cuda(), audio_mask=None if am is None else torch.from_numpy(am).cuda(),
                      prev_blendshapes=pb if prev else None, return_attention=True)
            pb = o["blendshapes"]
            outs.append({"b": o["blendshapes"].cpu().numpy(), "r": o["raw_blendshapes"].cpu().numpy(),
                         "a": [w.cpu().numpy() for w in o["attention_weights"]]})
        res[mode] = outs
    _lib.check(lib.km_set_option(h, b"kmm_no_fuse", 0))
    state, pb = None, None
    for f in range(2):
        mel, emo = synth.normal(300 + f, (B, T, 80)), synth.normal(400 + f, (B, T, 256))
        g = okm.koemorph_forward(params, cfg, mel, emo, audio_mask=am, prev_blendshapes=pb if prev else None, smoother_state=state)
        g = {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in g.items()}
        state, pb = g["smoother_state"], g["blendshapes"]
        for mode in ("fused", "chain"):
            o = res[mode][f]
            assert_same(o["b"], g["blendshapes"], TOL)
            assert_same(o["r"], g["raw_blendshapes"], TOL)
            for li in range(cfg.num_attention_layers):
                assert_same(o["a"][li], np.asarray(g["attention_weights"][li]), TOL)


@pytest.mark.parametrize("B,T", [(6, 4), (5, 30), (40, 1)])
def test_fused_fully_masked_window_stays_in_its_window(B, T):
    """A window whose every frame is padding is NaN (softmax over no key, as torch) -- and only that window: with T <= 16 several
    windows share an MFMA row tile in the fused encoder, where 0 x NaN of a neighbour's value rows would otherwise leak."""
    from koemorph_amd import synth
    cfg = okm.KoeMorphConfig()
    params = okm.make_koemorph_params(23, cfg)
    m = build(cfg, params)
    am = np.ones((B, T), dtype=bool)
    am[2, :] = False
    am[B - 1, :] = False
    mel, emo = synth.normal(310, (B, T, 80)), synth.normal(410, (B, T, 256))
    with torch.no_grad():
        o = m(torch.from_numpy(mel).cuda(), torch.from_numpy(emo).cuda(), audio_mask=torch.from_numpy(am).cuda(), return_attention=True)
    g = okm.koemorph_forward(params, cfg, mel, emo, audio_mask=am)
    got = o["blendshapes"].cpu().numpy()
    want = np.asarray(g["blendshapes"])
    dead = np.isnan(want).all(axis=1)
    assert dead[2] and dead[B - 1] and dead.sum() == 2
    assert_same(got, want, TOL)
    for li in range(cfg.num_attention_layers):
        assert_same(o["attention_weights"][li].cpu().numpy(), np.asarray(g["attention_weights"][li]), TOL)
