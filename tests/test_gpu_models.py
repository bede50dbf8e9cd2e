"""GPU parity of the reference-interface mirrors (torch.nn.Module classes and the rt.py loop) against the
oracle and the reference's golden outputs.  Everything computes through the C-ABI."""
import json

import numpy as np
import pytest
import torch

from conftest import golden_case
from koemorph_amd import synth
from koemorph_amd.features import MelSlidingWindowExtractor, MelSpectrogramExtractor
from koemorph_amd.model import (DualStreamCrossAttention, SequentialDualStreamModel, SimplifiedDualStreamModel,
                                EXPRESSION_INDICES, MOUTH_INDICES)
from koemorph_amd.scripts import rt
from oracle import mel as omel, models

pytestmark = pytest.mark.gpu


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def full_state(params, alpha=0.8):
    sd = {"dual_stream_attention." + k: torch.from_numpy(v) for k, v in params.items()}
    sd["smoothing_alpha"] = torch.tensor(alpha)
    return sd


def test_module_forward_matches_reference_golden():
    c, params, (mel, short, emo), g = golden_case("core_d256_T256_H8_trained")
    m = DualStreamCrossAttention().cuda().eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    o = m(dev(mel), dev(short), dev(emo), return_attention=True)
    assert set(o) == {"blendshapes", "mel_attention_weights", "emotion_attention_weights", "mel_blendshapes",
                      "emotion_blendshapes"}
    for k in o:
        np.testing.assert_allclose(o[k].cpu().numpy(), g[k], atol=2e-5 if "_blendshapes" in k else 2e-6)
    # weights changed in place -> the engine re-folds them
    with torch.no_grad():
        m.mel_weights.add_(1.0)
    o2 = m(dev(mel), dev(short), dev(emo))["blendshapes"]
    assert not torch.equal(o2, o["blendshapes"])
    with pytest.raises(RuntimeError, match="eval"):
        m.train()(dev(mel), dev(short), dev(emo))


def test_simplified_model_forward_and_smoothing():
    params = synth.make_core_params(51, style="trained")
    m = SimplifiedDualStreamModel().cuda().eval()
    m.load_state_dict(full_state(params, 0.3))
    orc = models.SimplifiedOracle(params, smoothing_alpha=0.3)
    for i in range(3):
        audio = synth.make_audio(60 + i, 3, 136448)
        emo = synth.normal(70 + i, (3, 256))
        want = orc.forward(audio, emo)["blendshapes"]
        got = m(dev(audio), emotion_features=dev(emo))["blendshapes"].cpu().numpy()
        assert np.abs(got - want).max() < 5e-6
    # a freshly constructed module is in train() mode: the fused path must refuse it like the attention module does
    fresh = SimplifiedDualStreamModel().cuda()
    with pytest.raises(RuntimeError, match="eval"):
        fresh(dev(audio), emotion_features=dev(emo))
    with torch.no_grad():
        assert fresh(dev(audio), emotion_features=dev(emo))["blendshapes"].shape == (3, 52)
    # attention path goes through the staged kernels and the stand-alone smoothing kernel
    m.reset_temporal_state(); orc.reset_temporal_state()
    audio = synth.make_audio(80, 2, 136448); emo = synth.normal(81, (2, 256))
    o = m(dev(audio), return_attention=True, emotion_features=dev(emo))
    w = orc.forward(audio, emo, return_attention=True)
    assert np.abs(o["blendshapes"].cpu().numpy() - w["blendshapes"]).max() < 5e-6
    assert np.abs(o["mel_attention_weights"].cpu().numpy() - w["mel_attention_weights"]).max() < 5e-6
    # provider hook, and the reference's dummy-feature fallback when there is none
    m.emotion_provider = lambda a: dev(emo)
    m.reset_temporal_state()
    via_provider = m(dev(audio))["blendshapes"]
    m.reset_temporal_state()
    assert torch.equal(via_provider, m(dev(audio), emotion_features=dev(emo))["blendshapes"])
    m.emotion_provider = None
    assert m(dev(audio))["blendshapes"].shape == (2, 52)
    long, short = m.extract_mel_features(dev(audio))
    assert long.shape == (2, 257, 80) and short.shape == (2, 3, 80)


@pytest.mark.parametrize("extra_hops,stride", [(5, 1), (9, 2), (-40, 1)])
def test_sequential_model_matches_oracle(extra_hops, stride):
    params = synth.make_core_params(52, style="trained")
    L = 136448 + 533 * extra_hops + 100               # not a multiple of the hop: exercises the tail logic
    audio = synth.make_audio(90, 2, L)
    emo = synth.normal(91, (2, 256))
    m = SequentialDualStreamModel(stride_frames=stride).cuda().eval()
    m.load_state_dict(full_state(params))
    orc = models.SequentialOracle(params, stride_frames=stride)
    want = orc.forward(audio, emo)
    got = m(dev(audio), emotion_features=dev(emo))
    assert got["num_frames"] == want["num_frames"] == got["blendshapes"].shape[1]
    assert got["fps"] == 30
    assert np.abs(got["blendshapes"].cpu().numpy() - want["blendshapes"]).max() < 5e-6
    if extra_hops == 5:
        ga = m(dev(audio), return_attention=True, emotion_features=dev(emo))
        assert torch.allclose(ga["blendshapes"], got["blendshapes"], atol=1e-6)
        assert ga["mel_attention_weights"].shape == (2, want["num_frames"], 28, 80)


def test_sequence_tiling_is_invisible():
    """More windows than the workspace tile: results must not depend on the tile size."""
    params = synth.make_core_params(53)
    audio = dev(synth.make_audio(92, 3, 136448 + 533 * 20))
    emo = dev(synth.normal(93, (3, 256)))
    m = DualStreamCrossAttention().cuda().eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    eng = m.engine()
    a = eng.sequence_forward(audio, emo, 1, True, max_tile=7)
    m2 = DualStreamCrossAttention().cuda().eval()
    m2.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    b = m2.engine().sequence_forward(audio, emo, 1, True, max_tile=4096)
    assert a.shape == (3, 21, 52) and torch.equal(a, b)


@pytest.mark.parametrize("extra_hops,stride,clips", [(20, 1, 3), (33, 3, 2), (-40, 1, 2), (0, 1, 1)])
def test_shared_frame_sequence_path_is_bit_identical_to_per_window(extra_hops, stride, clips):
    """km_sequence_forward computes the clip's STFT once and the two boundary frames of every window separately;
    the per-window evaluation (the reference's schedule, option seq_per_window) must give the same bits."""
    params = synth.make_core_params(54, style="trained")
    audio = dev(synth.make_audio(94, clips, 136448 + 533 * extra_hops + 77))
    emo = dev(synth.normal(95, (clips, 256)))
    m = DualStreamCrossAttention().cuda().eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    eng = m.engine()
    shared = eng.sequence_forward(audio, emo, stride, True, max_tile=16)
    eng.set_option("seq_per_window", 1)
    try:
        per_window = eng.sequence_forward(audio, emo, stride, True, max_tile=16)
    finally:
        eng.set_option("seq_per_window", 0)
    assert shared.shape[0] == clips and shared.shape[2] == 52
    assert torch.equal(shared, per_window)


def test_mel_spectrogram_extractor_mirror():
    ex = MelSpectrogramExtractor().cuda()
    assert ex.hop_length == 533 and ex.get_output_length(16000) == 31
    audio = synth.make_audio(94, 2, 32000)
    got = ex(dev(audio)).cpu().numpy()
    want = omel.mel_torchaudio(audio)
    assert got.shape == want.shape == (2, 60, 80)
    np.testing.assert_allclose(np.exp(got), np.exp(want), rtol=2e-3, atol=2e-8)
    assert ex(dev(audio[0])).shape == (1, 60, 80)                       # 1-D input
    assert ex(dev(audio[:, :533])).shape == (2, 0, 80)                  # rt.py's degenerate 533-sample call
    with pytest.raises(ValueError):
        ex(dev(audio)[None])


def test_realtime_path_and_rt_loop(tmp_path):
    params = synth.make_core_params(54, style="trained")
    m = SimplifiedDualStreamModel(real_time_mode=True).cuda().eval()
    m.load_state_dict(full_state(params))
    clock = [0.0]
    m.mel_extractor._clock = lambda: clock[0]
    stream = synth.make_audio(95, 1, 533 * 262)[0]
    emo = synth.normal(96, (1, 256))
    from oracle import buffers, core, smoothing
    ob = buffers.MelAudioBufferOracle()
    sm = smoothing.TemporalSmootherOracle(0.8)
    n_out = 0
    for i in range(262):
        frame = stream[i * 533:(i + 1) * 533]
        clock[0] += 1.0 / 30
        got = m.process_audio_frame_realtime(frame, emotion_features=dev(emo))
        ob.add_audio_frame(frame)
        win = ob.get_current_audio()
        assert (got is None) == (win is None)
        if got is None:
            continue
        n_out += 1
        if i % 3 == 0 or i == 261:
            feats = omel.mel_sliding_window(win, n_fft=1024, hop=533)                   # (255, 80) dB
            want = core.core_forward_np(params, feats[None], feats[None, -3:], emo)["blendshapes"]
            want = sm(want)
            assert np.abs(got.cpu().numpy() - want[0]).max() < 2e-5
        else:
            feats = omel.mel_sliding_window(win, n_fft=1024, hop=533)
            sm(core.core_forward_np(params, feats[None], feats[None, -3:], emo)["blendshapes"])
    assert n_out == 262 - 255
    # the rt.py loop end to end in --no_audio mode, JSONL out
    out = tmp_path / "frames.jsonl"
    m.reset_realtime_state()
    m.mel_extractor._clock = lambda: clock.__setitem__(0, clock[0] + 1.0) or clock[0]
    m.emotion_provider = lambda a: dev(emo)
    inf = rt.RealTimeInference(None, model=m, device="cuda")
    args = rt.build_parser().parse_args(["--model_path", "unused", "--no_audio", "--output_mode", "file",
                                         "--output_file", str(out), "--duration", "1000", "--chunk_size", "533"])
    streamer = rt.BlendshapeStreamer("file", output_file=str(out))
    import queue
    sent = 0
    q = queue.Queue(maxsize=100)
    for _ in range(270):                              # drive the loop body deterministically
        inf.process_audio_chunk(np.random.randn(533).astype(np.float32) * 0.01)
        bs = inf.inference_step()
        if bs is not None:
            streamer.send(bs, 0.0); sent += 1
    streamer.close()
    lines = out.read_text().splitlines()
    assert sent == len(lines) == 270 - 255
    rec = json.loads(lines[-1])
    assert len(rec["blendshapes"]) == 52 and all(0.0 <= v <= 1.0 for v in rec["blendshapes"])


def test_60fps_long_context_configuration_end_to_end():
    """BASELINE config 4: target_fps 60 (hop 266), window 512 frames, d_model 512 -- whole model from audio,
    and the sequential wrapper, on the shape-generic path."""
    params = synth.make_core_params(55, 512, 512, style="trained")
    kw = dict(d_model=512, num_heads=8, target_fps=60, mel_sequence_length=512)
    m = SimplifiedDualStreamModel(**kw).cuda().eval()
    m.load_state_dict(full_state(params))
    assert m.hop_length == 266
    orc = models.SimplifiedOracle(params, num_heads=8, mel_sequence_length=512, target_fps=60)
    for i in range(2):
        audio = synth.make_audio(97 + i, 2, 512 * 266)
        emo = synth.normal(98 + i, (2, 256))
        want = orc.forward(audio, emo)["blendshapes"]
        got = m(dev(audio), emotion_features=dev(emo))["blendshapes"].cpu().numpy()
        assert np.abs(got - want).max() < 5e-6
    long, short = m.extract_mel_features(dev(audio))
    assert long.shape == (2, 513, 80)
    seq = SequentialDualStreamModel(stride_frames=2, **kw).cuda().eval()
    seq.load_state_dict(full_state(params))
    sorc = models.SequentialOracle(params, num_heads=8, mel_sequence_length=512, target_fps=60, stride_frames=2)
    audio = synth.make_audio(99, 2, 512 * 266 + 266 * 6 + 50)
    emo = synth.normal(100, (2, 256))
    w = sorc.forward(audio, emo)
    g = seq(dev(audio), emotion_features=dev(emo))
    assert g["num_frames"] == w["num_frames"] == 4 and g["fps"] == 60
    assert np.abs(g["blendshapes"].cpu().numpy() - w["blendshapes"]).max() < 5e-6


@pytest.mark.parametrize("d,T,H,fps,L", [(512, 512, 16, 60, 512 * 266), (512, 512, 8, 60, 300 * 266 + 5), (64, 32, 4, 30, 33 * 533),
                                         (64, 32, 4, 30, 2 * 533 + 1)])
def test_generic_packed_encoder_path_matches_staged_path(d, T, H, fps, L):
    """km_forward_audio on generic shapes: log-mel written straight into the packed encoder input + encoder_tn_kernel,
    against the staged path (mel_log_kernel + strided GEMM + K=3 accumulate), incl. windows shorter than T (zero rows)
    and shorter than 3 frames' worth of context."""
    import os
    from koemorph_amd.engine import Engine, MelConfig
    eng = Engine(d_model=d, num_heads=H, mel_sequence_length=T, mel=MelConfig.model_batch(target_fps=fps))
    eng.load_state_dict(synth.make_core_params(56, d, T, 256, "trained"))
    eng.finalize()
    B = 5
    eng.reserve(B, L)
    audio, emo = dev(synth.make_audio(101, B, L)), dev(synth.normal(102, (B, 256)))
    packed = eng.forward_audio(audio, emo).clone()
    eng.set_option("generic_staged", 1)
    try:
        staged = eng.forward_audio(audio, emo).clone()
    finally:
        eng.set_option("generic_staged", 0)
    assert packed.shape == (B, 52) and float((packed - staged).abs().max()) < 1e-6


def test_legacy_simplified_koemorph_model():
    """the legacy single-stream model (52 queries over 257 encoded mel frames)."""
    from koemorph_amd.model import SimplifiedKoeMorphModel
    from oracle import legacy
    params = legacy.make_legacy_params(7)
    m = SimplifiedKoeMorphModel().cuda().eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    mel = synth.uniform(8, (3, 257, 80), 0, 1)
    got = m.forward_mel(dev(mel)).cpu().numpy()
    want = legacy.legacy_forward_mel(params, mel)
    assert got.shape == (3, 52) and np.abs(got - want).max() < 5e-6
    audio = synth.make_audio(9, 2, 136448)
    got = m(dev(audio)).cpu().numpy()
    want = legacy.legacy_forward(params, audio)
    assert np.abs(got - want).max() < 1e-4          # contract
    assert np.abs(got - want).max() < 2e-5
    assert m.extract_mel_features(dev(audio)).shape == (2, 257, 80)
    with pytest.raises(ValueError):
        m(dev(audio)[0])
    # the attention as one register-resident kernel (the default at 32-wide heads) against two batched products + a row softmax,
    # at frame counts that fill the last key tile, leave one key in it, and fit in one tile
    from koemorph_amd import _lib
    lib, h, _ = m._handle()
    for T in (257, 256, 250, 16, 5):
        melT = synth.uniform(10 + T, (5, T, 80), 0, 1)
        res = {}
        for mode in (0, 1, 2, 4, 7, 8):   # bit 0: attention as batched products, 1: encoder + K / V projections, 2: out_proj + decoder as GEMM launches,
            _lib.check(lib.km_set_option(h, b"legacy_no_attn_fusion", mode & 1))           # 3: attention and tail as two launches instead of one
            _lib.check(lib.km_set_option(h, b"legacy_no_enc_fusion", (mode >> 1) & 1))
            _lib.check(lib.km_set_option(h, b"legacy_no_tail_fusion", (mode >> 2) & 1))
            _lib.check(lib.km_set_option(h, b"legacy_no_merge", mode >> 3))
            res[mode] = m.forward_mel(dev(melT)).cpu().numpy()
        for name in (b"legacy_no_attn_fusion", b"legacy_no_enc_fusion", b"legacy_no_tail_fusion", b"legacy_no_merge"):
            _lib.check(lib.km_set_option(h, name, 0))
        assert np.array_equal(res[0], res[8]), T          # one launch or two: the same two bodies
        want = legacy.legacy_forward_mel(params, melT)
        for mode in res:
            assert np.abs(res[mode] - want).max() < 5e-6, (T, mode)
