"""Host-side mirrors of the reference interface (no GPU): ring buffers against the oracle, state-dict
layout, wire format, CLI flags, and the no-CPU-fallback rule."""
import json
import socket

import numpy as np
import pytest
import torch

from koemorph_amd import synth
from koemorph_amd.features.mel_sliding_window import MelAudioBuffer
from koemorph_amd.model import DualStreamCrossAttention, SequentialDualStreamModel, SimplifiedDualStreamModel
from koemorph_amd.scripts import rt
from oracle import buffers


def test_ringbuffer_matches_oracle_on_random_traffic():
    rng = np.random.default_rng(0)
    a, b = rt.RingBuffer(1000), buffers.RingBufferOracle(1000)
    for _ in range(400):
        if rng.random() < 0.55:
            d = rng.standard_normal(int(rng.integers(1, 400))).astype(np.float32)
            a.write(d); b.write(d)
        else:
            n = int(rng.integers(1, 600))
            x, y = a.read(n), b.read(n)
            assert (x is None) == (y is None)
            if x is not None:
                assert np.array_equal(x, y)
        assert (a.available, a.read_ptr, a.write_ptr) == (b.available, b.read_ptr, b.write_ptr)


def test_mel_audio_buffer_matches_oracle():
    rng = np.random.default_rng(1)
    a, b = MelAudioBuffer(), buffers.MelAudioBufferOracle()
    assert a.hop_length == b.hop_length == 532 and a.buffer_size == 136000
    for i in range(300):
        n = int(rng.choice([531, 532, 533, 533, 533, 500]))
        f = rng.standard_normal(n).astype(np.float32)
        assert a.add_audio_frame(f) == b.add_audio_frame(f)
        x, y = a.get_current_audio(), b.get_current_audio()
        assert (x is None) == (y is None)
        if x is not None and i % 17 == 0:
            assert np.array_equal(x, y)
    assert a.get_stats()["total_frames_added"] == b.total_frames_added


def test_state_dict_layout_matches_reference():
    m = DualStreamCrossAttention()
    want = synth.core_param_shapes()
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == dict(want)
    assert sum(p.numel() for p in m.parameters()) == 837737
    full = SimplifiedDualStreamModel()
    keys = set(full.state_dict().keys())
    assert keys == {"smoothing_alpha"} | {"dual_stream_attention." + k for k in want}
    m512 = DualStreamCrossAttention(d_model=512, mel_sequence_length=512, num_heads=16)
    assert sum(p.numel() for p in m512.parameters()) == 3182697
    # a reference-layout checkpoint loads strictly
    sd = {"dual_stream_attention." + k: torch.from_numpy(v) for k, v in synth.make_core_params(3).items()}
    sd["smoothing_alpha"] = torch.tensor(0.5)
    full.load_state_dict(sd, strict=True)
    seq = SequentialDualStreamModel(stride_frames=2)
    assert (seq.window_samples, seq.stride_samples, seq.hop_length) == (136448, 1066, 533)
    assert SequentialDualStreamModel(target_fps=60, mel_sequence_length=512).hop_length == 266


def test_no_cpu_fallback():
    m = DualStreamCrossAttention().eval()
    mel, short, emo = (torch.from_numpy(x) for x in synth.make_core_inputs(1, 2, 257))
    with pytest.raises(RuntimeError, match="GPU only"):
        m(mel, short, emo)


def test_streamer_wire_format(tmp_path):
    bs = synth.uniform(5, (52,), 0, 1)
    p = tmp_path / "out.jsonl"
    s = rt.BlendshapeStreamer("file", output_file=str(p))
    s.send(bs, 12.5); s.send(bs, 13.0); s.close()
    lines = p.read_text().splitlines()
    rec = json.loads(lines[0])
    assert len(lines) == 2 and list(rec) == ["timestamp", "blendshapes"] and rec["timestamp"] == 12.5
    assert len(rec["blendshapes"]) == 52 and rec["blendshapes"] == bs.tolist()
    rx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    rx.bind(("127.0.0.1", 0)); rx.settimeout(2.0)
    u = rt.BlendshapeStreamer("udp", port=rx.getsockname()[1])
    u.send(bs, 1.0)
    got = json.loads(rx.recv(65536).decode("utf-8"))
    assert got == {"timestamp": 1.0, "blendshapes": bs.tolist()}
    u.close(); rx.close()
    with pytest.raises(ValueError):
        rt.BlendshapeStreamer("file")
    with pytest.raises(ValueError):
        rt.BlendshapeStreamer("smoke-signals")


def test_cli_flags_match_reference():
    p = rt.build_parser()
    a = p.parse_args(["--model_path", "x.pt"])
    assert (a.sample_rate, a.target_fps, a.chunk_size, a.output_mode, a.host, a.port, a.device, a.no_audio) == \
        (16000, 30.0, 1024, "udp", "127.0.0.1", 9001, "auto", False)
    flags = {o for act in p._actions for o in act.option_strings}
    assert {"--model_path", "--config_path", "--sample_rate", "--target_fps", "--chunk_size", "--output_mode",
            "--host", "--port", "--output_file", "--device", "--duration", "--no_audio"} <= flags
    with pytest.raises(SystemExit):
        p.parse_args([])                                              # --model_path is required


def test_streamer_batch_is_byte_identical_to_per_frame_json_dumps(tmp_path):
    """A tick of many streams encoded by one C call: the same bytes the reference's per-frame json.dumps produces,
    over UDP (one datagram per stream, optional per-stream ports) and as JSONL."""
    frames = synth.uniform(6, (16, 52), 0, 0.03)
    want = [json.dumps({"timestamp": 99.25, "blendshapes": r.tolist()}) for r in frames]
    p = tmp_path / "tick.jsonl"
    s = rt.BlendshapeStreamer("file", output_file=str(p))
    s.send_batch(frames, 99.25); s.close()
    assert p.read_text() == "".join(w + "\n" for w in want)
    rx = [socket.socket(socket.AF_INET, socket.SOCK_DGRAM) for _ in range(2)]
    for r in rx:
        r.bind(("127.0.0.1", 0)); r.settimeout(2.0)
    u = rt.BlendshapeStreamer("udp", port=rx[0].getsockname()[1])
    u.send_batch(frames[:3], 99.25)
    assert [rx[0].recv(65536).decode() for _ in range(3)] == want[:3]
    u.send_batch(frames[:2], 99.25, ports=[r.getsockname()[1] for r in rx])
    assert rx[0].recv(65536).decode() == want[0] and rx[1].recv(65536).decode() == want[1]
    u.close()
    for r in rx:
        r.close()


def test_audio_file_reader_and_readme_flags(tmp_path):
    from scipy.io import wavfile
    x = (synth.uniform(7, (2500,), -0.5, 0.5) * 32767).astype(np.int16)
    wavfile.write(tmp_path / "a.wav", 16000, x)
    chunks = list(rt.AudioFileReader(str(tmp_path / "a.wav"), 16000, 1024))
    assert len(chunks) == 3 and all(c.shape == (1024,) and c.dtype == np.float32 for c in chunks)
    np.testing.assert_allclose(np.concatenate(chunks)[:2500], x.astype(np.float32) / 32768.0)
    assert not np.concatenate(chunks)[2500:].any()                      # last chunk zero padded (rt_simplified.py:134-136)
    a = rt.build_parser().parse_args(["--model_path", "m.pt", "--input_audio", "a.wav", "--output_json", "o.jsonl"])
    assert a.input_audio == "a.wav" and a.output_json == "o.jsonl"      # README.md:128-131


def test_dataset_host_helpers_and_oracle_windows(tmp_path):
    """Host-side pieces of the sequential dataset mirror (the window producer itself is a GPU test)."""
    from koemorph_amd.data import SequentialKoeMorphDataset, detect_source_fps, load_jsonl_labels
    from oracle import dataset as od
    labels = synth.uniform(8, (40, 52), 0, 1)
    with open(tmp_path / "l.jsonl", "w") as f:
        for i, row in enumerate(labels):
            f.write(json.dumps({"timestamp": i / 60.0, "blendshapes": row.tolist()}) + "\n")
    rows, ts = load_jsonl_labels(tmp_path / "l.jsonl")
    assert rows.dtype == np.float32 and np.array_equal(rows, labels) and detect_source_fps(ts) == 60.0
    assert detect_source_fps([0.0, 1 / 29.0, 2 / 29.0]) == 30.0 and detect_source_fps([]) == 30.0
    assert abs(detect_source_fps([0.0, 0.04, 0.08]) - 25.0) < 1e-9             # non-standard rates are kept
    r = od.resample_blendshapes(labels, 60.0, 30)
    assert r.shape == (20, 52) and np.array_equal(r[0], labels[0]) and np.array_equal(r[-1], labels[-1])
    a = np.arange(300 * 533 + 17, dtype=np.float32)
    b = synth.uniform(9, (304, 52), 0, 1)                                       # 4 frames more than the audio has
    w = list(od.windows(a, b, 256, 10, 533))
    assert [x[0] for x in w] == [0, 1, 2, 3, 4] and w[4][1] == 40 and w[4][2][0] == 40 * 533 and len(w[4][2]) == 136448
    with pytest.raises(RuntimeError):
        SequentialKoeMorphDataset(tmp_path, device="cpu")                       # clips live in HBM: no CPU fallback


def test_train_sequential_cli():
    from koemorph_amd.scripts import train_sequential as ts
    a = ts.build_parser().parse_args(["--data_dir", "d"])
    assert (a.epochs, a.batch_size, a.window_frames, a.stride_frames, a.learning_rate, a.weight_decay, a.gradient_clip) == \
        (10, 8, 256, 1, 1e-4, 1e-5, 1.0)                               # train_sequential.py:73-86 defaults
    with pytest.raises(SystemExit):
        ts.build_parser().parse_args([])


def test_rt_rejects_legacy_koemorph_checkpoint(tmp_path):
    """A checkpoint of the legacy multi-layer model is refused with a pointer to its mirror (the reference's own rt.py loop
    cannot drive that class either: it passes prosody features to a three-argument inference_step)."""
    import torch
    from koemorph_amd.model import KoeMorphModel
    m = KoeMorphModel(d_model=64, d_query=64, num_heads=4, num_encoder_layers=1, num_attention_layers=1, decoder_hidden_dim=32,
                      emotion_dim=8)
    path = tmp_path / "legacy.pt"
    torch.save({"model_state_dict": m.state_dict()}, path)
    r = rt.RealTimeInference.__new__(rt.RealTimeInference)
    r.sample_rate, r.target_fps, r.device = 16000, 30.0, torch.device("cpu")
    with pytest.raises(ValueError, match="create_koemorph_model"):
        r._load_model(str(path), None, None)
