"""Device-side window producer against the restated reference pipeline in oracle/dataset.py (numpy's own
linspace / interp, i.e. the library the reference calls): bit-exact labels and windows."""
import json

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from koemorph_amd import synth
from koemorph_amd.data import AdaptiveSequentialDataset, SequentialKoeMorphDataset, create_adaptive_dataloader, detect_source_fps
from oracle import dataset as od

pytestmark = pytest.mark.gpu


def write_pair(d, name, seconds, label_fps, seed, extra_label_frames=0):
    n = int(seconds * 16000)
    audio = synth.uniform(seed, (n,), -0.5, 0.5).astype(np.float32)
    wavfile.write(d / f"{name}.wav", 16000, audio)
    F = int(seconds * label_fps) + extra_label_frames
    labels = synth.uniform(seed + 1, (F, 52), 0, 1).astype(np.float32)
    with open(d / f"{name}.jsonl", "w") as f:
        for i in range(F):
            f.write(json.dumps({"timestamp": i / label_fps, "blendshapes": labels[i].tolist()}) + "\n")
    return audio, labels


@pytest.mark.parametrize("n_src,n_dst", [(600, 300), (601, 300), (2, 7), (1, 3), (977, 488), (300, 600)])
def test_label_resampling_is_bit_identical_to_numpy(n_src, n_dst, tmp_path):
    write_pair(tmp_path, "x", 9.0, 30, 1)
    ds = SequentialKoeMorphDataset(tmp_path, shuffle_files=False, loop_dataset=False)
    src = synth.uniform(3 + n_src, (n_src, 52), 0, 1).astype(np.float32)
    ratio = n_dst / n_src
    ds.target_fps = 30.0
    source_fps = 30.0 / ratio
    want = od.resample_blendshapes(src, source_fps, 30.0)
    got = ds.resample_labels(torch.from_numpy(src).cuda(), source_fps).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)


def test_windows_match_reference_pipeline(tmp_path):
    """Two clips: 30 fps labels (aligned) and 60 fps labels with a frame-count mismatch (resampling + truncation)."""
    a0, l0 = write_pair(tmp_path, "a_first", 9.3, 30, 10)
    a1, l1 = write_pair(tmp_path, "b_second", 9.0, 60, 20, extra_label_frames=9)
    ds = SequentialKoeMorphDataset(tmp_path, stride_frames=3, shuffle_files=False, loop_dataset=False, batch_size=5)
    assert ds.hop_length == 533 and ds.window_samples == 136448 and [p[0].stem for p in ds.file_pairs] == ["a_first", "b_second"]
    assert detect_source_fps([i / 60 for i in range(50)]) == 60.0 and detect_source_fps([0.0]) == 30.0
    ref = []
    for fi, (a, l, fps) in enumerate([(a0, l0, 30.0), (a1, l1, 60.0)]):
        lab = od.resample_blendshapes(l, fps, 30)
        # the int16 WAV round trip quantises the audio: compare against what the file holds
        a = wavfile.read(tmp_path / (["a_first", "b_second"][fi] + ".wav"))[1].astype(np.float32)
        for i, sf, aw, bw in od.windows(a, lab, 256, 3, 533):
            ref.append((fi, i, sf, aw, bw))
    got = []
    for batch in ds:
        B = batch["audio"].shape[0]
        assert B <= 5 and batch["blendshapes"].shape == (B, 256, 52) and batch["target"].shape == (B, 52)
        for b in range(B):
            got.append((int(batch["file_indices"][b]), int(batch["window_indices"][b]), int(batch["start_frames"][b]),
                        batch["audio"][b].cpu().numpy(), batch["blendshapes"][b].cpu().numpy(), batch["target"][b].cpu().numpy(),
                        batch["file_names"][b]))
    assert len(got) == len(ref) == ds.get_num_windows() and len(ref) > 10
    for g, r in zip(got, ref):
        assert g[:3] == r[:3]
        assert np.array_equal(g[3], r[3]) and np.array_equal(g[4], r[4]) and np.array_equal(g[5], r[4][-1])
    assert got[0][6] == "a_first" and got[-1][6] == "b_second"


def test_float_wav_is_kept_exact_and_feeds_the_train_step(tmp_path):
    from koemorph_amd.engine import Engine
    from koemorph_amd.training import Trainer
    audio = synth.uniform(30, (int(9.0 * 16000),), -0.5, 0.5).astype(np.float32)
    wavfile.write(tmp_path / "c.wav", 16000, audio)
    labels = synth.uniform(31, (270, 52), 0, 1).astype(np.float32)
    with open(tmp_path / "c.jsonl", "w") as f:
        for i in range(270):
            f.write(json.dumps({"timestamp": i / 30.0, "blendshapes": labels[i].tolist()}) + "\n")
    ds = SequentialKoeMorphDataset(tmp_path, shuffle_files=False, loop_dataset=False, batch_size=4)
    batch = next(iter(ds))
    assert np.array_equal(batch["audio"][1].cpu().numpy(), audio[533:533 + 136448])
    eng = Engine(); eng.load_state_dict(synth.make_core_params(0)); eng.finalize(); eng.reserve(4, 136448)
    tr = Trainer(eng, max_windows=4)
    emo = torch.from_numpy(synth.normal(32, (4, 256))).cuda()
    l0 = float(tr.step(batch["audio"], emo, batch["target"]).item())
    l1 = float(tr.step(batch["audio"], emo, batch["target"]).item())
    assert np.isfinite(l0) and np.isfinite(l1) and l1 < l0          # one optimisation step on a device-made batch


def test_window_producer_argument_errors():
    import ctypes
    from koemorph_amd._lib import KoeMorphError, check, load
    lib = load()
    x = torch.zeros(1000, device="cuda"); starts = torch.zeros(4, dtype=torch.int32, device="cuda"); out = torch.zeros(4, 100, device="cuda")
    with pytest.raises(KoeMorphError):
        check(lib.km_gather_windows(x.data_ptr(), 1000, None, 4, 10, 100, out.data_ptr(), None, 0, 0, 0, None, None, None))
    with pytest.raises(KoeMorphError):
        check(lib.km_gather_windows(x.data_ptr(), 1000, starts.data_ptr(), 70000, 10, 100, out.data_ptr(), None, 0, 0, 0, None, None, None))
    with pytest.raises(KoeMorphError):      # label outputs requested without a label track
        check(lib.km_gather_windows(x.data_ptr(), 1000, starts.data_ptr(), 4, 10, 100, out.data_ptr(), None, 0, 8, 52, out.data_ptr(), None, None))
    with pytest.raises(KoeMorphError):
        check(lib.km_resample_labels(x.data_ptr(), 0, 52, 10, out.data_ptr(), None))
    check(lib.km_resample_labels(x.data_ptr(), 10, 52, 0, out.data_ptr(), None))               # nothing to do is fine
    # windows starting past the end of the clip are zero filled, not out-of-bounds reads
    starts[:] = torch.tensor([0, 50, 99, 200], dtype=torch.int32)
    clip = torch.arange(1000, dtype=torch.float32, device="cuda")
    check(lib.km_gather_windows(clip.data_ptr(), 1000, starts.data_ptr(), 4, 10, 100, out.data_ptr(), None, 0, 0, 0, None, None, None))
    torch.cuda.synchronize()
    assert torch.equal(out[1], clip[500:600]) and torch.equal(out[2, :10], clip[990:]) and not out[2, 10:].any() and not out[3].any()


def test_sequential_trainer_script_end_to_end(tmp_path):
    """koemorph_amd.scripts.train_sequential: two epochs over a tiny two-clip data set, validation, checkpoint in the
    reference's format (loads into the model mirror with strict=True), resume."""
    from koemorph_amd.engine import Engine
    from koemorph_amd.model import SimplifiedDualStreamModel
    from koemorph_amd.scripts import train_sequential as ts
    for name, seed in (("a", 40), ("b", 50)):
        write_pair(tmp_path, name, 8.9, 30, seed)
    eng = Engine(); eng.load_state_dict(synth.make_core_params(0)); eng.finalize()
    kw = dict(stride_frames=4, shuffle_files=False, loop_dataset=False, batch_size=4)
    st = ts.SequentialTrainer(eng, SequentialKoeMorphDataset(tmp_path, **kw), SequentialKoeMorphDataset(tmp_path, **kw),
                              learning_rate=1e-3, l1_weight=0.1,
                              extra_loss_terms=dict(sparsity_weight=0.01, smoothness_weight=0.1))
    v0 = st.validate()["total"]
    m1 = st.train_epoch(); m2 = st.train_epoch()
    v2 = st.validate()["total"]
    assert m1["batches"] == m2["batches"] >= 2 and m2["total"] < m1["total"] and v2 < v0 and st.epoch == 2
    ck = tmp_path / "ck" / "checkpoint_epoch_2.pth"
    st.save_checkpoint(ck, is_best=True)
    assert (tmp_path / "ck" / "best_model.pth").exists()
    ckpt = torch.load(ck, weights_only=True)
    assert {"epoch", "global_step", "model_state_dict", "best_val_loss"} <= set(ckpt)
    model = SimplifiedDualStreamModel().cuda().eval()
    model.load_state_dict(ckpt["model_state_dict"], strict=True)
    batch = next(iter(SequentialKoeMorphDataset(tmp_path, **kw)))
    emo = st._emotion(batch)
    st.trainer.sync_inference_weights()
    model.reset_temporal_state()
    a = model(batch["audio"], emotion_features=emo)["blendshapes"]
    b = eng.forward_audio(batch["audio"], emo)
    assert float((a - b).abs().max()) < 1e-6                                   # the checkpoint IS the trained model
    eng2 = Engine(); eng2.load_state_dict(synth.make_core_params(0)); eng2.finalize()
    st2 = ts.SequentialTrainer(eng2, SequentialKoeMorphDataset(tmp_path, **kw), learning_rate=1e-3, l1_weight=0.1,
                               extra_loss_terms=dict(sparsity_weight=0.01, smoothness_weight=0.1))
    st2.load_checkpoint(ck)
    assert st2.epoch == 2 and st2.global_step == st.global_step
    for k, v in st2.state_dict().items():
        assert torch.equal(v, ckpt["model_state_dict"][k]), k
    # resuming continues exactly where the uninterrupted run goes: weights, AdamW moments, step counters, schedule
    m3a = st.train_epoch(); m3b = st2.train_epoch()
    assert m3a["total"] == m3b["total"] and m3a["lr"] == m3b["lr"]
    for (k, a), (_, b) in zip(st.state_dict().items(), st2.state_dict().items()):
        assert torch.equal(a, b), k
    # a checkpoint with a FOREIGN optimizer_state_dict (the reference's trainer stores torch.optim.AdamW's dict, keyed by
    # parameter index; src/train_sequential.py:303-339) resumes from the weights with fresh moments instead of aborting
    foreign = dict(ckpt)
    foreign["optimizer_state_dict"] = {"state": {0: {"step": torch.tensor(3.0)}}, "param_groups": [{"lr": 1e-3, "params": [0]}]}
    fpath = tmp_path / "ck" / "foreign.pth"
    torch.save(foreign, fpath)
    eng3 = Engine(); eng3.load_state_dict(synth.make_core_params(0)); eng3.finalize()
    st3 = ts.SequentialTrainer(eng3, SequentialKoeMorphDataset(tmp_path, **kw), learning_rate=1e-3, l1_weight=0.1)
    with pytest.warns(RuntimeWarning, match="fresh AdamW moments"):
        st3.load_checkpoint(fpath)
    assert st3.epoch == 2 and st3.trainer.step_count == 0 and st3.trainer.epoch == 2
    for k, v in st3.state_dict().items():
        assert torch.equal(v, ckpt["model_state_dict"][k]), k
    assert np.isfinite(st3.train_epoch()["total"])


@pytest.mark.parametrize("mode,kw", [("dense", {}), ("sparse", {"initial_stride": 16}), ("progressive", {"initial_stride": 12, "final_stride": 2, "epoch": 3, "max_epochs": 7}),
                                     ("mixed", {"initial_stride": 20, "dense_sampling_ratio": 0.25})])
def test_adaptive_dataset_modes_match_reference_pipeline(mode, kw, tmp_path):
    """AdaptiveSequentialDataset mirror (src/data/adaptive_sequential_dataset.py): window order, stride per mode, keys and
    bit-exact windows against the restated pipeline (oracle/dataset.py)."""
    a0, l0 = write_pair(tmp_path, "a_first", 9.3, 30, 10)
    a1, l1 = write_pair(tmp_path, "b_second", 9.6, 30, 20, extra_label_frames=7)      # label / audio mismatch -> truncation
    ds = AdaptiveSequentialDataset(tmp_path, stride_mode=mode, shuffle_files=False, loop_dataset=False, batch_size=5, **kw)
    stride = od.adaptive_stride(mode, kw.get("initial_stride", 32), kw.get("final_stride", 1), kw.get("epoch", 0), kw.get("max_epochs", 100))
    assert ds.current_stride == stride
    ref = []
    np.random.seed(99)
    for fi, (name, l) in enumerate([("a_first", l0), ("b_second", l1)]):
        a = wavfile.read(tmp_path / (name + ".wav"))[1].astype(np.float32)
        for i, sf, dense, aw, bw in od.adaptive_windows(a, l, mode, stride, kw.get("initial_stride", 32),
                                                        kw.get("dense_sampling_ratio", 0.1), 256, 533):
            ref.append((fi, i, sf, dense, aw, bw))
    got = []
    np.random.seed(99)
    for batch in ds:
        B = batch["audio"].shape[0]
        assert set(batch) == {"audio", "blendshapes", "target", "file_indices", "window_indices", "start_frames", "file_names", "is_dense"}
        assert B <= 5 and batch["blendshapes"].shape == (B, 256, 52)
        for b in range(B):
            got.append((int(batch["file_indices"][b]), int(batch["window_indices"][b]), int(batch["start_frames"][b]),
                        bool(batch["is_dense"][b]), batch["audio"][b].cpu().numpy(), batch["blendshapes"][b].cpu().numpy(),
                        batch["target"][b].cpu().numpy()))
    assert len(got) == len(ref) and len(ref) >= 4
    for g, r in zip(got, ref):
        assert g[:4] == r[:4]
        assert np.array_equal(g[4], r[4]) and np.array_equal(g[5], r[5]) and np.array_equal(g[6], r[5][-1])
    if mode == "progressive":                       # set_epoch moves the stride along the schedule (:128-132)
        ds.set_epoch(6)
        assert ds.current_stride == 2 and len(ds.plan(0)) == (279 - 256) // 2 + 1
    assert isinstance(create_adaptive_dataloader(tmp_path, batch_size=2, stride_mode="dense"), AdaptiveSequentialDataset)
