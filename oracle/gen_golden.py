"""Generate tests/golden/core_*.npz from the REFERENCE's own DualStreamCrossAttention.

Runs ONLY in the build container (it imports /root/reference, which does not exist on the
GPU box).  Usage:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

For every case it
  1. builds the state dict and inputs from koemorph_amd.synth (seeded, reproducible),
  2. instantiates the reference module (src/model/dual_stream_attention.py:48), loads the
     state dict, runs eval-mode forward(return_attention=True),
  3. for the gradient cases also runs autograd of MSE(blendshapes, target) in eval mode,
  4. stores config + seeds + the reference outputs (a fixture is data: no weights, no
     reference source).
The fixtures pin oracle/core.py (tests/test_oracle_core.py) and, on the GPU, the HIP path.
"""

from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = [
    # name, d_model, T_seq, heads, batch, t_in, param style, input style, seed, grads
    dict(name="core_d256_T256_H8_init", d=256, T=256, H=8, B=4, t_in=257, pstyle="init", istyle="mel01", seed=11, grads=False),
    dict(name="core_d256_T256_H8_trained", d=256, T=256, H=8, B=4, t_in=257, pstyle="trained", istyle="mel01", seed=12, grads=False),
    dict(name="core_d256_T256_H8_randn", d=256, T=256, H=8, B=3, t_in=256, pstyle="trained", istyle="randn", seed=13, grads=False),
    dict(name="core_d256_pad_T100", d=256, T=256, H=8, B=2, t_in=100, pstyle="trained", istyle="mel01", seed=14, grads=False),
    dict(name="core_d256_trunc_T300", d=256, T=256, H=8, B=2, t_in=300, pstyle="trained", istyle="mel01", seed=15, grads=False),
    dict(name="core_d256_rt_T255", d=256, T=256, H=8, B=2, t_in=255, pstyle="init", istyle="mel01", seed=16, grads=False),
    dict(name="core_d256_pad_T1", d=256, T=256, H=8, B=1, t_in=1, pstyle="trained", istyle="randn", seed=17, grads=False),
    dict(name="core_d256_trunc_T700", d=256, T=256, H=8, B=3, t_in=700, pstyle="init", istyle="randn", seed=18, grads=False),
    dict(name="core_d512_T512_H8", d=512, T=512, H=8, B=2, t_in=513, pstyle="trained", istyle="mel01", seed=21, grads=False),
    dict(name="core_d512_T512_H16", d=512, T=512, H=16, B=2, t_in=513, pstyle="trained", istyle="mel01", seed=22, grads=False),
    dict(name="core_d64_T32_H4_small", d=64, T=32, H=4, B=5, t_in=33, pstyle="trained", istyle="randn", seed=31, grads=True),
    dict(name="core_d256_T256_H8_grads", d=256, T=256, H=8, B=8, t_in=257, pstyle="trained", istyle="mel01", seed=41, grads=True),
    # full KoeMorphLoss (src/model/losses.py, default weights) with prev tensors and a seeded landmark matrix
    dict(name="core_d64_T32_H4_fullloss", d=64, T=32, H=4, B=5, t_in=33, pstyle="trained", istyle="randn", seed=51, grads="full"),
    dict(name="core_d256_T256_H8_fullloss", d=256, T=256, H=8, B=8, t_in=257, pstyle="trained", istyle="mel01", seed=52, grads="full"),
    # TRAINING mode (model.train(), dropout 0.1 in both attentions and the decoder, src/train_sequential.py:118): the three
    # dropout masks are reproduced from the seed (see train_masks) and stored with the outputs and gradients
    dict(name="core_d64_T32_H4_train", d=64, T=32, H=4, B=5, t_in=33, pstyle="trained", istyle="randn", seed=71, grads=True, train=0.1),
    dict(name="core_d256_T256_H8_train", d=256, T=256, H=8, B=8, t_in=257, pstyle="trained", istyle="mel01", seed=72, grads=True, train=0.1),
    # ... and the full KoeMorphLoss WITH audio_features, i.e. including the audio-visual term of PerceptualBlendshapeLoss
    dict(name="core_d256_T256_H8_train_fullloss_av", d=256, T=256, H=8, B=8, t_in=257, pstyle="trained", istyle="mel01", seed=73,
         grads="full", train=0.1, av=True),
    dict(name="core_d64_T32_H4_fullloss_av", d=64, T=32, H=4, B=5, t_in=33, pstyle="trained", istyle="randn", seed=74, grads="full", av=True),
    # gradients at the 60 fps TRAINING shape (configs/experiment/dual_stream_60fps.yaml:7-24: d_model 512, window 512), eval and
    # training mode
    dict(name="core_d512_T512_H8_grads", d=512, T=512, H=8, B=4, t_in=513, pstyle="trained", istyle="mel01", seed=81, grads=True),
    dict(name="core_d512_T512_H8_train", d=512, T=512, H=8, B=4, t_in=513, pstyle="trained", istyle="mel01", seed=82, grads=True, train=0.1),
]


def train_masks(seed, B, H, d, p):
    """The keep masks torch draws in DualStreamCrossAttention.forward under model.train() after torch.manual_seed(seed), in
    call order: attention-weight dropout of mel_attention (B,H,28,80), of emotion_attention (B,H,24,1), then the
    decoder's nn.Dropout on (B,52,d/2).  F.dropout on a tensor of ones yields keep/(1-p) with the same generator
    consumption; main() asserts that the oracle fed with these masks reproduces the reference's training-mode output."""
    torch.manual_seed(seed)
    out = {}
    for key, shape in (("mel", (B, H, 28, 80)), ("emo", (B, H, 24, 1)), ("dec", (B, 52, d // 2))):
        out[key] = (torch.nn.functional.dropout(torch.ones(shape), p, True) > 0).numpy()
    return out


def full_loss_inputs(synth, seed, B):
    """target, prev_pred, prev_target (B,52) in [0,1] and the landmark matrix (136,52) ~ N(0, 0.01^2), all seeded."""
    return (synth.uniform(seed * 3 + 1, (B, 52), 0.0, 1.0), synth.uniform(seed * 3 + 2, (B, 52), 0.0, 1.0),
            synth.uniform(seed * 3 + 3, (B, 52), 0.0, 1.0), (0.01 * synth.normal(seed * 3 + 4, (136, 52))).astype(np.float32))


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from src.model.dual_stream_attention import DualStreamCrossAttention  # reference, read-only
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from koemorph_amd import synth

    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    for c in CASES:
        params = synth.make_core_params(c["seed"], c["d"], c["T"], 256, c["pstyle"])
        mel, short, emo = synth.make_core_inputs(c["seed"], c["B"], c["t_in"], style=c["istyle"])
        if os.environ.get("KM_GOLDEN_ONLY") and os.environ["KM_GOLDEN_ONLY"] not in c["name"]:
            continue
        m = DualStreamCrossAttention(d_model=c["d"], num_heads=c["H"], mel_sequence_length=c["T"],
                                     dropout=c.get("train", 0.1)).eval()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        with torch.no_grad():
            o = m(torch.from_numpy(mel), torch.from_numpy(short), torch.from_numpy(emo), return_attention=True)
        rec = {
            "config": json.dumps({k: c[k] for k in ("d", "T", "H", "B", "t_in", "pstyle", "istyle", "seed")}),
            "params_checksum": np.float64(synth.params_checksum(params)),
            "blendshapes": o["blendshapes"].numpy(),
            "mel_attention_weights": o["mel_attention_weights"].numpy(),
            "emotion_attention_weights": o["emotion_attention_weights"].numpy(),
            "mel_blendshapes": o["mel_blendshapes"].numpy(),
            "emotion_blendshapes": o["emotion_blendshapes"].numpy(),
        }
        if c["grads"]:
            m.zero_grad()
            masks = None
            if c.get("train"):
                m.train()
                torch.manual_seed(1000 + c["seed"])
            out = m(torch.from_numpy(mel), torch.from_numpy(short), torch.from_numpy(emo))["blendshapes"]
            if c.get("train"):
                from oracle import core as ocore
                masks = train_masks(1000 + c["seed"], c["B"], c["H"], c["d"], c["train"])
                chk = ocore.core_forward(params, mel, short, emo, num_heads=c["H"], mel_sequence_length=c["T"],
                                         dropout_p=c["train"], drop_masks=masks)["blendshapes"]
                err = float((chk - out.detach()).abs().max())
                assert err < 2e-6, f"{c['name']}: the reproduced dropout masks do not explain the reference's training-mode output ({err})"
                rec["train_blendshapes"] = out.detach().numpy()
                rec["dropout_p"] = np.float64(c["train"])
                for mk, mv in masks.items():
                    rec["mask/" + mk] = np.packbits(mv.reshape(-1))
                    rec["maskshape/" + mk] = np.array(mv.shape, np.int64)
            af = None
            if c.get("av"):      # audio_features (B, T, D): the reference's trainers pass the mel features here
                af = synth.make_av_features(c["seed"], c["B"])
            if c["grads"] == "full":
                from src.model.losses import KoeMorphLoss  # reference, read-only
                target, prev_pred, prev_target, lw = full_loss_inputs(synth, c["seed"], c["B"])
                crit = KoeMorphLoss()
                with torch.no_grad():
                    crit.landmark_loss.bs_to_landmark_weights.copy_(torch.from_numpy(lw))
                loss, metrics = crit(out, torch.from_numpy(target), prev_pred=torch.from_numpy(prev_pred),
                                     prev_target=torch.from_numpy(prev_target),
                                     audio_features=None if af is None else torch.from_numpy(af.astype(np.float32)))
                for mk in ("mse", "l1", "perceptual", "temporal", "velocity", "sparsity", "smoothness", "landmark"):
                    rec["metric/" + mk] = np.float64(metrics[mk])
            else:
                target = synth.uniform(c["seed"] * 3 + 1, (c["B"], 52), 0.0, 1.0)
                loss = torch.nn.functional.mse_loss(out, torch.from_numpy(target))
            loss.backward()
            rec["loss"] = np.float64(loss.item())
            for k, p in m.named_parameters():
                g = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
                if g.size <= 20000:
                    rec["grad/" + k] = g.astype(np.float32)
                else:                                    # large tensors: norm + strided sample
                    rec["gradnorm/" + k] = np.float64(np.sqrt(np.sum(g.astype(np.float64) ** 2)))
                    rec["gradsample/" + k] = g.ravel()[::97].astype(np.float32).copy()
        path = os.path.join(OUT, c["name"] + ".npz")
        np.savez_compressed(path, **rec)
        print(f"{c['name']}: bs[0,:3]={rec['blendshapes'][0,:3]}  ->  {os.path.getsize(path)/1024:.1f} KB")


# ---- legacy KoeMorphModel (src/model/gaussian_face.py): two chained frames per case ------------------------------
KOEMORPH_CASES = [
    dict(name="koemorph_d64_T20", cfg=dict(d_model=64, num_heads=4, num_encoder_layers=1, num_attention_layers=2,
                                           decoder_hidden_dim=32, decoder_layers=2, emotion_dim=24), B=3, T=20, seed=61),
    dict(name="koemorph_d256_T30_default", cfg=dict(), B=2, T=30, seed=62),
    dict(name="koemorph_d128_T75_open", cfg=dict(d_model=128, num_heads=8, num_encoder_layers=2, num_attention_layers=3,
                                                 decoder_hidden_dim=64, decoder_layers=1, decoder_activation="relu",
                                                 causal=False, window_size=None, use_constraints=False, emotion_dim=88),
         B=2, T=75, seed=63),
    # padded batch: audio_mask with 20 / 13 / 7 valid frames
    dict(name="koemorph_d64_T20_padded", cfg=dict(d_model=64, num_heads=4, num_encoder_layers=2, num_attention_layers=2,
                                                  decoder_hidden_dim=32, decoder_layers=1, emotion_dim=24), B=3, T=20, seed=65,
         valid=[20, 13, 7]),
    # default mask at T = 256: rows >= 5 have every key masked -> NaN in the reference (SURVEY: "default config is broken")
    dict(name="koemorph_d64_T256_masked", cfg=dict(d_model=64, num_heads=4, num_encoder_layers=1, num_attention_layers=1,
                                                   decoder_hidden_dim=32, decoder_layers=1, emotion_dim=24), B=1, T=256, seed=64),
    # decoder activations swish / leaky_relu (decoder.py:68-75) and output activations tanh / none (:162-167), three chained
    # frames.  The gaussian and median TemporalSmoother methods have NO fixture: the reference raises on their first call
    # (decoder.py:339 assigns a Python int to the registered buffer history_ptr -> TypeError), so they cannot be pinned.
    dict(name="koemorph_d64_T20_swish", cfg=dict(d_model=64, num_heads=4, num_encoder_layers=1, num_attention_layers=2,
                                                 decoder_hidden_dim=32, decoder_layers=2, emotion_dim=24, decoder_activation="swish"),
         B=3, T=20, seed=66, frames=3),
    dict(name="koemorph_d64_T20_leaky_tanh", cfg=dict(d_model=64, num_heads=4, num_encoder_layers=1, num_attention_layers=2,
                                                      decoder_hidden_dim=32, decoder_layers=1, emotion_dim=24,
                                                      decoder_activation="leaky_relu", output_activation="tanh", use_constraints=False),
         B=2, T=20, seed=67, frames=3),
    # the default width (the two fused kernels of km_kmmf.hip) with a padded batch, and with windows of <= 16 frames that share
    # MFMA row tiles in the fused encoder (ragged padding, no causal / local-window mask, swish decoder)
    dict(name="koemorph_d256_T30_padded", cfg=dict(), B=3, T=30, seed=69, valid=[30, 19, 8]),
    dict(name="koemorph_d256_T12_open", cfg=dict(causal=False, window_size=None, decoder_activation="swish"), B=5, T=12, seed=70,
         valid=[12, 12, 7, 12, 3], frames=3),
    dict(name="koemorph_d64_T20_none", cfg=dict(d_model=64, num_heads=4, num_encoder_layers=1, num_attention_layers=1,
                                                decoder_hidden_dim=32, decoder_layers=1, emotion_dim=24, output_activation="none"),
         B=2, T=20, seed=68, frames=3),
]


def koemorph_inputs(synth, seed, B, T, mel_dim, emotion_dim):
    return synth.normal(seed * 7 + 1, (B, T, mel_dim), std=1.0), synth.normal(seed * 7 + 2, (B, T, emotion_dim), std=1.0)


def main_koemorph():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from src.model.gaussian_face import KoeMorphModel  # reference, read-only
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from koemorph_amd import synth
    from oracle.koemorph_model import KoeMorphConfig, make_koemorph_params

    torch.set_num_threads(4)
    for c in KOEMORPH_CASES:
        if os.environ.get("KM_GOLDEN_ONLY") and os.environ["KM_GOLDEN_ONLY"] not in c["name"]:
            continue
        kc = KoeMorphConfig(**c["cfg"])
        params = make_koemorph_params(c["seed"], kc)
        m = KoeMorphModel(mel_dim=kc.mel_dim, emotion_dim=kc.emotion_dim, d_model=kc.d_model, d_query=kc.d_model,
                          num_heads=kc.num_heads, num_encoder_layers=kc.num_encoder_layers,
                          num_attention_layers=kc.num_attention_layers, decoder_hidden_dim=kc.decoder_hidden_dim,
                          decoder_layers=kc.decoder_layers, decoder_activation=kc.decoder_activation,
                          output_activation=kc.output_activation, smoothing_method=kc.smoothing_method,
                          use_temporal_smoothing=kc.use_temporal_smoothing, use_constraints=kc.use_constraints,
                          causal=kc.causal, window_size=kc.window_size).eval()
        sd = m.state_dict()
        missing = [k for k in sd if k not in params]          # buffers of the smoother / constraints keep their initial values
        assert all(k.startswith(("temporal_smoother.", "constraints.")) for k in missing), missing
        sd.update({k: torch.from_numpy(np.asarray(v)) for k, v in params.items()})
        m.load_state_dict(sd, strict=True)
        m.reset_temporal_state()
        assert m.temporal_smoother.window_size == kc.smoothing_window if kc.use_temporal_smoothing else True
        nf = c.get("frames", 2)
        rec = {"config": json.dumps(dict(cfg=kc.to_dict(), B=c["B"], T=c["T"], seed=c["seed"], frames=nf))}
        am = None
        if c.get("valid"):
            am = torch.arange(c["T"])[None, :] < torch.tensor(c["valid"])[:, None]
            rec["valid"] = np.asarray(c["valid"], dtype=np.int32)
        prev = None
        for fi in range(nf):                                   # frame i: inputs seeded seed + 100 i, previous frame's output fed back
            mel, emo = koemorph_inputs(synth, c["seed"] + 100 * fi, c["B"], c["T"], kc.mel_dim, kc.emotion_dim)
            with torch.no_grad():
                o = m(torch.from_numpy(mel), torch.from_numpy(emo), audio_mask=am, prev_blendshapes=prev, return_attention=True)
            prev = o["blendshapes"]
            tag = f"f{fi + 1}"
            rec[tag + "/blendshapes"] = o["blendshapes"].numpy()
            rec[tag + "/raw_blendshapes"] = o["raw_blendshapes"].numpy()
            if fi < 2:
                for li, w in enumerate(o["attention_weights"]):
                    rec[f"{tag}/attn{li}"] = w.numpy()[:, :, ::13, :].copy()        # query rows 0, 13, 26, 39 of every head
        path = os.path.join(OUT, c["name"] + ".npz")
        np.savez_compressed(path, **rec)
        print(f"{c['name']}: f1[0,:3]={rec['f1/blendshapes'][0,:3]} f2[0,:3]={rec['f2/blendshapes'][0,:3]} -> {os.path.getsize(path)/1024:.1f} KB")


if __name__ == "__main__":
    if os.environ.get("KM_GOLDEN_GROUP", "all") in ("all", "core"):
        main()
    if os.environ.get("KM_GOLDEN_GROUP", "all") in ("all", "koemorph"):
        main_koemorph()
