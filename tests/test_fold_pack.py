"""CPU checks of the host half of the library (no GPU needed):

1. the folded weights (km_host.cpp) reproduce the oracle when evaluated in float64;
2. a lane-level numpy model of the fused gfx950 kernel -- same packed operand images, same
   fragment maps of v_mfma_f32_16x16x4_f32, same LDS index expressions as km_core.hip --
   reproduces the oracle.  This pins the packing / layout logic before the code ever
   reaches a GPU.
"""
import numpy as np
import pytest

from conftest import golden_case
from koemorph_amd import synth
from koemorph_amd.engine import Engine
from oracle import core

MOUTH = synth.MOUTH_INDICES
EXPR = synth.EXPRESSION_INDICES


def make_engine(params, d=256, T=256, H=8):
    e = Engine(d_model=d, num_heads=H, mel_sequence_length=T)
    e.load_state_dict(params)
    e.finalize_host()
    return e


def layer_norm(x, g, b):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + 1e-5) * g + b


def folded_forward(e, params, mel, short, emo, d, T, H):
    """Evaluate the FOLDED network in float64 from the library's debug buffers."""
    f = lambda n: e.debug_buffer(n).astype(np.float64)
    DH = d // 2
    qk = f("qk").reshape(H, 28, d)
    wf, bf = f("wf").reshape(d, DH), f("bf")
    we2, be2 = f("we2").reshape(d, DH), f("be2")
    wee_t = f("wee_t").reshape(-1, d)
    w2, b2, wsum = f("w2"), f("b2"), f("wsum")
    B = mel.shape[0]
    x = np.zeros((B, 80, T + 3))
    tv = min(T, mel.shape[1])
    x[:, :, :tv] = mel[:, :tv].transpose(0, 2, 1)
    x[:, :, T:] = short.transpose(0, 2, 1)
    P = {k: v.astype(np.float64) for k, v in params.items()}
    y = layer_norm(x @ P["mel_channel_encoder.weight"].T + P["mel_channel_encoder.bias"],
                   P["mel_norm.weight"], P["mel_norm.bias"])                          # (B,80,d)
    wv = P["mel_attention.in_proj_weight"][2 * d:]
    hd = d // H
    out = np.zeros((B, 52))
    attn = np.zeros((B, 28, 80))
    for b in range(B):
        o = np.zeros((28, d))
        for h in range(H):
            s = qk[h] @ y[b].T                                                       # (28,80)
            p = np.exp(s - s.max(-1, keepdims=True))
            p /= p.sum(-1, keepdims=True)
            attn[b] += p / H
            o[:, h * hd:(h + 1) * hd] = p @ (y[b] @ wv[h * hd:(h + 1) * hd].T)
        z = np.maximum(o @ wf + bf, 0) @ w2 + b2[0]
        e1 = layer_norm(emo[b].astype(np.float64) @ wee_t + f("bee"), f("eln_g"), f("eln_b"))
        ze = np.maximum(e1 @ we2 + be2, 0) @ w2 + b2[0]
        bs = np.zeros(52)
        bs[MOUTH] = 1 / (1 + np.exp(-z))
        bs[EXPR] = 1 / (1 + np.exp(-ze))
        out[b] = np.clip(wsum * bs, 0, 1)
    return out, attn


@pytest.mark.parametrize("name", ["core_d256_T256_H8_trained", "core_d256_pad_T100", "core_d512_T512_H16",
                                  "core_d64_T32_H4_small"])
def test_folded_weights_reproduce_reference(name):
    c, params, (mel, short, emo), g = golden_case(name)
    e = make_engine(params, c["d"], c["T"], c["H"])
    out, attn = folded_forward(e, params, mel, short, emo, c["d"], c["T"], c["H"])
    np.testing.assert_allclose(out, g["blendshapes"], atol=3e-7)
    np.testing.assert_allclose(attn, g["mel_attention_weights"], atol=3e-7)


def test_param_errors_mirror_load_state_dict():
    from koemorph_amd._lib import KoeMorphError
    e = Engine()
    with pytest.raises(KoeMorphError, match="size mismatch"):
        e.load_param("mel_norm.weight", np.zeros(3, np.float32))
    with pytest.raises(KoeMorphError, match="unexpected key"):
        e.load_param("nope", np.zeros(3, np.float32))
    with pytest.raises(KoeMorphError, match="never loaded"):
        e.finalize_host()
    e.load_param("dual_stream_attention.mel_norm.weight", np.ones(256, np.float32))   # full-model prefix accepted
    assert e.param_count() == (29, 2)       # smoothing_alpha has a default


# ------------------------------------------------------------------------------------------
# lane-level model of core_fused_kernel
# ------------------------------------------------------------------------------------------
L = np.arange(64)
G, J = L >> 4, L & 15


def mfma(a, b, c):
    """v_mfma_f32_16x16x4_f32: lane l supplies A[l&15][l>>4], B[l>>4][l&15]; C/D col=l&15, row=4(l>>4)+reg."""
    A = np.zeros((16, 4)); Bm = np.zeros((4, 16))
    A[J, G] = a
    Bm[G, J] = b
    D = A @ Bm
    out = c.copy()
    for r in range(4):
        out[:, r] += D[4 * G + r, J]
    return out


def test_lane_level_model_of_fused_kernel():
    c, params, (mel, short, emo), g = golden_case("core_d256_pad_T100")   # also exercises the zero-pad rows
    e = make_engine(params)
    T, NK, KTP, YS, KP, KB, NW = 256, 80, 264, 264, 33, 16, 8
    wce_p = e.debug_buffer("wce_p").astype(np.float64).reshape(NW, KP, 64, 4)
    qk_p = e.debug_buffer("qk_p").astype(np.float64).reshape(8, KB, 2, 64, 4)
    wv_p = e.debug_buffer("wv_p").astype(np.float64).reshape(8, KB, 2, 64, 4)
    wf_p = e.debug_buffer("wf_p").astype(np.float64).reshape(8, KB, 64, 4)
    bce, ln_g, ln_b = (e.debug_buffer(n).astype(np.float64) for n in ("bce", "ln_g", "ln_b"))
    bf, w2, b2, wsum = (e.debug_buffer(n).astype(np.float64) for n in ("bf", "w2", "b2", "wsum"))
    b = 1
    t_in = mel.shape[1]
    # phase 0
    R1 = np.zeros(KTP * NK)
    tv = min(t_in, T)
    R1[:tv * 80] = mel[b, :tv].ravel()
    R1[T * 80:T * 80 + 240] = short[b].ravel()
    # phase 1
    acc = np.zeros((NW, 5, 2, 64, 4))
    for w in range(NW):
        for kp in range(KP):
            bw = wce_p[w, kp]
            for ds in range(2):
                xr = (4 * (2 * kp + ds) + G) * NK + J
                for mt in range(5):
                    av = R1[xr + 16 * mt]
                    acc[w, mt, 0] = mfma(av, bw[:, 2 * ds + 0], acc[w, mt, 0])
                    acc[w, mt, 1] = mfma(av, bw[:, 2 * ds + 1], acc[w, mt, 1])
        n0 = 32 * w + J
        acc[w, :, 0] += bce[n0][None, :, None]
        acc[w, :, 1] += bce[n0 + 16][None, :, None]
    # LayerNorm via the same row bookkeeping: row = 16mt + 4g + r, columns spread over waves / lanes j
    Yfull = np.zeros((80, 256))
    for w in range(NW):
        for mt in range(5):
            for t in range(2):
                for r in range(4):
                    Yfull[16 * mt + 4 * G + r, 32 * w + 16 * t + J] = acc[w, mt, t][:, r]
    Yn = layer_norm(Yfull, ln_g, ln_b)
    ref = core.core_forward(params, mel, short, emo, return_intermediates=True, return_attention=True)
    np.testing.assert_allclose(Yn, ref["_y"][b].numpy(), atol=2e-5)
    R1 = np.zeros(KTP * NK)
    for row in range(80):
        R1[row * YS:row * YS + 256] = Yn[row]
    # phases 2-4 per wave/head, O image
    O_img = np.zeros(32 * YS)
    attn = np.zeros((28, 80))
    for w in range(NW):
        S = np.zeros((5, 2, 64, 4)); V = np.zeros((5, 2, 64, 4))
        for kb in range(KB):
            ya = [np.stack([R1[(16 * mt + J) * YS + 16 * kb + 4 * G + s] for s in range(4)], 1) for mt in range(5)]
            for s in range(4):
                for mt in range(5):
                    av = ya[mt][:, s]
                    S[mt, 0] = mfma(av, qk_p[w, kb, 0][:, s], S[mt, 0])
                    S[mt, 1] = mfma(av, qk_p[w, kb, 1][:, s], S[mt, 1])
                    V[mt, 0] = mfma(av, wv_p[w, kb, 0][:, s], V[mt, 0])
                    V[mt, 1] = mfma(av, wv_p[w, kb, 1][:, s], V[mt, 1])
        for qt in range(2):
            m = S[:, qt].max(axis=(0, 2))                      # in-lane over (mt, r)
            m = np.maximum(m, m[L ^ 16]); m = np.maximum(m, m[L ^ 32])
            ex = np.exp(S[:, qt] - m[None, :, None])
            sm = ex.sum(axis=(0, 2))
            sm = sm + sm[L ^ 16]; sm = sm + sm[L ^ 32]
            S[:, qt] = ex / sm[None, :, None]
        O = np.zeros((2, 2, 64, 4))
        for mt in range(5):
            for r in range(4):
                for dt in range(2):
                    for qt in range(2):
                        O[dt, qt] = mfma(V[mt, dt][:, r], S[mt, qt][:, r], O[dt, qt])
        for dt in range(2):
            for qt in range(2):
                base = (16 * qt + J) * YS + 32 * w + 16 * dt + 4 * G
                for r in range(4):
                    O_img[base + r] = O[dt, qt][:, r]
        for qt in range(2):
            for mt in range(5):
                for r in range(4):
                    q = 16 * qt + J
                    ok = q < 28
                    attn[q[ok], (16 * mt + 4 * G + r)[ok]] += S[mt, qt][ok, r] / 8
    np.testing.assert_allclose(attn, g["mel_attention_weights"][b], atol=3e-7)
    # phase 5
    R2 = np.zeros((NW, 32))
    for w in range(NW):
        Z = np.zeros((2, 64, 4))
        for kb in range(KB):
            wa = wf_p[w, kb]
            o0 = np.stack([O_img[J * YS + 16 * kb + 4 * G + s] for s in range(4)], 1)
            o1 = np.stack([O_img[(16 + J) * YS + 16 * kb + 4 * G + s] for s in range(4)], 1)
            for s in range(4):
                Z[0] = mfma(wa[:, s], o0[:, s], Z[0])
                Z[1] = mfma(wa[:, s], o1[:, s], Z[1])
        zp = np.zeros((2, 64))
        for r in range(4):
            hid = 16 * w + 4 * G + r
            for qt in range(2):
                zp[qt] += np.maximum(Z[qt][:, r] + bf[hid], 0) * w2[hid]
        for qt in range(2):
            zp[qt] = zp[qt] + zp[qt][L ^ 16]
            zp[qt] = zp[qt] + zp[qt][L ^ 32]
            R2[w, 16 * qt + J[G == 0]] = zp[qt][G == 0]
    z = R2.sum(0)[:28] + b2[0]
    np.testing.assert_allclose(z, ref["_z"][b, MOUTH].numpy(), atol=2e-5)
    bs = 1 / (1 + np.exp(-z))
    np.testing.assert_allclose(np.clip(wsum[MOUTH] * bs, 0, 1), g["blendshapes"][b, MOUTH], atol=3e-7)


def test_scalar_parameter_roundtrip():
    e = Engine()
    e.load_param("smoothing_alpha", np.float32(0.25))
    e.load_param("smoothing_alpha", __import__("torch").tensor(0.5))
    assert float(e.get_param("smoothing_alpha", ())) == 0.5
