"""CPU checks of the weight blobs the fused KoeMorphModel kernels read (km_host.cpp build_kmmf_blobs, layout km_kmmf.h): every
nn.Linear weight is stored in MFMA fragment order -- image[(t * K/16 + kb) * 64 + lane][e] = W[16 t + (lane & 15)][16 kb +
4 (lane >> 4) + e] -- with biases and LayerNorm affines at fixed per-layer offsets.  A lane-level numpy model of one product
(the operand a lane feeds to the four MFMAs of a k block, lane group g contracting k = 4 g + s in MFMA s) reproduces x W^T."""
import ctypes as C

import numpy as np

from koemorph_amd import _lib
from koemorph_amd._lib import check
from oracle import koemorph_model as okm

D, FF, HID = 256, 1024, 128
ENC_HEAD = D * D + 3 * D
EL = dict(WIN=0, BIN=3 * D * D, WO=3 * D * D + 3 * D)
EL["BO"] = EL["WO"] + D * D
EL["N1G"] = EL["BO"] + D
EL["N1B"] = EL["N1G"] + D
EL["W1"] = EL["N1B"] + D
EL["B1"] = EL["W1"] + FF * D
EL["W2"] = EL["B1"] + FF
EL["B2"] = EL["W2"] + D * FF
EL["N2G"] = EL["B2"] + D
EL["N2B"] = EL["N2G"] + D
ENC_LAYER = EL["N2B"] + D
CROSS_LAYER = 4 * (D * D + D) + 2 * D


def unpack(img, N, K):
    """fragment image -> W (N, K)"""
    img = np.asarray(img).reshape(N // 16, K // 16, 64, 4)
    W = np.zeros((N, K), dtype=img.dtype)
    for lane in range(64):
        j, g = lane & 15, lane >> 4
        for e in range(4):
            W[j::16, 4 * g + e::16] = img[:, :, lane, e]
    return W


def test_blobs_hold_every_weight_in_fragment_order():
    from koemorph_amd.model import KoeMorphModel
    cfg = okm.KoeMorphConfig()
    params = okm.make_koemorph_params(33, cfg)
    m = KoeMorphModel(d_query=cfg.d_model)
    lib = _lib.load()
    h = C.c_void_p()
    check(lib.km_koemorph_create(C.byref(m._c_config()), C.byref(h)))
    try:
        import torch
        sd = m.state_dict()
        sd.update({k: torch.from_numpy(np.asarray(v)) for k, v in params.items()})
        m.load_state_dict(sd)
        for k, v in m.named_parameters():                 # what KoeMorphModel._handle loads
            a = np.ascontiguousarray(v.detach().cpu().numpy(), dtype=np.float32)
            shape = (C.c_int64 * max(1, v.dim()))(*v.shape)
            check(lib.km_load_param(h, k.encode(), a.ctypes.data_as(C.c_void_p), shape, v.dim()))
        check(lib.km_finalize_host(h))

        def buf(name):
            n = C.c_int64(0)
            check(lib.km_debug_buffer(h, name.encode(), None, C.byref(n)))
            out = np.empty(n.value, dtype=np.float32)
            check(lib.km_debug_buffer(h, name.encode(), out.ctypes.data_as(C.c_void_p), C.byref(n)))
            return out
        enc, cross, dec = buf("kmf_enc"), buf("kmf_cross"), buf("kmf_dec")
        L = cfg.num_encoder_layers
        stream = ENC_HEAD + L * ENC_LAYER
        assert enc.size == 2 * stream and cross.size == cfg.num_attention_layers * CROSS_LAYER
        for s, st in enumerate(("mel", "emotion")):
            b = enc[s * stream:(s + 1) * stream]
            W0 = unpack(b[:D * D], D, D)
            k_in = cfg.mel_dim if s == 0 else cfg.emotion_dim
            assert np.array_equal(W0[:, :k_in], params[f"audio_encoder.{st}_encoder.0.weight"]) and not W0[:, k_in:].any()
            assert np.array_equal(b[D * D:D * D + D], params[f"audio_encoder.{st}_encoder.0.bias"])
            for i in range(L):
                l = b[ENC_HEAD + i * ENC_LAYER:ENC_HEAD + (i + 1) * ENC_LAYER]
                p = f"audio_encoder.{st}_transformer.layers.{i}."
                assert np.array_equal(unpack(l[EL["WIN"]:EL["BIN"]], 3 * D, D), params[p + "self_attn.in_proj_weight"])
                assert np.array_equal(l[EL["BIN"]:EL["WO"]], params[p + "self_attn.in_proj_bias"])
                assert np.array_equal(unpack(l[EL["WO"]:EL["BO"]], D, D), params[p + "self_attn.out_proj.weight"])
                assert np.array_equal(unpack(l[EL["W1"]:EL["B1"]], FF, D), params[p + "linear1.weight"])
                assert np.array_equal(unpack(l[EL["W2"]:EL["B2"]], D, FF), params[p + "linear2.weight"])
                assert np.array_equal(l[EL["N2G"]:EL["N2B"]], params[p + "norm2.weight"])
        for i in range(cfg.num_attention_layers):
            l = cross[i * CROSS_LAYER:(i + 1) * CROSS_LAYER]
            for q, nm in enumerate(("q_proj", "k_proj", "v_proj", "out_proj")):
                o = q * (D * D + D)
                assert np.array_equal(unpack(l[o:o + D * D], D, D), params[f"cross_attention_layers.{i}.{nm}.weight"])
                assert np.array_equal(l[o + D * D:o + D * D + D], params[f"cross_attention_layers.{i}.{nm}.bias"])
            assert np.array_equal(l[4 * (D * D + D):4 * (D * D + D) + D], params[f"attention_layer_norms.{i}.weight"])
        assert np.array_equal(unpack(dec[:HID * D], HID, D), params["decoder.input_proj.weight"])
        o = HID * D + HID
        assert np.array_equal(unpack(dec[o:o + HID * HID], HID, HID), params["decoder.hidden_layers.0.weight"])

        # lane-level model of mm_cols on one tile: lane (g, j) holds A = x[row j][16 kb + 4 g + s], B = image[(t, kb), lane][s]
        rng = np.random.default_rng(5)
        x = rng.standard_normal((16, D)).astype(np.float32)
        img = cross[:D * D].reshape(D // 16, D // 16, 64, 4)            # q_proj of layer 0
        t = 3
        acc = np.zeros((16, 16))
        for kb in range(D // 16):
            for s_ in range(4):
                for g in range(4):
                    a = x[:, 16 * kb + 4 * g + s_].astype(np.float64)                          # A[i][k = g]
                    bv = img[t, kb, 16 * g:16 * g + 16, s_].astype(np.float64)                 # B[k = g][n = j]
                    acc += np.outer(a, bv)
        want = x.astype(np.float64) @ params["cross_attention_layers.0.q_proj.weight"][16 * t:16 * t + 16].astype(np.float64).T
        assert np.abs(acc - want).max() < 1e-9
    finally:
        lib.km_destroy(h)


def test_legacy_blobs_hold_the_weights_in_fragment_order():
    """SimplifiedKoeMorphModel's fused kernels read lgf_enc (audio encoder + key / value projections) and lgf_tail (out_proj +
    decoder, the 52-row output weight padded to 64 rows)."""
    from koemorph_amd.model import SimplifiedKoeMorphModel
    from oracle import legacy
    import torch
    params = legacy.make_legacy_params(3)
    m = SimplifiedKoeMorphModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    lib = _lib.load()
    h = C.c_void_p()
    from koemorph_amd._lib import KMLegacyConfig
    from koemorph_amd.engine import MelConfig
    mel = MelConfig.model_batch(m.sample_rate, m.target_fps, m.n_fft)
    mel.hop_length = m.hop_length
    cfg = KMLegacyConfig(_lib.KM_ABI_VERSION, m.d_model, m.num_heads, m.decoder_hidden, m.num_blendshapes, mel.to_c())
    check(lib.km_legacy_create(C.byref(cfg), C.byref(h)))
    try:
        for k, v in m.state_dict().items():               # what SimplifiedKoeMorphModel._handle loads
            a = np.ascontiguousarray(v.detach().cpu().numpy(), dtype=np.float32)
            shape = (C.c_int64 * max(1, a.ndim))(*a.shape)
            check(lib.km_load_param(h, k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
        check(lib.km_finalize_host(h))

        def buf(name):
            n = C.c_int64(0)
            check(lib.km_debug_buffer(h, name.encode(), None, C.byref(n)))
            out = np.empty(n.value, dtype=np.float32)
            check(lib.km_debug_buffer(h, name.encode(), out.ctypes.data_as(C.c_void_p), C.byref(n)))
            return out
        enc, tail = buf("lgf_enc"), buf("lgf_tail")
        o = 0
        assert np.array_equal(unpack(enc[o:o + D * 80], D, 80), params["audio_encoder.0.weight"]); o += D * 80
        assert np.array_equal(enc[o:o + D], params["audio_encoder.0.bias"]); o += D
        assert np.array_equal(unpack(enc[o:o + D * D], D, D), params["audio_encoder.3.weight"]); o += D * D + D
        inw = params["attention.in_proj_weight"]
        assert np.array_equal(unpack(enc[o:o + D * D], D, D), inw[D:2 * D]); o += D * D
        assert np.array_equal(enc[o:o + D], params["attention.in_proj_bias"][D:2 * D]); o += D
        assert np.array_equal(unpack(enc[o:o + D * D], D, D), inw[2 * D:]); o += D * D + D
        assert o == enc.size
        o = D * D + D + HID * D + HID + HID * HID + HID
        w6 = unpack(tail[o:o + 64 * HID], 64, HID)
        assert np.array_equal(w6[:52], params["decoder.6.weight"]) and not w6[52:].any()
        assert np.array_equal(tail[o + 64 * HID:o + 64 * HID + 52], params["decoder.6.bias"]) and tail.size == o + 64 * HID + 64
    finally:
        lib.km_destroy(h)
