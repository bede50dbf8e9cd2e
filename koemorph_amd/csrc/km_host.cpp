// Host side of libkoemorph_hip.so: parameter store, weight folding and packing, mel plans.
// No HIP calls in this file, so it also runs on a machine without a GPU (CPU tests).
//
// Folding (eval mode, done once in double precision):
//   * the mouth queries are input independent (dual_stream_attention.py:221), so
//     Q = mouth_queries Wq^T + bq is precomputed, scaled by 1/sqrt(hd) and multiplied into the key
//     projection:  scores_h = (Q_h Wk_h) Y^T + Q_h bk_h.  The second term is constant along the
//     key axis and cancels in the softmax (:225-230), so it is dropped.
//   * out_proj -> mel_output_proj -> decoder[0] (:231, :248, :150-151) are consecutive affine maps
//     with no non-linearity in between: folded into one (d x d/2) matrix.  The value bias bv rides
//     through because softmax rows sum to one.
//   * the emotion stream attends over ONE key (:234-239): softmax == 1, so its output is
//     out_proj(Wv e + bv) for every query; everything after the emotion LayerNorm folds into one
//     (d x d/2) matrix.
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "km_context.h"
#include "km_kmmf.h"

namespace km {

const int kMouthIdx[kNumMouth] = {14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27,
                                  28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 51};
const int kExprIdx[kNumExpr] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13,
                                41, 42, 43, 44, 45, 46, 47, 48, 49, 50};

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

const char* last_error() { return g_err; }

using dvec = std::vector<double>;

static dvec to_d(const std::vector<float>& v) { return dvec(v.begin(), v.end()); }

// C[M x N] = A[M x K] * B[K x N]
static dvec mm(const dvec& A, const dvec& B, int M, int K, int N) {
    dvec C((size_t)M * N, 0.0);
    for (int i = 0; i < M; ++i)
        for (int k = 0; k < K; ++k) {
            const double a = A[(size_t)i * K + k];
            const double* b = &B[(size_t)k * N];
            double* c = &C[(size_t)i * N];
            for (int j = 0; j < N; ++j) c[j] += a * b[j];
        }
    return C;
}

// bf16 by round-to-nearest-even on the fp32 bits (what v_cvt_pk_bf16_f32 does for finite values) and back
static inline uint16_t bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf16_to_float(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

static dvec transpose(const dvec& A, int R, int Cn) {
    dvec T((size_t)R * Cn);
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < Cn; ++j) T[(size_t)j * R + i] = A[(size_t)i * Cn + j];
    return T;
}

static std::vector<float> to_f(const dvec& v) { return std::vector<float>(v.begin(), v.end()); }

static const std::vector<float>& P(Context* c, const char* k) { return c->params.at(k).data; }

static void put(Context* c, const char* name, std::vector<float>&& v) {
    c->packed[name].host = std::move(v);
}

static dvec softmax_d(const std::vector<float>& w, double temp) {
    dvec r(w.size());
    double m = -1e300;
    for (float x : w) m = std::fmax(m, x / temp);
    double s = 0;
    for (size_t i = 0; i < w.size(); ++i) { r[i] = std::exp(w[i] / temp - m); s += r[i]; }
    for (double& x : r) x /= s;
    return r;
}

// Chain y = ((x Wa^T + ba) Wb^T + bb) W1^T + b1 folded into x F + f, with an input bias bin that
// is pushed through as well (x -> x + bin first).  Wa, Wb are (d x d) [out][in], W1 is (DH x d).
static void fold_chain(const dvec& bin, const dvec& Wa, const dvec& ba, const dvec& Wb, const dvec& bb,
                       const dvec& W1, const dvec& b1, int d, int DH, dvec& F, dvec& f) {
    dvec WaT = transpose(Wa, d, d), WbT = transpose(Wb, d, d), W1T = transpose(W1, DH, d);  // [in][out]
    dvec M2 = mm(WaT, WbT, d, d, d);
    F = mm(M2, W1T, d, d, DH);                                                               // d x DH
    dvec t1 = mm(bin, WaT, 1, d, d);
    for (int i = 0; i < d; ++i) t1[i] += ba[i];
    dvec t2 = mm(t1, WbT, 1, d, d);
    for (int i = 0; i < d; ++i) t2[i] += bb[i];
    f = mm(t2, W1T, 1, d, DH);
    for (int i = 0; i < DH; ++i) f[i] += b1[i];
}

static void pack_frag(const std::vector<float>& W, int N, int K, int Kpad, float* dst);

// SimplifiedKoeMorphModel (simplified_model.py:44-77): the queries are input independent, so
// Q = (blendshape_queries Wq^T + bq) / sqrt(hd) is precomputed; everything else keeps the reference's layers.
int finalize_host_legacy(Context* c) {
    const int d = c->d, hd = c->hd, NQ = c->NB;
    const dvec inw = to_d(P(c, "attention.in_proj_weight")), inb = to_d(P(c, "attention.in_proj_bias"));
    dvec Wq(inw.begin(), inw.begin() + (size_t)d * d);
    dvec Q = mm(to_d(P(c, "blendshape_queries")), transpose(Wq, d, d), NQ, d, d);
    const double scale = 1.0 / std::sqrt((double)hd);
    for (int q = 0; q < NQ; ++q)
        for (int i = 0; i < d; ++i) Q[(size_t)q * d + i] = (Q[(size_t)q * d + i] + inb[i]) * scale;
    put(c, "l_q", to_f(Q));
    put(c, "l_wk", std::vector<float>(P(c, "attention.in_proj_weight").begin() + (size_t)d * d,
                                      P(c, "attention.in_proj_weight").begin() + (size_t)2 * d * d));
    put(c, "l_wv", std::vector<float>(P(c, "attention.in_proj_weight").begin() + (size_t)2 * d * d,
                                      P(c, "attention.in_proj_weight").end()));
    put(c, "l_bk", std::vector<float>(P(c, "attention.in_proj_bias").begin() + d, P(c, "attention.in_proj_bias").begin() + 2 * d));
    put(c, "l_bv", std::vector<float>(P(c, "attention.in_proj_bias").begin() + 2 * d, P(c, "attention.in_proj_bias").end()));
    const char* raw[][2] = {{"l_w0", "audio_encoder.0.weight"}, {"l_b0", "audio_encoder.0.bias"},
                            {"l_w3", "audio_encoder.3.weight"}, {"l_b3", "audio_encoder.3.bias"},
                            {"l_wo", "attention.out_proj.weight"}, {"l_bo", "attention.out_proj.bias"},
                            {"l_d0w", "decoder.0.weight"}, {"l_d0b", "decoder.0.bias"},
                            {"l_d3w", "decoder.3.weight"}, {"l_d3b", "decoder.3.bias"},
                            {"l_d6w", "decoder.6.weight"}, {"l_d6b", "decoder.6.bias"}};
    for (auto& r : raw) put(c, r[0], std::vector<float>(P(c, r[1])));
    // the audio encoder and the key / value projections as ONE rows-resident kernel (km_kmmf.hip legacy_encoder_kernel): weights in
    // MFMA fragment order (km_kmmf.h), at the reference's width
    c->legacy_fused = false;
    if (d == kmmf::D && c->NK == kmmf::LG_MEL && c->H == kmmf::HEADS) {
        using namespace kmmf;
        std::vector<float> blob((size_t)LG_FLOATS, 0.f);
        pack_frag(P(c, "audio_encoder.0.weight"), D, LG_MEL, LG_MEL, blob.data() + LG_W0);
        std::copy(P(c, "audio_encoder.0.bias").begin(), P(c, "audio_encoder.0.bias").end(), blob.begin() + LG_B0);
        pack_frag(P(c, "audio_encoder.3.weight"), D, D, D, blob.data() + LG_W3);
        std::copy(P(c, "audio_encoder.3.bias").begin(), P(c, "audio_encoder.3.bias").end(), blob.begin() + LG_B3);
        const std::vector<float>& inw_f = P(c, "attention.in_proj_weight");
        const std::vector<float>& inb_f = P(c, "attention.in_proj_bias");
        pack_frag(std::vector<float>(inw_f.begin() + (size_t)D * D, inw_f.begin() + (size_t)2 * D * D), D, D, D, blob.data() + LG_WK);
        std::copy(inb_f.begin() + D, inb_f.begin() + 2 * D, blob.begin() + LG_BK);
        pack_frag(std::vector<float>(inw_f.begin() + (size_t)2 * D * D, inw_f.end()), D, D, D, blob.data() + LG_WV);
        std::copy(inb_f.begin() + 2 * D, inb_f.end(), blob.begin() + LG_BV);
        put(c, "lgf_enc", std::move(blob));
        c->legacy_fused = true;
        c->legacy_tail_fused = false;
        if (c->legacy_hidden == HID && NQ == kmmf::NQ) {     // out_proj + decoder + sigmoid + mean over the query rows: legacy_tail_kernel
            std::vector<float> t((size_t)LT_FLOATS, 0.f);
            pack_frag(P(c, "attention.out_proj.weight"), D, D, D, t.data() + LT_WO);
            std::copy(P(c, "attention.out_proj.bias").begin(), P(c, "attention.out_proj.bias").end(), t.begin() + LT_BO);
            pack_frag(P(c, "decoder.0.weight"), HID, D, D, t.data() + LT_W0);
            std::copy(P(c, "decoder.0.bias").begin(), P(c, "decoder.0.bias").end(), t.begin() + LT_B0);
            pack_frag(P(c, "decoder.3.weight"), HID, HID, HID, t.data() + LT_W3);
            std::copy(P(c, "decoder.3.bias").begin(), P(c, "decoder.3.bias").end(), t.begin() + LT_B3);
            std::vector<float> w6((size_t)64 * HID, 0.f);
            std::copy(P(c, "decoder.6.weight").begin(), P(c, "decoder.6.weight").end(), w6.begin());
            pack_frag(w6, 64, HID, HID, t.data() + LT_W6);
            std::copy(P(c, "decoder.6.bias").begin(), P(c, "decoder.6.bias").end(), t.begin() + LT_B6);
            put(c, "lgf_tail", std::move(t));
            c->legacy_tail_fused = true;
        }
    }
    c->fused_ok = false;
    c->host_finalized = true;
    return KM_OK;
}

// Weight blobs of the fused KoeMorphModel kernels (km_kmmf.hip; layout and the fragment packing: km_kmmf.h).  Built when the
// model has the width those kernels are written for (the reference's defaults); anything else runs the launch-per-step chain.
static void pack_frag(const std::vector<float>& W, int N, int K, int Kpad, float* dst) {
    const int KB = Kpad / 16;
    for (int t = 0; t < N / 16; ++t)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int n = 16 * t + (lane & 15), k = 16 * kb + 4 * (lane >> 4) + e;
                    dst[(((size_t)t * KB + kb) * 64 + lane) * 4 + e] = k < K ? W[(size_t)n * K + k] : 0.f;
                }
}

static bool build_kmmf_blobs(Context* c) {
    using namespace kmmf;
    const km_koemorph_config& k = c->kmm;
    if (c->d != D || c->H != HEADS || c->NB != NQ || k.decoder_hidden_dim != HID || k.mel_dim % 16 || k.emotion_dim % 16 ||
        k.mel_dim > D || k.emotion_dim > D || k.mel_dim <= 0 || k.emotion_dim <= 0 || k.num_attention_layers < 1)
        return false;
    auto Pv = [&](const std::string& n) -> const std::vector<float>& { return c->params.at(n).data; };
    auto copy = [](const std::vector<float>& v, float* dst) { std::copy(v.begin(), v.end(), dst); };
    const int LE = k.num_encoder_layers, LC = k.num_attention_layers, LD = k.decoder_layers;
    std::vector<float> enc((size_t)2 * enc_stream_floats(LE), 0.f);
    for (int s = 0; s < 2; ++s) {
        float* b = enc.data() + (size_t)s * enc_stream_floats(LE);
        const std::string st = s ? "emotion" : "mel";
        const std::string e = "audio_encoder." + st + "_encoder.";
        pack_frag(Pv(e + "0.weight"), D, s ? k.emotion_dim : k.mel_dim, D, b + ENC_W0);
        copy(Pv(e + "0.bias"), b + ENC_B0);
        copy(Pv(e + "3.weight"), b + ENC_LNG);
        copy(Pv(e + "3.bias"), b + ENC_LNB);
        for (int i = 0; i < LE; ++i) {
            float* l = b + ENC_HEAD + (size_t)i * ENC_LAYER;
            const std::string p = "audio_encoder." + st + "_transformer.layers." + std::to_string(i) + ".";
            pack_frag(Pv(p + "self_attn.in_proj_weight"), 3 * D, D, D, l + EL_WIN);
            copy(Pv(p + "self_attn.in_proj_bias"), l + EL_BIN);
            pack_frag(Pv(p + "self_attn.out_proj.weight"), D, D, D, l + EL_WO);
            copy(Pv(p + "self_attn.out_proj.bias"), l + EL_BO);
            copy(Pv(p + "norm1.weight"), l + EL_N1G);
            copy(Pv(p + "norm1.bias"), l + EL_N1B);
            pack_frag(Pv(p + "linear1.weight"), FF, D, D, l + EL_W1);
            copy(Pv(p + "linear1.bias"), l + EL_B1);
            pack_frag(Pv(p + "linear2.weight"), D, FF, FF, l + EL_W2);
            copy(Pv(p + "linear2.bias"), l + EL_B2);
            copy(Pv(p + "norm2.weight"), l + EL_N2G);
            copy(Pv(p + "norm2.bias"), l + EL_N2B);
        }
    }
    std::vector<float> cross((size_t)LC * CROSS_LAYER, 0.f);
    for (int i = 0; i < LC; ++i) {
        float* l = cross.data() + (size_t)i * CROSS_LAYER;
        const std::string p = "cross_attention_layers." + std::to_string(i) + ".";
        pack_frag(Pv(p + "q_proj.weight"), D, D, D, l + CL_WQ);  copy(Pv(p + "q_proj.bias"), l + CL_BQ);
        pack_frag(Pv(p + "k_proj.weight"), D, D, D, l + CL_WK);  copy(Pv(p + "k_proj.bias"), l + CL_BK);
        pack_frag(Pv(p + "v_proj.weight"), D, D, D, l + CL_WV);  copy(Pv(p + "v_proj.bias"), l + CL_BV);
        pack_frag(Pv(p + "out_proj.weight"), D, D, D, l + CL_WO);  copy(Pv(p + "out_proj.bias"), l + CL_BO);
        copy(Pv("attention_layer_norms." + std::to_string(i) + ".weight"), l + CL_LNG);
        copy(Pv("attention_layer_norms." + std::to_string(i) + ".bias"), l + CL_LNB);
    }
    std::vector<float> dec((size_t)DC_HEAD + (size_t)LD * DEC_LAYER, 0.f);
    pack_frag(Pv("decoder.input_proj.weight"), HID, D, D, dec.data() + DC_WI);
    copy(Pv("decoder.input_proj.bias"), dec.data() + DC_BI);
    for (int i = 0; i < LD; ++i) {
        float* l = dec.data() + DC_HEAD + (size_t)i * DEC_LAYER;
        const std::string n = std::to_string(i);
        pack_frag(Pv("decoder.hidden_layers." + n + ".weight"), HID, HID, HID, l + DL_W);
        copy(Pv("decoder.hidden_layers." + n + ".bias"), l + DL_B);
        copy(Pv("decoder.layer_norms." + n + ".weight"), l + DL_LNG);
        copy(Pv("decoder.layer_norms." + n + ".bias"), l + DL_LNB);
    }
    put(c, "kmf_enc", std::move(enc));
    put(c, "kmf_cross", std::move(cross));
    put(c, "kmf_dec", std::move(dec));
    return true;
}

// KoeMorphModel (gaussian_face.py:29-173): every tensor keeps the reference's layout (nn.Linear weights are (out, in) =
// the B^T operand of the NT GEMM); the key / value projections of ALL cross-attention layers read the same encoded
// audio, so they are stacked into one (2 L d, d) weight: one GEMM instead of 2 L.
int finalize_host_koemorph(Context* c) {
    const int d = c->d, L = c->kmm.num_attention_layers;
    for (const auto& k : c->param_order) put(c, k.c_str(), std::vector<float>(P(c, k.c_str())));
    std::vector<float> w((size_t)2 * L * d * d), b((size_t)2 * L * d);
    for (int i = 0; i < L; ++i)
        for (int kv = 0; kv < 2; ++kv) {
            const std::string base = "cross_attention_layers." + std::to_string(i) + (kv ? ".v_proj." : ".k_proj.");
            const std::vector<float>& W = c->params.at(base + "weight").data;
            const std::vector<float>& Bv = c->params.at(base + "bias").data;
            std::copy(W.begin(), W.end(), w.begin() + (size_t)(2 * i + kv) * d * d);
            std::copy(Bv.begin(), Bv.end(), b.begin() + (size_t)(2 * i + kv) * d);
        }
    put(c, "kmm_kv_w", std::move(w));
    put(c, "kmm_kv_b", std::move(b));
    c->kmm_fused = build_kmmf_blobs(c);
    c->fused_ok = false;
    c->host_finalized = true;
    return KM_OK;
}

int finalize_host(Context* c) {
    for (const auto& k : c->param_order)
        if (!c->params[k].loaded) return fail(KM_ERR_NOT_FINALIZED, "parameter '%s' was never loaded", k.c_str());
    if (c->kind == 1) return finalize_host_legacy(c);
    if (c->kind == 2) return finalize_host_koemorph(c);
    const int d = c->d, H = c->H, hd = c->hd, KT = c->KT, ED = c->ED, DH = c->DH, NB = c->NB;
    const int NQ = kNumMouth;

    const dvec inw = to_d(P(c, "mel_attention.in_proj_weight")), inb = to_d(P(c, "mel_attention.in_proj_bias"));
    dvec Wq(inw.begin(), inw.begin() + (size_t)d * d), Wk(inw.begin() + (size_t)d * d, inw.begin() + (size_t)2 * d * d),
        Wv(inw.begin() + (size_t)2 * d * d, inw.end());
    dvec bq(inb.begin(), inb.begin() + d), bv(inb.begin() + 2 * d, inb.end());

    // Q = mouth_queries Wq^T + bq ; Qk[h][q][k] = 1/sqrt(hd) * sum_e Q[q][h*hd+e] Wk[h*hd+e][k]
    dvec Q = mm(to_d(P(c, "mouth_queries")), transpose(Wq, d, d), NQ, d, d);
    for (int q = 0; q < NQ; ++q)
        for (int i = 0; i < d; ++i) Q[(size_t)q * d + i] += bq[i];
    const double scale = 1.0 / std::sqrt((double)hd);
    dvec Qk((size_t)H * NQ * d, 0.0);
    for (int h = 0; h < H; ++h)
        for (int q = 0; q < NQ; ++q)
            for (int e = 0; e < hd; ++e) {
                const double qv = Q[(size_t)q * d + h * hd + e] * scale;
                const double* wk = &Wk[(size_t)(h * hd + e) * d];
                double* o = &Qk[((size_t)h * NQ + q) * d];
                for (int k = 0; k < d; ++k) o[k] += qv * wk[k];
            }

    // mouth chain: out_proj -> mel_output_proj -> decoder[0], value bias pushed through
    dvec Wf, bf;
    fold_chain(bv, to_d(P(c, "mel_attention.out_proj.weight")), to_d(P(c, "mel_attention.out_proj.bias")),
               to_d(P(c, "mel_output_proj.weight")), to_d(P(c, "mel_output_proj.bias")),
               to_d(P(c, "blendshape_decoder.0.weight")), to_d(P(c, "blendshape_decoder.0.bias")), d, DH, Wf, bf);

    // emotion chain: Wv_e -> out_proj -> emotion_output_proj -> decoder[0]
    const dvec einw = to_d(P(c, "emotion_attention.in_proj_weight")), einb = to_d(P(c, "emotion_attention.in_proj_bias"));
    dvec Wve(einw.begin() + (size_t)2 * d * d, einw.end()), bve(einb.begin() + 2 * d, einb.end());
    dvec WfE, bfE;
    fold_chain(bve, to_d(P(c, "emotion_attention.out_proj.weight")), to_d(P(c, "emotion_attention.out_proj.bias")),
               to_d(P(c, "emotion_output_proj.weight")), to_d(P(c, "emotion_output_proj.bias")),
               to_d(P(c, "blendshape_decoder.0.weight")), to_d(P(c, "blendshape_decoder.0.bias")), d, DH, WfE, bfE);
    dvec We2 = mm(transpose(Wve, d, d), WfE, d, d, DH);   // [n][hid] = sum_i Wve[i][n] WfE[i][hid]

    // stream weights: 0.5 * (softmax(mel_w / tau) + softmax(emo_w / tau))   (:252-253, :264-267)
    dvec wm = softmax_d(P(c, "mel_weights"), c->cfg.temperature), we = softmax_d(P(c, "emotion_weights"), c->cfg.temperature);
    std::vector<float> wsum(NB);
    for (int i = 0; i < NB; ++i) {
        // the reference evaluates w_m*bs*0.5 + w_e*bs*0.5 in fp32; the folded form differs by <= 1 ulp
        wsum[i] = (float)(0.5 * wm[i] + 0.5 * we[i]);
    }
    c->alpha = (float)(1.0 / (1.0 + std::exp(-(double)P(c, "smoothing_alpha")[0])));

    // ---- plain (unpacked) folded buffers: used by the generic kernels and by the tests ----
    put(c, "qk", to_f(Qk));                                        // (H, 28, d)
    put(c, "wf", to_f(Wf));                                        // (d, DH)
    put(c, "wf_t", to_f(transpose(Wf, d, DH)));                    // (DH, d): K-contiguous for gemm_nt_kernel
    if (DH % 32 == 0 && d % 16 == 0) {
        // decoder[0] fold as MFMA A-operand image for attn_out_kernel (km_generic.hip): wave w owns hidden units
        // 32 w + 16 rt + j; [w][rt][kb][lane][s] = Wf[k = 16 kb + 4 g + s][hid]  (one coalesced 1 KiB load per wave)
        const int NWv = DH / 32, KBv = d / 16;
        std::vector<float> wf_pg((size_t)NWv * 2 * KBv * 64 * 4);
        for (int w = 0; w < NWv; ++w)
            for (int rt = 0; rt < 2; ++rt)
                for (int kb = 0; kb < KBv; ++kb)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 4; ++e) {
                            const int g = l >> 4, j = l & 15, k = 16 * kb + 4 * g + e;
                            wf_pg[((((size_t)w * 2 + rt) * KBv + kb) * 64 + l) * 4 + e] = (float)Wf[(size_t)k * DH + 32 * w + 16 * rt + j];
                        }
        put(c, "wf_pg", std::move(wf_pg));
    }
    if (d % 16 == 0) {
        // folded query-key matrix of ALL heads as MFMA A-operand image for scores_softmax_kernel: row tile mt covers rows
        // 16 mt .. 16 mt + 15 of the (H * 28, d) matrix (zero rows past the end); [mt][kb][lane][s] = Qk[16 mt + j][16 kb + 4 g + s]
        const int rows = H * NQ, MT = (rows + 15) / 16, KBv = d / 16;
        std::vector<float> qk_pg((size_t)MT * KBv * 64 * 4, 0.0f);
        for (int mt = 0; mt < MT; ++mt)
            for (int kb = 0; kb < KBv; ++kb)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 4; ++e) {
                        const int g = l >> 4, j = l & 15, r = 16 * mt + j, k = 16 * kb + 4 * g + e;
                        if (r < rows) qk_pg[(((size_t)kb * MT + mt) * 64 + l) * 4 + e] = (float)Qk[(size_t)r * d + k];
                    }
        put(c, "qk_pg", std::move(qk_pg));
    }
    if (d == 512) {
        // value projection as MFMA B-operand image for attn_out_vr_kernel: V = Y Wv^T with wave w owning columns 64 w + 16 ct + j;
        // [kb][w][ct][lane][s] = Wv[64 w + 16 ct + j][16 kb + 4 g + s]
        const int NWv = d / 64, KBv = d / 16;
        std::vector<float> wv_bg((size_t)d * d);
        for (int kb = 0; kb < KBv; ++kb)
            for (int w2 = 0; w2 < NWv; ++w2)
                for (int ct = 0; ct < 4; ++ct)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 4; ++e) {
                            const int g = l >> 4, j = l & 15;
                            wv_bg[((((size_t)kb * NWv + w2) * 4 + ct) * 64 + l) * 4 + e] =
                                (float)Wv[(size_t)(64 * w2 + 16 * ct + j) * d + 16 * kb + 4 * g + e];
                        }
        put(c, "wv_bg", std::move(wv_bg));
    }
    {   // channel encoder weight with K padded to a multiple of 16 (zeros): rows stay 16-byte aligned and the long
        // and short-term columns form ONE contraction for encoder_tn_kernel (km_generic.hip)
        const int KT = c->KT, KP = (KT + 15) / 16 * 16;
        const std::vector<float>& w = P(c, "mel_channel_encoder.weight");          // (d, KT)
        std::vector<float> wp((size_t)d * KP, 0.0f);
        for (int n = 0; n < d; ++n) std::copy(w.begin() + (size_t)n * KT, w.begin() + (size_t)(n + 1) * KT, wp.begin() + (size_t)n * KP);
        if (d == 512 || d == 256 || d == 64) {
            // the same weight as MFMA B-operand image for encoder_ln_kernel<NW, CT> (km_encoder_dev.h): wave w, column tile
            // ct, lane (g, j) holds column n = 16 CT w + CT j + ct; [k block][w][ct][lane][s] = W[n][16 kb + 4 g + s]
            const int NWv = d == 64 ? 2 : 8, CT = d / (16 * NWv), KBv = KP / 16;
            std::vector<float> pg((size_t)d * KP);
            for (int kb = 0; kb < KBv; ++kb)
                for (int w2 = 0; w2 < NWv; ++w2)
                    for (int ct = 0; ct < CT; ++ct)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 4; ++e) {
                                const int g = l >> 4, j = l & 15, n = 16 * CT * w2 + CT * j + ct;
                                pg[((((size_t)kb * NWv + w2) * CT + ct) * 64 + l) * 4 + e] = wp[(size_t)n * KP + 16 * kb + 4 * g + e];
                            }
            put(c, "wce_pg", std::move(pg));
        }
        put(c, "wce_pad", std::move(wp));
    }
    put(c, "bf", to_f(bf));                                        // (DH)
    put(c, "we2", to_f(We2));                                      // (d, DH)
    put(c, "be2", to_f(bfE));                                      // (DH)
    put(c, "wee_t", to_f(transpose(to_d(P(c, "emotion_encoder.weight")), d, ED)));   // (ED, d)
    if (d == 256 && ED <= 256) {   // rows zero-padded to 256 k: the emotion step fused into the front end loads without guards
        std::vector<float> wpad((size_t)256 * d, 0.0f);
        const std::vector<float>& wt = c->packed.at("wee_t").host;
        std::copy(wt.begin(), wt.end(), wpad.begin());
        put(c, "wee_t256", std::move(wpad));
    }
    put(c, "bee", std::vector<float>(P(c, "emotion_encoder.bias")));
    put(c, "eln_g", std::vector<float>(P(c, "emotion_norm.weight")));
    put(c, "eln_b", std::vector<float>(P(c, "emotion_norm.bias")));
    put(c, "bce", std::vector<float>(P(c, "mel_channel_encoder.bias")));
    put(c, "ln_g", std::vector<float>(P(c, "mel_norm.weight")));
    put(c, "ln_b", std::vector<float>(P(c, "mel_norm.bias")));
    put(c, "w2", std::vector<float>(P(c, "blendshape_decoder.3.weight")));
    put(c, "b2", std::vector<float>(P(c, "blendshape_decoder.3.bias")));
    put(c, "wsum", std::move(wsum));
    put(c, "wce_raw", std::vector<float>(P(c, "mel_channel_encoder.weight")));          // (d, KT), generic path
    put(c, "wv_raw", to_f(Wv));                                                           // (d, d), generic path

    // ---- packed MFMA operand images for the fused gfx950 kernel -------------------------
    // v_mfma_f32_16x16x4_f32 operand maps: lane l = 16*g + j supplies A[i=j][k=g] and B[k=g][n=j].
    c->fused_ok = (d == 256 && H == 8 && c->T == 256 && c->NK == 80 && NB == 52 && c->cfg.mel_temporal_frames == 3);
    if (c->fused_ok) {
        const std::vector<float>& Wce = P(c, "mel_channel_encoder.weight");   // (d, KT)
        const int NW = 8, KP = 33;                                            // 66 k-steps of 4 >= KT=259
        std::vector<float> wce_p((size_t)NW * KP * 64 * 4);
        for (int w = 0; w < NW; ++w)
            for (int kp = 0; kp < KP; ++kp)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 4; ++e) {
                        const int g = l >> 4, j = l & 15, ds = e >> 1, t = e & 1;
                        const int k = 4 * (2 * kp + ds) + g, n = 16 * (2 * w + t) + j;
                        wce_p[(((size_t)w * KP + kp) * 64 + l) * 4 + e] = k < KT ? Wce[(size_t)n * KT + k] : 0.0f;
                    }
        put(c, "wce_p", std::move(wce_p));

        const int KB = d / 16;
        std::vector<float> qk_p((size_t)H * KB * 2 * 64 * 4), wv_p((size_t)H * KB * 2 * 64 * 4), wf_p((size_t)8 * KB * 64 * 4);
        for (int h = 0; h < H; ++h)
            for (int kb = 0; kb < KB; ++kb)
                for (int t = 0; t < 2; ++t)
                    for (int l = 0; l < 64; ++l)
                        for (int s = 0; s < 4; ++s) {
                            const int g = l >> 4, j = l & 15, k = 16 * kb + 4 * g + s;
                            const size_t o = ((((size_t)h * KB + kb) * 2 + t) * 64 + l) * 4 + s;
                            const int q = 16 * t + j;
                            qk_p[o] = q < NQ ? (float)Qk[((size_t)h * NQ + q) * d + k] : 0.0f;
                            wv_p[o] = (float)Wv[(size_t)(h * hd + 16 * t + j) * d + k];
                        }
        for (int w = 0; w < 8; ++w)
            for (int kb = 0; kb < KB; ++kb)
                for (int l = 0; l < 64; ++l)
                    for (int s = 0; s < 4; ++s) {
                        const int g = l >> 4, j = l & 15, k = 16 * kb + 4 * g + s;
                        wf_p[(((size_t)w * KB + kb) * 64 + l) * 4 + s] = (float)Wf[(size_t)k * DH + 16 * w + j];
                    }
        put(c, "qk_p", std::move(qk_p));
        put(c, "wv_p", std::move(wv_p));
        put(c, "wf_p", std::move(wf_p));
        // Experimental split-bf16 form of the same phase-2/3 operands (opt-in through KM_CORE_SPLIT): every
        // fp32 weight as NP bf16 pieces w = p0 + p1 (+ p2), B fragments of v_mfma_f32_16x16x32_bf16 (lane 16 g + j holds
        // k = 8 g .. 8 g + 7 of column j), [head][k block of 32][tile: Qk 0, Qk 1, Wv 0, Wv 1][piece][lane][8 bf16]
        for (int NP = 2; NP <= 3; ++NP) {
            std::vector<uint16_t> img((size_t)H * 8 * 4 * NP * 64 * 8);
            for (int h = 0; h < H; ++h)
                for (int kb = 0; kb < 8; ++kb)
                    for (int t = 0; t < 4; ++t)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                const int g = l >> 4, j = l & 15, k = 32 * kb + 8 * g + e, q = 16 * (t & 1) + j;
                                float w = t < 2 ? (q < NQ ? (float)Qk[((size_t)h * NQ + q) * d + k] : 0.0f)
                                                : (float)Wv[(size_t)(h * hd + 16 * (t & 1) + j) * d + k];
                                for (int pc = 0; pc < NP; ++pc) {
                                    const uint16_t b16 = bf16_rne(w);
                                    img[(((((size_t)h * 8 + kb) * 4 + t) * NP + pc) * 64 + l) * 8 + e] = b16;
                                    w -= bf16_to_float(b16);          // exact: the remainder has <= 16 significant bits
                                }
                            }
            std::vector<float> bits(img.size() / 2);
            std::memcpy(bits.data(), img.data(), img.size() * sizeof(uint16_t));
            put(c, NP == 2 ? "qkv_s2" : "qkv_s3", std::move(bits));
            // the channel encoder the same way: [wave][k block of 32 (9: K = 259 padded to 288)][column tile 0 / 1][piece][lane][8 bf16]
            std::vector<uint16_t> enc((size_t)8 * 9 * 2 * NP * 64 * 8);
            for (int w = 0; w < 8; ++w)
                for (int kb = 0; kb < 9; ++kb)
                    for (int t = 0; t < 2; ++t)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                const int g = l >> 4, j = l & 15, k = 32 * kb + 8 * g + e, n = 32 * w + 16 * t + j;
                                float wv = k < KT ? Wce[(size_t)n * KT + k] : 0.0f;
                                for (int pc = 0; pc < NP; ++pc) {
                                    const uint16_t b16 = bf16_rne(wv);
                                    enc[(((((size_t)w * 9 + kb) * 2 + t) * NP + pc) * 64 + l) * 8 + e] = b16;
                                    wv -= bf16_to_float(b16);
                                }
                            }
            std::vector<float> ebits(enc.size() / 2);
            std::memcpy(ebits.data(), enc.data(), enc.size() * sizeof(uint16_t));
            put(c, NP == 2 ? "wce_s2" : "wce_s3", std::move(ebits));
        }
    }
    c->host_finalized = true;
    return KM_OK;
}

// ---------------------------------------------------------------------------------------
// mel plans
// ---------------------------------------------------------------------------------------
static double hz_to_mel(double f, bool htk) {
    if (htk) return 2595.0 * std::log10(1.0 + f / 700.0);
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}

static double mel_to_hz(double m, bool htk) {
    if (htk) return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0);
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

MelPlan* build_mel_plan(const km_mel_config& cfg) {
    MelPlan* p = new MelPlan();
    p->cfg = cfg;
    const int N = cfg.n_fft, nf = N / 2 + 1, nm = cfg.n_mels;
    p->n_freq = nf;
    const double PI = 3.14159265358979323846;
    // periodic Hann (scipy get_window('hann', N, fftbins=True) == torch.hann_window(N, periodic=True))
    std::vector<double> w(N);
    double sw2 = 0;
    for (int n = 0; n < N; ++n) { w[n] = 0.5 - 0.5 * std::cos(2.0 * PI * n / N); sw2 += w[n] * w[n]; }
    const double wn = cfg.window_norm ? 1.0 / std::sqrt(sw2) : 1.0;
    p->window.resize(N);
    for (int n = 0; n < N; ++n) p->window[n] = (float)(w[n] * wn);
    p->twiddle.resize((size_t)2 * N);
    for (int q = 0; q < N; ++q) {
        p->twiddle[2 * q] = (float)std::cos(2.0 * PI * q / N);
        p->twiddle[2 * q + 1] = (float)(-std::sin(2.0 * PI * q / N));
    }
    // triangular filters: librosa.filters.mel (slaney/htk, norm) == torchaudio melscale_fbanks
    const bool htk = cfg.mel_scale == KM_MEL_HTK;
    std::vector<double> mel_f(nm + 2);
    const double m0 = hz_to_mel(cfg.f_min, htk), m1 = hz_to_mel(cfg.f_max, htk);
    for (int i = 0; i < nm + 2; ++i) mel_f[i] = mel_to_hz(m0 + (m1 - m0) * i / (nm + 1), htk);
    p->fb_start.assign(nm, 0);
    p->fb_count.assign(nm, 0);
    p->fb_offset.assign(nm, 0);
    for (int i = 0; i < nm; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
        const double enorm = cfg.slaney_norm ? 2.0 / (mel_f[i + 2] - mel_f[i]) : 1.0;
        int first = -1, last = -1;
        std::vector<float> row(nf, 0.0f);
        for (int k = 0; k < nf; ++k) {
            const double fk = (double)k * cfg.sample_rate / N;
            const double lower = (fk - mel_f[i]) / fd0, upper = (mel_f[i + 2] - fk) / fd1;
            const double v = std::fmax(0.0, std::fmin(lower, upper));
            const float v32 = (float)v;                      // librosa stores the triangle in float32 ...
            const float wv = (float)((double)v32 * enorm);   // ... and scales it in place
            row[k] = wv;
            if (wv != 0.0f) { if (first < 0) first = k; last = k; }
        }
        p->fb_offset[i] = (int32_t)p->fb_weight.size();
        if (first >= 0) {
            p->fb_start[i] = first;
            p->fb_count[i] = last - first + 1;
            for (int k = first; k <= last; ++k) p->fb_weight.push_back(row[k]);
        }
    }
    if (p->fb_weight.empty()) p->fb_weight.push_back(0.0f);
    // second image of the same filters for mel_power_rp_kernel: groups of four consecutive filters; a lane owns one
    // (frame, filter) pair and walks its filter four bins (one 16-byte LDS read of powers, one 16-byte load of taps) per
    // step from the filter's first bin rounded down to a multiple of 4; every filter of a group is zero-padded to the
    // group's step count.  The taps carry the 1/4 of |X|^2 = |2X|^2 / 4 (exact).  Groups go to the waves longest first,
    // each to the wave with the least steps so far (<= kMelRpGroups per wave).
    {
        const int ng = (nm + 3) / 4;
        std::vector<int> gsteps(ng, 0), order(ng);
        auto fsteps = [&](int i) { return (i < nm && p->fb_count[i] > 0) ? ((p->fb_start[i] & 3) + p->fb_count[i] + 3) / 4 : 0; };
        for (int g = 0; g < ng; ++g) {
            for (int s = 0; s < 4; ++s) gsteps[g] = std::max(gsteps[g], fsteps(4 * g + s));
            order[g] = g;
        }
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return gsteps[x] > gsteps[y]; });
        std::vector<std::vector<int>> mine(kMelRpWaves);
        std::vector<int> load(kMelRpWaves, 0);
        const bool fits = ng <= kMelRpWaves * kMelRpGroups;
        for (int oi = 0; fits && oi < ng; ++oi) {
            int best = -1;
            for (int w = 0; w < kMelRpWaves; ++w)
                if ((int)mine[w].size() < kMelRpGroups && (best < 0 || load[w] < load[best])) best = w;
            mine[best].push_back(order[oi]);
            load[best] += gsteps[order[oi]];
        }
        if (fits) {
            p->fbg_gid.assign((size_t)kMelRpWaves * kMelRpGroups, -1);
            p->fbg_desc.assign((size_t)kMelRpWaves * kMelRpGroups * 4, 0);
            for (int w = 0; w < kMelRpWaves; ++w)
                for (size_t gi = 0; gi < mine[w].size(); ++gi) {
                    const int g = mine[w][gi], steps = gsteps[g];
                    p->fbg_gid[(size_t)w * kMelRpGroups + gi] = g;
                    for (int s = 0; s < 4; ++s) {
                        const int i = 4 * g + s;
                        const bool live = i < nm && p->fb_count[i] > 0;
                        const int st4 = live ? (p->fb_start[i] & ~3) : 0, lead = live ? p->fb_start[i] - st4 : 0;
                        const size_t off = p->fbg_weight.size();
                        p->fbg_weight.resize(off + (size_t)4 * steps, 0.0f);
                        for (int k = 0; live && k < p->fb_count[i]; ++k) p->fbg_weight[off + lead + k] = 0.25f * p->fb_weight[p->fb_offset[i] + k];
                        p->fbg_desc[((size_t)w * kMelRpGroups + gi) * 4 + s] =
                            (int32_t)((uint32_t)(st4 / 4) | ((uint32_t)steps << 8) | ((uint32_t)(off / 4) << 16));
                        p->fbg_extent = std::max(p->fbg_extent, st4 + 4 * steps);
                    }
                }
            if (p->fbg_extent > kMelRpRow || p->fbg_weight.size() / 4 > 65535) { p->fbg_gid.clear(); p->fbg_desc.clear(); }
        }
        if (p->fbg_weight.empty()) p->fbg_weight.assign(4, 0.0f);
    }
    return p;
}

static bool same_cfg(const km_mel_config& a, const km_mel_config& b) { return std::memcmp(&a, &b, sizeof(a)) == 0; }

MelPlan* find_or_add_plan(Context* c, const km_mel_config& cfg) {
    for (MelPlan* p : c->mel_plans)
        if (same_cfg(p->cfg, cfg)) return p;
    MelPlan* p = build_mel_plan(cfg);
    c->mel_plans.push_back(p);
    return p;
}

}  // namespace km

// ---------------------------------------------------------------------------------------
// C-ABI: host-only entry points
// ---------------------------------------------------------------------------------------
using namespace km;

namespace km { const char* last_error(); }

namespace km {
namespace {
struct OptName { const char* name; int Options::*field; };
const OptName kOptNames[] = {
    {"core_split", &Options::core_split}, {"seq_per_window", &Options::seq_per_window},
    {"generic_staged", &Options::generic_staged}, {"mel_two_frame", &Options::mel_two_frame},
    {"emotion_separate", &Options::emotion_separate}, {"no_ln_fusion", &Options::no_ln_fusion},
    {"no_db_fusion", &Options::no_db_fusion}, {"no_score_fusion", &Options::no_score_fusion},
    {"no_out_fusion", &Options::no_out_fusion}, {"no_v_fusion", &Options::no_v_fusion}, {"no_core_merge", &Options::no_core_merge}, {"legacy_no_merge", &Options::legacy_no_merge}, {"train_chain", &Options::train_chain}, {"kmm_no_fuse", &Options::kmm_no_fuse}, {"legacy_no_attn_fusion", &Options::legacy_no_attn_fusion}, {"legacy_no_enc_fusion", &Options::legacy_no_enc_fusion}, {"legacy_no_tail_fusion", &Options::legacy_no_tail_fusion},
    {"train_no_split", &Options::train_no_split}, {"train_dwce_parts", &Options::train_dwce_parts},
    {"train_tail_groups", &Options::train_tail_groups}, {"train_bm32_below", &Options::train_bm32_below}, {"train_op_per_launch", &Options::train_op_per_launch}, {"train_split_min_k", &Options::train_split_min_k}, {"train_no_dma", &Options::train_no_dma}, {"train_attn_regs", &Options::train_attn_regs}, {"train_no_fe_pack", &Options::train_no_fe_pack}, {"train_colsum_gemm", &Options::train_colsum_gemm}, {"train_no_ln_fuse", &Options::train_no_ln_fuse}, {"train_ln_fuse_rows", &Options::train_ln_fuse_rows}, {"train_no_dy_split", &Options::train_no_dy_split}, {"train_alone_max", &Options::train_alone_max},
};
}  // namespace

void options_from_env(Options& o) {
    for (const OptName& n : kOptNames) {
        std::string env = "KM_";
        for (const char* p = n.name; *p; ++p) env += (char)std::toupper((unsigned char)*p);
        if (const char* v = std::getenv(env.c_str())) o.*(n.field) = *v ? std::atoi(v) : 1;
    }
    // historical names of the generic-chain switches
    if (std::getenv("KM_GENERIC_NO_LN_FUSION")) o.no_ln_fusion = 1;
    if (std::getenv("KM_GENERIC_NO_DB_FUSION")) o.no_db_fusion = 1;
    if (std::getenv("KM_GENERIC_NO_SCORE_FUSION")) o.no_score_fusion = 1;
    if (std::getenv("KM_GENERIC_NO_OUT_FUSION")) o.no_out_fusion = 1;
    if (std::getenv("KM_GENERIC_NO_V_FUSION")) o.no_v_fusion = 1;
}

int set_option(Context* c, const char* name, long long value) {
    for (const OptName& n : kOptNames)
        if (std::strcmp(n.name, name) == 0) {
            c->opt.*(n.field) = (int)value;
            return KM_OK;
        }
    return fail(KM_ERR_INVALID_ARG, "km_set_option: unknown option '%s'", name);
}
}  // namespace km

extern "C" {

int km_abi_version(void) { return KM_ABI_VERSION; }
const char* km_last_error(void) { return km::last_error(); }

static void expect(Context* c, const char* key, std::vector<int64_t> shape) {
    HostParam hp;
    hp.shape = std::move(shape);
    size_t n = 1;
    for (int64_t s : hp.shape) n *= (size_t)s;
    hp.data.assign(n, 0.0f);
    c->params[key] = std::move(hp);
    c->param_order.push_back(key);
}

static int check_mel_cfg(const km_mel_config& m) {
    if (m.n_fft != 512 && m.n_fft != 1024) return fail(KM_ERR_UNSUPPORTED, "n_fft must be 512 or 1024 (got %d)", m.n_fft);
    if (m.hop_length <= 0) return fail(KM_ERR_INVALID_ARG, "invalid hop_length %d", m.hop_length);   // stft.py:78-81
    if (m.n_mels <= 0 || m.n_mels > 128) return fail(KM_ERR_UNSUPPORTED, "n_mels must be in 1..128 (got %d)", m.n_mels);
    if (!(m.f_max > m.f_min) || m.f_min < 0 || m.f_max > m.sample_rate / 2.0f + 1e-3f)
        return fail(KM_ERR_INVALID_ARG, "need 0 <= f_min < f_max <= sr/2");
    return KM_OK;
}


int km_set_option(km_handle h, const char* name, int64_t value) {
    if (!h || !name) return km::fail(KM_ERR_INVALID_ARG, "km_set_option: NULL argument");
    return km::set_option(h, name, (long long)value);
}

int km_create(const km_config* cfg, km_handle* out) {
    if (!cfg || !out) return fail(KM_ERR_INVALID_ARG, "km_create: NULL argument");
    if (cfg->abi_version != KM_ABI_VERSION) return fail(KM_ERR_INVALID_ARG, "km_config.abi_version %d != %d", cfg->abi_version, KM_ABI_VERSION);
    if (cfg->d_model <= 0 || cfg->num_heads <= 0 || cfg->d_model % cfg->num_heads != 0)
        return fail(KM_ERR_INVALID_ARG, "embed_dim %d must be divisible by num_heads %d", cfg->d_model, cfg->num_heads);
    if (cfg->d_model % 2 != 0 || cfg->mel_sequence_length <= 0 || cfg->emotion_dim <= 0)
        return fail(KM_ERR_INVALID_ARG, "bad d_model / mel_sequence_length / emotion_dim");
    if (cfg->num_blendshapes != 52 || cfg->num_mel_channels <= 0 || cfg->mel_temporal_frames != 3)
        return fail(KM_ERR_UNSUPPORTED, "num_blendshapes must be 52 and mel_temporal_frames 3 (ARKit grouping, dual_stream_attention.py:14-45)");
    if (cfg->num_mel_channels != cfg->mel.n_mels) return fail(KM_ERR_INVALID_ARG, "num_mel_channels != mel.n_mels");
    if (int rc = check_mel_cfg(cfg->mel)) return rc;
    km_context* c = new km_context();
    options_from_env(c->opt);
    c->cfg = *cfg;
    c->d = cfg->d_model; c->H = cfg->num_heads; c->hd = c->d / c->H; c->T = cfg->mel_sequence_length;
    c->KT = c->T + cfg->mel_temporal_frames; c->ED = cfg->emotion_dim; c->DH = c->d / 2;
    c->NB = cfg->num_blendshapes; c->NK = cfg->num_mel_channels;
    const int64_t d = c->d;
    // state-dict layout of DualStreamCrossAttention (dual_stream_attention.py:102-159)
    expect(c, "mouth_queries", {kNumMouth, d});
    expect(c, "expression_queries", {kNumExpr, d});
    expect(c, "mel_weights", {c->NB});
    expect(c, "emotion_weights", {c->NB});
    expect(c, "mel_channel_encoder.weight", {d, c->KT});
    expect(c, "mel_channel_encoder.bias", {d});
    expect(c, "mel_attention.in_proj_weight", {3 * d, d});
    expect(c, "mel_attention.in_proj_bias", {3 * d});
    expect(c, "mel_attention.out_proj.weight", {d, d});
    expect(c, "mel_attention.out_proj.bias", {d});
    expect(c, "emotion_encoder.weight", {d, c->ED});
    expect(c, "emotion_encoder.bias", {d});
    expect(c, "emotion_attention.in_proj_weight", {3 * d, d});
    expect(c, "emotion_attention.in_proj_bias", {3 * d});
    expect(c, "emotion_attention.out_proj.weight", {d, d});
    expect(c, "emotion_attention.out_proj.bias", {d});
    expect(c, "mel_output_proj.weight", {d, d});
    expect(c, "mel_output_proj.bias", {d});
    expect(c, "emotion_output_proj.weight", {d, d});
    expect(c, "emotion_output_proj.bias", {d});
    expect(c, "blendshape_decoder.0.weight", {d / 2, d});
    expect(c, "blendshape_decoder.0.bias", {d / 2});
    expect(c, "blendshape_decoder.3.weight", {1, d / 2});
    expect(c, "blendshape_decoder.3.bias", {1});
    expect(c, "mel_norm.weight", {d});
    expect(c, "mel_norm.bias", {d});
    expect(c, "emotion_norm.weight", {d});
    expect(c, "emotion_norm.bias", {d});
    // SimplifiedDualStreamModel.smoothing_alpha (simplified_dual_stream_model.py:163); optional, default 0.8
    expect(c, "smoothing_alpha", {});
    c->params["smoothing_alpha"].data[0] = 0.8f;
    c->params["smoothing_alpha"].loaded = true;
    c->mel_plans.push_back(build_mel_plan(cfg->mel));
    *out = c;
    return KM_OK;
}

int km_koemorph_create(const km_koemorph_config* cfg, km_handle* out) {
    if (!cfg || !out) return fail(KM_ERR_INVALID_ARG, "km_koemorph_create: NULL argument");
    if (cfg->abi_version != KM_ABI_VERSION) return fail(KM_ERR_INVALID_ARG, "km_koemorph_config.abi_version %d != %d", cfg->abi_version, KM_ABI_VERSION);
    if (cfg->d_model <= 0 || cfg->num_heads <= 0 || cfg->d_model % cfg->num_heads != 0)     // attention.py:68-71
        return fail(KM_ERR_INVALID_ARG, "d_model (%d) must be divisible by num_heads (%d)", cfg->d_model, cfg->num_heads);
    if (cfg->d_model % 8 != 0) return fail(KM_ERR_INVALID_ARG, "d_model (%d) must be divisible by the encoder's 8 heads", cfg->d_model);
    if (cfg->mel_dim <= 0 || cfg->emotion_dim <= 0 || cfg->decoder_hidden_dim <= 0 || cfg->num_encoder_layers < 0 ||
        cfg->num_attention_layers < 0 || cfg->decoder_layers < 0 || cfg->num_blendshapes <= 0 || cfg->num_blendshapes > 64)
        return fail(KM_ERR_INVALID_ARG, "km_koemorph_create: bad dimension");
    if (cfg->decoder_activation < 0 || cfg->decoder_activation > 3)
        return fail(KM_ERR_INVALID_ARG, "decoder_activation: 0 relu, 1 gelu, 2 swish, 3 leaky_relu (got %d)", cfg->decoder_activation);
    if (cfg->output_activation < 0 || cfg->output_activation > 2)
        return fail(KM_ERR_INVALID_ARG, "output_activation: 0 sigmoid, 1 tanh, 2 none (got %d)", cfg->output_activation);
    if (cfg->smoothing_method < 0 || cfg->smoothing_method > 2)
        return fail(KM_ERR_INVALID_ARG, "smoothing_method: 0 exponential, 1 gaussian, 2 median (got %d)", cfg->smoothing_method);
    if (cfg->use_temporal_smoothing && cfg->smoothing_method != 0 && (cfg->smoothing_window < 1 || cfg->smoothing_window > 16))
        return fail(KM_ERR_INVALID_ARG, "smoothing_window has to be 1..16, got %d", cfg->smoothing_window);
    if (cfg->use_constraints && cfg->num_blendshapes < 27)
        return fail(KM_ERR_INVALID_ARG, "the default exclusion pairs (25, 26), (20, 21) need >= 27 blendshapes");
    km_context* c = new km_context();
    options_from_env(c->opt);
    c->kind = 2;
    c->kmm = *cfg;
    c->cfg.abi_version = KM_ABI_VERSION; c->cfg.d_model = cfg->d_model; c->cfg.num_heads = cfg->num_heads;
    c->cfg.num_blendshapes = cfg->num_blendshapes; c->cfg.temperature = 1.0f;
    c->d = cfg->d_model; c->H = cfg->num_heads; c->hd = c->d / c->H; c->NB = cfg->num_blendshapes; c->NK = cfg->mel_dim;
    c->ED = cfg->emotion_dim;
    const int64_t d = c->d, hid = cfg->decoder_hidden_dim, nb = c->NB;
    auto key = [](const std::string& s) { return s; };
    // state-dict layout of KoeMorphModel (gaussian_face.py:111-173)
    const char* streams[2] = {"mel", "emotion"};
    for (int si = 0; si < 2; ++si) {
        const std::string p = std::string("audio_encoder.") + streams[si] + "_encoder.";
        expect(c, key(p + "0.weight").c_str(), {d, si ? (int64_t)cfg->emotion_dim : (int64_t)cfg->mel_dim});
        expect(c, key(p + "0.bias").c_str(), {d});
        expect(c, key(p + "3.weight").c_str(), {d});
        expect(c, key(p + "3.bias").c_str(), {d});
    }
    for (int si = 0; si < 2; ++si)
        for (int i = 0; i < cfg->num_encoder_layers; ++i) {
            const std::string p = std::string("audio_encoder.") + streams[si] + "_transformer.layers." + std::to_string(i) + ".";
            expect(c, (p + "self_attn.in_proj_weight").c_str(), {3 * d, d});
            expect(c, (p + "self_attn.in_proj_bias").c_str(), {3 * d});
            expect(c, (p + "self_attn.out_proj.weight").c_str(), {d, d});
            expect(c, (p + "self_attn.out_proj.bias").c_str(), {d});
            expect(c, (p + "linear1.weight").c_str(), {4 * d, d});
            expect(c, (p + "linear1.bias").c_str(), {4 * d});
            expect(c, (p + "linear2.weight").c_str(), {d, 4 * d});
            expect(c, (p + "linear2.bias").c_str(), {d});
            for (const char* n : {"norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"}) expect(c, (p + n).c_str(), {d});
        }
    expect(c, "query_embeddings.query_embeddings", {nb, d});
    expect(c, "query_embeddings.conditioning_net.0.weight", {d / 2, nb});
    expect(c, "query_embeddings.conditioning_net.0.bias", {d / 2});
    expect(c, "query_embeddings.conditioning_net.3.weight", {d, d / 2});
    expect(c, "query_embeddings.conditioning_net.3.bias", {d});
    for (int i = 0; i < cfg->num_attention_layers; ++i) {
        const std::string p = "cross_attention_layers." + std::to_string(i) + ".";
        for (const char* n : {"q_proj", "k_proj", "v_proj", "out_proj"}) {
            expect(c, (p + n + ".weight").c_str(), {d, d});
            expect(c, (p + n + ".bias").c_str(), {d});
        }
    }
    for (int i = 0; i < cfg->num_attention_layers; ++i) {
        expect(c, ("attention_layer_norms." + std::to_string(i) + ".weight").c_str(), {d});
        expect(c, ("attention_layer_norms." + std::to_string(i) + ".bias").c_str(), {d});
    }
    expect(c, "decoder.input_proj.weight", {hid, d});
    expect(c, "decoder.input_proj.bias", {hid});
    for (int i = 0; i < cfg->decoder_layers; ++i) {
        expect(c, ("decoder.hidden_layers." + std::to_string(i) + ".weight").c_str(), {hid, hid});
        expect(c, ("decoder.hidden_layers." + std::to_string(i) + ".bias").c_str(), {hid});
    }
    for (int i = 0; i < cfg->decoder_layers; ++i) {
        expect(c, ("decoder.layer_norms." + std::to_string(i) + ".weight").c_str(), {hid});
        expect(c, ("decoder.layer_norms." + std::to_string(i) + ".bias").c_str(), {hid});
    }
    expect(c, "decoder.output_proj.weight", {nb, hid});
    expect(c, "decoder.output_proj.bias", {nb});
    if (cfg->use_temporal_smoothing && cfg->smoothing_method == 0) expect(c, "temporal_smoother.alpha", {});
    if (cfg->use_temporal_smoothing && cfg->smoothing_method == 1) expect(c, "temporal_smoother.gaussian_weights", {(int64_t)cfg->smoothing_window});
    *out = c;
    return KM_OK;
}

int km_legacy_create(const km_legacy_config* cfg, km_handle* out) {
    if (!cfg || !out) return fail(KM_ERR_INVALID_ARG, "km_legacy_create: NULL argument");
    if (cfg->abi_version != KM_ABI_VERSION) return fail(KM_ERR_INVALID_ARG, "km_legacy_config.abi_version %d != %d", cfg->abi_version, KM_ABI_VERSION);
    if (cfg->d_model <= 0 || cfg->num_heads <= 0 || cfg->d_model % cfg->num_heads != 0)
        return fail(KM_ERR_INVALID_ARG, "embed_dim %d must be divisible by num_heads %d", cfg->d_model, cfg->num_heads);
    if (cfg->decoder_hidden <= 0 || cfg->num_blendshapes <= 0 || cfg->num_blendshapes > 64)
        return fail(KM_ERR_INVALID_ARG, "bad decoder_hidden / num_blendshapes");
    if (int rc = check_mel_cfg(cfg->mel)) return rc;
    km_context* c = new km_context();
    options_from_env(c->opt);
    c->kind = 1;
    c->cfg.abi_version = KM_ABI_VERSION; c->cfg.d_model = cfg->d_model; c->cfg.num_heads = cfg->num_heads;
    c->cfg.num_mel_channels = cfg->mel.n_mels; c->cfg.num_blendshapes = cfg->num_blendshapes; c->cfg.mel = cfg->mel;
    c->cfg.temperature = 1.0f;
    c->d = cfg->d_model; c->H = cfg->num_heads; c->hd = c->d / c->H; c->NB = cfg->num_blendshapes; c->NK = cfg->mel.n_mels;
    c->legacy_hidden = cfg->decoder_hidden;
    const int64_t d = c->d, hid = cfg->decoder_hidden;
    // state-dict layout of SimplifiedKoeMorphModel (simplified_model.py:44-77)
    expect(c, "audio_encoder.0.weight", {d, c->NK});
    expect(c, "audio_encoder.0.bias", {d});
    expect(c, "audio_encoder.3.weight", {d, d});
    expect(c, "audio_encoder.3.bias", {d});
    expect(c, "attention.in_proj_weight", {3 * d, d});
    expect(c, "attention.in_proj_bias", {3 * d});
    expect(c, "attention.out_proj.weight", {d, d});
    expect(c, "attention.out_proj.bias", {d});
    expect(c, "decoder.0.weight", {hid, d});
    expect(c, "decoder.0.bias", {hid});
    expect(c, "decoder.3.weight", {hid, hid});
    expect(c, "decoder.3.bias", {hid});
    expect(c, "decoder.6.weight", {c->NB, hid});
    expect(c, "decoder.6.bias", {c->NB});
    expect(c, "blendshape_queries", {c->NB, d});
    c->mel_plans.push_back(build_mel_plan(cfg->mel));
    *out = c;
    return KM_OK;
}

static const char* strip_prefix(const char* key) {
    static const char pre[] = "dual_stream_attention.";
    return std::strncmp(key, pre, sizeof(pre) - 1) == 0 ? key + sizeof(pre) - 1 : key;
}

int km_load_param(km_handle h, const char* key, const float* data, const int64_t* shape, int32_t ndim) {
    if (!h || !key || !data) return fail(KM_ERR_INVALID_ARG, "km_load_param: NULL argument");
    Context* c = h;
    key = strip_prefix(key);
    auto it = c->params.find(key);
    if (it == c->params.end()) return fail(KM_ERR_INVALID_ARG, "unexpected key '%s' in state dict", key);
    HostParam& hp = it->second;
    size_t n = 1;
    bool ok = (size_t)ndim == hp.shape.size();
    for (int i = 0; ok && i < ndim; ++i) ok = shape[i] == hp.shape[i];
    if (!ok) {
        // torch also reports the two shapes on a size mismatch
        char want[64] = "", got[64] = "";
        for (int64_t s : hp.shape) snprintf(want + strlen(want), sizeof(want) - strlen(want), "%lld,", (long long)s);
        for (int i = 0; i < ndim; ++i) snprintf(got + strlen(got), sizeof(got) - strlen(got), "%lld,", (long long)shape[i]);
        return fail(KM_ERR_INVALID_ARG, "size mismatch for %s: expected (%s) got (%s)", key, want, got);
    }
    for (int64_t s : hp.shape) n *= (size_t)s;
    std::memcpy(hp.data.data(), data, n * sizeof(float));
    hp.loaded = true;
    c->host_finalized = false;
    c->dev_finalized = false;
    return KM_OK;
}

int km_get_param(km_handle h, const char* key, float* out, int64_t n) {
    if (!h || !key || !out) return fail(KM_ERR_INVALID_ARG, "km_get_param: NULL argument");
    auto it = h->params.find(strip_prefix(key));
    if (it == h->params.end()) return fail(KM_ERR_INVALID_ARG, "unknown key '%s'", key);
    if ((size_t)n != it->second.data.size()) return fail(KM_ERR_INVALID_ARG, "km_get_param: size mismatch for %s", key);
    std::memcpy(out, it->second.data.data(), (size_t)n * sizeof(float));
    return KM_OK;
}

int km_param_count(km_handle h, int32_t* expected, int32_t* loaded) {
    if (!h) return fail(KM_ERR_INVALID_ARG, "NULL handle");
    int e = 0, l = 0;
    for (auto& kv : h->params) { ++e; l += kv.second.loaded ? 1 : 0; }
    if (expected) *expected = e;
    if (loaded) *loaded = l;
    return KM_OK;
}

int km_finalize_host(km_handle h) {
    if (!h) return fail(KM_ERR_INVALID_ARG, "NULL handle");
    return finalize_host(h);
}

int km_debug_buffer(km_handle h, const char* name, float* out, int64_t* n) {
    if (!h || !name || !n) return fail(KM_ERR_INVALID_ARG, "km_debug_buffer: NULL argument");
    if (!h->host_finalized) return fail(KM_ERR_NOT_FINALIZED, "km_debug_buffer before km_finalize_host");
    auto it = h->packed.find(name);
    if (it == h->packed.end()) return fail(KM_ERR_INVALID_ARG, "no buffer named '%s'", name);
    if (!out) { *n = (int64_t)it->second.host.size(); return KM_OK; }
    if (*n != (int64_t)it->second.host.size()) return fail(KM_ERR_INVALID_ARG, "km_debug_buffer: size mismatch");
    std::memcpy(out, it->second.host.data(), it->second.host.size() * sizeof(float));
    return KM_OK;
}

}  // extern "C"
