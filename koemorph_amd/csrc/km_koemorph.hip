// Legacy multi-layer model: KoeMorphModel.forward (src/model/gaussian_face.py:175-268), eval mode, on the shape-generic
// fp32-MFMA GEMMs of km_generic.hip plus the row kernels below.  See include/koemorph.h (km_koemorph_*) for the layer
// list with the reference lines each step follows.  Off the production path: built for coverage and
// parity, one launch per layer step (~75 launches per forward at the default depth), no fusion beyond GEMM epilogues.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <string>

#include "km_context.h"
#include "km_gemm.h"

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return km::fail(KM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace km {

#include "km_kmm_tail.h"

static const float* dv(Context* c, const std::string& name) { return c->packed.at(name).dev; }

// km_kmmf.hip: the fused kernels for the reference's default width
bool koemorph_fused_ok(Context* c, int64_t B, int64_t T, const float* mel, const float* emo);
int launch_kmmf_encoder(Context* c, const float* mel, const float* emo, int64_t B, int64_t T, const unsigned char* kvalid, float* xm,
                        float* xe, void* stream);
int launch_kmmf_decode(Context* c, const float* xm, const float* xe, int64_t B, int64_t T, const unsigned char* kvalid, const float* prev,
                       float* attn, const KmmTail& tail, void* stream);

// LayerNorm over the last dimension, in place, one wave per row (two-pass, eps 1e-5)
__global__ __launch_bounds__(256) void kmm_ln_rows_kernel(float* __restrict__ x, int64_t rows, int d,
                                                          const float* __restrict__ gam, const float* __restrict__ bet) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* p = x + row * d;
    float s = 0.f;
    for (int i = lane; i < d; i += 64) s += p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / d;
    float v = 0.f;
    for (int i = lane; i < d; i += 64) { const float t = p[i] - mean; v += t * t; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const float rstd = 1.0f / sqrtf(v / d + 1e-5f);
    for (int i = lane; i < d; i += 64) p[i] = (p[i] - mean) * rstd * gam[i] + bet[i];
}

// softmax over the T keys of each (b, h, q) row with the causal / window masks of attention.py:208-246 applied on the
// fly: key j of query q is masked when j > q (causal) or outside [kp - w/2, kp + w/2] with kp = floor(q T / NQ).
// A fully masked row becomes NaN, as torch's softmax over -inf only; a NaN score makes its whole row NaN.
// kvalid (B, T), 1 = attend, or null: the key padding mask (attention.py:196-200; src_key_padding_mask of the encoder),
// looked up per batch element = row / rows_per_b.
template <int G>   // G lanes per row (a power of two <= 64): 64 / G rows per wave, so short key axes (T = 1 per tick) fill the wave
__global__ __launch_bounds__(256) void kmm_masked_softmax_kernel(float* __restrict__ x, int64_t rows, int T, int NQ, int causal, int window,
                                                                 const unsigned char* __restrict__ kvalid, int rows_per_b) {
    constexpr int RPW = 64 / G;
    const int lane = threadIdx.x & 63, sub = lane & (G - 1);
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / G;
    const bool live = row < rows;                       // dead rows still take part in the shuffles
    const int q = live ? (int)(row % NQ) : 0;
    int lo = 0, hi = T;
    if (window >= 0) {
        const int kp = (int)(((int64_t)q * T) / NQ);
        lo = kp - window / 2 > 0 ? kp - window / 2 : 0;
        hi = kp + window / 2 + 1 < T ? kp + window / 2 + 1 : T;
    }
    if (causal && q + 1 < hi) hi = q + 1;
    if (!live) hi = lo;
    float* p = x + (live ? row : 0) * T;
    const unsigned char* kv = kvalid ? kvalid + (live ? row / rows_per_b : 0) * T : nullptr;
    float m = -INFINITY;
    for (int i = lo + sub; i < hi; i += G) m = fmaxf(m, (!kv || kv[i]) ? p[i] : -INFINITY);
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int i = lo + sub; i < hi; i += G) s += expf(((!kv || kv[i]) ? p[i] : -INFINITY) - m);
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (!live) return;
    // masked keys go through the same arithmetic as torch's masked_fill(-inf) + softmax: 0 for a regular row, NaN when the
    // row has no key at all (exp(-inf - -inf)) or holds a NaN score (0 * NaN)
    const float inv = 1.0f / s;
    for (int i = sub; i < T; i += G) {
        const bool in = i >= lo && i < hi && (!kv || kv[i]);
        p[i] = expf((in ? p[i] : -INFINITY) - m) * inv;
    }
}

static int masked_softmax(float* S, int64_t rows, int T, int NQ, int causal, int window, const unsigned char* kvalid, int rows_per_b,
                          hipStream_t st) {
#define KMM_MS(G)                                                                                                              \
    hipLaunchKernelGGL(kmm_masked_softmax_kernel<G>, dim3((unsigned)((rows + 4 * (64 / G) - 1) / (4 * (64 / G)))), dim3(256), 0, st, S, \
                       rows, T, NQ, causal, window, kvalid, rows_per_b)
    if (T <= 1) KMM_MS(1);
    else if (T <= 4) KMM_MS(4);
    else if (T <= 16) KMM_MS(16);
    else KMM_MS(64);
#undef KMM_MS
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

// y = 0.5 (a + b)
__global__ void kmm_avg_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (a[i] + b[i]) / 2.0f;
}

// x[b][q][:] = emb[q][:] (+ cond[b][:])
__global__ void kmm_query_init_kernel(const float* __restrict__ emb, const float* __restrict__ cond, float* __restrict__ x,
                                      int64_t B, int NQ, int d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * NQ * d) return;
    const int col = (int)(i % d);
    const int q = (int)((i / d) % NQ);
    const int64_t b = i / ((int64_t)d * NQ);
    x[i] = emb[(int64_t)q * d + col] + (cond ? cond[b * d + col] : 0.f);
}

// decoder hidden layer tail: x = act(x) + res   (decoder.py:147-152, after the LayerNorm)
__global__ void kmm_act_residual_kernel(float* __restrict__ x, const float* __restrict__ res, int64_t n, int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    x[i] = gemm_act(v, act) + res[i];
}

// decoder output (diagonal of output_proj: row q only needs Wout[q]) and the output tail (km_kmm_tail.h).  One workgroup of
// 64 threads per batch element (NB <= 64).
__global__ __launch_bounds__(64) void kmm_tail_kernel(KmmTail a) {
    __shared__ float ys[64];
    const int64_t b = blockIdx.x;
    const int q = threadIdx.x, NB = a.NB;
    float z = 0.f;
    if (q < NB) {
        const float* hr = a.h + (b * NB + q) * a.hid;
        const float* wr = a.wout + (int64_t)q * a.hid;
        for (int k = 0; k < a.hid; ++k) z = fmaf(hr[k], wr[k], z);
        z += a.bout[q];
    }
    kmm_tail_dev(a, b, q, z, ys);
}

static GemmArgs lin(const float* A, int64_t a_rs, const float* W, int K, float* C, int64_t c_rs, int64_t rows, int N,
                    const float* bias, int act, float beta = 0.f) {
    GemmArgs g{};      // C (rows x N) = act(A (rows x K) W^T + bias + beta C) with W stored (N x K) like nn.Linear
    g.alpha = 1.f; g.beta = beta; g.batch2 = 1;
    g.A = A; g.a_rs = a_rs; g.a_cs = 1;
    g.B = W; g.b_rs = 1; g.b_cs = K;
    g.C = C; g.c_rs = c_rs; g.M = (int)rows; g.N = N; g.K = K; g.bias = bias; g.bias_mode = bias ? 1 : 0; g.relu = act;
    return g;
}

static int ln_rows(float* x, int64_t rows, int d, const float* g, const float* b, hipStream_t st) {
    hipLaunchKernelGGL(kmm_ln_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, rows, d, g, b);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

int64_t koemorph_ws_floats(Context* c, int64_t T) {
    const int64_t d = c->d, H = c->H, NB = c->NB, hid = c->kmm.decoder_hidden_dim, L = c->kmm.num_attention_layers;
    const int64_t HE = 8;
    const int64_t s_enc = HE * T * T, s_x = H * NB * T;
    return 2 * T * d /* xm, xe */ + 3 * T * d /* qkv */ + (s_enc > s_x ? s_enc : s_x) /* scores */ + T * d /* O */ + 4 * T * d /* ffn */ +
           2 * L * T * d /* K, V of every layer */ + 3 * NB * d /* x, Q, O2 */ + 2 * NB * hid + (d / 2 + d) /* conditioning */;
}

// one stream of DualStreamEncoder (dual_stream_attention.py:369-388): x = LN(ReLU(in W0^T + b0)), then the post-norm layers
static int encode_stream(Context* c, const char* stream, const float* in, int in_dim, int64_t B, int64_t T, const unsigned char* kvalid,
                         float* x, float* qkv, float* S, float* O, float* ffn, hipStream_t st) {
    const int d = c->d, HE = 8, hde = d / HE;
    const int64_t R = B * T;
    const std::string p = std::string("audio_encoder.") + stream + "_encoder.";
    if (int rc = launch_gemm(lin(in, in_dim, dv(c, p + "0.weight"), in_dim, x, d, R, d, dv(c, p + "0.bias"), 1), 1, st)) return rc;
    if (int rc = ln_rows(x, R, d, dv(c, p + "3.weight"), dv(c, p + "3.bias"), st)) return rc;
    const float scale = 1.0f / sqrtf((float)hde);
    for (int i = 0; i < c->kmm.num_encoder_layers; ++i) {
        const std::string l = std::string("audio_encoder.") + stream + "_transformer.layers." + std::to_string(i) + ".";
        if (int rc = launch_gemm(lin(x, d, dv(c, l + "self_attn.in_proj_weight"), d, qkv, 3 * d, R, 3 * d, dv(c, l + "self_attn.in_proj_bias"), 0), 1, st)) return rc;
        GemmArgs g{};                                       // S[b][h] (T x T) = Q_h K_h^T / sqrt(hd)
        g.alpha = scale;
        g.A = qkv; g.a_rs = 3 * d; g.a_cs = 1; g.a_bs1 = T * 3 * d; g.a_bs2 = hde;
        g.B = qkv + d; g.b_rs = 1; g.b_cs = 3 * d; g.b_bs1 = T * 3 * d; g.b_bs2 = hde;
        g.C = S; g.c_rs = T; g.c_bs1 = (int64_t)HE * T * T; g.c_bs2 = T * T;
        g.M = (int)T; g.N = (int)T; g.K = hde; g.batch2 = HE;
        if (int rc = launch_gemm(g, (int)(B * HE), st)) return rc;
        if (kvalid) { if (int rc = masked_softmax(S, B * HE * T, (int)T, (int)T, 0, -1, kvalid, (int)(HE * T), st)) return rc; }
        else if (int rc = launch_softmax_rows(S, B * HE * T, (int)T, st)) return rc;
        g = GemmArgs{};                                     // O[b][:, h] (T x hd) = P V_h
        g.alpha = 1.f;
        g.A = S; g.a_rs = T; g.a_cs = 1; g.a_bs1 = (int64_t)HE * T * T; g.a_bs2 = T * T;
        g.B = qkv + 2 * d; g.b_rs = 3 * d; g.b_cs = 1; g.b_bs1 = T * 3 * d; g.b_bs2 = hde;
        g.C = O; g.c_rs = d; g.c_bs1 = T * d; g.c_bs2 = hde;
        g.M = (int)T; g.N = hde; g.K = (int)T; g.batch2 = HE;
        if (int rc = launch_gemm(g, (int)(B * HE), st)) return rc;
        // x = LN1(x + out_proj(O));  x = LN2(x + W2 gelu(W1 x + b1) + b2)
        if (int rc = launch_gemm(lin(O, d, dv(c, l + "self_attn.out_proj.weight"), d, x, d, R, d, dv(c, l + "self_attn.out_proj.bias"), 0, 1.f), 1, st)) return rc;
        if (int rc = ln_rows(x, R, d, dv(c, l + "norm1.weight"), dv(c, l + "norm1.bias"), st)) return rc;
        if (int rc = launch_gemm(lin(x, d, dv(c, l + "linear1.weight"), d, ffn, 4 * d, R, 4 * d, dv(c, l + "linear1.bias"), 2), 1, st)) return rc;
        if (int rc = launch_gemm(lin(ffn, 4 * d, dv(c, l + "linear2.weight"), 4 * d, x, d, R, d, dv(c, l + "linear2.bias"), 0, 1.f), 1, st)) return rc;
        if (int rc = ln_rows(x, R, d, dv(c, l + "norm2.weight"), dv(c, l + "norm2.bias"), st)) return rc;
    }
    return KM_OK;
}

int launch_koemorph(Context* c, const float* mel, const float* emo, int64_t B, int64_t T, const unsigned char* kvalid, const float* prev,
                    float* state, int apply_constraints, float* out, float* raw, float* attn, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const km_koemorph_config& k = c->kmm;
    const int d = c->d, H = c->H, hd = c->hd, NB = c->NB, hid = k.decoder_hidden_dim, L = k.num_attention_layers;
    const int64_t R = B * T, RQ = B * NB;
    const int64_t s_enc = 8 * T * T, s_x = (int64_t)H * NB * T;
    float* xm = c->ws_generic;
    float* xe = xm + R * d;
    float* qkv = xe + R * d;
    float* S = qkv + R * 3 * d;
    float* O = S + B * (s_enc > s_x ? s_enc : s_x);
    float* ffn = O + R * d;
    float* KV = ffn + R * 4 * d;
    float* x = KV + R * 2 * L * d;
    float* Q = x + RQ * d;
    float* O2 = Q + RQ * d;
    float* dA = O2 + RQ * d;
    float* dB = dA + RQ * hid;
    float* cond = dB + RQ * hid;
    float* cond1 = cond + B * d;
    // ---- DualStreamEncoder on both streams, then the average (gaussian_face.py:203-209) ----
    const bool fused = koemorph_fused_ok(c, B, T, mel, emo);
    if (fused) {
        if (int rc = launch_kmmf_encoder(c, mel, emo, B, T, kvalid, xm, xe, stream)) return rc;
    } else {
        if (int rc = encode_stream(c, "mel", mel, k.mel_dim, B, T, kvalid, xm, qkv, S, O, ffn, st)) return rc;
        if (int rc = encode_stream(c, "emotion", emo, k.emotion_dim, B, T, kvalid, xe, qkv, S, O, ffn, st)) return rc;
    }
    const bool smooth = state != nullptr && k.use_temporal_smoothing;
    KmmTail ta{};
    ta.hid = hid; ta.NB = NB; ta.wout = dv(c, "decoder.output_proj.weight"); ta.bout = dv(c, "decoder.output_proj.bias");
    ta.prev = prev; ta.out_act = k.output_activation; ta.smooth = smooth ? k.smoothing_method : -1; ta.window = k.smoothing_window;
    ta.sm_param = !smooth ? nullptr : (k.smoothing_method == 0 ? dv(c, "temporal_smoother.alpha")
                                       : (k.smoothing_method == 1 ? dv(c, "temporal_smoother.gaussian_weights") : nullptr));
    ta.state = smooth ? state : nullptr; ta.constraints = (apply_constraints && k.use_constraints) ? 1 : 0; ta.out = out; ta.raw = raw;
    if (fused) return launch_kmmf_decode(c, xm, xe, B, T, kvalid, prev, attn, ta, stream);
    hipLaunchKernelGGL(kmm_avg_kernel, dim3((unsigned)((R * d + 255) / 256)), dim3(256), 0, st, xm, xe, xm, R * d);
    // ---- queries (attention.py:481-514) ----
    if (prev) {
        if (int rc = launch_gemm(lin(prev, NB, dv(c, "query_embeddings.conditioning_net.0.weight"), NB, cond1, d / 2, B, d / 2,
                                     dv(c, "query_embeddings.conditioning_net.0.bias"), 1), 1, st)) return rc;
        if (int rc = launch_gemm(lin(cond1, d / 2, dv(c, "query_embeddings.conditioning_net.3.weight"), d / 2, cond, d, B, d,
                                     dv(c, "query_embeddings.conditioning_net.3.bias"), 0), 1, st)) return rc;
    }
    hipLaunchKernelGGL(kmm_query_init_kernel, dim3((unsigned)((RQ * d + 255) / 256)), dim3(256), 0, st,
                       dv(c, "query_embeddings.query_embeddings"), prev ? cond : (const float*)nullptr, x, B, NB, d);
    HIP_TRY(hipGetLastError());
    // ---- keys and values of every cross-attention layer in one product: KV (R x 2 L d), layer i at columns 2 i d ----
    if (L > 0)
        if (int rc = launch_gemm(lin(xm, d, dv(c, "kmm_kv_w"), d, KV, 2 * L * d, R, 2 * L * d, dv(c, "kmm_kv_b"), 0), 1, st)) return rc;
    const float scale = 1.0f / sqrtf((float)hd);                       // (head_dim * temperature)^-0.5, temperature 1
    for (int i = 0; i < L; ++i) {
        const std::string p = "cross_attention_layers." + std::to_string(i) + ".";
        if (int rc = launch_gemm(lin(x, d, dv(c, p + "q_proj.weight"), d, Q, d, RQ, d, dv(c, p + "q_proj.bias"), 0), 1, st)) return rc;
        const float* Kl = KV + (int64_t)(2 * i) * d;
        const float* Vl = Kl + d;
        GemmArgs g{};                                       // S[b][h] (NB x T) = scale Q_h K_h^T
        g.alpha = scale;
        g.A = Q; g.a_rs = d; g.a_cs = 1; g.a_bs1 = (int64_t)NB * d; g.a_bs2 = hd;
        g.B = Kl; g.b_rs = 1; g.b_cs = 2 * L * d; g.b_bs1 = T * 2 * L * d; g.b_bs2 = hd;
        g.C = S; g.c_rs = T; g.c_bs1 = (int64_t)H * NB * T; g.c_bs2 = (int64_t)NB * T;
        g.M = NB; g.N = (int)T; g.K = hd; g.batch2 = H;
        if (int rc = launch_gemm(g, (int)(B * H), st)) return rc;
        if (int rc = masked_softmax(S, B * H * NB, (int)T, NB, k.causal, k.window_size, kvalid, H * NB, st)) return rc;
        if (attn)
            HIP_TRY(hipMemcpyAsync(attn + (int64_t)i * B * H * NB * T, S, (size_t)(B * H * NB * T) * sizeof(float), hipMemcpyDeviceToDevice, st));
        g = GemmArgs{};                                     // O2[b][:, h] (NB x hd) = P V_h
        g.alpha = 1.f;
        g.A = S; g.a_rs = T; g.a_cs = 1; g.a_bs1 = (int64_t)H * NB * T; g.a_bs2 = (int64_t)NB * T;
        g.B = Vl; g.b_rs = 2 * L * d; g.b_cs = 1; g.b_bs1 = T * 2 * L * d; g.b_bs2 = hd;
        g.C = O2; g.c_rs = d; g.c_bs1 = (int64_t)NB * d; g.c_bs2 = hd;
        g.M = NB; g.N = hd; g.K = (int)T; g.batch2 = H;
        if (int rc = launch_gemm(g, (int)(B * H), st)) return rc;
        // x = LN(out_proj(O2) + x)   (gaussian_face.py:230-231)
        if (int rc = launch_gemm(lin(O2, d, dv(c, p + "out_proj.weight"), d, x, d, RQ, d, dv(c, p + "out_proj.bias"), 0, 1.f), 1, st)) return rc;
        if (int rc = ln_rows(x, RQ, d, dv(c, "attention_layer_norms." + std::to_string(i) + ".weight"),
                             dv(c, "attention_layer_norms." + std::to_string(i) + ".bias"), st)) return rc;
    }
    // ---- BlendshapeDecoder (decoder.py:131-177) ----
    const int act = k.decoder_activation == 0 ? 1 : (k.decoder_activation == 1 ? 2 : (k.decoder_activation == 2 ? 3 : 4));   // gemm_act codes
    if (int rc = launch_gemm(lin(x, d, dv(c, "decoder.input_proj.weight"), d, dA, hid, RQ, hid, dv(c, "decoder.input_proj.bias"), act), 1, st)) return rc;
    float *cur = dA, *nxt = dB;
    for (int i = 0; i < k.decoder_layers; ++i) {
        const std::string n = std::to_string(i);
        if (int rc = launch_gemm(lin(cur, hid, dv(c, "decoder.hidden_layers." + n + ".weight"), hid, nxt, hid, RQ, hid,
                                     dv(c, "decoder.hidden_layers." + n + ".bias"), 0), 1, st)) return rc;
        if (int rc = ln_rows(nxt, RQ, hid, dv(c, "decoder.layer_norms." + n + ".weight"), dv(c, "decoder.layer_norms." + n + ".bias"), st)) return rc;
        hipLaunchKernelGGL(kmm_act_residual_kernel, dim3((unsigned)((RQ * hid + 255) / 256)), dim3(256), 0, st, nxt, cur, RQ * hid, act);
        HIP_TRY(hipGetLastError());
        float* t = cur; cur = nxt; nxt = t;
    }
    ta.h = cur;
    hipLaunchKernelGGL(kmm_tail_kernel, dim3((unsigned)B), dim3(64), 0, st, ta);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

}  // namespace km
