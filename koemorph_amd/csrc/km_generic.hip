// Shape-generic path of the attention core for gfx950: any d_model / mel_sequence_length / head count
// (BASELINE config 4: d_model 512, window 512, 8 or 16 heads; the small test configuration d=64).
//
// The production shape (256/256/8) runs the single fused kernel of km_core.hip, whose Y image must fit in LDS.
// At d=512 it does not (80 x 512 x 4 B = 160 KB), so this path runs the same FOLDED network (km_host.cpp) as a
// short chain of strided-batched exact-fp32 MFMA GEMMs with the intermediates in an L2-resident workspace:
//
//   Y0 = X^T Wce^T + b      (per window: A is the caller's (T,80) mel read transposed through strides, K = T, then
//                            a K=3 accumulate for the short-term rows -- no (B,80,T+3) copy is ever built)
//   Y  = LayerNorm(Y0)                                               ln_rows_kernel
//   S  = Qk_h Y^T           (batch = windows x heads)                 gemm
//   P  = softmax(S)                                                   softmax_rows_kernel
//   V  = Y Wv^T                                                       gemm
//   O  = P V_h              (batch = windows x heads, strided into the concatenated head layout)
//   Hd = relu(O Wf + bf)                                              gemm + epilogue
//   z  = Hd w2 + b2 -> sigmoid -> stream weights -> clamp             decoder_tail_kernel
//
// gemm_kernel: C = alpha * A B + beta * C (+ bias, ReLU) with arbitrary row/column strides for A and B (so
// transposes are free), two batch dimensions with independent strides, 64x64x16 tiles, 256 threads (2x2 waves,
// 2x2 v_mfma_f32_16x16x4_f32 tiles each), operands staged k-major in LDS with a row stride of 80 floats
// (= 16 banks mod 32, so the two k rows of a ds_read_b32 half-wave are conflict free).
#include <hip/hip_runtime.h>

#include "km_context.h"
#include "km_device.h"
#include "km_gemm.h"

namespace km {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define KM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(KM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace gg {
constexpr int BM = 64, BN = 64, BK = 32, LDT = 80;   // LDS tile row stride (floats)
constexpr int EPT = BM * BK / 256;                   // staged elements per thread per operand
}

// Global -> register staging of one (64 x BK) A tile and one (BK x 64) B tile; the unit-stride dimension of each
// operand is the fast thread index (coalescing).  Out-of-range elements read as zero.
__device__ __forceinline__ void gemm_stage(const GemmArgs& g, const float* A, const float* Bp, int m0, int n0, int k0, int tid,
                                           bool a_kfast, bool b_kfast, float (&ra)[gg::EPT], float (&rb)[gg::EPT]) {
    using namespace gg;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int idx = tid + 256 * e;
        int m, k;
        if (a_kfast) { k = idx & (BK - 1); m = idx / BK; } else { m = idx & 63; k = idx >> 6; }
        const int gm = m0 + m, gk = k0 + k;
        ra[e] = (gm < g.M && gk < g.K) ? A[gm * g.a_rs + gk * g.a_cs] : 0.f;
        int n, kb;
        if (b_kfast) { kb = idx & (BK - 1); n = idx / BK; } else { n = idx & 63; kb = idx >> 6; }
        const int gn = n0 + n, gkb = k0 + kb;
        rb[e] = (gn < g.N && gkb < g.K) ? Bp[gkb * g.b_rs + gn * g.b_cs] : 0.f;
    }
}

__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    using namespace gg;
    __shared__ float As[BK * LDT];
    __shared__ float Bs[BK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lg = lane >> 4, lj = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;                 // 2 x 2 waves, 32 x 32 each
    const int z1 = blockIdx.z / g.batch2, z2 = blockIdx.z - z1 * g.batch2;
    const float* A = g.A + z1 * g.a_bs1 + z2 * g.a_bs2;
    const float* Bp = g.B + z1 * g.b_bs1 + z2 * g.b_bs2;
    float* C = g.C + z1 * g.c_bs1 + z2 * g.c_bs2;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const bool a_kfast = g.a_cs == 1;                        // A is K-contiguous (row-major M x K)
    const bool b_kfast = g.b_rs == 1;                        // B is K-contiguous (stored N x K, i.e. a transposed weight)
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { acc[i][0] = f32x4{0, 0, 0, 0}; acc[i][1] = f32x4{0, 0, 0, 0}; }

    // software pipeline (register staging): the next tile's global loads are in flight while the MFMAs of the
    // current tile run; the flattened iteration space is (contraction batch) x (k tiles)
    const int kbn = g.kb_count > 0 ? g.kb_count : 1;
    const int kt = (g.K + BK - 1) / BK, total = kbn * kt;
    float ra[EPT], rb[EPT];
    gemm_stage(g, A, Bp, m0, n0, 0, tid, a_kfast, b_kfast, ra, rb);
    for (int it = 0; it < total; ++it) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int idx = tid + 256 * e;
            int m, k;
            if (a_kfast) { k = idx & (BK - 1); m = idx / BK; } else { m = idx & 63; k = idx >> 6; }
            As[k * LDT + m] = ra[e];
            int n, kb;
            if (b_kfast) { kb = idx & (BK - 1); n = idx / BK; } else { n = idx & 63; kb = idx >> 6; }
            Bs[kb * LDT + n] = rb[e];
        }
        __syncthreads();
        if (it + 1 < total) {
            const int nb = (it + 1) / kt, nk = (it + 1) - nb * kt;
            gemm_stage(g, A + nb * g.a_kbs, Bp + nb * g.b_kbs, m0, n0, nk * BK, tid, a_kfast, b_kfast, ra, rb);
        }
#pragma unroll
        for (int s = 0; s < BK / 4; ++s) {
            const float* ar = As + (4 * s + lg) * LDT + 32 * wm + lj;
            const float* br = Bs + (4 * s + lg) * LDT + 32 * wn + lj;
            const float a0 = ar[0], a1 = ar[16], b0 = br[0], b1 = br[16];
            acc[0][0] = KM_MFMA(a0, b0, acc[0][0]);
            acc[0][1] = KM_MFMA(a0, b1, acc[0][1]);
            acc[1][0] = KM_MFMA(a1, b0, acc[1][0]);
            acc[1][1] = KM_MFMA(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 32 * wm + 16 * i + 4 * lg + r, n = n0 + 32 * wn + 16 * jn + lj;
                if (m < g.M && n < g.N) {
                    float v = g.alpha * acc[i][jn][r];
                    if (g.bias_mode == 1) v += g.bias[n];
                    else if (g.bias_mode == 2) v += g.bias[m];
                    float* cp = C + (int64_t)m * g.c_rs + n;
                    if (g.beta != 0.f) v += g.beta * (*cp);
                    if (g.relu) v = fmaxf(v, 0.f);
                    *cp = v;
                }
            }
}

int launch_softmax_rows(float* x, int64_t rows, int w, void* stream);

int launch_gemm(const GemmArgs& g, int batch, void* stream) {
    const dim3 grid((unsigned)((g.N + gg::BN - 1) / gg::BN), (unsigned)((g.M + gg::BM - 1) / gg::BM), (unsigned)batch);
    hipLaunchKernelGGL(gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, g);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

// LayerNorm(eps 1e-5) over the last dimension, one wave per row, two-pass; optional saved statistics
__global__ __launch_bounds__(256) void ln_rows_kernel(float* __restrict__ x, int64_t rows, int d,
                                                      const float* __restrict__ gam, const float* __restrict__ bet,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out) {
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
    if (mean_out && lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// softmax over rows of width w <= 128 (80 keys), one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, int64_t rows, int w) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* p = x + row * w;
    const float a = lane < w ? p[lane] : -INFINITY, b = lane + 64 < w ? p[lane + 64] : -INFINITY;
    float m = fmaxf(a, b);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float ea = lane < w ? expf(a - m) : 0.f, eb = lane + 64 < w ? expf(b - m) : 0.f;
    float s = ea + eb;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.0f / s;
    if (lane < w) p[lane] = ea * inv;
    if (lane + 64 < w) p[lane + 64] = eb * inv;
}

// softmax over rows of any width, one wave per row (three passes over the row, which sits in L2)
__global__ __launch_bounds__(256) void softmax_rows_wide_kernel(float* __restrict__ x, int64_t rows, int w) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* p = x + row * w;
    float m = -INFINITY;
    for (int i = lane; i < w; i += 64) m = fmaxf(m, p[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int i = lane; i < w; i += 64) { const float e = expf(p[i] - m); p[i] = e; s += e; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.0f / s;
    for (int i = lane; i < w; i += 64) p[i] *= inv;
}

// out[b][j] = mean over the NQ query rows of sigmoid(x[b][q][j])   (simplified_model.py:72,147)
__global__ __launch_bounds__(64) void sigmoid_mean_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int NQ, int NB) {
    const int b = blockIdx.x, j = threadIdx.x;
    if (j >= NB) return;
    const float* p = x + (int64_t)b * NQ * NB + j;
    float s = 0.f;
    for (int q = 0; q < NQ; ++q) s += 1.0f / (1.0f + expf(-p[(int64_t)q * NB]));
    out[(int64_t)b * NB + j] = s / NQ;
}

int launch_softmax_rows(float* x, int64_t rows, int w, void* stream) {
    if (w <= 128)
        hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, rows, w);
    else
        hipLaunchKernelGGL(softmax_rows_wide_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, rows, w);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

// head-averaged attention weights: (B, H, 28, 80) -> (B, 28, 80)
__global__ void head_mean_kernel(const float* __restrict__ p, float* __restrict__ out, int64_t B, int H, int per) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * per) return;
    const int64_t b = i / per, r = i - b * per;
    float s = 0.f;
    for (int h = 0; h < H; ++h) s += p[(b * H + h) * per + r];
    out[i] = s / H;
}

__device__ __forceinline__ int gen_mouth_slot(int i) { return (i >= 14 && i <= 40) ? i - 14 : (i == 51 ? 27 : -1); }

// z = Hd w2 + b2 per mouth row, sigmoid, stream weights, clamp (+ the emotion logit for the expression rows)
__global__ __launch_bounds__(64) void decoder_tail_kernel(const float* __restrict__ hd, int DH, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, const float* __restrict__ zemo,
                                                          const float* __restrict__ wsum, float* __restrict__ out,
                                                          float* __restrict__ raw) {
    const int b = blockIdx.x, i = threadIdx.x;
    if (i >= 52) return;
    const int slot = gen_mouth_slot(i);
    float z;
    if (slot >= 0) {
        const float* h = hd + ((int64_t)b * 28 + slot) * DH;
        z = b2[0];
        for (int m = 0; m < DH; ++m) z = fmaf(h[m], w2[m], z);
    } else {
        z = zemo[b];
    }
    const float bs = 1.0f / (1.0f + expf(-z));
    if (raw) raw[(int64_t)b * 52 + i] = bs;
    out[(int64_t)b * 52 + i] = fminf(fmaxf(wsum[i] * bs, 0.f), 1.f);
}

__global__ void gather_clip_logits_kernel(const float* __restrict__ zclip, float* __restrict__ zwin, int64_t nw, int64_t w0, int wpc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nw) zwin[i] = zclip[(w0 + i) / wpc];
}

int launch_gather_clip_logits(Context* c, const float* zclip, float* zwin, int64_t nw, int64_t w0, int wins_per_clip, void* stream) {
    (void)c;
    hipLaunchKernelGGL(gather_clip_logits_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, (hipStream_t)stream, zclip,
                       zwin, nw, w0, wins_per_clip);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

static const float* dv(Context* c, const char* name) { return c->packed.at(name).dev; }

// workspace floats per window for the generic forward
int64_t generic_ws_floats(Context* c) {
    const int64_t d = c->d, H = c->H, NKk = c->NK;
    return 2 * NKk * d /* Y, V */ + H * 28 * NKk /* S */ + 28 * d /* O */ + 28 * (d / 2) /* Hd */;
}

int launch_core_generic(Context* c, const float* mel, int64_t B, int64_t T_in, const float* mel_short, const float* zemo,
                        float* out, float* raw, float* attn, void* stream) {
    const int d = c->d, H = c->H, hd = c->hd, T = c->T, KT = c->KT, DH = c->DH, NKk = c->NK;
    if (NKk > 128) return fail(KM_ERR_UNSUPPORTED, "more than 128 mel channels");
    float* Y = c->ws_generic;
    float* V = Y + B * NKk * d;
    float* S = V + B * NKk * d;
    float* O = S + B * H * 28 * NKk;
    float* Hd = O + B * 28 * d;
    const float* Wce = dv(c, "wce_raw");
    GemmArgs g{};
    g.alpha = 1.f; g.batch2 = 1;
    // Y0 = X^T Wce^T + b : long rows (zero pad / truncate to T, dual_stream_attention.py:193-202) ...
    const int tv = (int)(T_in < T ? T_in : T);
    g.A = mel; g.a_rs = 1; g.a_cs = NKk; g.a_bs1 = T_in * NKk;
    g.B = Wce; g.b_rs = 1; g.b_cs = KT;
    g.C = Y; g.c_rs = d; g.c_bs1 = (int64_t)NKk * d;
    g.M = NKk; g.N = d; g.K = tv; g.bias = dv(c, "bce"); g.bias_mode = 1; g.beta = 0.f;
    if (int rc = launch_gemm(g, (int)B, stream)) return rc;
    // ... plus the 3 short-term rows (:205-208)
    g.A = mel_short; g.a_bs1 = 3 * NKk; g.B = Wce + T; g.K = 3; g.bias_mode = 0; g.beta = 1.f;
    if (int rc = launch_gemm(g, (int)B, stream)) return rc;
    hipLaunchKernelGGL(ln_rows_kernel, dim3((unsigned)((B * NKk + 3) / 4)), dim3(256), 0, (hipStream_t)stream, Y,
                       B * NKk, d, dv(c, "ln_g"), dv(c, "ln_b"), (float*)nullptr, (float*)nullptr);
    // S[b,h] = Qk_h Y_b^T
    g = GemmArgs{}; g.alpha = 1.f;
    g.A = dv(c, "qk"); g.a_rs = d; g.a_cs = 1; g.a_bs1 = 0; g.a_bs2 = (int64_t)28 * d;
    g.B = Y; g.b_rs = 1; g.b_cs = d; g.b_bs1 = (int64_t)NKk * d; g.b_bs2 = 0;
    g.C = S; g.c_rs = NKk; g.c_bs1 = (int64_t)H * 28 * NKk; g.c_bs2 = (int64_t)28 * NKk;
    g.M = 28; g.N = NKk; g.K = d; g.batch2 = H;
    if (int rc = launch_gemm(g, (int)(B * H), stream)) return rc;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((B * H * 28 + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S,
                       B * H * 28, NKk);
    if (attn) {
        const int64_t n = B * 28 * NKk;
        hipLaunchKernelGGL(head_mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, S, attn,
                           B, H, 28 * NKk);
    }
    // V = Y Wv^T   (the value bias is folded into bf)
    g = GemmArgs{}; g.alpha = 1.f; g.batch2 = 1;
    g.A = Y; g.a_rs = d; g.a_cs = 1;
    g.B = dv(c, "wv_raw"); g.b_rs = 1; g.b_cs = d;
    g.C = V; g.c_rs = d; g.M = (int)(B * NKk); g.N = d; g.K = d;
    if (int rc = launch_gemm(g, 1, stream)) return rc;
    // O[b][:, h*hd:(h+1)*hd] = P[b,h] V[b][:, h*hd:(h+1)*hd]
    g = GemmArgs{}; g.alpha = 1.f;
    g.A = S; g.a_rs = NKk; g.a_cs = 1; g.a_bs1 = (int64_t)H * 28 * NKk; g.a_bs2 = (int64_t)28 * NKk;
    g.B = V; g.b_rs = d; g.b_cs = 1; g.b_bs1 = (int64_t)NKk * d; g.b_bs2 = hd;
    g.C = O; g.c_rs = d; g.c_bs1 = (int64_t)28 * d; g.c_bs2 = hd;
    g.M = 28; g.N = hd; g.K = NKk; g.batch2 = H;
    if (int rc = launch_gemm(g, (int)(B * H), stream)) return rc;
    // Hd = relu(O Wf + bf)
    g = GemmArgs{}; g.alpha = 1.f; g.batch2 = 1;
    g.A = O; g.a_rs = d; g.a_cs = 1;
    g.B = dv(c, "wf"); g.b_rs = DH; g.b_cs = 1;
    g.C = Hd; g.c_rs = DH; g.M = (int)(B * 28); g.N = DH; g.K = d; g.bias = dv(c, "bf"); g.bias_mode = 1; g.relu = 1;
    if (int rc = launch_gemm(g, 1, stream)) return rc;
    hipLaunchKernelGGL(decoder_tail_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, Hd, DH, dv(c, "w2"),
                       dv(c, "b2"), zemo, dv(c, "wsum"), out, raw);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Legacy single-stream model (SimplifiedKoeMorphModel.forward, simplified_model.py:114-149), eval mode
// ---------------------------------------------------------------------------------------------------------
int64_t legacy_ws_floats(Context* c, int64_t F) {
    const int64_t d = c->d, H = c->H, NQ = c->NB, hid = c->legacy_hidden;
    return 4 * F * d /* E1, E, K, V */ + H * NQ * F /* S */ + 2 * NQ * d /* O, A1 */ + 2 * NQ * hid + NQ * NQ;
}

static GemmArgs lin(const float* A, int64_t a_rs, const float* W, int K, float* C, int64_t rows, int N, const float* bias, int relu) {
    GemmArgs g{};      // C (rows x N) = A (rows x K) W^T (+ bias) with W stored (N x K) like nn.Linear
    g.alpha = 1.f; g.batch2 = 1;
    g.A = A; g.a_rs = a_rs; g.a_cs = 1;
    g.B = W; g.b_rs = 1; g.b_cs = K;
    g.C = C; g.c_rs = N; g.M = (int)rows; g.N = N; g.K = K; g.bias = bias; g.bias_mode = bias ? 1 : 0; g.relu = relu;
    return g;
}

int launch_legacy(Context* c, const float* mel, int64_t B, int64_t Tm, float* out, void* stream) {
    const int d = c->d, H = c->H, hd = c->hd, NQ = c->NB, hid = c->legacy_hidden, NKk = c->NK;
    float* E1 = c->ws_generic;
    float* E = E1 + B * Tm * d;
    float* Kp = E + B * Tm * d;
    float* Vp = Kp + B * Tm * d;
    float* S = Vp + B * Tm * d;
    float* O = S + B * H * NQ * Tm;
    float* A1 = O + B * NQ * d;
    float* D1 = A1 + B * NQ * d;
    float* D2 = D1 + B * NQ * hid;
    float* D3 = D2 + B * NQ * hid;
    // audio_encoder: Linear(80,d) ReLU [Dropout] Linear(d,d) ReLU [Dropout]   (:44-51, :129)
    if (int rc = launch_gemm(lin(mel, NKk, dv(c, "l_w0"), NKk, E1, B * Tm, d, dv(c, "l_b0"), 1), 1, stream)) return rc;
    if (int rc = launch_gemm(lin(E1, d, dv(c, "l_w3"), d, E, B * Tm, d, dv(c, "l_b3"), 1), 1, stream)) return rc;
    // nn.MultiheadAttention(query = 52 learnable rows, key = value = encoded frames)   (:136-141)
    if (int rc = launch_gemm(lin(E, d, dv(c, "l_wk"), d, Kp, B * Tm, d, dv(c, "l_bk"), 0), 1, stream)) return rc;
    if (int rc = launch_gemm(lin(E, d, dv(c, "l_wv"), d, Vp, B * Tm, d, dv(c, "l_bv"), 0), 1, stream)) return rc;
    GemmArgs g{};
    g.alpha = 1.f;
    g.A = dv(c, "l_q"); g.a_rs = d; g.a_cs = 1; g.a_bs1 = 0; g.a_bs2 = hd;                       // Q_h (NQ x hd), pre-scaled
    g.B = Kp; g.b_rs = 1; g.b_cs = d; g.b_bs1 = Tm * d; g.b_bs2 = hd;                           // K_h^T
    g.C = S; g.c_rs = Tm; g.c_bs1 = (int64_t)H * NQ * Tm; g.c_bs2 = (int64_t)NQ * Tm;
    g.M = NQ; g.N = (int)Tm; g.K = hd; g.batch2 = H;
    if (int rc = launch_gemm(g, (int)(B * H), stream)) return rc;
    hipLaunchKernelGGL(softmax_rows_wide_kernel, dim3((unsigned)((B * H * NQ + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S,
                       B * H * NQ, (int)Tm);
    g = GemmArgs{};
    g.alpha = 1.f;
    g.A = S; g.a_rs = Tm; g.a_cs = 1; g.a_bs1 = (int64_t)H * NQ * Tm; g.a_bs2 = (int64_t)NQ * Tm;
    g.B = Vp; g.b_rs = d; g.b_cs = 1; g.b_bs1 = Tm * d; g.b_bs2 = hd;
    g.C = O; g.c_rs = d; g.c_bs1 = (int64_t)NQ * d; g.c_bs2 = hd;
    g.M = NQ; g.N = hd; g.K = (int)Tm; g.batch2 = H;
    if (int rc = launch_gemm(g, (int)(B * H), stream)) return rc;
    if (int rc = launch_gemm(lin(O, d, dv(c, "l_wo"), d, A1, B * NQ, d, dv(c, "l_bo"), 0), 1, stream)) return rc;
    // decoder: Linear(d,hid) ReLU Linear(hid,hid) ReLU Linear(hid,52) Sigmoid, then mean over the query rows (:63-72, :144-147)
    if (int rc = launch_gemm(lin(A1, d, dv(c, "l_d0w"), d, D1, B * NQ, hid, dv(c, "l_d0b"), 1), 1, stream)) return rc;
    if (int rc = launch_gemm(lin(D1, hid, dv(c, "l_d3w"), hid, D2, B * NQ, hid, dv(c, "l_d3b"), 1), 1, stream)) return rc;
    if (int rc = launch_gemm(lin(D2, hid, dv(c, "l_d6w"), hid, D3, B * NQ, NQ, dv(c, "l_d6b"), 0), 1, stream)) return rc;
    hipLaunchKernelGGL(sigmoid_mean_rows_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, D3, out, NQ, NQ);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

}  // namespace km
