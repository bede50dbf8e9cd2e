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

#include <cstdint>
#include <cstdlib>

#include "km_context.h"
#include "km_device.h"
#include "km_attn_dev.h"
#include "km_legacy_attn_dev.h"
#include "km_encoder_dev.h"
#include "km_gemm.h"

namespace km {


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
                    v = gemm_act(v, g.relu);
                    *cp = v;
                }
            }
}

int launch_softmax_rows(float* x, int64_t rows, int w, void* stream);

// ---------------------------------------------------------------------------------------------------------
// gemm_nt_kernel: the fast path for C = alpha A B^T (+ bias[n], ReLU, + beta C) when BOTH operands are K-contiguous
// (A row-major M x K, B stored N x K like an nn.Linear weight), K a multiple of 16 and rows 16-byte aligned -- every
// large product of the generic inference chain.  128 x (32 NT) tile per 256 threads, 2 x 2 waves, each wave 64 x (16 NT):
// 4 x NT accumulators, BK = 16.
//   * LDS layout [k / 4][row][k % 4]: a lane's fragment for FOUR consecutive MFMAs is one ds_read_b128 (lane group g
//     contracts k = 4 g + s in MFMA s: any partition of the 16 k into 4 MFMAs is valid as long as A and B agree), and
//     the 16 lanes of a b128 beat always hold 16 different rows = all 64 banks: (4 + NT) reads per 16 NT MFMAs
//     instead of one ds_read_b32 per MFMA.
//   * global -> LDS: one float4 along k per ds_write_b128, no transposition; consecutive lanes take consecutive rows
//     (conflict-free stores).  Register-staged double buffer: the next tile's loads fly under the current MFMAs, one
//     barrier per k tile.
// ---------------------------------------------------------------------------------------------------------
namespace gnt {
constexpr int BK = 16;
}

// Tiles are fetched with buffer loads: an out-of-range row points past the descriptor's range and reads zeros without a
// memory access, so no select touches the loaded values before the commit -- with plain loads + `ok ? v : 0` the
// compiler waited for the tile (vmcnt(0)) and wrote it to LDS after the first five MFMAs of a k step instead of the last.
typedef unsigned int nt_u32x4 __attribute__((ext_vector_type(4)));
template <int E>
__device__ __forceinline__ void nt_stage(__amdgpu_buffer_rsrc_t rs, const unsigned (&off)[E], int k0, nt_u32x4 (&r)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[e] + (unsigned)k0 * 4u, 0, 0);
}
template <int E>
__device__ __forceinline__ void nt_commit(float* dst, int tid, const nt_u32x4 (&r)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) *reinterpret_cast<nt_u32x4*>(dst + (tid + 256 * e) * 4) = r[e];
}

// PIN: scheduling barriers hold the next tile's loads at the top of a k step and its LDS commit at the bottom.  Short
// contractions (K <= 256: 16 steps, the legacy model's products) gain 7 % from it; long ones (K = 512, C4's value
// projection on 128-column tiles) lose 2 % to the extra live registers, so the launcher picks per product.
template <int MT, int NT, bool PIN>   // 16-row / 16-column MFMA tiles per wave: MT 4 -> BM 128, 2 -> BM 64; NT 4 -> BN 128, 2 -> BN 64
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs g) {
    using namespace gnt;
    constexpr int BM = 32 * MT;
    constexpr int BN = 32 * NT;
    __shared__ __attribute__((aligned(16))) float As[2][4 * BM * 4];
    __shared__ __attribute__((aligned(16))) float Bs[2][4 * BN * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lg = lane >> 4, lj = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    const int z1 = blockIdx.z / g.batch2, z2 = blockIdx.z - z1 * g.batch2;
    const float* A = g.A + z1 * g.a_bs1 + z2 * g.a_bs2;
    const float* Bp = g.B + z1 * g.b_bs1 + z2 * g.b_bs2;
    float* C = g.C + z1 * g.c_bs1 + z2 * g.c_bs2;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // staging map: element e of this thread is row (tid + 256 e) & (rows - 1), k group (tid + 256 e) / rows
    constexpr int EA = BM * 4 / 256, EB = BN * 4 / 256;
    static_assert(EA >= 1 && EB >= 1, "tile too small for 256 threads");
    // byte offsets from the (batch's) operand base; gemm_nt_ok() keeps both operands under 2 GiB
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)((((int64_t)g.M - 1) * g.a_rs + g.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bp), 0, (unsigned)((((int64_t)g.N - 1) * g.b_cs + g.K) * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned aoff[EA], boff[EB];
#pragma unroll
    for (int e = 0; e < EA; ++e) {
        const int idx = tid + 256 * e, m = idx & (BM - 1), kg = idx / BM;
        aoff[e] = m0 + m < g.M ? (unsigned)(((int64_t)(m0 + m) * g.a_rs + 4 * kg) * 4) : OOB;
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) {
        const int idx = tid + 256 * e, n = idx & (BN - 1), kg = idx / BN;
        boff[e] = n0 + n < g.N ? (unsigned)(((int64_t)(n0 + n) * g.b_cs + 4 * kg) * 4) : OOB;
    }
    nt_u32x4 ra[EA], rb[EB];

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) acc[i][jn] = f32x4{0, 0, 0, 0};

    const int kt = g.K / BK;
    nt_stage<EA>(ars, aoff, 0, ra);
    nt_stage<EB>(brs, boff, 0, rb);
    nt_commit<EA>(As[0], tid, ra);
    nt_commit<EB>(Bs[0], tid, rb);
    __syncthreads();
    for (int it = 0; it < kt; ++it) {
        const int buf = it & 1;
        const int knext = (it + 1 < kt ? it + 1 : it) * BK;      // last iteration: a harmless reload of the current tile
        nt_stage<EA>(ars, aoff, knext, ra);
        nt_stage<EB>(brs, boff, knext, rb);
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);   // ... and the loads stay ahead of them (the scheduler sinks them to save registers)
        const float* as = As[buf];
        const float* bs = Bs[buf];
        f32x4 af[MT], bf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(as + (lg * BM + 16 * MT * wm + 16 * i + lj) * 4);
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) bf[jn] = *reinterpret_cast<const f32x4*>(bs + (lg * BN + 16 * NT * wn + 16 * jn + lj) * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) acc[i][jn] = KM_MFMA(af[i][s], bf[jn][s], acc[i][jn]);
        // the other buffer: its last readers passed the previous barrier (after the last tile: written, never read)
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);   // the commit (and the wait for the tile) stays behind the MFMAs
        nt_commit<EA>(As[buf ^ 1], tid, ra);
        nt_commit<EB>(Bs[buf ^ 1], tid, rb);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * MT * wm + 16 * i + 4 * lg + r, n = n0 + 16 * NT * wn + 16 * jn + lj;
                if (m < g.M && n < g.N) {
                    float v = g.alpha * acc[i][jn][r];
                    if (g.bias_mode == 1) v += g.bias[n];
                    else if (g.bias_mode == 2) v += g.bias[m];
                    float* cp = C + (int64_t)m * g.c_rs + n;
                    if (g.beta != 0.f) v += g.beta * (*cp);
                    v = gemm_act(v, g.relu);
                    *cp = v;
                }
            }
}

static bool gemm_nt_ok(const GemmArgs& g) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const int64_t lim = (int64_t)1 << 29;     // floats: buffer-load offsets are 32-bit byte offsets with the top bit reserved for "out of range"
    if (((int64_t)g.M - 1) * g.a_rs + g.K >= lim || ((int64_t)g.N - 1) * g.b_cs + g.K >= lim) return false;
    return g.a_cs == 1 && g.b_rs == 1 && g.K >= 16 && g.K % 16 == 0 && g.kb_count <= 1 && g.M >= 64 && al(g.A) && al(g.B) &&
           g.a_rs % 4 == 0 && g.b_cs % 4 == 0 && g.a_bs1 % 4 == 0 && g.a_bs2 % 4 == 0 && g.b_bs1 % 4 == 0 && g.b_bs2 % 4 == 0;
}

int launch_gemm(const GemmArgs& g, int batch, void* stream) {
    static const bool fast = std::getenv("KM_GEMM_GENERIC_ONLY") == nullptr;
#ifndef KM_NT_MIN_WGS
#define KM_NT_MIN_WGS 192
#endif
    // largest tile that still gives the chip >= 192 workgroups: 128 x 128, 128 x 64, then 64 x 64 (16 accumulator registers:
    // several workgroups per CU, which a 16-step contraction needs to hide its barriers)
    const int64_t mt_wgs = (int64_t)batch * ((g.M + 127) / 128), mt64_wgs = (int64_t)batch * ((g.M + 63) / 64);
    const int64_t wgs4 = mt_wgs * ((g.N + 127) / 128), wgs2 = mt_wgs * ((g.N + 63) / 64), wgs22 = mt64_wgs * ((g.N + 63) / 64);
    static const bool mid = std::getenv("KM_GEMM_NO_NT2_MID") == nullptr;
    static const bool small = std::getenv("KM_GEMM_NO_64_TILE") == nullptr;
    static const int64_t small_below = std::getenv("KM_GEMM_64_BELOW") ? atoll(std::getenv("KM_GEMM_64_BELOW")) : 4096;   // < 16 big tiles per CU (1024 until round 3: the legacy model's 1028-tile products run 12 % faster on 64 x 64 tiles, eight waves per SIMD against three)
    const bool use4 = g.N > 64 && wgs4 >= KM_NT_MIN_WGS;
    const bool use2 = !use4 && wgs2 >= KM_NT_MIN_WGS && (mid || g.N <= 64);
    const bool use22 = small && gemm_nt_ok(g) && ((!use4 && !use2 && wgs22 >= KM_NT_MIN_WGS) || ((use4 ? wgs4 : wgs2) < small_below && (use4 || use2)));
    if (fast && use22) {
        const dim3 grid((unsigned)((g.N + 63) / 64), (unsigned)((g.M + 63) / 64), (unsigned)batch);
        hipLaunchKernelGGL((gemm_nt_kernel<2, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, g);
        HIP_TRY(hipGetLastError());
        return KM_OK;
    }
    if (fast && gemm_nt_ok(g) && (use4 || use2)) {
        if (use4) {
            const dim3 grid((unsigned)((g.N + 127) / 128), (unsigned)((g.M + 127) / 128), (unsigned)batch);
            if (g.K <= 256) hipLaunchKernelGGL((gemm_nt_kernel<4, 4, true>), grid, dim3(256), 0, (hipStream_t)stream, g);
            else hipLaunchKernelGGL((gemm_nt_kernel<4, 4, false>), grid, dim3(256), 0, (hipStream_t)stream, g);
        } else {
            const dim3 grid((unsigned)((g.N + 63) / 64), (unsigned)((g.M + 127) / 128), (unsigned)batch);
            hipLaunchKernelGGL((gemm_nt_kernel<4, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, g);   // 80 registers either way
        }
        HIP_TRY(hipGetLastError());
        return KM_OK;
    }
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
    const int64_t KP = (c->KT + 15) / 16 * 16;
    return 2 * NKk * d /* Y, V */ + H * 28 * NKk /* S */ + 28 * d /* O */ + 28 * (d / 2) /* Hd */ + KP * NKk /* packed X */;
}

// ---------------------------------------------------------------------------------------------------------
// encoder_tn_kernel: Y_b (80 x d) = Xp_b^T Wce_pad^T + b for the packed input image Xp (B, KP, 80) that the
// front end writes (rows = frames: T long rows, 3 short-term rows, zero rows up to KP = multiple of 16) and the
// zero-padded weight (d, KP).  A is k-major (a frame's 80 channels are contiguous): a thread loads a 4 x 4 block
// (4 frames x 4 channels) and stores its transpose as four ds_write_b128 into the [k/4][row][k%4] layout of
// gemm_nt_kernel; the 80 rows are exactly 5 MFMA tiles, so nothing is padded (a 64- or 128-row tile wastes 37 %).
// 80 x 128 tile per 256 threads, wave w owns columns 32 w .. +31: 10 accumulators, 7 ds_read_b128 per 40 MFMAs.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void encoder_tn_kernel(const float* __restrict__ xp, const float* __restrict__ wpad,
                                                         const float* __restrict__ bias, float* __restrict__ Y, int d, int KP) {
    constexpr int NKc = 80, BN = 128;
    __shared__ __attribute__((aligned(16))) float As[2][4 * NKc * 4];
    __shared__ __attribute__((aligned(16))) float Bs[2][4 * BN * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lg = lane >> 4, lj = lane & 15;
    const int b = blockIdx.y, n0 = blockIdx.x * BN;
    const float* X = xp + (int64_t)b * KP * NKc;
    // A staging: threads 0..79 -> (k group, channel quad); B staging: two (row, k group) float4 per thread
    const bool a_thr = tid < 80;
    const int akg = tid / 20, am4 = tid - akg * 20;
    const float* apt = X + (int64_t)(4 * akg) * NKc + 4 * am4;
    const float* bpt[2];
    bool bok[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = tid + 256 * e, n = idx & (BN - 1), kg = idx >> 7;
        bok[e] = n0 + n < d;
        bpt[e] = wpad + (int64_t)(bok[e] ? n0 + n : 0) * KP + 4 * kg;
    }
    float4 ra[4], rb[2];
    auto stage = [&](int k0, float4 (&a4)[4], float4 (&b2)[2]) {
        if (a_thr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a4[i] = *reinterpret_cast<const float4*>(apt + (int64_t)(k0 + i) * NKc);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float4 v = *reinterpret_cast<const float4*>(bpt[e] + k0);
            b2[e] = bok[e] ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](int buf, const float4 (&a4)[4], const float4 (&b2)[2]) {
        if (a_thr) {   // transpose the 4 x 4 block: row (channel 4 am4 + j) gets the 4 frames of k group akg
            float* dst = &As[buf][(akg * NKc + 4 * am4) * 4];
            *reinterpret_cast<float4*>(dst + 0) = make_float4(a4[0].x, a4[1].x, a4[2].x, a4[3].x);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(a4[0].y, a4[1].y, a4[2].y, a4[3].y);
            *reinterpret_cast<float4*>(dst + 8) = make_float4(a4[0].z, a4[1].z, a4[2].z, a4[3].z);
            *reinterpret_cast<float4*>(dst + 12) = make_float4(a4[0].w, a4[1].w, a4[2].w, a4[3].w);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) *reinterpret_cast<float4*>(&Bs[buf][(tid + 256 * e) * 4]) = b2[e];
    };
    f32x4 acc[5][2];
#pragma unroll
    for (int i = 0; i < 5; ++i) { acc[i][0] = f32x4{0, 0, 0, 0}; acc[i][1] = f32x4{0, 0, 0, 0}; }
    const int kt = KP / 16;
    stage(0, ra, rb);
    commit(0, ra, rb);
    __syncthreads();
    for (int it = 0; it < kt; ++it) {
        const int buf = it & 1;
        stage((it + 1 < kt ? it + 1 : it) * 16, ra, rb);
        asm volatile("" ::: "memory");   // pins the prefetch loads above the MFMAs (the compiler otherwise sinks them into commit())
        f32x4 af[5], bf[2];
#pragma unroll
        for (int i = 0; i < 5; ++i) af[i] = *reinterpret_cast<const f32x4*>(&As[buf][(lg * NKc + 16 * i + lj) * 4]);
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) bf[jn] = *reinterpret_cast<const f32x4*>(&Bs[buf][(lg * BN + 32 * wave + 16 * jn + lj) * 4]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                acc[i][0] = KM_MFMA(af[i][s], bf[0][s], acc[i][0]);
                acc[i][1] = KM_MFMA(af[i][s], bf[1][s], acc[i][1]);
            }
        commit(buf ^ 1, ra, rb);
        __syncthreads();
    }
    float* Yb = Y + (int64_t)b * NKc * d;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int n = n0 + 32 * wave + 16 * jn + lj;
            if (n < d) {
                const float bb = bias[n];
#pragma unroll
                for (int r = 0; r < 4; ++r) Yb[(int64_t)(16 * i + 4 * lg + r) * d + n] = acc[i][jn][r] + bb;
            }
        }
}

template <int NW, int CT, bool FUSE_DB>
static int launch_encoder_ln(Context* c, int64_t B, hipStream_t st, const float* xp, int KP, const EncSrc& src) {
    hipLaunchKernelGGL((encoder_ln_kernel<NW, CT, FUSE_DB>), dim3((unsigned)B), dim3(64 * NW), 0, st, xp, dv(c, "wce_pg"), dv(c, "bce"),
                       dv(c, "ln_g"), dv(c, "ln_b"), c->ws_generic, KP, src);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

// ---------------------------------------------------------------------------------------------------------
// core512_kernel: the whole d_model 512 core of one window in ONE workgroup -- encoder + LayerNorm, scores + softmax, value
// projection + P V + decoder + tail: the three kernels of round 2 / 3 as three stages of one launch (their bodies, unchanged).  Each of
// them was one 512-thread workgroup per window already, i.e. one workgroup per CU at the C4 batch: every kernel boundary drained and
// refilled the whole chip (profiles/r04_c4_pmc_*.txt: CUs busy 0.89 / 0.78 / 0.91 of the three launches).  Y_b and the softmaxed
// scores still go through memory -- written and read by the SAME workgroup now, so they come back from its own L2 -- and a
// __syncthreads() orders the stages.  LDS: the encoder's 26 KB static + the 109 KB dynamic image of the last stage, whose head doubles
// as the scores stage's Y buffers.
// ---------------------------------------------------------------------------------------------------------
struct Core512Args {
    const float* xp; const float* wce_pg; const float* bce; const float* ln_g; const float* ln_b; float* Y; int KP; EncSrc src;
    const float* qk_pg; float* S; int rows;
    const float* wv_bg; const float* wf_pg; const float* bf; const float* w2; const float* b2; const float* zemo; const float* wsum;
    float* out; float* raw;
};
template <int TPW, int HPW, bool FUSE_DB>
__global__ __launch_bounds__(512) void core512_kernel(Core512Args a) {
    __shared__ EncLds<8> el;
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    const int b = (int)blockIdx.x;
    encoder_ln_body<8, 4, FUSE_DB>(a.xp, a.wce_pg, a.bce, a.ln_g, a.ln_b, a.Y, a.KP, a.src, b, el);
    __syncthreads();          // Y_b is in memory for every wave of this workgroup
    scores_softmax_body<512, TPW>(a.Y, a.qk_pg, a.S, a.rows, b, *reinterpret_cast<float (*)[2][kScoresYsFloats]>(gsm));
    __syncthreads();          // ... and the window's attention weights
    attn_out_vr_body<512, HPW>(a.S, a.Y, a.wv_bg, a.wf_pg, a.bf, a.w2, a.b2, a.zemo, a.wsum, a.out, a.raw, b, gsm);
}

static bool core512_merge_ok(Context* c, const float* attn) {
    return c->d == 512 && c->NK == 80 && c->DH == 256 && (c->H == 8 || c->H == 16) && !attn && !c->opt.no_ln_fusion &&
           !c->opt.no_score_fusion && !c->opt.no_out_fusion && !c->opt.no_v_fusion && !c->opt.no_core_merge && c->packed.count("qk_pg") &&
           c->packed.count("wf_pg") && c->packed.count("wv_bg");
}

template <bool FUSE_DB>
static int launch_core512(Context* c, int64_t B, const float* xp, int KP, const EncSrc& src, const float* zemo, float* out, float* raw,
                          hipStream_t st) {
    const int d = c->d, H = c->H, NKk = c->NK;
    float* Y = c->ws_generic;
    float* S = Y + 2 * B * NKk * d;
    Core512Args a{xp, dv(c, "wce_pg"), dv(c, "bce"), dv(c, "ln_g"), dv(c, "ln_b"), Y, KP, src, dv(c, "qk_pg"), S, H * 28,
                  dv(c, "wv_bg"), dv(c, "wf_pg"), dv(c, "bf"), dv(c, "w2"), dv(c, "b2"), zemo, dv(c, "wsum"), out, raw};
    constexpr int lds = (2 * 16 * 81 * 4 + 32 * (512 + 8) + 8 * 32) * (int)sizeof(float);
    static PerDeviceOnce once;
    if (once.first(c->device)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&core512_kernel<2, 1, FUSE_DB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&core512_kernel<4, 2, FUSE_DB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    if (H == 8) hipLaunchKernelGGL((core512_kernel<2, 1, FUSE_DB>), dim3((unsigned)B), dim3(512), lds, st, a);
    else hipLaunchKernelGGL((core512_kernel<4, 2, FUSE_DB>), dim3((unsigned)B), dim3(512), lds, st, a);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

// d_model -> (waves, column tiles per wave) of the fused encoder + LayerNorm kernel
template <bool FUSE_DB>
static int launch_encoder_ln_for(Context* c, int64_t B, hipStream_t st, const float* xp, int KP, const EncSrc& src) {
    switch (c->d) {
        case 512: return launch_encoder_ln<8, 4, FUSE_DB>(c, B, st, xp, KP, src);
        case 256: return launch_encoder_ln<8, 2, FUSE_DB>(c, B, st, xp, KP, src);
        case 64:  return launch_encoder_ln<2, 2, FUSE_DB>(c, B, st, xp, KP, src);
    }
    return fail(KM_ERR_UNSUPPORTED, "no fused encoder for d_model=%d", c->d);
}

static int core_generic_after_encoder(Context* c, int64_t B, const float* zemo, float* out, float* raw, float* attn, void* stream,
                                      bool ln_done = false);

// the generic core fed by the packed image of the front end (km_forward_audio on generic shapes)
int launch_core_generic_packed(Context* c, const float* xp, int64_t B, const float* zemo, float* out, float* raw, float* attn,
                               void* stream) {
    const int d = c->d, NKk = c->NK, KP = (c->KT + 15) / 16 * 16;
    if (NKk != 80) return fail(KM_ERR_UNSUPPORTED, "packed encoder path needs 80 mel channels");
    float* Y = c->ws_generic;
    const bool fuse_ln = !c->opt.no_ln_fusion;
    hipStream_t st = (hipStream_t)stream;
    bool ln_done = fuse_ln;
    if (core512_merge_ok(c, attn)) return launch_core512<false>(c, B, xp, KP, EncSrc{}, zemo, out, raw, st);
    if (fuse_ln && (d == 512 || d == 256 || d == 64)) {
        if (int rc = launch_encoder_ln_for<false>(c, B, st, xp, KP, EncSrc{})) return rc;
    } else {
        ln_done = false;
        hipLaunchKernelGGL(encoder_tn_kernel, dim3((unsigned)((d + 127) / 128), (unsigned)B), dim3(256), 0, st, xp,
                           dv(c, "wce_pad"), dv(c, "bce"), Y, d, KP);
    }
    HIP_TRY(hipGetLastError());
    return core_generic_after_encoder(c, B, zemo, out, raw, attn, stream, ln_done);
}

// the generic core fed directly by the front end's power-mel workspace (dB conversion inside the encoder kernel);
// returns KM_ERR_UNSUPPORTED for shapes without an encoder_ln_kernel instantiation (callers then use the packed image)
bool generic_core_takes_power(Context* c) {
    return c->NK == 80 && (c->d == 512 || c->d == 256 || c->d == 64) && !c->opt.no_ln_fusion && !c->opt.no_db_fusion;
}

LogParams plan_log_params(MelPlan* p);

int launch_core_generic_power(Context* c, MelPlan* plan, int64_t B, int64_t n_frames, const float* zemo, float* out,
                              float* raw, float* attn, void* stream) {
    if (!generic_core_takes_power(c)) return fail(KM_ERR_UNSUPPORTED, "no fused encoder for d_model=%d", c->d);
    const int KP = (c->KT + 15) / 16 * 16;
    hipStream_t st = (hipStream_t)stream;
    EncSrc src{c->ws_melpow, c->ws_melmax, (int)n_frames, c->T, plan_log_params(plan)};
    if (core512_merge_ok(c, attn)) {
        if (int rc = launch_core512<true>(c, B, nullptr, KP, src, zemo, out, raw, st)) return rc;
        c->melmax_dirty = false;
        return KM_OK;
    }
    if (int rc = launch_encoder_ln_for<true>(c, B, st, nullptr, KP, src)) return rc;
    c->melmax_dirty = false;     // every slot the encoder read was just written by the front end and is re-zeroed by the encoder
    return core_generic_after_encoder(c, B, zemo, out, raw, attn, stream, true);
}

float* generic_packed_x(Context* c, int64_t B) {     // the packed-X slot behind the other intermediates of B windows
    const int64_t d = c->d, H = c->H, NKk = c->NK;
    return c->ws_generic + B * (2 * NKk * d + H * 28 * NKk + 28 * d + 28 * (d / 2));
}

// caller-provided log-mel (B, T_in, 80) + short-term rows (B, 3, 80) -> packed encoder input (B, KP, 80):
// zero-pad / truncate the time axis to T (dual_stream_attention.py:193-202), then the 3 short rows, then zero rows
__global__ __launch_bounds__(256) void pack_x_kernel(const float* __restrict__ mel, const float* __restrict__ mel_short, int T_in,
                                                     int T, int KP, float* __restrict__ xp) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;        // float4 index inside the window's image
    if (i >= KP * 20) return;
    const int r = i / 20, c4 = i - r * 20;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < T) { if (r < T_in) v = reinterpret_cast<const float4*>(mel + ((int64_t)b * T_in + r) * 80)[c4]; }
    else if (r < T + 3) v = reinterpret_cast<const float4*>(mel_short + ((int64_t)b * 3 + (r - T)) * 80)[c4];
    reinterpret_cast<float4*>(xp + (int64_t)b * KP * 80)[i] = v;
}

int launch_core_generic(Context* c, const float* mel, int64_t B, int64_t T_in, const float* mel_short, const float* zemo,
                        float* out, float* raw, float* attn, void* stream) {
    const int d = c->d, T = c->T, KT = c->KT, NKk = c->NK;
    if (NKk > 128) return fail(KM_ERR_UNSUPPORTED, "more than 128 mel channels");
    if (NKk == 80 && !c->opt.generic_staged && (reinterpret_cast<uintptr_t>(mel) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(mel_short) & 15) == 0) {
        const int KP = (KT + 15) / 16 * 16;
        float* xp = generic_packed_x(c, B);
        hipLaunchKernelGGL(pack_x_kernel, dim3((unsigned)((KP * 20 + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream, mel,
                           mel_short, (int)T_in, T, KP, xp);
        HIP_TRY(hipGetLastError());
        return launch_core_generic_packed(c, xp, B, zemo, out, raw, attn, stream);
    }
    float* Y = c->ws_generic;
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
    return core_generic_after_encoder(c, B, zemo, out, raw, attn, stream);
}

static int core_generic_after_encoder(Context* c, int64_t B, const float* zemo, float* out, float* raw, float* attn, void* stream,
                                      bool ln_done) {
    const int d = c->d, H = c->H, hd = c->hd, DH = c->DH, NKk = c->NK;
    float* Y = c->ws_generic;
    float* V = Y + B * NKk * d;
    float* S = V + B * NKk * d;
    float* O = S + B * H * 28 * NKk;
    float* Hd = O + B * 28 * d;
    GemmArgs g{};
    if (!ln_done)
        hipLaunchKernelGGL(ln_rows_kernel, dim3((unsigned)((B * NKk + 3) / 4)), dim3(256), 0, (hipStream_t)stream, Y,
                           B * NKk, d, dv(c, "ln_g"), dv(c, "ln_b"), (float*)nullptr, (float*)nullptr);
    const bool fused_scores = d == 512 && NKk == 80 && c->packed.count("qk_pg") && !c->opt.no_score_fusion;
    if (fused_scores) {
        // up to 16 row tiles (8 heads): two per wave; more (16 heads: 28): four per wave, still one sweep over Y
        if (H * 28 <= 256) hipLaunchKernelGGL((scores_softmax_kernel<512, 2>), dim3((unsigned)B), dim3(512), 0, (hipStream_t)stream, Y, dv(c, "qk_pg"), S, H * 28);
        else hipLaunchKernelGGL((scores_softmax_kernel<512, 4>), dim3((unsigned)B), dim3(512), 0, (hipStream_t)stream, Y, dv(c, "qk_pg"), S, H * 28);
        HIP_TRY(hipGetLastError());
    }
    // S[b] (H*28 x 80) = Qk (H*28 x d) Y_b^T: all heads of a window in ONE product -- the folded query matrix of every
    // head spans the full d, so the heads are just row blocks (28-row tiles would waste 56 % of a 64-row MFMA tile)
    g = GemmArgs{}; g.alpha = 1.f; g.batch2 = 1;
    g.A = dv(c, "qk"); g.a_rs = d; g.a_cs = 1; g.a_bs1 = 0;
    g.B = Y; g.b_rs = 1; g.b_cs = d; g.b_bs1 = (int64_t)NKk * d;
    g.C = S; g.c_rs = NKk; g.c_bs1 = (int64_t)H * 28 * NKk;
    g.M = H * 28; g.N = NKk; g.K = d;
    if (!fused_scores) {
        if (int rc = launch_gemm(g, (int)B, stream)) return rc;
        hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((B * H * 28 + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S,
                           B * H * 28, NKk);
    }
    if (attn) {
        const int64_t n = B * 28 * NKk;
        hipLaunchKernelGGL(head_mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, S, attn,
                           B, H, 28 * NKk);
    }
    if (d == 512 && DH == 256 && NKk == 80 && (H == 8 || H == 16) && c->packed.count("wf_pg") && c->packed.count("wv_bg") &&
        !c->opt.no_out_fusion && !c->opt.no_v_fusion) {
        // the value projection, P V and the decoder in one kernel per window, V in registers (8 or 16 heads)
        constexpr int lds = (2 * 16 * 81 * 4 + 32 * (512 + 8) + 8 * 32) * (int)sizeof(float);
        static PerDeviceOnce once;
        if (once.first(c->device)) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_out_vr_kernel<512, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_out_vr_kernel<512, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        }
        if (H == 8)
            hipLaunchKernelGGL((attn_out_vr_kernel<512, 1>), dim3((unsigned)B), dim3(512), lds, (hipStream_t)stream, S, Y, dv(c, "wv_bg"),
                               dv(c, "wf_pg"), dv(c, "bf"), dv(c, "w2"), dv(c, "b2"), zemo, dv(c, "wsum"), out, raw);
        else
            hipLaunchKernelGGL((attn_out_vr_kernel<512, 2>), dim3((unsigned)B), dim3(512), lds, (hipStream_t)stream, S, Y, dv(c, "wv_bg"),
                               dv(c, "wf_pg"), dv(c, "bf"), dv(c, "w2"), dv(c, "b2"), zemo, dv(c, "wsum"), out, raw);
        HIP_TRY(hipGetLastError());
        return KM_OK;
    }
    // V = Y Wv^T   (the value bias is folded into bf)
    g = GemmArgs{}; g.alpha = 1.f; g.batch2 = 1;
    g.A = Y; g.a_rs = d; g.a_cs = 1;
    g.B = dv(c, "wv_raw"); g.b_rs = 1; g.b_cs = d;
    g.C = V; g.c_rs = d; g.M = (int)(B * NKk); g.N = d; g.K = d;
    if (int rc = launch_gemm(g, 1, stream)) return rc;
    if (d == 512 && DH == 256 && NKk == 80 && (H == 8 || H == 16) && c->packed.count("wf_pg") &&
        !c->opt.no_out_fusion) {
        // P V, the decoder fold and the tail in one kernel per window
        constexpr int lds = (32 * (512 + 8) + 8 * 32) * (int)sizeof(float);
        static PerDeviceOnce once;
        if (once.first(c->device))
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_out_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));

        hipLaunchKernelGGL(attn_out_kernel<512>, dim3((unsigned)B), dim3(512), lds, (hipStream_t)stream, S, V, dv(c, "wf_pg"), dv(c, "bf"),
                           dv(c, "w2"), dv(c, "b2"), zemo, dv(c, "wsum"), out, raw, H);
        HIP_TRY(hipGetLastError());
        return KM_OK;
    }
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
    g.B = dv(c, "wf_t"); g.b_rs = 1; g.b_cs = d;
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

// the stand-alone attention launch (km_legacy_attn_dev.h: legacy_attention_body); km_kmmf.hip runs the same body ahead of the tail
__global__ __launch_bounds__(256) void legacy_attention_kernel(const float* __restrict__ Qs, const float* __restrict__ Kp,
                                                              const float* __restrict__ Vp, float* __restrict__ O, int64_t BH, int Tm,
                                                              int H, int NQ, unsigned* __restrict__ zero_max) {
    const int64_t bh = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bh >= BH) return;                       // wave-uniform; no barrier below
    legacy_attention_body(Qs, Kp, Vp, O, bh, Tm, H, NQ, zero_max);
}

int launch_legacy_encoder_fused(Context* c, const float* mel, int64_t rows, float* Kp, float* Vp, void* stream, const unsigned* melmax,
                                int Tm, const LogParams* lp);      // km_kmmf.hip
int launch_legacy_tail_fused(Context* c, const float* O, int64_t B, float* out, void* stream);
int launch_legacy_attn_tail_fused(Context* c, const float* Kp, const float* Vp, float* O, int64_t B, int Tm, unsigned* zero_max, float* out,
                                  void* stream);      // km_kmmf.hip

// pow_src != null: `mel` is ignored, the fused encoder reads the front end's power-mel + window maxima and converts on the fly
// (the caller has checked legacy_pow_ok); the attention kernel then puts the maxima back to zero
bool legacy_pow_ok(Context* c) {
    return c->legacy_fused && !c->opt.legacy_no_enc_fusion && !c->opt.legacy_no_attn_fusion && c->hd == 32 && c->NB <= 64 &&
           (reinterpret_cast<uintptr_t>(c->ws_melpow) & 15) == 0;
}
int launch_legacy(Context* c, const float* mel, int64_t B, int64_t Tm, float* out, void* stream, const LegacyPowSrc* pow_src) {
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
    unsigned* zero_max = nullptr;
    if (pow_src) {
        if (int rc = launch_legacy_encoder_fused(c, pow_src->melpow, B * Tm, Kp, Vp, stream, pow_src->melmax, (int)Tm, pow_src->lp)) return rc;
        zero_max = pow_src->melmax;
    } else if (c->legacy_fused && !c->opt.legacy_no_enc_fusion && (reinterpret_cast<uintptr_t>(mel) & 15) == 0) {
        // audio_encoder + key / value projections with the hidden activations resident in LDS (km_kmmf.hip)
        if (int rc = launch_legacy_encoder_fused(c, mel, B * Tm, Kp, Vp, stream, nullptr, (int)Tm, nullptr)) return rc;
    } else {
    // audio_encoder: Linear(80,d) ReLU [Dropout] Linear(d,d) ReLU [Dropout]   (:44-51, :129)
    if (int rc = launch_gemm(lin(mel, NKk, dv(c, "l_w0"), NKk, E1, B * Tm, d, dv(c, "l_b0"), 1), 1, stream)) return rc;
    if (int rc = launch_gemm(lin(E1, d, dv(c, "l_w3"), d, E, B * Tm, d, dv(c, "l_b3"), 1), 1, stream)) return rc;
    // nn.MultiheadAttention(query = 52 learnable rows, key = value = encoded frames)   (:136-141)
    if (int rc = launch_gemm(lin(E, d, dv(c, "l_wk"), d, Kp, B * Tm, d, dv(c, "l_bk"), 0), 1, stream)) return rc;
    if (int rc = launch_gemm(lin(E, d, dv(c, "l_wv"), d, Vp, B * Tm, d, dv(c, "l_bv"), 0), 1, stream)) return rc;
    }
    if (hd == 32 && H == 8 && NQ == 52 && !c->opt.legacy_no_attn_fusion && c->legacy_fused && c->legacy_tail_fused && !c->opt.legacy_no_tail_fusion &&
        !c->opt.legacy_no_merge)       // attention + tail of a window in one workgroup, one launch
        return launch_legacy_attn_tail_fused(c, Kp, Vp, O, B, (int)Tm, zero_max, out, stream);
    if (hd == 32 && NQ <= 64 && !c->opt.legacy_no_attn_fusion) {
        hipLaunchKernelGGL(legacy_attention_kernel, dim3((unsigned)((B * H + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dv(c, "l_q"), Kp, Vp, O,
                           B * H, (int)Tm, H, NQ, zero_max);
        HIP_TRY(hipGetLastError());
    } else {
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
    }
    if (c->legacy_fused && c->legacy_tail_fused && !c->opt.legacy_no_tail_fusion)      // out_proj + decoder + mean with the 52 rows resident in LDS
        return launch_legacy_tail_fused(c, O, B, out, stream);
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
