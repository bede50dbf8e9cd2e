// Device side of the strided exact-fp32 MFMA GEMM for SMALL problems (training step at 8 windows per GPU: 4 to 80 tiles per
// product): one 64 x 64 tile per 256 threads, BK = 32, operands through arbitrary strides like gemm_kernel, but with the
// global loads of the next D k-tiles in flight in registers (gemm_kernel keeps one).  A product of 8 k-steps on a handful
// of workgroups is bound by the load -> LDS -> MFMA chain of every step, not by arithmetic: with one tile of prefetch a
// step costs a full L2 round trip, with D = 4 the round trips overlap.  LDS tiles are double buffered (one barrier per
// k-step).  Included inside namespace km of a .hip translation unit that defines f32x4 and KM_MFMA.
#pragma once

#include "km_gemm.h"

namespace ggd {
constexpr int BM = 64, BN = 64, BK = 32, LDT = 80;
constexpr int EPT = BM * BK / 256;
constexpr int LDS_FLOATS = 2 * 2 * BK * LDT;          // A and B tiles, two buffers each: 40 KB
}

struct GemmTileCtx {
    const float* A; const float* Bp;
    int m0, n0, tid;
    bool a_kfast, b_kfast;
};

__device__ __forceinline__ void gemm_stage_d(const GemmArgs& g, const GemmTileCtx& c, const float* A, const float* Bp, int k0,
                                             float (&ra)[ggd::EPT], float (&rb)[ggd::EPT]) {
    using namespace ggd;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int idx = c.tid + 256 * e;
        int m, k;
        if (c.a_kfast) { k = idx & (BK - 1); m = idx / BK; } else { m = idx & 63; k = idx >> 6; }
        const int gm = c.m0 + m, gk = k0 + k;
        ra[e] = (gm < g.M && gk < g.K) ? A[gm * g.a_rs + gk * g.a_cs] : 0.f;
        int n, kb;
        if (c.b_kfast) { kb = idx & (BK - 1); n = idx / BK; } else { n = idx & 63; kb = idx >> 6; }
        const int gn = c.n0 + n, gkb = k0 + kb;
        rb[e] = (gn < g.N && gkb < g.K) ? Bp[gkb * g.b_rs + gn * g.b_cs] : 0.f;
    }
}

// tile (bx, by) of output batch bz; smem = ggd::LDS_FLOATS floats
template <int D>
__device__ __forceinline__ void gemm_tile_dev(const GemmArgs& g, int bx, int by, int bz, float* smem) {
    using namespace ggd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lg = lane >> 4, lj = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;                 // 2 x 2 waves, 32 x 32 each
    const int z1 = bz / g.batch2, z2 = bz - z1 * g.batch2;
    GemmTileCtx c;
    c.A = g.A + z1 * g.a_bs1 + z2 * g.a_bs2;
    c.Bp = g.B + z1 * g.b_bs1 + z2 * g.b_bs2;
    float* C = g.C + z1 * g.c_bs1 + z2 * g.c_bs2;
    c.m0 = by * BM; c.n0 = bx * BN; c.tid = tid;
    c.a_kfast = g.a_cs == 1;
    c.b_kfast = g.b_rs == 1;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { acc[i][0] = f32x4{0, 0, 0, 0}; acc[i][1] = f32x4{0, 0, 0, 0}; }
    const int kbn = g.kb_count > 0 ? g.kb_count : 1;
    const int kt = (g.K + BK - 1) / BK, total = kbn * kt;
    float ra[D][EPT], rb[D][EPT];
    auto stage = [&](int it, float (&a)[EPT], float (&b)[EPT]) {
        const int nb = it / kt, nk = it - nb * kt;
        gemm_stage_d(g, c, c.A + nb * g.a_kbs, c.Bp + nb * g.b_kbs, nk * BK, a, b);
    };
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < total) stage(s, ra[s], rb[s]);
    for (int it0 = 0; it0 < total; it0 += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int it = it0 + s;
            if (it < total) {                                  // workgroup-uniform
                float* As = smem + (it & 1) * (2 * BK * LDT);
                float* Bs = As + BK * LDT;
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const int idx = tid + 256 * e;
                    int m, k;
                    if (c.a_kfast) { k = idx & (BK - 1); m = idx / BK; } else { m = idx & 63; k = idx >> 6; }
                    As[k * LDT + m] = ra[s][e];
                    int n, kb;
                    if (c.b_kfast) { kb = idx & (BK - 1); n = idx / BK; } else { n = idx & 63; kb = idx >> 6; }
                    Bs[kb * LDT + n] = rb[s][e];
                }
                // one barrier per step: the tile written two steps from now reuses this buffer, and every wave passes the
                // NEXT step's barrier (behind its own MFMAs of this step) before anyone gets there
                __syncthreads();
                if (it + D < total) stage(it + D, ra[s], rb[s]);
#pragma unroll
                for (int q = 0; q < BK / 4; ++q) {
                    const float* ar = As + (4 * q + lg) * LDT + 32 * wm + lj;
                    const float* br = Bs + (4 * q + lg) * LDT + 32 * wn + lj;
                    const float a0 = ar[0], a1 = ar[16], b0 = br[0], b1 = br[16];
                    acc[0][0] = KM_MFMA(a0, b0, acc[0][0]);
                    acc[0][1] = KM_MFMA(a0, b1, acc[0][1]);
                    acc[1][0] = KM_MFMA(a1, b0, acc[1][0]);
                    acc[1][1] = KM_MFMA(a1, b1, acc[1][1]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = c.m0 + 32 * wm + 16 * i + 4 * lg + r, n = c.n0 + 32 * wn + 16 * jn + lj;
                if (m < g.M && n < g.N) {
                    float v = g.alpha * acc[i][jn][r];
                    if (g.bias_mode == 1) v += g.bias[n];
                    else if (g.bias_mode == 2) v += g.bias[m];
                    float* cp = C + (int64_t)m * g.c_rs + n;
                    if (g.beta != 0.f) v += g.beta * (*cp);
                    if (g.relu == 1) v = v < 0.f ? 0.f : v;
                    else if (g.relu == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                    if (g.drop) v = g.drop[gemm_drop_row(g.drop_map, m) * g.N + n] ? v * g.drop_scale : 0.f;
                    *cp = v;
                }
            }
    __syncthreads();      // the caller may reuse smem (another tile of the same workgroup)
}
