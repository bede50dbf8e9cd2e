// Device code of the shape-generic attention chain behind the encoder (d_model 512): scores + softmax, and the value
// projection + P V + decoder + tail.  Kernels only (km_device.h is the one dependency), so that km_generic.hip and the
// stand-alone timing harnesses under tools/micro/ instantiate the same code.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "km_device.h"
#include "km_encoder_dev.h"

namespace km {

// blendshape index -> row of the 28 mouth queries (mouth set 14..40 + 51), -1 for the expression set
__device__ __forceinline__ int gen_mouth_slot(int i) { return (i >= 14 && i <= 40) ? i - 14 : (i == 51 ? 27 : -1); }

// ---------------------------------------------------------------------------------------------------------
// scores_softmax_kernel<D>: P_b (H*28 x 80) = softmax_rows(Qk Y_b^T) for one window per workgroup (8 waves).  Wave w owns
// row tiles 2 w' ... of the stacked heads (a row's 80 keys are 5 column tiles in ONE wave: the row maximum and sum are
// an in-lane pass over the 5 tiles plus a DPP reduction over the 16 key lanes -- no exchange), A = the packed folded
// query-key image (qk_pg, one coalesced KiB per wave and k block, k block outermost so a workgroup's fetch per step is
// one contiguous run), B = Y rows staged through LDS in 64-k chunks.  Replaces the scores GEMM + softmax_rows_kernel
// and the round trip of the raw scores.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 as_f32x4(u32x4 v) { return __builtin_bit_cast(f32x4, v); }

__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
    return v;
}

// KM_SC_SKIP (timing harness only, tools/micro/attn_bench.hip; 0 in the library): bit 0 no MFMAs, bit 1 no S store,
// bit 2 no softmax arithmetic, bit 4 whole row tiles only (no k split at 8 heads), bit 3 no query-key image loads after the
// prologue (half the A ring is then never written,
// so the compiler may drop MFMAs: an upper bound on what the loads cost, not a measurement).  Reading 8 or 64 COPIES of
// the image from different workgroups changed nothing (49.4 / 84.4 us either way): the loads do not contend on addresses.
#ifndef KM_SC_SKIP
#define KM_SC_SKIP 0
#endif

constexpr int kScoresYsFloats = 16 * (80 + 1) * 4;      // one Y chunk image [k / 4][row, padded to 81][k % 4]

// window b of the launch; Ys: the workgroup's two Y chunk buffers (the kernel's static array, or the head of core512_kernel's dynamic LDS)
template <int D, int TPW>
__device__ __forceinline__ void scores_softmax_body(const float* __restrict__ Y, const float* __restrict__ qk_pg,
                                                    float* __restrict__ S, int rows /* H * 28 */, int b, float (&Ys)[2][kScoresYsFloats]) {
    constexpr int NKc = 80, KB = D / 16, CH = 4, NCH = KB / CH, QS = NKc + 1;
    static_assert(16 * QS * 4 == kScoresYsFloats, "Y chunk image");
    static_assert(KB % CH == 0 && CH == 4, "k blocks come in chunks of four (the A prefetch ring has four slots)");
    // Y_b reaches the MFMAs through LDS in chunks of 64 k ([k / 4][row, padded to 81][k % 4]: conflict-free b128 on both
    // sides), double buffered: read from L2 once per workgroup instead of once per wave.  The KEY ROWS are stored permuted
    // (key 4 j + t of the first 64 at row 16 t + j), so that the lane that reads row 16 t + j for column tile t ends up with
    // keys 4 j .. 4 j + 3 in its four accumulators: the softmaxed row leaves as one 16-byte store + one dword (keys 64..79)
    // per lane instead of five 64-byte segments.
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = lane >> 4, lj = lane & 15;
    const int MT = (rows + 15) >> 4;
    const float* Yb = Y + (int64_t)b * NKc * D;
    float* Sb = S + (int64_t)b * rows * NKc;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Yb), 0, (unsigned)(NKc * D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qk_pg), 0, (unsigned)(MT * KB * 1024), 0x00020000);
    constexpr unsigned OOB = 0x7fffffffu;
    u32x4 yst[3];
    auto ystage = [&](int ch) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = tid + 512 * j, row = idx >> 4, q = idx & 15;
            yst[j] = __builtin_amdgcn_raw_buffer_load_b128(yr, idx < NKc * 16 ? (unsigned)((row * D + 64 * ch + 4 * q) * 4) : OOB, 0, 0);
        }
    };
    auto ycommit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = tid + 512 * j, row = idx >> 4, q = idx & 15;
            const int prow = row < 64 ? 16 * (row & 3) + (row >> 2) : row;
            asm volatile("" ::"v"(yst[j]));     // unconditional use: the load stays out of the branch
            if (idx < NKc * 16) *reinterpret_cast<u32x4*>(&Ys[buf][(q * QS + prow) * 4]) = yst[j];
        }
    };
    // Row tiles are dealt in passes of up to 8 TPW, each wave a contiguous run of floor / ceil(tiles / 8) of them with the
    // longer runs on waves 0..: waves w and w + 4 share a SIMD, so 28 tiles (16 heads, TPW = 4: ONE sweep over Y) are 4 + 3
    // = 7 per SIMD.  14 tiles (8 heads, TPW = 2) would be 4 / 4 / 3 / 3 whole tiles; instead two of them are SPLIT along k
    // between the second waves of two SIMDs (k blocks 0-1 / 2-3 of every chunk, so every chunk stays balanced) and every
    // SIMD carries 3.5: waves 0..3 hold tiles 3 w and 3 w + 1, waves 4..7 hold tile 3 (w - 4) + 2 and one half of tile 12
    // (waves 4, 5) or 13 (waves 6, 7); the halves meet in LDS after the sweep.
    for (int base = 0; base < MT; base += 8 * TPW) {
        const int ntile = MT - base < 8 * TPW ? MT - base : 8 * TPW;      // tiles of this pass, wave-uniform
        const bool split14 = TPW == 2 && ntile == 14 && !(KM_SC_SKIP & 16);
        const int per = ntile >> 3, extra = ntile & 7;
        const int cnt = split14 ? 2 : per + (wave < extra ? 1 : 0);        // this wave's tiles (0 .. TPW)
        const int mt0 = base + (split14 ? (wave < 4 ? 3 * wave : 3 * (wave - 4) + 2) : wave * per + (wave < extra ? wave : extra));
        const int mt1 = split14 && wave >= 4 ? base + 12 + ((wave - 4) >> 1) : mt0 + 1;      // second tile (TPW = 2 only uses it)
        const int half = split14 && wave >= 4 ? (wave & 1) : -1;          // which k blocks of a chunk the second tile takes
        f32x4 acc[TPW][5];
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) acc[i][nt] = f32x4{0, 0, 0, 0};
        // qk_pg is [k block][row tile][lane][4]: what the workgroup's waves fetch for one k block is ONE contiguous
        // 1 KiB x MT run (with the row tile outermost the 14 runs sat 32 KiB apart -- one L2 channel for all of them)
        const unsigned a0o = (unsigned)((mt0 * 64 + lane) * 16), a1d = (unsigned)((mt1 - mt0) * 1024);
        auto lda = [&](int kb, int i) {
            const bool ok = i < cnt && kb < KB && (i != 1 || half < 0 || ((kb >> 1) & 1) == half);
            return as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(ar, ok ? a0o + (i == 1 ? a1d : (unsigned)(i * 1024)) + (unsigned)(kb * MT) * 1024u : OOB, 0, 0));
        };
        auto ldy = [&](int buf, int kk, int nt) { return *reinterpret_cast<const f32x4*>(&Ys[buf][((4 * kk + lg) * QS + 16 * nt + lj) * 4]); };
        ystage(0);
        f32x4 av[4][TPW], yb[2][5];                                        // A fragments two k blocks ahead, B one ahead
#pragma unroll
        for (int i = 0; i < TPW; ++i) { av[0][i] = lda(0, i); av[1][i] = lda(1, i); }
        ycommit(0);
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) yb[0][nt] = ldy(0, 0, nt);
        for (int ch = 0; ch < NCH; ++ch) {
            const int buf = ch & 1;
            const bool more = ch + 1 < NCH;
            if (more) ystage(ch + 1);
#pragma unroll
            for (int kk = 0; kk < CH; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1, kb = CH * ch + kk;
                if (!(KM_SC_SKIP & 8)) {
#pragma unroll
                    for (int i = 0; i < TPW; ++i) av[(kk + 2) & 3][i] = lda(kb + 2, i);
                }
                if (kk + 1 < CH) {
#pragma unroll
                    for (int nt = 0; nt < 5; ++nt) yb[nxt][nt] = ldy(buf, kk + 1, nt);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(KM_SC_SKIP & 1)) {
#pragma unroll
                    for (int i = 0; i < TPW; ++i) {
                        if (i < cnt && (i != 1 || half < 0 || (kk >> 1) == half)) {      // wave-uniform
#pragma unroll
                            for (int s = 0; s < 4; ++s)
#pragma unroll
                                for (int nt = 0; nt < 5; ++nt) acc[i][nt] = KM_MFMA(av[kk][i][s], yb[cur][nt][s], acc[i][nt]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) ycommit(buf ^ 1);
            __syncthreads();
            if (more) {
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) yb[0][nt] = ldy(buf ^ 1, 0, nt);
            }
        }
        if (split14) {
            // the two halves of a split tile meet: the wave that took k blocks 2-3 hands its accumulators over through the
            // (now idle) Y buffer, the wave that took k blocks 0-1 adds them and finishes the tile
            f32x4* X = reinterpret_cast<f32x4*>(&Ys[0][0]) + ((wave - 4) >> 1) * (5 * 64) + lane;
            if (half == 1) {
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) X[nt * 64] = acc[1][nt];
            }
            __syncthreads();
            if (half == 0) {
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) acc[1][nt] += X[nt * 64];
            }
            __syncthreads();      // (a later pass would restage Y into this buffer)
        }
        // softmax over the 80 keys of every row: C/D layout puts row 4 lg + r of a tile in lanes lj = 0..15 x 5 tiles;
        // this lane's accumulator nt < 4 is key 4 lj + nt (the staging permutation), nt = 4 is key 64 + lj
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            if (i >= cnt || (i == 1 && half == 1)) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float e[5];
                if (!(KM_SC_SKIP & 4)) {
                    float m = acc[i][0][r];
#pragma unroll
                    for (int nt = 1; nt < 5; ++nt) m = fmaxf(m, acc[i][nt][r]);
                    m = row16_max(m);
                    float sum = 0.f;
#pragma unroll
                    for (int nt = 0; nt < 5; ++nt) { e[nt] = __builtin_amdgcn_exp2f((acc[i][nt][r] - m) * 1.44269504088896341f); sum += e[nt]; }
                    sum = row16_sum(sum);
                    const float inv = 1.0f / sum;
#pragma unroll
                    for (int nt = 0; nt < 5; ++nt) e[nt] *= inv;
                } else {
#pragma unroll
                    for (int nt = 0; nt < 5; ++nt) e[nt] = acc[i][nt][r];
                }
                const int row = 16 * (i == 1 ? mt1 : mt0 + i) + 4 * lg + r;
                if ((KM_SC_SKIP & 2) && e[0] != 12345.f) continue;
                if (row < rows) {
                    *reinterpret_cast<float4*>(Sb + (int64_t)row * NKc + 4 * lj) = make_float4(e[0], e[1], e[2], e[3]);
                    Sb[(int64_t)row * NKc + 64 + lj] = e[4];
                }
            }
        }
    }
}

template <int D, int TPW>
__global__ __launch_bounds__(512) void scores_softmax_kernel(const float* __restrict__ Y, const float* __restrict__ qk_pg,
                                                             float* __restrict__ S, int rows /* H * 28 */) {
    __shared__ __attribute__((aligned(16))) float Ys[2][kScoresYsFloats];
    scores_softmax_body<D, TPW>(Y, qk_pg, S, rows, (int)blockIdx.x, Ys);
}

// ---------------------------------------------------------------------------------------------------------
// attn_out_kernel<D>: everything after the softmax for one window in one workgroup (8 waves): O = P V per head straight
// from the softmaxed scores and the value projection in L2 (A fragments as one b128 per 16 keys, B as four coalesced
// dword rows), O -> LDS [32 q][D + 8], hidden^T = Wf^T O^T with the packed fold (wf_pg), ReLU . w2, cross-wave sum in
// wave order, sigmoid, stream weights, clamp.  Replaces the batched P V product (windows x heads tiny GEMMs), the fold
// GEMM and decoder_tail_kernel: 9.6 MFLOP per window that the three launches spent 107 us of latency on at C4.
// D = 512 (decoder hidden 256 = 8 waves x 32 units); heads of 64 or 32 columns.
// ---------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(512) void attn_out_kernel(const float* __restrict__ S, const float* __restrict__ V,
                                                       const float* __restrict__ wf_pg, const float* __restrict__ bf,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       const float* __restrict__ zemo, const float* __restrict__ wsum,
                                                       float* __restrict__ out, float* __restrict__ raw, int H) {
    constexpr int NKc = 80, OS = D + 8, KB = D / 16, NWv = 8;
    static_assert(D / 2 == 32 * NWv && D == 64 * NWv, "one wave per 64 output columns and per 32 hidden units");
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float* Os = gsm;                   // [32][OS]
    float* R2 = Os + 32 * OS;          // [NWv][32]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = lane >> 4, lj = lane & 15;
    const int hd = D / H, tph = hd / 16;                   // column tiles per head: 4 (hd 64) or 2 (hd 32)
    const float* Pb = S + (int64_t)b * H * 28 * NKc;
    const float* Vb = V + (int64_t)b * NKc * D;
    // ---- O[:, 64 w .. 64 w + 63] = P_h V_h for the head(s) that own these columns ----
    f32x4 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[mt][ct] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int half = 0; half < 2; ++half) {                 // two passes of two column tiles; a head spans one or both
        const int h = (64 * wave + 32 * half) / hd;
        f32x4 a[2][5];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int q = 16 * mt + lj;
#pragma unroll
            for (int kb = 0; kb < 5; ++kb)
                a[mt][kb] = q < 28 ? *reinterpret_cast<const f32x4*>(Pb + ((int64_t)h * 28 + q) * NKc + 16 * kb + 4 * lg)
                                   : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int ct = 2 * half + c2;
            const float* vcol = Vb + 64 * wave + 16 * ct + lj;
#pragma unroll
            for (int kb = 0; kb < 5; ++kb) {
                float bv[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) bv[s] = vcol[(int64_t)(16 * kb + 4 * lg + s) * D];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc[0][ct] = KM_MFMA(a[0][kb][s], bv[s], acc[0][ct]);
                    acc[1][ct] = KM_MFMA(a[1][kb][s], bv[s], acc[1][ct]);
                }
            }
        }
    }
    (void)tph;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) Os[(16 * mt + 4 * lg + r) * OS + 64 * wave + 16 * ct + lj] = acc[mt][ct][r];
    __syncthreads();
    // ---- hidden^T (256 x 32 q) = Wf^T O^T; wave w owns hidden units 32 w .. 32 w + 31 ----
    f32x4 Z[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) { Z[rt][0] = f32x4{0, 0, 0, 0}; Z[rt][1] = f32x4{0, 0, 0, 0}; }
    {
        const f32x4* fp = reinterpret_cast<const f32x4*>(wf_pg) + (size_t)wave * 2 * KB * 64 + lane;
#pragma unroll 4
        for (int kb = 0; kb < KB; ++kb) {
            const f32x4 w0 = fp[(size_t)kb * 64], w1 = fp[(size_t)(KB + kb) * 64];
            const f32x4 o0 = *reinterpret_cast<const f32x4*>(Os + lj * OS + 16 * kb + 4 * lg);
            const f32x4 o1 = *reinterpret_cast<const f32x4*>(Os + (16 + lj) * OS + 16 * kb + 4 * lg);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                Z[0][0] = KM_MFMA(w0[s], o0[s], Z[0][0]);
                Z[0][1] = KM_MFMA(w0[s], o1[s], Z[0][1]);
                Z[1][0] = KM_MFMA(w1[s], o0[s], Z[1][0]);
                Z[1][1] = KM_MFMA(w1[s], o1[s], Z[1][1]);
            }
        }
    }
    {
        float zp[2] = {0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hid = 32 * wave + 16 * rt + 4 * lg + r;
                const float bfv = bf[hid], w2v = w2[hid];
                zp[0] += fmaxf(Z[rt][0][r] + bfv, 0.f) * w2v;
                zp[1] += fmaxf(Z[rt][1][r] + bfv, 0.f) * w2v;
            }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            zp[qt] += __shfl_xor(zp[qt], 16);
            zp[qt] += __shfl_xor(zp[qt], 32);
        }
        if (lg == 0) { R2[wave * 32 + lj] = zp[0]; R2[wave * 32 + 16 + lj] = zp[1]; }
    }
    __syncthreads();
    if (tid < 52) {
        const int slot = gen_mouth_slot(tid);
        float z;
        if (slot >= 0) {
            z = b2[0];
#pragma unroll
            for (int w = 0; w < NWv; ++w) z += R2[w * 32 + slot];
        } else {
            z = zemo[b];
        }
        const float bs = 1.0f / (1.0f + expf(-z));
        if (raw) raw[(int64_t)b * 52 + tid] = bs;
        out[(int64_t)b * 52 + tid] = fminf(fmaxf(wsum[tid] * bs, 0.f), 1.f);
    }
}

// ---------------------------------------------------------------------------------------------------------
// attn_out_vr_kernel<D, HPW>: value projection, P V, the decoder fold and the tail for one window per workgroup, with the
// value matrix never leaving REGISTERS.  Wave w owns value columns 64 w .. 64 w + 63 = HPW whole heads (one of 64 columns
// or two of 32):
//   1. V[:, 64 w ..] (80 x 64, 20 accumulators) = Y Wv^T in the encoder's loop shape: A = Y through LDS in chunks of
//      64 k (the image scores_softmax_kernel uses), fragments read one k block ahead; B = wv_bg, the MFMA operand image
//      of Wv ([k block][wave][column tile][lane][4], km_host.cpp), one coalesced KiB per wave and tile straight into
//      registers, one k block ahead.  8 barriers for 2560 MFMAs per wave.
//   2. O_h = P_h V_h with the accumulators of step 1 as the B operand: the C/D layout of the 16 x 16 x 4 MFMA (column =
//      lane & 15, row = 4 (lane >> 4) + register) is its own B layout when the contraction runs over the row index (the
//      key), as in the fused d = 256 core.  A = the softmaxed scores, one 16-byte load per 16 keys (requested before
//      step 1 for the first head).
//   3. O -> LDS [32 q][D + 8], hidden^T = Wf^T O^T, ReLU . w2, cross-wave sum in wave order, sigmoid, stream weights,
//      clamp: attn_out_kernel's tail.
// One path for 8 and 16 heads (3232 MFMAs per wave either way).  It replaced a kernel that folded the value projection per
// head, O_h = (P_h Y) Wv_h^T (H = 8 only: 2816 MFMAs per wave but two barriers per head and every operand fetched right
// before use -- 0.66 of the MFMA rate, 5 us slower at C4) and, at H = 16, a separate value GEMM + attn_out_kernel (156 us).
// ---------------------------------------------------------------------------------------------------------
// KM_VR_SKIP (timing harness only): bit 0 no V-product MFMAs, bit 1 no fold, bit 2 no P V, bit 3 no weight loads in the V loop
#ifndef KM_VR_SKIP
#define KM_VR_SKIP 0
#endif

// window b of the launch; gsm: the workgroup's dynamic LDS
template <int D, int HPW>
__device__ __forceinline__ void attn_out_vr_body(const float* __restrict__ S, const float* __restrict__ Y,
                                                 const float* __restrict__ wv_bg, const float* __restrict__ wf_pg,
                                                 const float* __restrict__ bf, const float* __restrict__ w2,
                                                 const float* __restrict__ b2, const float* __restrict__ zemo,
                                                 const float* __restrict__ wsum, float* __restrict__ out,
                                                 float* __restrict__ raw, int b, float* gsm) {
    constexpr int NKc = 80, KB = D / 16, CH = 4, NCH = KB / CH, QS = NKc + 1, OS = D + 8, NWv = 8, H = NWv * HPW;
    constexpr int TPH = 4 / HPW;                               // column tiles per head
    static_assert(D == 64 * NWv && KB % CH == 0 && CH % 2 == 0, "one wave per 64 columns; k blocks in chunks of four");
    float* Ys = gsm;                                           // [2][16 * QS * 4]  Y chunk image [k / 4][row, padded][k % 4]
    float* Os = Ys + 2 * 16 * QS * 4;                          // [32][OS]
    float* R2 = Os + 32 * OS;                                  // [NWv][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = lane >> 4, lj = lane & 15;
    const float* Yb = Y + (int64_t)b * NKc * D;
    const float* Pb = S + (int64_t)b * H * 28 * NKc;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Yb), 0, (unsigned)(NKc * D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wv_bg), 0, (unsigned)(D * D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Pb), 0, (unsigned)(H * 28 * NKc * 4), 0x00020000);
    constexpr unsigned OOB = 0x7fffffffu;
    u32x4 yst[3];
    auto ystage = [&](int ch) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = tid + 512 * j, row = idx >> 4, q = idx & 15;
            yst[j] = __builtin_amdgcn_raw_buffer_load_b128(yr, idx < NKc * 16 ? (unsigned)((row * D + 64 * ch + 4 * q) * 4) : OOB, 0, 0);
        }
    };
    auto ycommit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = tid + 512 * j, row = idx >> 4, q = idx & 15;
            asm volatile("" ::"v"(yst[j]));     // unconditional use: the load stays out of the branch
            if (idx < NKc * 16) *reinterpret_cast<u32x4*>(&Ys[buf * (16 * QS * 4) + (q * QS + row) * 4]) = yst[j];
        }
    };
    // the attention weights of one head as A fragments: ap[mt][i] = P_h[q = 16 mt + lj][keys 16 i + 4 lg .. + 3]
    f32x4 ap[2][5];
    auto load_p = [&](int h) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int q = 16 * mt + lj;
#pragma unroll
            for (int i = 0; i < 5; ++i)
                ap[mt][i] = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(pr, q < 28 ? (unsigned)(((h * 28 + q) * NKc + 16 * i + 4 * lg) * 4) : OOB, 0, 0));
        }
    };
    f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[i][ct] = f32x4{0, 0, 0, 0};
    f32x4 ay[2][5], bw[2][4];
    const unsigned wo = (unsigned)((wave * 4 * 64 + lane) * 16);
    auto ldw = [&](int kb, int slot) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            bw[slot][ct] = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(wr, kb < KB ? wo + (unsigned)(kb * NWv * 4 + ct) * 1024u : OOB, 0, 0));
    };
    auto ldy = [&](int buf, int kk, int slot) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
            ay[slot][i] = *reinterpret_cast<const f32x4*>(&Ys[buf * (16 * QS * 4) + ((4 * kk + lg) * QS + 16 * i + lj) * 4]);
    };
    ystage(0);
    ldw(0, 0);
    load_p(wave * HPW);
    ycommit(0);
    __syncthreads();
    ldy(0, 0, 0);
    for (int ch = 0; ch < NCH; ++ch) {
        const int buf = ch & 1;
        const bool more = ch + 1 < NCH;
        if (more) ystage(ch + 1);
#pragma unroll
        for (int kk = 0; kk < CH; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (!(KM_VR_SKIP & 8)) ldw(CH * ch + kk + 1, nxt);
            if (kk + 1 < CH) ldy(buf, kk + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
            if (!(KM_VR_SKIP & 1)) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[i][ct] = KM_MFMA(ay[cur][i][s], bw[cur][ct][s], acc[i][ct]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) ycommit(buf ^ 1);
        __syncthreads();
        if (more) ldy(buf ^ 1, 0, 0);
    }
    // ---- O[:, 64 w ..] = P_h V_h for this wave's head(s): B operand = the accumulators ----
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh) {
        if (hh > 0) load_p(wave * HPW + hh);
        f32x4 o[2][TPH];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int c2 = 0; c2 < TPH; ++c2) o[mt][c2] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < ((KM_VR_SKIP & 4) ? 1 : 5); ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int c2 = 0; c2 < TPH; ++c2) {
                    o[0][c2] = KM_MFMA(ap[0][i][s], acc[i][TPH * hh + c2][s], o[0][c2]);
                    o[1][c2] = KM_MFMA(ap[1][i][s], acc[i][TPH * hh + c2][s], o[1][c2]);
                }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int c2 = 0; c2 < TPH; ++c2)
#pragma unroll
                for (int r = 0; r < 4; ++r) Os[(16 * mt + 4 * lg + r) * OS + 64 * wave + 16 * (TPH * hh + c2) + lj] = o[mt][c2][r];
    }
    // the first group of fold weights is requested before the barrier that publishes O
    const f32x4* fp = reinterpret_cast<const f32x4*>(wf_pg) + (size_t)wave * 2 * KB * 64 + lane;
    constexpr int FS = 8;
    f32x4 wq[2][FS][2];
#pragma unroll
    for (int j = 0; j < FS; ++j) { wq[0][j][0] = fp[(size_t)j * 64]; wq[0][j][1] = fp[(size_t)(KB + j) * 64]; }
    __syncthreads();
    // ---- hidden^T (256 x 32 q) = Wf^T O^T; wave w owns hidden units 32 w .. 32 w + 31 ----
    f32x4 Z[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) { Z[rt][0] = f32x4{0, 0, 0, 0}; Z[rt][1] = f32x4{0, 0, 0, 0}; }
    {
        // eight k blocks per step: their 16 weight fragments (coalesced KiB each, L2) are requested together, one step ahead
#pragma unroll
        for (int st = 0; st < ((KM_VR_SKIP & 2) ? 1 : KB / FS); ++st) {
            const int cur = st & 1, nxt = cur ^ 1;
            if (st + 1 < KB / FS) {
#pragma unroll
                for (int j = 0; j < FS; ++j) {
                    wq[nxt][j][0] = fp[(size_t)(FS * (st + 1) + j) * 64];
                    wq[nxt][j][1] = fp[(size_t)(KB + FS * (st + 1) + j) * 64];
                }
            }
#pragma unroll
            for (int j = 0; j < FS; ++j) {
                const int kb = FS * st + j;
                const f32x4 o0 = *reinterpret_cast<const f32x4*>(Os + lj * OS + 16 * kb + 4 * lg);
                const f32x4 o1 = *reinterpret_cast<const f32x4*>(Os + (16 + lj) * OS + 16 * kb + 4 * lg);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    Z[0][0] = KM_MFMA(wq[cur][j][0][s], o0[s], Z[0][0]);
                    Z[0][1] = KM_MFMA(wq[cur][j][0][s], o1[s], Z[0][1]);
                    Z[1][0] = KM_MFMA(wq[cur][j][1][s], o0[s], Z[1][0]);
                    Z[1][1] = KM_MFMA(wq[cur][j][1][s], o1[s], Z[1][1]);
                }
            }
        }
    }
    {
        float zp[2] = {0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hid = 32 * wave + 16 * rt + 4 * lg + r;
                const float bfv = bf[hid], w2v = w2[hid];
                zp[0] += fmaxf(Z[rt][0][r] + bfv, 0.f) * w2v;
                zp[1] += fmaxf(Z[rt][1][r] + bfv, 0.f) * w2v;
            }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            zp[qt] += __shfl_xor(zp[qt], 16);
            zp[qt] += __shfl_xor(zp[qt], 32);
        }
        if (lg == 0) { R2[wave * 32 + lj] = zp[0]; R2[wave * 32 + 16 + lj] = zp[1]; }
    }
    __syncthreads();
    if (tid < 52) {
        const int slot = gen_mouth_slot(tid);
        float z;
        if (slot >= 0) {
            z = b2[0];
#pragma unroll
            for (int w = 0; w < NWv; ++w) z += R2[w * 32 + slot];
        } else {
            z = zemo[b];
        }
        const float bs = 1.0f / (1.0f + expf(-z));
        if (raw) raw[(int64_t)b * 52 + tid] = bs;
        out[(int64_t)b * 52 + tid] = fminf(fmaxf(wsum[tid] * bs, 0.f), 1.f);
    }
}

template <int D, int HPW>
__global__ __launch_bounds__(512) void attn_out_vr_kernel(const float* __restrict__ S, const float* __restrict__ Y,
                                                          const float* __restrict__ wv_bg, const float* __restrict__ wf_pg,
                                                          const float* __restrict__ bf, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, const float* __restrict__ zemo,
                                                          const float* __restrict__ wsum, float* __restrict__ out,
                                                          float* __restrict__ raw) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    attn_out_vr_body<D, HPW>(S, Y, wv_bg, wf_pg, bf, w2, b2, zemo, wsum, out, raw, (int)blockIdx.x, gsm);
}

}  // namespace km
