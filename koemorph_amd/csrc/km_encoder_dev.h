// Fused encoder product + LayerNorm of the shape-generic attention core (device code only, so that the kernel can be
// instantiated by km_generic.hip and by the stand-alone timing harness tools/micro/enc_bench.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "km_device.h"

namespace km {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// KM_ENC_SKIP (timing harness only, tools/micro/enc_bench.hip; 0 in the library): bit 0 no MFMAs, bit 1 no Y store,
// bit 2 no LayerNorm, bit 3 no tile loads / commits after the prologue (bit 4: no weight loads, bit 5: no A tiles)
#ifndef KM_ENC_SKIP
#define KM_ENC_SKIP 0
#endif
#ifndef KM_MFMA
#define KM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif

// ---------------------------------------------------------------------------------------------------------
// encoder_ln_kernel<NW>: the same product as encoder_tn_kernel with ONE workgroup of NW = d / 32 waves per window, so
// that a row's d columns are all in the workgroup and LayerNorm (two-pass, DPP row sums + a [row][wave] exchange, as in
// the fused d=256 kernel) runs in the epilogue: no Y0 round trip, no separate ln_rows launch.
// ---------------------------------------------------------------------------------------------------------
// FUSE_DB: the input is the front end's POWER-mel (B, n_frames, 80) + window maxima; the dB / log conversion and the
// row packing (T long rows, 3 short-term rows, zero rows) happen while a tile is staged, so no packed log-mel image is
// written and read back (mel_log_packed_kernel disappears from km_forward_audio).
struct EncSrc {
    const float* melpow; unsigned* melmax; int n_frames, T; LogParams lp;   // melmax[b] is re-zeroed once every thread has read it
};

// NW waves of CT column tiles each: D = 16 CT NW columns (d = 512 -> 8 waves x 64 columns, two waves per SIMD with 20
// accumulators each; d = 256 -> 8 x 32; d = 64 -> 2 x 32).  What the timing harness showed about the first version
// (16 waves x 32 columns, both operands through LDS, 114 us at C4): the MFMA loop alone runs at 0.83 of the nominal rate,
// the tile traffic (global -> registers -> LDS, after the MFMAs of the same iteration, then the barrier) cost 19 us on top
// because every wave of the workgroup reaches that tail at the same time -- nobody is left to issue MFMAs under it.  So:
//  * B (the weight) never touches LDS: wce_pg is the MFMA operand image [k block][wave][column tile][lane][4]
//    (km_host.cpp), one coalesced 1 KiB load per wave and tile, fetched one k block ahead straight into registers;
//  * A (80 channels x 16 frames per k block, converted to dB on the way) is the only LDS traffic: 5 KB per k block,
//    committed at the TOP of an iteration from registers loaded an iteration earlier, so its writes and the reads of the
//    next block's fragments run under the iteration's 80 MFMAs;
//  * a lane's CT column tiles are CT adjacent columns (n = 16 CT wave + CT lj + ct), so the epilogue stores 16-byte
//    vectors, 256 contiguous bytes per row and wave, instead of 64-byte segments.
// the kernel's LDS as one object, so that the body can also run as the first stage of core512_kernel (km_generic.hip)
template <int NW>
struct EncLds {
    __attribute__((aligned(16))) float As[2 * 2 * 4 * 80 * 4];     // two buffers of two k blocks each, [k / 4][row][k % 4] per k block
    __attribute__((aligned(16))) float Ps[2][80 * NW];             // LayerNorm partial sums [pass][row][wave]
    __attribute__((aligned(16))) float Ts[2][80];                  // LayerNorm row totals [pass][row]
};

// window b of the launch; every thread of the workgroup runs the whole body (barriers inside)
template <int NW, int CT, bool FUSE_DB>
__device__ __forceinline__ void encoder_ln_body(const float* __restrict__ xp, const float* __restrict__ wpg,
                                                const float* __restrict__ bias, const float* __restrict__ gam,
                                                const float* __restrict__ bet, float* __restrict__ Y, int KP,
                                                const EncSrc& src, int b, EncLds<NW>& L) {
    constexpr int NKc = 80, NTHR = 64 * NW, D = 16 * CT * NW;
    static_assert(NKc % NW == 0, "rows must divide evenly among the waves");
    static_assert(CT == 2 || CT == 4, "a lane's columns are stored as one 8- or 16-byte vector");
    constexpr int ABUF = 4 * NKc * 4;                 // floats of one A tile image [k / 4][row][k % 4]
    float* As = L.As;
    float (&Ps)[2][NKc * NW] = L.Ps;
    float (&Ts)[2][NKc] = L.Ts;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = lane >> 4, lj = lane & 15;
    const int kt = KP / 16;
    // Tiles are fetched with buffer loads: a thread (or a row, or a k block past the end) with nothing to fetch points
    // past the descriptor's range and gets zeros without a memory access, so no load sits under a branch.
    const float* X = FUSE_DB ? src.melpow + (int64_t)b * src.n_frames * NKc : xp + (int64_t)b * KP * NKc;
    const unsigned x_bytes = (unsigned)((FUSE_DB ? src.n_frames : KP) * NKc * 4);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wpg), 0, (unsigned)(D * KP * 4), 0x00020000);
    constexpr unsigned OOB = 0x7fffffffu;
    // A tiles are staged 32 frames (two k blocks) at a time: 640 float4, one per thread and a second one for the first
    // 640 - NTHR threads (8 waves: waves 0 and 1, so the dB conversion weighs 3 / 3 / 2 / 2 on the four SIMDs per 32 frames
    // where one k block at a time put 2 / 1 / 1 / 1 on them per 16), and ONE barrier per two k blocks.
    constexpr int NA = (640 + NTHR - 1) / NTHR;                        // staged float4 per thread and super-block
    float ref_db = 0.f, floor_db = 0.f;
    if (FUSE_DB) log_window_consts(src.lp, __uint_as_float(src.melmax[b]), ref_db, floor_db);
    u32x4 ra[NA];
    unsigned ravalid = 0;        // which of the staged float4 hold a real frame (zero rows, rows past K stay zero)
    auto load_a = [&](int sb) {  // super-block sb = packed rows 32 sb .. 32 sb + 31
        ravalid = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + NTHR * i, fr = idx / 20, c4 = idx - fr * 20;
            const int r = 32 * sb + fr;                               // packed row of this load
            int f = r < KP ? r : -1;
            if constexpr (FUSE_DB) {                                  // packed row -> frame (mel_log_packed_kernel's mapping)
                f = -1;
                if (r < src.T) f = r < src.n_frames ? r : -1;
                else if (r < src.T + 3) {
                    const int q = r - src.T;
                    if (src.n_frames >= 3) f = src.n_frames - 3 + q; else if (q < src.n_frames) f = q;
                }
            }
            const bool ok = idx < 640 && f >= 0;
            ravalid |= (ok ? 1u : 0u) << i;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)(f * NKc + 4 * c4) * 4u : OOB, 0, 0);
        }
    };
    auto commit_a = [&](int buf) {   // [half = frame / 16][(frame % 16) / 4][channel][frame % 4]: a k block's fragment image per half
        float* Ab = As + buf * (2 * ABUF);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + NTHR * i, fr = idx / 20, c4 = idx - fr * 20;
            asm volatile("" ::"v"(ra[i]));                            // an unconditional use: keeps the load out of the branch below
            if (idx < 640) {
                float4 a4 = make_float4(__uint_as_float(ra[i].x), __uint_as_float(ra[i].y), __uint_as_float(ra[i].z), __uint_as_float(ra[i].w));
                if constexpr (FUSE_DB) {
                    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (src.lp.log_mode == KM_LOG_LN_EPS) a4 = (ravalid >> i) & 1 ? log_four_t<KM_LOG_LN_EPS>(src.lp, a4, ref_db, floor_db) : z4;
                    else a4 = (ravalid >> i) & 1 ? log_four_t<KM_LOG_DB_MAX>(src.lp, a4, ref_db, floor_db) : z4;
                }
                float* dst = &Ab[(fr >> 4) * ABUF + (((fr & 15) >> 2) * NKc + 4 * c4) * 4 + (fr & 3)];
                dst[0] = a4.x; dst[4] = a4.y; dst[8] = a4.z; dst[12] = a4.w;
            }
        }
    };
    f32x4 acc[5][CT];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[i][ct] = f32x4{0, 0, 0, 0};
    f32x4 af[2][5], bf[2][CT];
    auto frags_a = [&](int buf, int half, int slot) {
        const float* Ab = As + buf * (2 * ABUF) + half * ABUF;
#pragma unroll
        for (int i = 0; i < 5; ++i) af[slot][i] = *reinterpret_cast<const f32x4*>(&Ab[(lg * NKc + 16 * i + lj) * 4]);
    };
    const unsigned bo = (unsigned)((wave * CT * 64 + lane) * 16);
    auto load_b = [&](int t, int slot) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            bf[slot][ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                wr, t < kt ? bo + (unsigned)(t * NW * CT + ct) * 1024u : OOB, 0, 0));
    };
    auto mfma_block = [&](int it, int slot) {
        if (it < kt && !(KM_ENC_SKIP & 1)) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[i][ct] = KM_MFMA(af[slot][i][s], bf[slot][ct][s], acc[i][ct]);
        }
    };
    load_a(0);
    load_b(0, 0);
    commit_a(0);
    load_a(1);
    __syncthreads();
    if constexpr (FUSE_DB) {
        if (tid == 0) src.melmax[b] = 0u;   // every thread has read it: a clean slot for the next front-end launch (no memset)
    }
    frags_a(0, 0, 0);
    // Super-iteration sb = k blocks 2 sb (slot 0) and 2 sb + 1 (slot 1).  Top: super-block sb + 1 (in registers since the
    // last super-iteration) -> buffer (sb + 1) & 1, whose last reader -- the fragments of block 2 sb - 1 -- ran before the
    // last barrier; request super-block sb + 2.  Each block's fragments and weight image are read one block ahead: block
    // 2 sb + 1's from this super-block's second half, block 2 sb + 2's from the buffer just committed, after the
    // super-iteration's ONE barrier.
    const int nsb = (kt + 1) / 2;
    for (int sb = 0; sb < nsb; ++sb) {
        const int cur = sb & 1;
        if (!(KM_ENC_SKIP & (8 | 32))) { commit_a(cur ^ 1); load_a(sb + 2); }
        frags_a(cur, 1, 1);
        if (!(KM_ENC_SKIP & (8 | 16))) load_b(2 * sb + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(2 * sb, 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        frags_a(cur ^ 1, 0, 0);
        if (!(KM_ENC_SKIP & (8 | 16))) load_b(2 * sb + 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(2 * sb + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // bias, then LayerNorm over the D columns of each of the 80 rows (this lane: rows 16 i + 4 lg + r, columns n0 + ct)
    const int n0 = 16 * CT * wave + CT * lj;
    {
        float bb[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) bb[ct] = bias[n0 + ct];
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][ct][r] += bb[ct];
    }
    float mean[5][4] = {}, rstd[5][4] = {};
#pragma unroll
    for (int pass = 0; pass < ((KM_ENC_SKIP & 4) ? 0 : 2); ++pass) {
        float* P = Ps[pass];
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = 0.f;
                if (pass == 0) {
#pragma unroll
                    for (int ct = 0; ct < CT; ct += 2) v += acc[i][ct][r] + acc[i][ct + 1][r];
                } else {
#pragma unroll
                    for (int ct = 0; ct < CT; ct += 2) {
                        const float d0 = acc[i][ct][r] - mean[i][r], d1 = acc[i][ct + 1][r] - mean[i][r];
                        v += d0 * d0 + d1 * d1;
                    }
                }
                v = row16_sum(v);
                if (lj == 0) P[(16 * i + 4 * lg + r) * NW + wave] = v;
            }
        __syncthreads();
        // row totals: wave w adds up the NW partials (in wave order) of rows w * RPW .. + RPW - 1, one row per lane
        constexpr int RPW = NKc / NW;
        if (lane < RPW) {
            const int row = wave * RPW + lane;
            float s = 0.f;
            for (int w = 0; w < NW; ++w) s += P[row * NW + w];
            Ts[pass][row] = s;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(&Ts[pass][16 * i + 4 * lg]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (pass == 0) mean[i][r] = t[r] * (1.0f / D);
                else rstd[i][r] = 1.0f / sqrtf(t[r] * (1.0f / D) + 1e-5f);
            }
        }
    }
    float* Yb = Y + (int64_t)b * NKc * D;
    if ((KM_ENC_SKIP & 2) && Yb[0] != 12345.f) return;     // keeps the accumulators live without the store traffic
    float g[CT], be[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { g[ct] = gam[n0 + ct]; be[ct] = bet[n0 + ct]; }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float o[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) o[ct] = (acc[i][ct][r] - mean[i][r]) * rstd[i][r] * g[ct] + be[ct];
            float* dst = Yb + (int64_t)(16 * i + 4 * lg + r) * D + n0;
            if constexpr (CT == 4) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            else *reinterpret_cast<float2*>(dst) = make_float2(o[0], o[1]);
        }
}

template <int NW, int CT, bool FUSE_DB>
__global__ __launch_bounds__(64 * NW) void encoder_ln_kernel(const float* __restrict__ xp, const float* __restrict__ wpg,
                                                             const float* __restrict__ bias, const float* __restrict__ gam,
                                                             const float* __restrict__ bet, float* __restrict__ Y, int KP,
                                                             EncSrc src) {
    __shared__ EncLds<NW> L;
    encoder_ln_body<NW, CT, FUSE_DB>(xp, wpg, bias, gam, bet, Y, KP, src, (int)blockIdx.x, L);
}

}  // namespace km
