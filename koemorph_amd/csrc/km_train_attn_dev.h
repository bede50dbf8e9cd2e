// Fused attention blocks of the phased training step (km_trainp.hip): one (window, head) of the mel stream per 256-thread
// workgroup, forward (scores, softmax, dropout, P V) and backward (dP, dV, softmax', dK, dQ).  A header of its own so that
// tools/micro/attn_train_bench.hip can time a block alone, with stamps.  Included inside namespace km of a translation unit that
// defines f32x4, KM_MFMA and (optionally) KM_TILE_STAMP, after km_device.h.
#pragma once

#ifndef KM_TILE_STAMP
#define KM_TILE_STAMP(i)
#endif

// arguments of an element-wise / attention operation of the training program (a union member of km_trainp.hip's Op)
struct ElemArgs {
    const float *p0, *p1, *p2, *p3, *p4;
    float *q0, *q1, *q2, *q3;
    const unsigned char* mask;
    unsigned char* mask_out;
    int64_t n0, n1;
    int i0, i1, i2, i3;
    float f0;
    unsigned u0, u1;
    LogParams lp;          // OP_LOGPACK
    // OP_REDUCE4: up to four reductions out_j[c] (+)= sum_{y < rS[j]} p_j[y * rn[j] + c], c < rn[j] (p0..p3 -> q0..q3); the
    // partial products of the split-K gradient GEMMs of the phase before
    // (separate fields, read through select chains: an array indexed at run time would move the whole kernel argument into
    // scratch memory -- 4 KB per thread and every phase six times slower)
    int rn0, rn1, rn2, rn3, rS0, rS1, rS2, rS3, racc;      // racc: bit j = reduction j accumulates into its output
};
static_assert(sizeof(ElemArgs) <= 200, "ElemArgs shares a union with GemmArgs (200 bytes): 17 operations + the phase's shared arguments in 4 KB");

// ---- fused attention blocks of the training step, on the matrix pipe ----------------------------------------------------------
// One (window, head) per 256-thread workgroup; every product is v_mfma_f32_16x16x4_f32 tiles fed from LDS images: the 28
// query rows are padded to two 16-row tiles (rows 28..31 zero), the keys to ceil(NK / 16) 16-column tiles (80 keys: five;
// padded keys are zero rows of K / V and get no softmax weight), a head is HD / 16 column tiles.  Operand fetches: lane (g = lane >> 4, j = lane & 15) supplies A[row j][k = 4 s + g] and
// B[k = 4 s + g][col j] of MFMA step s.  A wave that owns a 16-row tile of scores holds all of its keys (five accumulator
// tiles), so softmax and its backward are in-lane passes plus a reduction over the 16 lanes of a row.  Rounds 1 - 2 ran these
// blocks as plain FMA loops reading two LDS values per multiply-add (20 - 28 us for the forward phase, 25 - 50 us for the
// backward phase at 8 - 64 windows); here the products are a few hundred MFMAs per block and the block is bound by its staging.
// (Both versions side by side in one kernel pushed the inlined code over a threshold beyond which the compiler copies the
// 4 KB kernel argument into scratch memory -- every phase five times slower --, so the FMA blocks are gone, not optional.)
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
    return v;
}
constexpr int kAttnMaxKT = 5;          // key tiles of 16: num_mel_channels <= 80 (with room for 128 the score / gradient tiles of a wave
                                       // need 240 registers and every phase of the step drops to one wave per SIMD)
__host__ __device__ constexpr int attn_mfma_lds_floats(int hd, int nk) {
    // backward: Q, dA [32][hd+4], K, V [nkp][hd+4], P, Pd, dS [32][nkp+4]; nkp = keys padded to 16
    return 2 * 32 * (hd + 4) + 2 * ((nk + 15) / 16 * 16) * (hd + 4) + 3 * 32 * ((nk + 15) / 16 * 16 + 4);
}
__host__ __device__ constexpr int attn_mfma_fwd_lds_floats(int hd, int nk) {
    // forward: Q [32][hd+4], K, V [nkp][hd+4], Pd [32][nkp+4]
    return 32 * (hd + 4) + 2 * ((nk + 15) / 16 * 16) * (hd + 4) + 32 * ((nk + 15) / 16 * 16 + 4);
}

template <int HD>
__device__ __forceinline__ void attn_fwd_mfma_dev(const ElemArgs& a, int vb, float* smem) {
    constexpr int hd = HD, QS = hd + 4, CT = hd / 16;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, j = lane & 15;
    const int d = a.i0, NKk = a.i2, H = a.i3, NKT = (NKk + 15) / 16, NKp = 16 * NKT, SS = NKp + 4;
    const int b = vb / H, h = vb - b * H;
    float* Qs = smem;                         // [32][QS], rows 28..31 zero
    float* Ks = Qs + 32 * QS;                 // [NK][QS]
    float* Vs = Ks + NKp * QS;                // [NKp][QS], rows >= NK zero
    float* Ss = Vs + NKp * QS;                // [32][SS]  Pd = P keep / (1 - p), rows 28..31 zero
    const float scale = __uint_as_float(a.u0);
    const float* kvb = a.p1 + (int64_t)b * NKk * 2 * d + h * hd;
    // The dropout bytes of the score elements this lane will own (waves 0 and 1: row 16 wv + 4 g + r, key 16 t + j) are requested
    // FIRST, with everything else of the block.  Round 3 loaded each byte inside the softmax loop, between the stores of P: a
    // byte load may alias any store, so the compiler kept program order -- twenty dependent memory round trips per lane made
    // this block (14 us alone on the chip) the longest operation of the forward pass.
    unsigned char keepb[4][kAttnMaxKT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < kAttnMaxKT; ++t) {
            const int row = 16 * (wv & 1) + 4 * g + r, k = 16 * t + j;
            const int64_t pi = (((int64_t)b * H + h) * 28 + (row < 28 ? row : 27)) * NKk + (k < NKk ? k : 0);
            keepb[r][t] = 1;
            if (a.mask) keepb[r][t] = a.mask[pi];
        }
    // Staging: Q (32 rows, 28 real), K and V (NKp rows, NK real) are consecutive [row][QS] images, so a float4 unit of any of
    // them is (row, c4) -> smem + row * QS + 4 c4.  ALL of a thread's loads are issued before its first LDS store: one memory
    // round trip for the block (three dependent ones made the block as slow as its plain-FMA predecessor).  Padding rows load
    // a real row and store zeros.
    {
        constexpr int C4 = hd / 4, MAXU = ((32 + 2 * 16 * kAttnMaxKT) * C4 + 255) / 256;
        const int total = (32 + 2 * NKp) * C4;
        float4 rv[MAXU];
        unsigned livem = 0;
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const int u = tid + 256 * i, uc = u < total ? u : 0;
            const int row = uc / C4, c4 = uc - row * C4;
            const int kk = row < 32 ? 0 : (row - 32 < NKp ? row - 32 : row - 32 - NKp);
            const bool live = row < 32 ? row < 28 : kk < NKk;
            const float* src = row < 32 ? a.p0 + (int64_t)(live ? row : 0) * d + h * hd
                                        : kvb + (int64_t)(live ? kk : 0) * 2 * d + (row - 32 < NKp ? 0 : d);
            rv[i] = *reinterpret_cast<const float4*>(src + 4 * c4);
            livem |= (live && u < total ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const int u = tid + 256 * i;
            if (u < total) {
                const int row = u / C4, c4 = u - row * C4;
                *reinterpret_cast<float4*>(smem + row * QS + 4 * c4) = (livem >> i) & 1 ? rv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    __syncthreads();
    if (wv < 2) {                             // query rows 16 wv .. 16 wv + 15 against every key
        f32x4 S[kAttnMaxKT];
#pragma unroll
        for (int t = 0; t < kAttnMaxKT; ++t) S[t] = f32x4{0, 0, 0, 0};
#pragma unroll 2
        for (int sk = 0; sk < hd / 4; ++sk) {
            const float av = Qs[(16 * wv + j) * QS + 4 * sk + g];
#pragma unroll
            for (int t = 0; t < kAttnMaxKT; ++t)
                if (t < NKT) S[t] = KM_MFMA(av, Ks[(16 * t + j) * QS + 4 * sk + g], S[t]);
        }
        // S[t][r] = score of (row 16 wv + 4 g + r, key 16 t + j): softmax over the keys = over t in the lane and over the 16 lanes j
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * wv + 4 * g + r;
            float m = -INFINITY;
#pragma unroll
            for (int t = 0; t < kAttnMaxKT; ++t)
                if (t < NKT) { S[t][r] = 16 * t + j < NKk ? S[t][r] * scale : -INFINITY; m = fmaxf(m, S[t][r]); }
            m = row16_max(m);
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < kAttnMaxKT; ++t)
                if (t < NKT) { S[t][r] = expf(S[t][r] - m); sum += S[t][r]; }
            sum = row16_sum(sum);
            const int64_t prow = (((int64_t)b * H + h) * 28 + row) * NKk;
#pragma unroll
            for (int t = 0; t < kAttnMaxKT; ++t)
                if (t < NKT) {
                    const int k = 16 * t + j;
                    const float pv = S[t][r] / sum;
                    const bool live = row < 28 && k < NKk;
                    const float pd = a.mask ? (keepb[r][t] ? pv * a.f0 : 0.f) : pv;
                    if (live) a.q0[prow + k] = pv;
                    Ss[row * SS + k] = live ? pd : 0.f;
                }
        }
    }
    __syncthreads();
    for (int tile = wv; tile < 2 * CT; tile += 4) {          // A_h = Pd V_h: 2 x CT tiles of 16 x 16, contraction over the keys
        const int rt = tile / CT, ct = tile - rt * CT;
        f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll 2
        for (int sk = 0; sk < NKp / 4; ++sk)
            acc = KM_MFMA(Ss[(16 * rt + j) * SS + 4 * sk + g], Vs[(4 * sk + g) * QS + 16 * ct + j], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * rt + 4 * g + r;
            if (row < 28) a.q1[((int64_t)b * 28 + row) * d + h * hd + 16 * ct + j] = acc[r];
        }
    }
    __syncthreads();
}

// ---- forward block for heads of 32 columns, staged by LDS-DMA (round 4) ------------------------------------------------------
// The block above moves Q, K, V through registers into padded [row][36] images and feeds every MFMA from two ds_read_b32.
// Here the three operands land by buffer_load ... lds in the swizzled images of km_gemm_dma_dev.h (Q and K k-contiguous:
// [row][8 chunks], chunk c of row r at c ^ ((r >> 1) & 7); V with its columns contiguous: [key][32], columns 16 .. 31 swapped
// with 0 .. 15 for keys whose (key >> 2) is odd), so a fragment of four consecutive head dimensions is ONE ds_read_b128 per
// lane (lane group g supplies k = 16 kb + 4 g + s to MFMA s on both sides) and nothing passes through registers on the way in.
// Pd = P keep / (1 - p) is written by the two softmax waves into the same k-contiguous image form over the keys (three
// k-tiles of 32, keys beyond num_mel_channels zero) and is the A operand of P V; V's fragments are four ds_read_b32 at
// compile-time offsets.  Arithmetic: the same MFMAs on the same operand values, with the 16 head dimensions of a k block dealt to
// the lane groups in another order, and the softmax's exp / division on the hardware's v_exp_f32 / v_rcp_f32 -- P within 1e-6
// of the block above.
constexpr int kAttnDmaFwdLdsFloats = 32 * 32 + 2 * 96 * 32 + 3 * 32 * 32;      // Q, K, V (96 rows each: three DMA instructions), Pd
typedef __attribute__((address_space(3))) void* km_attn_lds_ptr;

// NKT: key tiles of 16 (compile time: a run-time count predicates every one of the five score tiles of a lane -- the softmax
// stretch of the block was ~1350 instructions per lane, on one wave per SIMD, for that reason and for its per-element 64-bit
// addresses and branches)
template <int NKT>
__device__ __forceinline__ void attn_fwd_dma32_dev(const ElemArgs& a, int vb, float* smem) {
    constexpr int hd = 32;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, j = lane & 15;
    const int d = a.i0, NKk = a.i2, H = a.i3;
    const int b = vb / H, h = vb - b * H;
    float* Qi = smem;                  // [32][32]   rows 28 .. 31 zero (out of the descriptor's range)
    float* Ki = Qi + 32 * 32;          // [96][32]   rows >= NK zero
    float* Vi = Ki + 96 * 32;          // [96][32]   rows >= NK zero
    float* Pi = Vi + 96 * 32;          // [3][32][32] Pd by k-tiles of 32 keys
    const float scale = __uint_as_float(a.u0);
    const unsigned OOB = 0x80000000u;
    // All FOUR waves compute scores: wave w the row tile rt = w & 1 (the two waves of a row tile run the same 8 NKT MFMAs -- half a
    // microsecond) and then the softmax of HALF of its lanes' rows, r = 2 (w >> 1) and 2 (w >> 1) + 1.
    const int rt = wv & 1, rh = wv >> 1;
    {   // staging: 1 + 3 + 3 DMA instructions per thread, requested before anything else
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.p0), 0, (unsigned)(28 * d * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.p1 + (int64_t)b * NKk * 2 * d), 0,
                                                                             (unsigned)(NKk * 2 * d * 4), 0x00020000);
        {
            const int row = tid >> 3, kg = (tid & 7) ^ ((row >> 1) & 7);
            const unsigned o = row < 28 ? (unsigned)((row * d + h * hd + 4 * kg) * 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (km_attn_lds_ptr)(Qi + 64 * wv * 4), 16, o, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int s = i * 256 + tid, key = s >> 3;
            const int kg = (s & 7) ^ ((key >> 1) & 7);                       // K: k-contiguous image
            const int grp = (s & 7) ^ (4 * ((key >> 2) & 1));                // V: column-contiguous image
            const unsigned ok = key < NKk ? (unsigned)((key * 2 * d + h * hd + 4 * kg) * 4) : OOB;
            const unsigned ov = key < NKk ? (unsigned)((key * 2 * d + d + h * hd + 4 * grp) * 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (km_attn_lds_ptr)(Ki + (i * 256 + 64 * wv) * 4), 16, ok, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (km_attn_lds_ptr)(Vi + (i * 256 + 64 * wv) * 4), 16, ov, 0, 0, 0);
        }
        // keys 64 .. 95 of Pd: what the softmax waves do not write (keys >= 16 NKT) must be zero
        reinterpret_cast<float4*>(Pi + 2 * 32 * 32)[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // what multiplies P on its way to Pd, per score element this lane will finish (rows 16 rt + 4 g + 2 rh + {0, 1}, keys
    // 16 t + j): keep / (1 - p) from the dropout bytes, 0 for the padding rows 28 .. 31 -- requested with the operands
    float kf[2][NKT];
    {
        const unsigned char* mh = a.mask ? a.mask + (int64_t)(b * H + h) * 28 * NKk : nullptr;
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
            for (int t = 0; t < NKT; ++t) {
                const int row = 16 * rt + 4 * g + 2 * rh + r2, k = 16 * t + j;
                unsigned char kb = 1;
                if (mh) kb = mh[(row < 28 ? row : 27) * NKk + (k < NKk ? k : 0)];
                kf[r2][t] = row < 28 ? (mh ? (kb ? a.f0 : 0.f) : 1.f) : 0.f;
            }
    }
    KM_TILE_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    KM_TILE_STAMP(2);
    {
        f32x4 S[NKT];
#pragma unroll
        for (int t = 0; t < NKT; ++t) S[t] = f32x4{0, 0, 0, 0};
        const int qrow = 16 * rt + j, qsw = (qrow >> 1) & 7, ksw = (j >> 1) & 7;       // (key >> 1) & 7 with key = 16 t + j
        const float* qp = Qi + qrow * 32;
        const float* kp = Ki + j * 32;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const f32x4 aq = *reinterpret_cast<const f32x4*>(qp + (((4 * kb + g) ^ qsw) << 2));
            const int ko = ((4 * kb + g) ^ ksw) << 2;
#pragma unroll
            for (int t = 0; t < NKT; ++t) {
                const f32x4 bk = *reinterpret_cast<const f32x4*>(kp + ko + t * 16 * 32);
#pragma unroll
                for (int q = 0; q < 4; ++q) S[t] = KM_MFMA(aq[q], bk[q], S[t]);
            }
        }
        KM_TILE_STAMP(4);
        // S[t][r] = score of (row 16 rt + 4 g + r, key 16 t + j): softmax over the keys = over t in the lane and over the 16 lanes j.
        // exp through v_exp_f32 (exp2 of the scaled difference) and one v_rcp_f32 per row, as the inference core does
        // (DESIGN 3.1): ~1 ulp each, P within 3e-7 relative of expf / division.  P leaves through a buffer descriptor over the
        // head's (28, NK) block (rows 28 .. 31 lie beyond it: dropped), Pd into the k-tiled image.
        const float sl2 = scale * 1.44269504088896341f;
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(a.q0 + (int64_t)(b * H + h) * 28 * NKk, 0, (unsigned)(28 * NKk * 4), 0x00020000);
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const int r = 2 * rh + r2, row = 16 * rt + 4 * g + r;
            float m = -INFINITY;
#pragma unroll
            for (int t = 0; t < NKT; ++t) {
                S[t][r] = (t + 1 < NKT || 16 * t + j < NKk) ? S[t][r] * sl2 : -INFINITY;       // only the last tile can hold keys past NK
                m = fmaxf(m, S[t][r]);
            }
            m = row16_max(m);
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NKT; ++t) { S[t][r] = __builtin_amdgcn_exp2f(S[t][r] - m); sum += S[t][r]; }
            sum = row16_sum(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);
            // image position of (row, key 16 t + j): k-tile t >> 1, chunk (4 (t & 1) + (j >> 2)) ^ psw, element j & 3
            const int psw = (row >> 1) & 7;
            const int pbase = row * 32 + ((((j >> 2) ^ (psw & 3))) << 2) + (j & 3), pflip = (psw >> 2) & 1;
            const unsigned gbase = (unsigned)((row * NKk + j) * 4);
#pragma unroll
            for (int t = 0; t < NKT; ++t) {
                const float pv = S[t][r] * inv;
                const unsigned go = (t + 1 < NKT || 16 * t + j < NKk) ? gbase + (unsigned)(64 * t) : OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pv), rp, go, 0, 0);
                Pi[pbase + (t >> 1) * 1024 + 16 * ((t & 1) ^ pflip)] = pv * kf[r2][t];
            }
        }
    }
    __syncthreads();
    KM_TILE_STAMP(3);
    {   // A_h = Pd V_h: output tile (ot, ct) = (wv >> 1, wv & 1), contraction over the keys in blocks of 16
        const int ot = wv >> 1, ct = wv & 1;
        const int prow = 16 * ot + j, psw = (prow >> 1) & 7;
        const int vcol = (16 * ct + j) ^ (16 * (g & 1));
        f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < NKT; ++kb) {
            const f32x4 ap = *reinterpret_cast<const f32x4*>(Pi + (kb >> 1) * 1024 + prow * 32 + (((4 * (kb & 1) + g) ^ psw) << 2));
            const float* vp = Vi + (16 * kb + 4 * g) * 32 + vcol;
            acc = KM_MFMA(ap[0], vp[0], acc);
            acc = KM_MFMA(ap[1], vp[32], acc);
            acc = KM_MFMA(ap[2], vp[64], acc);
            acc = KM_MFMA(ap[3], vp[96], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * ot + 4 * g + r;
            if (row < 28) a.q1[((int64_t)b * 28 + row) * d + h * hd + 16 * ct + j] = acc[r];
        }
    }
    __syncthreads();
}

template <int HD>
__device__ __forceinline__ void attn_bwd_mfma_dev(const ElemArgs& a, int vb, float* smem) {
    constexpr int hd = HD, QS = hd + 4, CT = hd / 16;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, j = lane & 15;
    const int d = a.i0, NKk = a.i2, H = a.i3, NKT = (NKk + 15) / 16, NKp = 16 * NKT, SS = NKp + 4;
    const int b = vb / H, h = vb - b * H;
    float* Qs = smem;                         // [32][QS]  rows 28..31 zero
    float* Gs = Qs + 32 * QS;                 // [32][QS]  dA_h, rows 28..31 zero
    float* Ks = Gs + 32 * QS;                 // [NK][QS]
    float* Vs = Ks + NKp * QS;                // [NKp][QS], rows >= NK zero
    float* Ps = Vs + NKp * QS;                // [32][SS]  P, zero beyond row 27 / key NK - 1
    float* Ds = Ps + 32 * SS;                 // [32][SS]  Pd = P keep / (1 - p), rows 28..31 zero
    float* Es = Ds + 32 * SS;                 // [32][SS]  dS, rows 28..31 zero
    const float scale = __uint_as_float(a.u0);
    const int64_t prow0 = ((int64_t)b * H + h) * 28 * NKk;
    const float* kvb = a.p1 + (int64_t)b * NKk * 2 * d + h * hd;
    const float* gab = a.p3 + (int64_t)b * 28 * d + h * hd;
    // dropout bytes of the lane's dP elements (waves 0 and 1), requested up front: see attn_fwd_mfma_dev
    unsigned char keepb[4][kAttnMaxKT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < kAttnMaxKT; ++t) {
            const int row = 16 * (wv & 1) + 4 * g + r, k = 16 * t + j;
            keepb[r][t] = 1;
            if (a.mask) keepb[r][t] = a.mask[prow0 + (int64_t)(row < 28 ? row : 27) * NKk + (k < NKk ? k : 0)];
        }
    // (rows 28..31: a real row is loaded and a zero selected -- a select between a load and a constant is compiled as a select
    // of ADDRESSES with the constant in scratch memory, which drags the whole kernel argument there)
    // Staging as in the forward block: Q, dA (32 rows each), K, V (NKp rows) are consecutive [row][QS] images; P and
    // Pd = P keep / (1 - p) are [32][SS] images made from the same loads of P (+ the keep bytes); everything is requested
    // before the first LDS store.
    {
        constexpr int C4 = hd / 4, MAXU = ((64 + 2 * 16 * kAttnMaxKT) * C4 + 255) / 256, MAXP = (32 * 4 * kAttnMaxKT + 255) / 256;
        const int total = (64 + 2 * NKp) * C4, P4 = NKp / 4, totalp = 32 * P4;
        float4 rv[MAXU], pvv[MAXP];
        unsigned keepw[MAXP];
        unsigned livem = 0, livep = 0;
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const int u = tid + 256 * i, uc = u < total ? u : 0;
            const int row = uc / C4, c4 = uc - row * C4;
            const int qq = row < 32 ? row : row - 32;
            const int kk = row < 64 ? 0 : (row - 64 < NKp ? row - 64 : row - 64 - NKp);
            const bool live = row < 64 ? qq < 28 : kk < NKk;
            const float* src = row < 32 ? a.p0 + (int64_t)(live ? qq : 0) * d + h * hd
                             : row < 64 ? gab + (int64_t)(live ? qq : 0) * d
                                        : kvb + (int64_t)(live ? kk : 0) * 2 * d + (row - 64 < NKp ? 0 : d);
            rv[i] = *reinterpret_cast<const float4*>(src + 4 * c4);
            livem |= (live && u < total ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int u = tid + 256 * i, uc = u < totalp ? u : 0;
            const int q = uc / P4, k4 = uc - q * P4;
            const bool live = u < totalp && q < 28 && 4 * k4 < NKk;          // NK is a multiple of 4: a float4 of keys is all real or all padding
            const int64_t off = prow0 + (live ? q * NKk + 4 * k4 : 0);
            pvv[i] = *reinterpret_cast<const float4*>(a.p2 + off);
            keepw[i] = a.mask ? *reinterpret_cast<const unsigned*>(a.mask + off) : 0x01010101u;
            livep |= (live ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const int u = tid + 256 * i;
            if (u < total) {
                const int row = u / C4, c4 = u - row * C4;
                *reinterpret_cast<float4*>(smem + row * QS + 4 * c4) = (livem >> i) & 1 ? rv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int u = tid + 256 * i;
            if (u < totalp) {
                const int q = u / P4, k4 = u - q * P4;
                const bool live = (livep >> i) & 1;
                const float4 pz = live ? pvv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                const unsigned kw = keepw[i];
                const float sc = a.mask ? a.f0 : 1.0f;
                *reinterpret_cast<float4*>(Ps + q * SS + 4 * k4) = pz;
                *reinterpret_cast<float4*>(Ds + q * SS + 4 * k4) =
                    make_float4((kw & 0xffu) ? pz.x * sc : 0.f, (kw & 0xff00u) ? pz.y * sc : 0.f, (kw & 0xff0000u) ? pz.z * sc : 0.f,
                                (kw & 0xff000000u) ? pz.w * sc : 0.f);
            }
        }
    }
    __syncthreads();
    if (wv < 2) {
        // dP = (dA_h V_h^T) keep / (1 - p) and dS = P (dP - sum_k dP P) for the query rows 16 wv .. + 15 (all keys in this wave)
        f32x4 G[kAttnMaxKT];
#pragma unroll
        for (int t = 0; t < kAttnMaxKT; ++t) G[t] = f32x4{0, 0, 0, 0};
#pragma unroll 2
        for (int sk = 0; sk < hd / 4; ++sk) {
            const float av = Gs[(16 * wv + j) * QS + 4 * sk + g];
#pragma unroll
            for (int t = 0; t < kAttnMaxKT; ++t)
                if (t < NKT) G[t] = KM_MFMA(av, Vs[(16 * t + j) * QS + 4 * sk + g], G[t]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * wv + 4 * g + r;
            float pv[kAttnMaxKT], sacc = 0.f;
#pragma unroll
            for (int t = 0; t < kAttnMaxKT; ++t)
                if (t < NKT) {
                    const int k = 16 * t + j;
                    pv[t] = Ps[row * SS + k];
                    float gv = G[t][r];
                    if (a.mask) gv = (row < 28 && k < NKk && keepb[r][t]) ? gv * a.f0 : 0.f;
                    G[t][r] = gv;
                    sacc += gv * pv[t];
                }
            sacc = row16_sum(sacc);
#pragma unroll
            for (int t = 0; t < kAttnMaxKT; ++t)
                if (t < NKT) Es[row * SS + 16 * t + j] = row < 28 ? pv[t] * (G[t][r] - sacc) : 0.f;
        }
    } else {
        // dV_h = Pd^T dA_h: NKT x CT tiles, contraction over the (padded) query rows; waves 2 and 3 share the key tiles
        for (int tile = wv - 2; tile < NKT * CT; tile += 2) {
            const int kt = tile / CT, ct = tile - kt * CT;
            f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll 2
            for (int sk = 0; sk < 8; ++sk)
                acc = KM_MFMA(Ds[(4 * sk + g) * SS + 16 * kt + j], Gs[(4 * sk + g) * QS + 16 * ct + j], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * kt + 4 * g + r < NKk) a.q0[((int64_t)b * NKk + 16 * kt + 4 * g + r) * 2 * d + d + h * hd + 16 * ct + j] = acc[r];
        }
    }
    __syncthreads();
    if (wv < 2) {
        // dK_h = scale dS^T Q_h: NKT x CT tiles over the query rows; waves 0 and 1 share the key tiles
        for (int tile = wv; tile < NKT * CT; tile += 2) {
            const int kt = tile / CT, ct = tile - kt * CT;
            f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll 2
            for (int sk = 0; sk < 8; ++sk)
                acc = KM_MFMA(Es[(4 * sk + g) * SS + 16 * kt + j], Qs[(4 * sk + g) * QS + 16 * ct + j], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * kt + 4 * g + r < NKk) a.q0[((int64_t)b * NKk + 16 * kt + 4 * g + r) * 2 * d + h * hd + 16 * ct + j] = scale * acc[r];
        }
    } else {
        // dQ_h of this window = scale dS K_h: 2 x CT tiles over the keys; waves 2 and 3 share them
        for (int tile = wv - 2; tile < 2 * CT; tile += 2) {
            const int rt = tile / CT, ct = tile - rt * CT;
            f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll 2
            for (int sk = 0; sk < NKp / 4; ++sk)
                acc = KM_MFMA(Es[(16 * rt + j) * SS + 4 * sk + g], Ks[(4 * sk + g) * QS + 16 * ct + j], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rt + 4 * g + r;
                if (row < 28) a.q1[((int64_t)b * 28 + row) * d + h * hd + 16 * ct + j] = scale * acc[r];
            }
        }
    }
    __syncthreads();
}

// ---- backward block for heads of 32 columns on LDS-DMA staged images (round 4) ------------------------------------------------
// Same recipe as attn_fwd_dma32_dev.  Operands and the image form each product wants (MFMA: out[M][N] += A[M][k] B[k][N];
// "k-contiguous" = the swizzled [row][8 chunks] image read with one ds_read_b128 per fragment, "row-contiguous" = [k][rows]
// read with four ds_read_b32 at compile-time offsets):
//   dP[q][key]  = dA V^T        A = dA k-contiguous, B = V k-contiguous                       (all four waves, as the scores)
//   dS          = P (dP' - sum_key dP' P), dP' = dP keep / (1 - p): in registers; P and the dropout bytes of a lane's ten
//                 elements come straight from memory into registers, requested with the operands
//   dV[key][c]  = Pd^T dA       A = Pd^T: the [q][84] image of Pd read row-contiguous, B = dA row-contiguous
//   dK[key][c]  = scale dS^T Q  A = dS^T: the [q][84] image of dS, B = Q row-contiguous
//   dQ[q][c]    = scale dS K    A = dS k-contiguous (the SAME [q][84] image: a row stride of 84 floats is conflict-free for
//                 both read forms), B = K row-contiguous
// dA is staged twice (one image per form: 4 KB each).  53 KB of LDS against 64.5 for the register-staged block.
constexpr int kAttnDmaBwdLdsFloats = 3 * 32 * 32 + 2 * 80 * 32 + 2 * 32 * 84;
template <int NKT>
__device__ __forceinline__ void attn_bwd_dma32_dev(const ElemArgs& a, int vb, float* smem) {
    constexpr int hd = 32, PS = 84;
    static_assert(NKT == 5, "the images below hold 80 keys");
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, j = lane & 15;
    const int d = a.i0, NKk = a.i2, H = a.i3;
    const int b = vb / H, h = vb - b * H;
    float* G0 = smem;                  // dA, k-contiguous   [32][32]  rows 28 .. 31 zero
    float* G1 = G0 + 32 * 32;          // dA, row-contiguous [32 q][32 cols]
    float* Q1 = G1 + 32 * 32;          // Q,  row-contiguous [32 q][32 cols]
    float* V0 = Q1 + 32 * 32;          // V,  k-contiguous   [80][32]
    float* K1 = V0 + 80 * 32;          // K,  row-contiguous [80 keys][32 cols]
    float* Dp = K1 + 80 * 32;          // Pd [32 q][84]
    float* Es = Dp + 32 * PS;          // dS [32 q][84]
    const float scale = __uint_as_float(a.u0);
    const unsigned OOB = 0x80000000u;
    const int rt = wv & 1, rh = wv >> 1;
    {
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.p0), 0, (unsigned)(28 * d * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.p3 + (int64_t)b * 28 * d), 0, (unsigned)(28 * d * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.p1 + (int64_t)b * NKk * 2 * d), 0,
                                                                             (unsigned)(NKk * 2 * d * 4), 0x00020000);
        {
            const int row = tid >> 3, cp = tid & 7;
            const int kg = cp ^ ((row >> 1) & 7);                           // k-contiguous image: chunk
            const int grp = cp ^ (4 * ((row >> 2) & 1));                    // row-contiguous image [k = row][cols]: column group
            const unsigned ok = row < 28 ? 0u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (km_attn_lds_ptr)(G0 + 64 * wv * 4), 16, ok | (unsigned)((row * d + h * hd + 4 * kg) * 4), 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (km_attn_lds_ptr)(G1 + 64 * wv * 4), 16, ok | (unsigned)((row * d + h * hd + 4 * grp) * 4), 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (km_attn_lds_ptr)(Q1 + 64 * wv * 4), 16, ok | (unsigned)((row * d + h * hd + 4 * grp) * 4), 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i < 2 || wv < 2) {                                           // keys 64 .. 79: waves 0 and 1 (an image holds 80 rows)
                const int s = i * 256 + tid, key = s >> 3;
                const int kg = (s & 7) ^ ((key >> 1) & 7);
                const int grp = (s & 7) ^ (4 * ((key >> 2) & 1));
                const unsigned ov = key < NKk ? (unsigned)((key * 2 * d + d + h * hd + 4 * kg) * 4) : OOB;
                const unsigned ok = key < NKk ? (unsigned)((key * 2 * d + h * hd + 4 * grp) * 4) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (km_attn_lds_ptr)(V0 + (i * 256 + 64 * wv) * 4), 16, ov, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (km_attn_lds_ptr)(K1 + (i * 256 + 64 * wv) * 4), 16, ok, 0, 0, 0);
            }
        }
    }
    // P and keep / (1 - p) of the ten elements this lane will finish (rows 16 rt + 4 g + 2 rh + {0, 1}, keys 16 t + j): rows
    // 28 .. 31 lie beyond the descriptor's range (P = 0), keys past NK too
    float pe[2][NKT], kf[2][NKT];
    {
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.p2 + (int64_t)(b * H + h) * 28 * NKk), 0,
                                                                             (unsigned)(28 * NKk * 4), 0x00020000);
        const unsigned char* mh = a.mask ? a.mask + (int64_t)(b * H + h) * 28 * NKk : nullptr;
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
            for (int t = 0; t < NKT; ++t) {
                const int row = 16 * rt + 4 * g + 2 * rh + r2, k = 16 * t + j;
                const unsigned o = (row < 28 && k < NKk) ? (unsigned)((row * NKk + k) * 4) : OOB;
                pe[r2][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rp, o, 0, 0));
                unsigned char kb = 1;
                if (mh) kb = mh[(row < 28 ? row : 27) * NKk + (k < NKk ? k : 0)];
                kf[r2][t] = (row < 28 && k < NKk) ? (mh ? (kb ? a.f0 : 0.f) : 1.f) : 0.f;
            }
    }
    KM_TILE_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    KM_TILE_STAMP(2);
    {   // dP for the row tile rt (both waves of a row tile), then dS and Pd of this wave's two rows per lane
        f32x4 Gp[NKT];
#pragma unroll
        for (int t = 0; t < NKT; ++t) Gp[t] = f32x4{0, 0, 0, 0};
        const int arow = 16 * rt + j, asw = (arow >> 1) & 7, ksw = (j >> 1) & 7;
        const float* ap = G0 + arow * 32;
        const float* vp = V0 + j * 32;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(ap + (((4 * kb + g) ^ asw) << 2));
            const int vo = ((4 * kb + g) ^ ksw) << 2;
#pragma unroll
            for (int t = 0; t < NKT; ++t) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(vp + vo + t * 16 * 32);
#pragma unroll
                for (int q = 0; q < 4; ++q) Gp[t] = KM_MFMA(av[q], bv[q], Gp[t]);
            }
        }
        KM_TILE_STAMP(4);
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const int r = 2 * rh + r2, row = 16 * rt + 4 * g + r;
            float gv[NKT], sacc = 0.f;
#pragma unroll
            for (int t = 0; t < NKT; ++t) { gv[t] = Gp[t][r] * kf[r2][t]; sacc += gv[t] * pe[r2][t]; }
            sacc = row16_sum(sacc);
#pragma unroll
            for (int t = 0; t < NKT; ++t) {
                Es[row * PS + 16 * t + j] = pe[r2][t] * (gv[t] - sacc);
                Dp[row * PS + 16 * t + j] = pe[r2][t] * kf[r2][t];
            }
        }
    }
    __syncthreads();
    KM_TILE_STAMP(3);
    // dV, dK: ten 16 x 16 tiles each (key tile kt, column tile ct), contraction over the 32 queries; dQ: four tiles over the keys.
    // Tiles are dealt to the waves round robin: 3 + 3 (or 2 + 2) + 1.
    float* dkv = a.q0 + (int64_t)b * NKk * 2 * d + h * hd;
    const int cb = 16 * (g & 1);                                              // the row-contiguous images' column swap for this lane group
#pragma unroll
    for (int ti = 0; ti < 3; ++ti) {
        const int tile = wv + 4 * ti;
        if (tile < 2 * NKT) {
            const int kt = tile >> 1, ct = tile & 1;
            f32x4 accv = f32x4{0, 0, 0, 0}, acck = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int q0 = 16 * kb + 4 * g;                                  // queries q0 .. q0 + 3 in MFMAs 0 .. 3
                const float* pd = Dp + q0 * PS + 16 * kt + j;
                const float* es = Es + q0 * PS + 16 * kt + j;
                const float* ga = G1 + q0 * 32 + ((16 * ct + j) ^ cb);
                const float* qa = Q1 + q0 * 32 + ((16 * ct + j) ^ cb);
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    accv = KM_MFMA(pd[s_ * PS], ga[s_ * 32], accv);
                    acck = KM_MFMA(es[s_ * PS], qa[s_ * 32], acck);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * g + r;
                if (key < NKk) {
                    dkv[(int64_t)key * 2 * d + d + 16 * ct + j] = accv[r];
                    dkv[(int64_t)key * 2 * d + 16 * ct + j] = scale * acck[r];
                }
            }
        }
    }
    {   // dQ tile (ot, ct) = (wv >> 1, wv & 1): contraction over the keys in blocks of 16
        const int ot = wv >> 1, ct = wv & 1;
        const float* er = Es + (16 * ot + j) * PS + 4 * g;
        const float* kc = K1 + 4 * g * 32 + ((16 * ct + j) ^ cb);
        f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < NKT; ++kb) {
            const f32x4 ev = *reinterpret_cast<const f32x4*>(er + 16 * kb);
            const float* kp = kc + 16 * kb * 32;
            acc = KM_MFMA(ev[0], kp[0], acc);
            acc = KM_MFMA(ev[1], kp[32], acc);
            acc = KM_MFMA(ev[2], kp[64], acc);
            acc = KM_MFMA(ev[3], kp[96], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * ot + 4 * g + r;
            if (row < 28) a.q1[((int64_t)b * 28 + row) * d + h * hd + 16 * ct + j] = scale * acc[r];
        }
    }
    __syncthreads();
}
