// Device helpers shared by km_mel.hip and km_core.hip.
#pragma once

#include <hip/hip_runtime.h>

#include "koemorph.h"

namespace km {

struct LogParams {
    int log_mode;
    float amin, top_db, db_add, db_scale, log_eps;
};

// 10 log10(max(amin, s)) and ln(x) on the hardware log2 (v_log_f32, 1 ulp): |error| < 2e-5 dB over [amin, 1e10], i.e.
// < 3e-7 after the 1/80 normalisation.  Operands are >= amin / eps (never denormal), so the library wrappers' denormal
// scaling, compensated constant multiply and inf/nan selects (25 instructions per value) buy nothing here; the clamps
// are v_med3_f32 (no canonicalising v_max pair).
__device__ __forceinline__ float clamp_lo(float x, float lo) { return __builtin_amdgcn_fmed3f(x, lo, __builtin_inff()); }
__device__ __forceinline__ float db10(float s, float amin) {
    return 3.01029995663981195f * __builtin_amdgcn_logf(clamp_lo(s, amin));
}

// per-window constants of librosa.power_to_db(ref=np.max): reference level and the top_db floor
__device__ __forceinline__ void log_window_consts(const LogParams& p, float ref, float& ref_db, float& floor_db) {
    ref_db = db10(ref, p.amin);
    // log_spec.max() - top_db: the window maximum is its own reference, so max_db = f(ref) - ref_db
    floor_db = (db10(ref, p.amin) - ref_db) - p.top_db;
}

// the part of librosa.power_to_db(ref=np.max) + affine that needs the window's reference, applied to a stored 10 log10(max(amin, s))
// (the training step's front end stores that; its readers finish the conversion: MelArgs::pack_*, DbXform)
__device__ __forceinline__ float db_finish(const LogParams& p, float dbs, float ref_db, float floor_db) {
    float v = dbs - ref_db;
    v = clamp_lo(v, floor_db);
    return (v + p.db_add) * p.db_scale;
}
// The same in two instructions, for top_db == db_add (the floor then maps to exactly 0): max(t, -A) + A == max(t + A, 0) and the
// positive scale commutes with the max, so x = max(dbs * scale + (A - ref_db) * scale, 0) -- one fma + one max per value, within
// an ulp or two of db_finish (the front end's own log is good to 3e-7 of the feature range).  c1 = db_fast_c1().
__device__ __forceinline__ float db_fast_c1(const LogParams& p, float ref_db) { return (p.db_add - ref_db) * p.db_scale; }
__device__ __forceinline__ float db_finish_fast(float dbs, float scale, float c1) { return fmaxf(fmaf(dbs, scale, c1), 0.f); }
// operand transform of a product whose A rows hold 10 log10(power): batch entry z uses the reference ref_bits[z] (float bits)
struct DbXform { LogParams lp; const unsigned* ref_bits; };

// LayerNorm applied by the READER of a product's output rows (km_trainp.hip, round 4): the producing tiles leave, per row and per 32
// columns, the mean of those 32 values and their sum of squared deviations from it (`stats`: float2 [row][npart], npart = d / 32);
// the reader combines the parts (Chan et al.: exact pooling of means and M2, no E[x^2] - mean^2 cancellation), and takes every value
// through ln_apply on its way to the MFMAs -- or, OP_LNAPPLY, into the stored Y.  Both use these two functions: the same bits.
struct LnXform { const float* stats; const float* gamma; const float* beta; int npart; float eps; };
__device__ __forceinline__ void ln_combine(const float2* __restrict__ st, int npart, float eps, float& rs, float& nmurs, float* mean_out = nullptr) {
    float msum = 0.f;
    for (int p = 0; p < npart; ++p) msum += st[p].x;
    const float mean = msum / (float)npart;
    float m2 = 0.f;
    for (int p = 0; p < npart; ++p) { const float dm = st[p].x - mean; m2 += st[p].y + 32.0f * dm * dm; }
    rs = 1.0f / sqrtf(m2 / (float)(32 * npart) + eps);
    nmurs = -mean * rs;
    if (mean_out) *mean_out = mean;
}
// the same from parts held in registers (up to 8: d <= 256); terms beyond npart add an exact zero: the same bits as ln_combine
__device__ __forceinline__ void ln_combine8(const float2 (&st)[8], int npart, float eps, float& rs, float& nmurs) {
    float msum = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) msum += p < npart ? st[p].x : 0.f;
    const float mean = msum / (float)npart;
    float m2 = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) { const float dm = st[p].x - mean; m2 += p < npart ? st[p].y + 32.0f * dm * dm : 0.f; }
    rs = 1.0f / sqrtf(m2 / (float)(32 * npart) + eps);
    nmurs = -mean * rs;
}
__device__ __forceinline__ float ln_apply(float x, float rs, float nmurs, float gam, float bet) { return fmaf(fmaf(x, rs, nmurs), gam, bet); }

template <int MODE>
__device__ __forceinline__ float log_one_t(const LogParams& p, float s, float ref_db, float floor_db) {
    if (MODE == KM_LOG_LN_EPS) return 0.693147180559945309f * __builtin_amdgcn_logf(s + p.log_eps);   // src/features/stft.py:123
    float v = db10(s, p.amin) - ref_db;                                // librosa.power_to_db
    v = clamp_lo(v, floor_db);
    return (v + p.db_add) * p.db_scale;                               // simplified_dual_stream_model.py:200
}

__device__ __forceinline__ float log_one(const LogParams& p, float s, float ref_db, float floor_db) {
    return p.log_mode == KM_LOG_LN_EPS ? log_one_t<KM_LOG_LN_EPS>(p, s, ref_db, floor_db)
                                       : log_one_t<KM_LOG_DB_MAX>(p, s, ref_db, floor_db);
}

template <int MODE>
__device__ __forceinline__ float4 log_four_t(const LogParams& p, float4 v, float ref_db, float floor_db) {
    return make_float4(log_one_t<MODE>(p, v.x, ref_db, floor_db), log_one_t<MODE>(p, v.y, ref_db, floor_db),
                       log_one_t<MODE>(p, v.z, ref_db, floor_db), log_one_t<MODE>(p, v.w, ref_db, floor_db));
}

// Sum over the 16 lanes of a DPP row, result in every lane: the same pairing (hence bit-identical sums) as the
// xor-1/2/4/8 butterfly, but four v_add_f32_dpp instead of four ds_bpermute round trips through the LDS crossbar.
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

// ---------------------------------------------------------------------------------------------
// Emotion stream of ONE window by one 512-thread workgroup (production shape: d_model 256, decoder hidden 128,
// emotion_dim <= 256), for workgroups of another kernel that have slack (the front end's second workgroup of a
// window).  Arithmetic and summation order are exactly those of emotion_kernel_d256 (km_core.hip): layer 1 in four
// k-quarters of 64, layer 2 in eight k-eighths of 32, partials combined in index order -- the two are bit-identical.
// ---------------------------------------------------------------------------------------------
struct EmoArgs {
    const float* emo = nullptr;   // (B, ED); nullptr = not fused
    int ED = 0;
    const float *wee_t = nullptr /* (256, 256): rows >= ED are zero */, *bee = nullptr, *lg = nullptr, *lb = nullptr, *we2 = nullptr, *be2 = nullptr,
                *w2 = nullptr, *b2 = nullptr;
    float* zemo = nullptr;        // (B)
    int d = 256, DH = 128;        // 256 / 128: emotion_window_d256 (wee_t is then the zero-padded (256, 256) image)
};

// scratch: 1536 floats of LDS.  Must be called by all 512 threads of the workgroup (contains barriers).
__device__ __forceinline__ void emotion_window_d256(const EmoArgs& e, int64_t b, float* scratch) {
    constexpr int d = 256, DH = 128;
    float* emo_s = scratch;           // [256]
    float* e1 = scratch + 256;        // [256]
    float* part = scratch + 512;      // [4][256] then [8][128]
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 256) emo_s[tid] = tid < e.ED ? e.emo[b * e.ED + tid] : 0.f;
    __syncthreads();
    {   // layer 1: column n, k-quarters 2h and 2h+1 (64 coalesced loads in flight per quarter; wee_t is the
        // zero-padded 256-row image, so no guards)
        const int n = tid & 255, h = __builtin_amdgcn_readfirstlane(tid >> 8);
#pragma unroll 1
        for (int q = 2 * h; q < 2 * h + 2; ++q) {
            const float* wp = e.wee_t + (size_t)64 * q * d + n;
            float w[64];
#pragma unroll
            for (int u = 0; u < 64; ++u) w[u] = wp[u * d];
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < 64; u += 4) {
                const float4 x = *reinterpret_cast<const float4*>(emo_s + 64 * q + u);
                acc = fmaf(x.x, w[u], acc);
                acc = fmaf(x.y, w[u + 1], acc);
                acc = fmaf(x.z, w[u + 2], acc);
                acc = fmaf(x.w, w[u + 3], acc);
            }
            part[q * d + n] = acc;
        }
    }
    __syncthreads();
    if (tid < 256) {
        float s = e.bee[tid];
#pragma unroll
        for (int q = 0; q < 4; ++q) s += part[q * d + tid];
        e1[tid] = s;
    }
    __syncthreads();
    if (tid < 64) {   // LayerNorm (eps 1e-5), two-pass, one wave
        float x[4], s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = e1[lane + 64 * i]; s += x[i]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / d);
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float t = x[i] - mean; v += t * t; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        const float rstd = 1.0f / sqrtf(v * (1.0f / d) + 1e-5f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            e1[c] = (x[i] - mean) * rstd * e.lg[c] + e.lb[c];
        }
    }
    __syncthreads();
    {   // layer 2: hidden unit m, k-eighths 2h and 2h+1
        const int m = tid & 127, h = tid >> 7;     // two values of h per wave
#pragma unroll 1
        for (int q = 2 * h; q < 2 * h + 2; ++q) {
            const float* wp = e.we2 + (size_t)32 * q * DH + m;
            float w[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) w[u] = wp[u * DH];
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < 32; u += 4) {
                const float4 x = *reinterpret_cast<const float4*>(e1 + 32 * q + u);
                acc = fmaf(x.x, w[u], acc);
                acc = fmaf(x.y, w[u + 1], acc);
                acc = fmaf(x.z, w[u + 2], acc);
                acc = fmaf(x.w, w[u + 3], acc);
            }
            part[q * DH + m] = acc;
        }
    }
    __syncthreads();
    if (tid < 64) {   // ReLU, dot with w2
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mm = lane + 64 * i;
            float hsum = e.be2[mm];
#pragma unroll
            for (int q = 0; q < 8; ++q) hsum += part[q * DH + mm];
            s += fmaxf(hsum, 0.f) * e.w2[mm];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) e.zemo[b] = s + e.b2[0];
    }
    __syncthreads();   // scratch is free again
}

// The same stream for the other shapes (d_model, decoder hidden: powers of two in [64, 1024]; emotion_dim <= 256), again
// for a 512-thread workgroup.  Both layers read their weights as float4 (four adjacent output columns per thread) with
// the contraction index split over 512 / (columns / 4) thread groups, 16 independent 16-byte loads in flight per thread;
// partial sums meet in LDS in group order (deterministic).  wee_t is (ED, d), we2 is (d, DH).
// scratch: 256 + d + 2048 floats of LDS.  Must be called by all 512 threads of the workgroup (contains barriers).
struct EmoShape { int d = 0, DH = 0; };

__device__ __forceinline__ void emotion_layer_f4(const float* __restrict__ w, int K, int N, const float* x, float* part) {
    const int tid = threadIdx.x;
    const int tpr = N >> 2;                        // threads per weight row
    const int G = 512 / tpr;                       // k groups
    const int c4 = tid & (tpr - 1), g = tid / tpr;
    const int kc = (K + G - 1) / G;
    const int k0 = g * kc, k1 = (k0 + kc) < K ? (k0 + kc) : K;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* wp = w + 4 * c4;
    for (int k = k0; k < k1; k += 16) {
        float4 wv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) wv[u] = (k + u) < k1 ? *reinterpret_cast<const float4*>(wp + (size_t)(k + u) * N) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float xv = (k + u) < k1 ? x[k + u] : 0.f;
            acc.x = fmaf(xv, wv[u].x, acc.x); acc.y = fmaf(xv, wv[u].y, acc.y);
            acc.z = fmaf(xv, wv[u].z, acc.z); acc.w = fmaf(xv, wv[u].w, acc.w);
        }
    }
    *reinterpret_cast<float4*>(part + g * N + 4 * c4) = acc;
}

__device__ __forceinline__ void emotion_window_generic(const EmoArgs& e, const EmoShape sh, int64_t b, float* scratch) {
    const int d = sh.d, DH = sh.DH;
    float* emo_s = scratch;           // [256]
    float* e1 = scratch + 256;        // [d]
    float* part = e1 + d;             // [512 * 4]: [G1][d], then [G2][DH]
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 256) emo_s[tid] = tid < e.ED ? e.emo[b * e.ED + tid] : 0.f;
    __syncthreads();
    emotion_layer_f4(e.wee_t, e.ED, d, emo_s, part);
    __syncthreads();
    {
        const int G1 = 2048 / d;
        for (int n = tid; n < d; n += 512) {
            float s = e.bee[n];
            for (int g = 0; g < G1; ++g) s += part[g * d + n];
            e1[n] = s;
        }
    }
    __syncthreads();
    if (tid < 64) {   // LayerNorm (eps 1e-5), two-pass, one wave
        float s = 0.f;
        for (int n = lane; n < d; n += 64) s += e1[n];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / d;
        float v = 0.f;
        for (int n = lane; n < d; n += 64) { const float t = e1[n] - mean; v += t * t; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        const float rstd = 1.0f / sqrtf(v / d + 1e-5f);
        for (int n = lane; n < d; n += 64) e1[n] = (e1[n] - mean) * rstd * e.lg[n] + e.lb[n];
    }
    __syncthreads();
    emotion_layer_f4(e.we2, d, DH, e1, part);
    __syncthreads();
    if (tid < 64) {   // ReLU, dot with w2
        const int G2 = 2048 / DH;
        float s = 0.f;
        for (int m = lane; m < DH; m += 64) {
            float h = e.be2[m];
            for (int g = 0; g < G2; ++g) h += part[g * DH + m];
            s += fmaxf(h, 0.f) * e.w2[m];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) e.zemo[b] = s + e.b2[0];
    }
    __syncthreads();   // scratch is free again
}

}  // namespace km
