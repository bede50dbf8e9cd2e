// Attention of the legacy SimplifiedKoeMorphModel (device code shared by km_generic.hip: the stand-alone kernel, and km_kmmf.hip: the
// attention + tail launch).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

namespace km {

// Attention of the legacy model for hd = 32: one WAVE per (window, head), nothing in LDS, no barrier.  Q_h^T (32 x 52, input
// independent, pre-scaled: l_q) sits in registers in the MFMA C layout, i.e. as the B operand of S^T = K_h Q_h^T; a key tile's
// K rows are the A operand straight from memory (one 16-byte load per lane and 16 dimensions), its V values are loaded in C layout
// (lane (g, j) = V[key 4 g + r][dim j]) = the A operand of O^T += V_h^T P^T with P^T = the softmaxed S^T as it stands.  The
// softmax over the Tm keys runs online over the key tiles (running maximum and sum per query column, accumulators rescaled:
// every rescale factor is one value per lane).  Replaces two batched strided products + a row softmax: 228 -> 69 us per 256 windows
// x 8 heads x 257 keys.  O (B, 52, d): O^T tiles leave as 16-byte stores.
// (window, head) pair bh of this WAVE; no barrier inside
__device__ __forceinline__ void legacy_attention_body(const float* __restrict__ Qs, const float* __restrict__ Kp,
                                                      const float* __restrict__ Vp, float* __restrict__ O, int64_t bh, int Tm,
                                                      int H, int NQ, unsigned* __restrict__ zero_max) {
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int64_t b = bh / H;
    const int h = (int)(bh - b * H), d = 32 * H;
    // the window maxima of the front end were consumed by the launch before this one (the encoder with the fused dB conversion):
    // clean slots for the next front-end launch, no memset on the step path
    if (zero_max && h == 0 && lane == 0) zero_max[b] = 0u;
    float qT[2][4][4];                          // [dim tile][query tile][reg]: Q^T[16 dt + 4 g + s][16 qt + j]
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            const int q = 16 * qt + j;
            const float4 v = q < NQ ? *reinterpret_cast<const float4*>(Qs + (int64_t)q * d + 32 * h + 16 * dt + 4 * g)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
            // scores in the base-2 domain: softmax(s) = 2^(s log2 e - max) / sum, and the exponentials below are one v_exp_f32 each
            // (expf: ~15 instructions; 20 per key tile and query tile made the softmax as long as the tile's 64 MFMAs)
            constexpr float kLog2e = 1.4426950408889634f;
            qT[dt][qt][0] = v.x * kLog2e; qT[dt][qt][1] = v.y * kLog2e; qT[dt][qt][2] = v.z * kLog2e; qT[dt][qt][3] = v.w * kLog2e;
        }
    typedef float f32x4l __attribute__((ext_vector_type(4)));
    f32x4l oT[2][4];
    float m[4], l[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        m[qt] = -INFINITY; l[qt] = 0.f;
        oT[0][qt] = f32x4l{0, 0, 0, 0}; oT[1][qt] = f32x4l{0, 0, 0, 0};
    }
    const float* Kb = Kp + (b * Tm) * (int64_t)d + 32 * h;
    const float* Vb = Vp + (b * Tm) * (int64_t)d + 32 * h;
    const int nkt = (Tm + 15) / 16;
    // operands of a key tile (keys past Tm: the last row again, masked below)
    auto load_tile = [&](int kt, float4 (&ka)[2], float (&va)[2][4]) {
        const int kr = 16 * kt + j < Tm ? 16 * kt + j : Tm - 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) ka[dt] = *reinterpret_cast<const float4*>(Kb + (int64_t)kr * d + 16 * dt + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * g + r < Tm ? 16 * kt + 4 * g + r : Tm - 1;
            va[0][r] = Vb[(int64_t)key * d + j];
            va[1][r] = Vb[(int64_t)key * d + 16 + j];
        }
    };
    // the NEXT key tile's operands are requested before the current tile's 64 MFMAs (measured: no change -- 68.6 us either way; what
    // helped was the base-2 softmax above, 68.6 -> 57.0 us.  Also measured and dropped: masking only in the last tile and skipping
    // the accumulator rescale when no lane's maximum moved: 58.8 us, the branches cost more than the 48 instructions they save)
    float4 ka[2], ka_n[2];
    float va[2][4], va_n[2][4];
    load_tile(0, ka, va);
    for (int kt = 0; kt < nkt; ++kt) {
        load_tile(kt + 1 < nkt ? kt + 1 : kt, ka_n, va_n);
        f32x4l S[4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            f32x4l cacc = f32x4l{0, 0, 0, 0};
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const float kv[4] = {ka[dt].x, ka[dt].y, ka[dt].z, ka[dt].w};
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) cacc = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[s_], qT[dt][qt][s_], cacc, 0, 0, 0);   // S^T[key][query]
            }
            S[qt] = cacc;
        }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            float tm = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * kt + 4 * g + r >= Tm) S[qt][r] = -INFINITY;
                tm = fmaxf(tm, S[qt][r]);
            }
            tm = fmaxf(tm, __shfl_xor(tm, 16));
            tm = fmaxf(tm, __shfl_xor(tm, 32));
            const float mn = fmaxf(m[qt], tm);
            const float alpha = mn == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f(m[qt] - mn);       // no key yet: nothing to rescale
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { S[qt][r] = mn == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(S[qt][r] - mn); ps += S[qt][r]; }
            l[qt] = l[qt] * alpha + ps;            // this lane's part of the row sum (same alpha in the four lanes of a column)
            m[qt] = mn;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                f32x4l o = oT[dt][qt];
                o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) o = __builtin_amdgcn_mfma_f32_16x16x4f32(va[dt][s_], S[qt][s_], o, 0, 0, 0);     // O^T[dim][query]
                oT[dt][qt] = o;
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            ka[dt] = ka_n[dt];
#pragma unroll
            for (int r = 0; r < 4; ++r) va[dt][r] = va_n[dt][r];
        }
    }
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        float ls = l[qt];
        ls += __shfl_xor(ls, 16);
        ls += __shfl_xor(ls, 32);
        const float inv = 1.0f / ls;
        const int q = 16 * qt + j;
        if (q < NQ) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const f32x4l o = oT[dt][qt];
                *reinterpret_cast<float4*>(O + (b * NQ + q) * (int64_t)d + 32 * h + 16 * dt + 4 * g) =
                    make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
            }
        }
    }
}

}  // namespace km
