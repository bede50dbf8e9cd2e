// Fused kernels of the legacy KoeMorphModel forward (src/model/gaussian_face.py:175-268) at the reference's default width
// (d_model 256, 8 heads, 52 queries, decoder 128, windows of <= 32 frames).  Two launches replace the ~75 of the
// launch-per-step chain in km_koemorph.hip:
//
//   kmmf_encoder_kernel   DualStreamEncoder (dual_stream_attention.py:369-388): one workgroup = 64 rows (4 MFMA row tiles) of ONE
//                         stream = two windows of 17 - 32 frames (window w on rows 32 w ..), or 4 .. 64 windows of fewer frames
//                         (slots of the power of two >= T); input projection + ReLU + LayerNorm and every post-norm
//                         transformer layer with the rows resident in LDS.
//   kmmf_decode_kernel    average of the two encodings, query embeddings (+ conditioning net), every cross-attention layer,
//                         BlendshapeDecoder and the output tail: one workgroup per window, the 52 query rows resident in LDS.
//
// All products are exact-fp32 v_mfma_f32_16x16x4_f32.  Activations are the A operand out of an LDS image [row][264] (one
// ds_read_b128 = the operand of four MFMAs, stride 264 = 8 mod 64 dwords: conflict-free for the instruction's lane groups);
// weights are the B operand straight from L2 in the fragment-packed images of km_kmmf.h (one k block prefetched), never
// staged: a workgroup owns all 64 rows, so every weight fragment is used by 4 row tiles x 4 MFMAs and nobody else in the
// workgroup needs it.  Attention never leaves the registers: Q_h^T and K_h^T are produced TRANSPOSED (W x^T: weight
// fragments as the A operand), so a C-layout tile holds 4 consecutive head dimensions per lane = the operand of the four
// MFMAs that contract them; S^T = K_h Q_h^T, the softmax runs over the rows of S^T (4 in-lane values x 2 tiles, then lanes
// l ^ 16, l ^ 32), P^T is the B operand of O^T = V_h^T P^T as it stands, and O^T leaves as 16-byte LDS stores.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <string>

#include "km_context.h"
#include "km_device.h"
#include "km_legacy_attn_dev.h"
#include "km_gemm.h"
#include "km_kmmf.h"

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return km::fail(KM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace km {

#include "km_kmm_tail.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define KM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace kf {
using namespace kmmf;
constexpr int NTH = 512;             // decode kernel: threads per workgroup (8 waves: wave = head in the attention phase, = 32 columns elsewhere)
constexpr int NW = 8;
constexpr int XS = 264;              // row stride of the LDS images (floats)
constexpr int KBD = D / 16;          // k blocks of a 256-wide contraction
constexpr int RED = 2 * 64 * NW + 2 * 64;   // LayerNorm partials [pass][row][wave] + totals [pass][row] of 64 rows x 8 waves
// encoder kernel: 256 threads (4 waves: wave = heads w and w + 4 in the attention phase, = 64 columns elsewhere) on 32 rows,
// 70 KB of LDS: TWO independent workgroups per CU instead of one 512-thread workgroup on 64 rows whose two waves per SIMD
// sit in the same barrier-separated phase.  Measured: the split alone changes nothing (0.785 against 0.786 ms per forward),
// but pinned weight prefetch (mm_cols PIN 1) gains 4 % in this form and loses in the other
constexpr int ENTH = 256;
constexpr int ENW = 4;
constexpr int EROWS = 32;
constexpr int EIMG = EROWS * XS;
constexpr int ERED = 2 * EROWS * ENW + 2 * EROWS;
constexpr int ENC_LDS_FLOATS = 2 * EIMG + ERED;

// timing experiments (tools/micro/lib_variant.sh; results are wrong): 1 GELU -> ReLU, 2 no LayerNorm arithmetic
#ifndef KM_KMMF_SKIP
#define KM_KMMF_SKIP 0
#endif
// KM_KMMF_STAMP (timing builds, tools/micro/kmmf_stamp.py): every encoder wave sums the shader-clock cycles it spends in each part
// of the kernel (s_memtime at the part boundaries) into km_kmmf_stamps[(workgroup, wave)][part]
#ifdef KM_KMMF_STAMP
__device__ unsigned long long km_kmmf_stamps[2048 * 4 * 16 + 1024 * 8 * 16];    // encoder, then decode
#define KF_STAMP(i)                                                   \
    do {                                                              \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        st_acc[i] += (unsigned)(now_ - st_last);                      \
        st_last = now_;                                               \
    } while (0)
#else
#define KF_STAMP(i) do {} while (0)
#endif

__device__ __forceinline__ float row16_sum(float v) {      // sum over the 16 lanes of a DPP row, in every lane
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
    return v;
}

// acc[mt][nt] += X[16 mt + ..][16 kb ..] . W[16 (t0 + nt) + ..][16 (kb0 + kb) ..]^T over nkb k blocks.
// X: LDS image (k block 0 of the contraction at column 0); wp: fragment-packed weight of kbs k blocks per tile.
// PIN 0: the k loop as the compiler schedules it (the next block's weight loads sink into the step and are waited for with
// vmcnt(0) a few MFMAs later).  PIN 1: the next k block's weight fragments are requested at the top of a step and held there
// by scheduling barriers, two register sets alternating.  PIN 2: the rows' LDS fragments as well.  Measured, 256 x 30 frames:
// the encoder (4 waves per workgroup, two workgroups per CU) gains 4 % from PIN 1 (PIN 2: 3 %); the 8-wave decode kernel LOSES
// 11 % to it (its two waves per SIMD run in lockstep: the sunk loads stagger them).
#ifndef KM_KMMF_ENC_PIN
#define KM_KMMF_ENC_PIN 1
#endif
#ifndef KM_KMMF_ENC_QKV_PIN
#define KM_KMMF_ENC_QKV_PIN 0
#endif
#ifndef KM_KMMF_DEC_PIN
#define KM_KMMF_DEC_PIN 0
#endif
#ifndef KM_LEGACY_ENC_PIN
#define KM_LEGACY_ENC_PIN 1      /* round 4, measured and dropped: weight fragments THREE k blocks ahead (four register sets in turn) in the legacy
                                   encoder: 0.453 against 0.451 ms per forward -- its waves do not wait for the weights */
#endif
template <int MT, int NTW, int PIN>
__device__ __forceinline__ void mm_cols(f32x4 (&acc)[MT][NTW], const float* X, const float* wp, int t0, int kbs, int kb0, int nkb, int lane) {
    const int g = lane >> 4, j = lane & 15;
    const float* xp = X + j * XS + 4 * g;
    const f32x4* w = reinterpret_cast<const f32x4*>(wp) + ((size_t)t0 * kbs + kb0) * 64 + lane;
    if constexpr (PIN == 0) {
    f32x4 b[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) b[nt] = w[(size_t)nt * kbs * 64];
    for (int kb = 0; kb < nkb; ++kb) {
        const int kn = kb + 1 < nkb ? kb + 1 : kb;
        f32x4 bn[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) bn[nt] = w[((size_t)nt * kbs + kn) * 64];
        f32x4 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(xp + 16 * mt * XS + 16 * kb);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = KM_MFMA(a[mt][s], b[nt][s], acc[mt][nt]);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) b[nt] = bn[nt];
    }
    } else {
    f32x4 b0[NTW], b1[NTW], a0[MT], a1[MT];
    auto fetch = [&](f32x4 (&b)[NTW], f32x4 (&a)[MT], int kb) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) b[nt] = w[((size_t)nt * kbs + kb) * 64];
        if constexpr (PIN == 2) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(xp + 16 * mt * XS + 16 * kb);
        }
    };
    auto step = [&](const f32x4 (&b)[NTW], f32x4 (&a)[MT], int kb) {
        if constexpr (PIN == 1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(xp + 16 * mt * XS + 16 * kb);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = KM_MFMA(a[mt][s], b[nt][s], acc[mt][nt]);
    };
    fetch(b0, a0, 0);
    for (int kb = 0; kb < nkb; kb += 2) {
        fetch(b1, a1, kb + 1 < nkb ? kb + 1 : kb);
        __builtin_amdgcn_sched_barrier(0);
        step(b0, a0, kb);
        __builtin_amdgcn_sched_barrier(0);
        fetch(b0, a0, kb + 2 < nkb ? kb + 2 : kb);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < nkb) step(b1, a1, kb + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    }
}

// LayerNorm (two-pass, eps 1e-5) over 16 MT rows whose DCOLS columns are spread over the NWV waves (NTW column tiles each, C layout),
// then (STORE) dst[row][col0 + 16 nt + j] = the normalised value (which v holds afterwards) for rows < row_limit.  Two barriers
// per pass (partials -> totals).  red: 2 (16 MT) NWV + 2 (16 MT) floats.  The caller fences dst against its readers.
template <int MT, int NTW, int DCOLS, int NWV, bool STORE = true>
__device__ __forceinline__ void ln_store(f32x4 (&v)[MT][NTW], float* red, const float* gam, const float* bet, float* dst, int col0,
                                         int row_limit, int wave, int lane, int tid) {
    constexpr int ROWS = 16 * MT;
    static_assert(NWV == 4 || NWV == 8, "partials of a row are one or two 16-byte reads");
    const int g = lane >> 4, j = lane & 15;
    float mean[MT][4], rstd[MT][4];
#pragma unroll
    for (int pass = (KM_KMMF_SKIP & 2) ? 2 : 0; pass < 2; ++pass) {
        float* P = red + pass * (ROWS * NWV);
        float part[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    if (pass == 0) s += v[mt][nt][r];
                    else { const float dlt = v[mt][nt][r] - mean[mt][r]; s += dlt * dlt; }
                }
                part[mt][r] = row16_sum(s);
            }
        if (j == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) P[(16 * mt + 4 * g + r) * NWV + wave] = part[mt][r];
        }
        __syncthreads();
        float* Tt = red + 2 * ROWS * NWV + pass * ROWS;
        if (tid < ROWS) {
            const f32x4* pr = reinterpret_cast<const f32x4*>(P + tid * NWV);
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < NWV / 4; ++q) { const f32x4 p4 = pr[q]; s += p4[0]; s += p4[1]; s += p4[2]; s += p4[3]; }
            Tt[tid] = s;
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(Tt + 16 * mt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (pass == 0) mean[mt][r] = t4[r] * (1.0f / DCOLS);
                else rstd[mt][r] = 1.0f / sqrtf(t4[r] * (1.0f / DCOLS) + 1e-5f);
            }
        }
    }
    float gm[NTW], bt[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) { gm[nt] = gam[col0 + 16 * nt + j]; bt[nt] = bet[col0 + 16 * nt + j]; }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + 4 * g + r;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                if (!(KM_KMMF_SKIP & 2)) v[mt][nt][r] = (v[mt][nt][r] - mean[mt][r]) * rstd[mt][r] * gm[nt] + bt[nt];
                if (STORE && row < row_limit) dst[row * XS + col0 + 16 * nt + j] = v[mt][nt][r];
            }
        }
}

// One head (= this wave) of an attention layer with everything in registers, in two steps.
//
// project_qkv: Q_h^T (scaled, as torch scales q after its bias), K_h^T and V_h of this head.
//   Xq: LDS image of the query rows (QT row tiles), Xk: image of the key / value rows (KTT row tiles; SELF: the same rows).
//   wq / wk / wv: fragment-packed (256-wide) projections already offset to this head's first tile (16 k blocks per tile),
//   bq / bk / bv: their biases offset to this head's first column.
template <int QT, int KTT, bool SELF, int PIN>
__device__ __forceinline__ void project_qkv(f32x4 (&qT)[2][QT], f32x4 (&kT)[2][KTT], f32x4 (&vv)[KTT][2], const float* Xq, const float* Xk,
                                            const float* wq, const float* wk, const float* wv, const float* bq, const float* bk,
                                            const float* bv, float scale, int lane) {
    const int g = lane >> 4, j = lane & 15;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
        for (int t = 0; t < QT; ++t) qT[dt][t] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < KTT; ++t) { kT[dt][t] = f32x4{0, 0, 0, 0}; vv[t][dt] = f32x4{0, 0, 0, 0}; }
    }
    {
        const f32x4* pq = reinterpret_cast<const f32x4*>(wq) + lane;
        const f32x4* pk = reinterpret_cast<const f32x4*>(wk) + lane;
        const f32x4* pv = reinterpret_cast<const f32x4*>(wv) + lane;
        const float* xq = Xq + j * XS + 4 * g;
        const float* xk = Xk + j * XS + 4 * g;
        if constexpr (PIN == 0) {
        f32x4 fq[2], fk[2], fv[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) { fq[dt] = pq[dt * KBD * 64]; fk[dt] = pk[dt * KBD * 64]; fv[dt] = pv[dt * KBD * 64]; }
        for (int kb = 0; kb < KBD; ++kb) {
            const int kn = kb + 1 < KBD ? kb + 1 : kb;
            f32x4 nq[2], nk[2], nv[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                nq[dt] = pq[(dt * KBD + kn) * 64]; nk[dt] = pk[(dt * KBD + kn) * 64]; nv[dt] = pv[(dt * KBD + kn) * 64];
            }
            f32x4 aq[QT], ak[KTT];
#pragma unroll
            for (int t = 0; t < QT; ++t) aq[t] = *reinterpret_cast<const f32x4*>(xq + 16 * t * XS + 16 * kb);
#pragma unroll
            for (int t = 0; t < KTT; ++t) {
                if constexpr (SELF) ak[t] = aq[t];
                else ak[t] = *reinterpret_cast<const f32x4*>(xk + 16 * t * XS + 16 * kb);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
                    for (int t = 0; t < QT; ++t) qT[dt][t] = KM_MFMA(fq[dt][s], aq[t][s], qT[dt][t]);       // Q^T = Wq x^T
#pragma unroll
                    for (int t = 0; t < KTT; ++t) kT[dt][t] = KM_MFMA(fk[dt][s], ak[t][s], kT[dt][t]);     // K^T = Wk x^T
#pragma unroll
                    for (int t = 0; t < KTT; ++t) vv[t][dt] = KM_MFMA(ak[t][s], fv[dt][s], vv[t][dt]);     // V = x Wv^T
                }
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) { fq[dt] = nq[dt]; fk[dt] = nk[dt]; fv[dt] = nv[dt]; }
        }
        } else {       // see mm_cols
            f32x4 f0[3][2], f1[3][2];            // [q, k, v][head dimension tile]
            auto fetch = [&](f32x4 (&f)[3][2], int kb) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) { f[0][dt] = pq[(dt * KBD + kb) * 64]; f[1][dt] = pk[(dt * KBD + kb) * 64]; f[2][dt] = pv[(dt * KBD + kb) * 64]; }
            };
            auto step = [&](const f32x4 (&f)[3][2], int kb) {
                f32x4 aq[QT], ak[KTT];
#pragma unroll
                for (int t = 0; t < QT; ++t) aq[t] = *reinterpret_cast<const f32x4*>(xq + 16 * t * XS + 16 * kb);
#pragma unroll
                for (int t = 0; t < KTT; ++t) {
                    if constexpr (SELF) ak[t] = aq[t];
                    else ak[t] = *reinterpret_cast<const f32x4*>(xk + 16 * t * XS + 16 * kb);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
                        for (int t = 0; t < QT; ++t) qT[dt][t] = KM_MFMA(f[0][dt][s], aq[t][s], qT[dt][t]);
#pragma unroll
                        for (int t = 0; t < KTT; ++t) kT[dt][t] = KM_MFMA(f[1][dt][s], ak[t][s], kT[dt][t]);
#pragma unroll
                        for (int t = 0; t < KTT; ++t) vv[t][dt] = KM_MFMA(ak[t][s], f[2][dt][s], vv[t][dt]);
                    }
                }
            };
            static_assert(KBD % 2 == 0, "two k blocks per turn");
            fetch(f0, 0);
            for (int kb = 0; kb < KBD; kb += 2) {
                fetch(f1, kb + 1);
                __builtin_amdgcn_sched_barrier(0);
                step(f0, kb);
                __builtin_amdgcn_sched_barrier(0);
                fetch(f0, kb + 2 < KBD ? kb + 2 : kb);
                __builtin_amdgcn_sched_barrier(0);
                step(f1, kb + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const f32x4 b4q = *reinterpret_cast<const f32x4*>(bq + 16 * dt + 4 * g);
        const f32x4 b4k = *reinterpret_cast<const f32x4*>(bk + 16 * dt + 4 * g);
        const float bvj = bv[16 * dt + j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int t = 0; t < QT; ++t) qT[dt][t][r] = (qT[dt][t][r] + b4q[r]) * scale;
#pragma unroll
            for (int t = 0; t < KTT; ++t) { kT[dt][t][r] += b4k[r]; vv[t][dt][r] += bvj; }
        }
    }
}

// attend: masked softmax and P V for groups of KW key tiles x QW query tiles: group w owns key tiles [KW w, KW w + KW) and
// query tiles [QW w, QW w + QW) (encoder, windows of 17 - 32 frames: two groups of 2 x 2 tiles; windows of <= 16 frames: four
// groups of 1 x 1, several windows inside a tile told apart by the mask; cross-attention: one group of 2 x 4).
//   mask(key_row, query_row) -> true when that key may be attended by that query (row indices in the images).
//   O^T -> Yo[query row][32 head + dim] for query rows < q_limit;  store_p(query_row, key_row, p) sees every probability.
template <int QT, int KTT, int KW, int QW, class Mask, class StoreP>
__device__ __forceinline__ void attend(const f32x4 (&qT)[2][QT], const f32x4 (&kT)[2][KTT], const f32x4 (&vv)[KTT][2], float* Yo, int head,
                                       int q_limit, int lane, Mask mask, StoreP store_p) {
    constexpr int NG = KTT / KW;
    static_assert(NG * QW == QT && NG * KW == KTT, "groups tile the images");
    const int g = lane >> 4, j = lane & 15;
#pragma unroll
    for (int w = 0; w < NG; ++w) {
        f32x4 S[KW][QW];
#pragma unroll
        for (int kt = 0; kt < KW; ++kt)
#pragma unroll
            for (int qt = 0; qt < QW; ++qt) {
                f32x4 c = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int s = 0; s < 4; ++s) c = KM_MFMA(kT[dt][KW * w + kt][s], qT[dt][QW * w + qt][s], c);   // S^T[key][query]
                S[kt][qt] = c;
            }
        // masked softmax over the keys (the rows of S^T) of each query column
#pragma unroll
        for (int qt = 0; qt < QW; ++qt) {
            const int q = 16 * (QW * w + qt) + j;
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < KW; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (!mask(16 * (KW * w + kt) + 4 * g + r, q)) S[kt][qt][r] = -INFINITY;
                    m = fmaxf(m, S[kt][qt][r]);
                }
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < KW; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {       // v_exp_f32 (arguments <= 0; a column without keys: -inf - -inf = NaN, as in torch)
                    S[kt][qt][r] = __builtin_amdgcn_exp2f((S[kt][qt][r] - m) * 1.44269504088896341f);
                    sum += S[kt][qt][r];
                }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int kt = 0; kt < KW; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    S[kt][qt][r] *= inv;
                    store_p(q, 16 * (KW * w + kt) + 4 * g + r, S[kt][qt][r]);
                }
        }
        // O^T[dim][query] = sum over keys V[key][dim] P^T[key][query]
#pragma unroll
        for (int qt = 0; qt < QW; ++qt)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                f32x4 o = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int kt = 0; kt < KW; ++kt)
#pragma unroll
                    for (int s = 0; s < 4; ++s) o = KM_MFMA(vv[KW * w + kt][dt][s], S[kt][qt][s], o);
                const int row = 16 * (QW * w + qt) + j;
                if (row < q_limit) *reinterpret_cast<f32x4*>(Yo + row * XS + 32 * head + 16 * dt + 4 * g) = o;
            }
    }
}

struct EncArgs {
    const float* in0; const float* in1;       // (B, T, in_dim) of the mel / emotion stream
    int in_dim0, in_dim1;
    const float* blob; int64_t stream_floats; int layers;
    const unsigned char* kvalid;              // (B, T), 1 = attend, or null (src_key_padding_mask)
    float* out0; float* out1;                 // (B, T, 256)
    int B, T;
    int groups;                               // workgroups per stream
    int slot_log2;                            // a window owns 1 << slot_log2 rows of the 32 (the power of two >= T): 1 .. 32 windows per workgroup
};

template <bool SMALL>     // SMALL: windows of <= 16 frames (slots of 1 .. 16 rows), else one window of 17 - 32 frames
__global__ __launch_bounds__(256) void kmmf_encoder_kernel(EncArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;
    float* Y = smem + EIMG;
    float* red = smem + 2 * EIMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int T = a.T, lg = a.slot_log2;
    const int stream = blockIdx.x & 1, widx = blockIdx.x >> 1;
    const int w0 = widx * (EROWS >> lg);                // first window of this workgroup
    const float* in = stream ? a.in1 : a.in0;
    const int in_dim = stream ? a.in_dim1 : a.in_dim0;
    float* out = stream ? a.out1 : a.out0;
    const float* blob = a.blob + stream * a.stream_floats;
    const int col0 = 64 * wave;
#ifdef KM_KMMF_STAMP
    unsigned st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_readcyclecounter();
#endif

    // row r = frame (r & (slot - 1)) of window w0 + (r >> lg).  Bit r of rowmask: row r holds a frame that may be attended
    // (inside the batch, inside the window, not padding: src_key_padding_mask)
    unsigned rowmask;
    {
        const int b = w0 + (lane >> lg), t = lane & ((1 << lg) - 1);
        rowmask = (unsigned)__ballot(lane < EROWS && b < a.B && t < T && (!a.kvalid || a.kvalid[(int64_t)b * T + t]));
    }
    // ---- input rows -> X (columns >= in_dim and rows without a frame are zero) ----
    for (int i = tid; i < EROWS * 64; i += ENTH) {
        const int r = i >> 6, c4 = i & 63, b = w0 + (r >> lg), t = r & ((1 << lg) - 1);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < a.B && t < T && 4 * c4 < in_dim) v = *reinterpret_cast<const float4*>(in + ((int64_t)b * T + t) * in_dim + 4 * c4);
        *reinterpret_cast<float4*>(X + r * XS + 4 * c4) = v;
    }
    __syncthreads();
    KF_STAMP(0);
    // ---- x = LayerNorm(ReLU(in W0^T + b0))   (dual_stream_attention.py:369-388) -> Y ----
    {
        f32x4 acc[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0, 0, 0, 0};
        mm_cols<2, 4, KM_KMMF_ENC_PIN>(acc, X, blob + ENC_W0, 4 * wave, KBD, 0, in_dim / 16, lane);
        KF_STAMP(1);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float bb = blob[ENC_B0 + col0 + 16 * nt + j];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float v = acc[mt][nt][r] + bb; acc[mt][nt][r] = v < 0.f ? 0.f : v; }
        }
        ln_store<2, 4, D, ENW>(acc, red, blob + ENC_LNG, blob + ENC_LNB, Y, col0, EROWS, wave, lane, tid);
    }
    __syncthreads();
    KF_STAMP(2);
    float* xc = Y;      // the rows
    float* xo = X;      // attention output / feed-forward hidden chunk
    const float scale = 1.0f / sqrtf((float)HD);
    for (int layer = 0; layer < a.layers; ++layer) {
        const float* L = blob + ENC_HEAD + (int64_t)layer * ENC_LAYER;
        // ---- self-attention: heads wave and wave + 4 ----
#pragma unroll 1
        for (int hh = 0; hh < 2; ++hh) {
            const int head = wave + 4 * hh;
            f32x4 qT[2][2], kT[2][2], vv[2][2];
            project_qkv<2, 2, true, KM_KMMF_ENC_QKV_PIN>(qT, kT, vv, xc, xc, L + EL_WIN + (int64_t)(2 * head) * KBD * 256,
                                    L + EL_WIN + (int64_t)(16 + 2 * head) * KBD * 256, L + EL_WIN + (int64_t)(32 + 2 * head) * KBD * 256,
                                    L + EL_BIN + 32 * head, L + EL_BIN + D + 32 * head, L + EL_BIN + 2 * D + 32 * head, scale, lane);
            KF_STAMP(3);
            auto nostore = [](int, int, float) {};
            if constexpr (!SMALL) {     // the window = both row tiles
                attend<2, 2, 2, 2>(qT, kT, vv, xo, head, EROWS, lane, [&](int key, int) { return ((rowmask >> key) & 1u) != 0; }, nostore);
            } else {                    // 1 .. 16 windows inside each row tile
                // A tile's P V product runs over the keys of ALL its windows with P = 0 for the foreign ones: the value rows of
                // frames nobody may attend (padding, windows past the batch -- NaN from the second layer on, when a window
                // has no key at all) are cleared, or 0 x NaN would leak into the tile's other windows.  (A non-finite
                // activation of an ATTENDED frame -- non-finite input features -- still spreads inside its row tile.)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (!((rowmask >> (16 * t + 4 * g + r)) & 1u)) { vv[t][0][r] = 0.f; vv[t][1][r] = 0.f; }
                attend<2, 2, 1, 1>(qT, kT, vv, xo, head, EROWS, lane,
                                   [&](int key, int q) { return ((rowmask >> key) & 1u) != 0 && (key >> lg) == (q >> lg); }, nostore);
            }
            KF_STAMP(4);
        }
        __syncthreads();
        KF_STAMP(5);
        // ---- x = LN1(x + out_proj(O)) ----
        {
            f32x4 acc[2][4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0, 0, 0, 0};
            mm_cols<2, 4, KM_KMMF_ENC_PIN>(acc, xo, L + EL_WO, 4 * wave, KBD, 0, KBD, lane);
            KF_STAMP(6);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float bb = L[EL_BO + col0 + 16 * nt + j];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mt][nt][r] += bb + xc[(16 * mt + 4 * g + r) * XS + col0 + 16 * nt + j];
            }
            ln_store<2, 4, D, ENW>(acc, red, L + EL_N1G, L + EL_N1B, xc, col0, EROWS, wave, lane, tid);
        }
        __syncthreads();
        KF_STAMP(7);
        // ---- x = LN2(x + W2 gelu(W1 x + b1) + b2), the hidden layer in four chunks of 256 through xo ----
        {
            f32x4 y2[2][4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) y2[mt][nt] = f32x4{0, 0, 0, 0};
            for (int c = 0; c < FF / D; ++c) {
                f32x4 h[2][4];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) h[mt][nt] = f32x4{0, 0, 0, 0};
                mm_cols<2, 4, KM_KMMF_ENC_PIN>(h, xc, L + EL_W1, 16 * c + 4 * wave, KBD, 0, KBD, lane);
                KF_STAMP(8);
                if (c > 0) __syncthreads();          // the previous chunk's readers of xo
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float bb = L[EL_B1 + D * c + col0 + 16 * nt + j];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            xo[(16 * mt + 4 * g + r) * XS + col0 + 16 * nt + j] = gemm_act(h[mt][nt][r] + bb, (KM_KMMF_SKIP & 1) ? 1 : 2);
                }
                __syncthreads();
                KF_STAMP(9);
                mm_cols<2, 4, KM_KMMF_ENC_PIN>(y2, xo, L + EL_W2, 4 * wave, FF / 16, KBD * c, KBD, lane);
                KF_STAMP(10);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float bb = L[EL_B2 + col0 + 16 * nt + j];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) y2[mt][nt][r] += bb + xc[(16 * mt + 4 * g + r) * XS + col0 + 16 * nt + j];
            }
            ln_store<2, 4, D, ENW>(y2, red, L + EL_N2G, L + EL_N2B, xc, col0, EROWS, wave, lane, tid);
        }
        __syncthreads();
        KF_STAMP(11);
    }
    // ---- rows -> (B, T, 256) ----
    for (int i = tid; i < EROWS * 64; i += ENTH) {
        const int r = i >> 6, c4 = i & 63, b = w0 + (r >> lg), t = r & ((1 << lg) - 1);
        if (b < a.B && t < T) *reinterpret_cast<float4*>(out + ((int64_t)b * T + t) * D + 4 * c4) = *reinterpret_cast<const float4*>(xc + r * XS + 4 * c4);
    }
#ifdef KM_KMMF_STAMP
    KF_STAMP(12);
    if (lane == 0 && blockIdx.x < 2048) {
        unsigned long long* o = km_kmmf_stamps + ((size_t)blockIdx.x * 4 + wave) * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = st_acc[i];
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// kmmf_decode_kernel: gaussian_face.py:209-268 for one window per workgroup.
// LDS: X = the 52 query rows [52][264], Y = attention output / decoder ping-pong [52][264], A = the averaged encoding
// [32][264] (keys and values of every layer), LayerNorm partials.  MFMA row tiles 3's rows 52..63 read whatever follows the
// image (rows are independent in every product, and nothing is stored for them).
// ---------------------------------------------------------------------------------------------------------
constexpr int QIMG = NQ * XS;
constexpr int DEC_LDS_FLOATS = 2 * QIMG + TMAX * XS + RED + 64;

struct DecArgs {
    const float* xm; const float* xe;          // (B, T, 256) the two encodings
    const float* emb;                          // query_embeddings (52, 256)
    const float* cw0; const float* cb0; const float* cw3; const float* cb3;    // conditioning_net (attention.py:481-514)
    const float* prev;                         // (B, 52) or null
    const float* cross; int cross_layers;
    const float* dec; int dec_layers; int act; // gemm_act code of the decoder
    const unsigned char* kvalid; int causal, window;
    float* attn;                               // (L, B, 8, 52, T) or null
    int B, T;
    KmmTail tail;
};

__global__ __launch_bounds__(512) void kmmf_decode_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;
    float* Y = smem + QIMG;
    float* A = smem + 2 * QIMG;
    float* red = A + TMAX * XS;
    float* ys = red + RED;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int b = blockIdx.x, T = a.T;
    const int col0 = 32 * wave;

    const unsigned kmask = (unsigned)__ballot(lane < T && (!a.kvalid || a.kvalid[(int64_t)b * T + lane]));
#ifdef KM_KMMF_STAMP
    unsigned st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_readcyclecounter();
#endif
    // ---- A = (xm + xe) / 2, rows >= T zero (requested first: the conditioning net below waits on its own loads) ----
    for (int i = tid; i < TMAX * 64; i += NTH) {
        const int t = i >> 6, c4 = i & 63;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < T) {
            const float4 m = *reinterpret_cast<const float4*>(a.xm + ((int64_t)b * T + t) * D + 4 * c4);
            const float4 e = *reinterpret_cast<const float4*>(a.xe + ((int64_t)b * T + t) * D + 4 * c4);
            v = make_float4((m.x + e.x) / 2.0f, (m.y + e.y) / 2.0f, (m.z + e.z) / 2.0f, (m.w + e.w) / 2.0f);
        }
        *reinterpret_cast<float4*>(A + t * XS + 4 * c4) = v;
    }
    // ---- conditioning (attention.py:500-512): cond = W3 relu(W0 prev + b0) + b3, through `red`; four lanes per hidden
    //      unit (13 of the 52 inputs each), then two lanes per output (64 of the 128 hidden units each) ----
    if (a.prev) {
        {
            const int o = tid >> 2, part = tid & 3;
            const float* w = a.cw0 + o * NQ + 13 * part;
            const float* p = a.prev + (int64_t)b * NQ + 13 * part;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 13; ++k) s = fmaf(w[k], p[k], s);
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += a.cb0[o];
            if (part == 0) red[o] = s < 0.f ? 0.f : s;
        }
        __syncthreads();
        {
            const int o = tid >> 1, part = tid & 1;
            const float4* w = reinterpret_cast<const float4*>(a.cw3 + o * 128 + 64 * part);
            const float4* hp = reinterpret_cast<const float4*>(red + 64 * part);
            float4 wv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) wv[k] = w[k];
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float4 hv = hp[k];
                s = fmaf(wv[k].x, hv.x, s); s = fmaf(wv[k].y, hv.y, s); s = fmaf(wv[k].z, hv.z, s); s = fmaf(wv[k].w, hv.w, s);
            }
            s += __shfl_xor(s, 1);
            if (part == 0) red[128 + o] = s + a.cb3[o];
        }
        __syncthreads();
    }
    // ---- x = embeddings (+ cond) ----
    for (int i = tid; i < NQ * 64; i += NTH) {
        const int q = i >> 6, c4 = i & 63;
        float4 v = *reinterpret_cast<const float4*>(a.emb + q * D + 4 * c4);
        if (a.prev) {
            const float4 cv = *reinterpret_cast<const float4*>(red + 128 + 4 * c4);
            v.x += cv.x; v.y += cv.y; v.z += cv.z; v.w += cv.w;
        }
        *reinterpret_cast<float4*>(X + q * XS + 4 * c4) = v;
    }
    __syncthreads();
    KF_STAMP(0);
    const float scale = 1.0f / sqrtf((float)HD);                       // (head_dim * temperature)^-0.5, temperature 1
    int qlo[4], qhi[4];               // keys [lo, hi) of this lane's query in each of the four query tiles
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int q = 16 * qt + j;
        int lo = 0, hi = T;
        if (a.window >= 0) {
            const int kp = (q * T) / NQ;
            lo = kp - a.window / 2 > 0 ? kp - a.window / 2 : 0;
            hi = kp + a.window / 2 + 1 < T ? kp + a.window / 2 + 1 : T;
        }
        if (a.causal && q + 1 < hi) hi = q + 1;
        qlo[qt] = lo; qhi[qt] = hi;
    }
    for (int layer = 0; layer < a.cross_layers; ++layer) {
        const float* L = a.cross + (int64_t)layer * CROSS_LAYER;
        float* ap = a.attn ? a.attn + ((((int64_t)layer * a.B + b) * HEADS + wave) * NQ) * T : nullptr;
        {
            f32x4 qT[2][4], kT[2][2], vv[2][2];
            project_qkv<4, 2, false, KM_KMMF_DEC_PIN>(qT, kT, vv, X, A, L + CL_WQ + (int64_t)(2 * wave) * KBD * 256, L + CL_WK + (int64_t)(2 * wave) * KBD * 256,
                                     L + CL_WV + (int64_t)(2 * wave) * KBD * 256, L + CL_BQ + 32 * wave, L + CL_BK + 32 * wave,
                                     L + CL_BV + 32 * wave, scale, lane);
            KF_STAMP(1);
            attend<4, 2, 2, 4>(
                qT, kT, vv, Y, wave, NQ, lane,
                [&](int key, int q) {          // attention.py:208-246: causal and local-window masks, key padding
                    const int qt = q >> 4;      // compile-time in the unrolled caller: bounds computed once per query tile
                    return key >= qlo[qt] && key < qhi[qt] && ((kmask >> key) & 1u) != 0;
                },
                [&](int q, int key, float p) { if (ap && q < NQ && key < T) ap[q * T + key] = p; });
            KF_STAMP(2);
        }
        __syncthreads();
        KF_STAMP(3);
        // ---- x = LN(out_proj(O) + x)   (gaussian_face.py:230-231) ----
        f32x4 acc[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { acc[mt][0] = f32x4{0, 0, 0, 0}; acc[mt][1] = f32x4{0, 0, 0, 0}; }
        mm_cols<4, 2, KM_KMMF_DEC_PIN>(acc, Y, L + CL_WO, 2 * wave, KBD, 0, KBD, lane);
        KF_STAMP(4);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float bb = L[CL_BO + col0 + 16 * nt + j];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][nt][r] += bb + X[(16 * mt + 4 * g + r) * XS + col0 + 16 * nt + j];
        }
        ln_store<4, 2, D, NW>(acc, red, L + CL_LNG, L + CL_LNB, X, col0, NQ, wave, lane, tid);
        __syncthreads();
        KF_STAMP(5);
    }
    // ---- BlendshapeDecoder (decoder.py:131-177): wave = 16 hidden units ----
    float* cur = Y;
    float* nxt = X;
    {
        f32x4 acc[4][1];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][0] = f32x4{0, 0, 0, 0};
        mm_cols<4, 1, KM_KMMF_DEC_PIN>(acc, X, a.dec + DC_WI, wave, KBD, 0, KBD, lane);
        const float bb = a.dec[DC_BI + 16 * wave + j];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * mt + 4 * g + r;
                if (row < NQ) cur[row * XS + 16 * wave + j] = gemm_act(acc[mt][0][r] + bb, a.act);
            }
    }
    __syncthreads();
    for (int layer = 0; layer < a.dec_layers; ++layer) {
        const float* L = a.dec + DC_HEAD + (int64_t)layer * DEC_LAYER;
        f32x4 acc[4][1];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][0] = f32x4{0, 0, 0, 0};
        mm_cols<4, 1, KM_KMMF_DEC_PIN>(acc, cur, L + DL_W, wave, HID / 16, 0, HID / 16, lane);
        const float bb = L[DL_B + 16 * wave + j];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][0][r] += bb;
        ln_store<4, 1, HID, NW, false>(acc, red, L + DL_LNG, L + DL_LNB, nullptr, 16 * wave, NQ, wave, lane, tid);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {                       // x = act(LN(.)) + x   (decoder.py:147-152)
                const int row = 16 * mt + 4 * g + r;
                if (row < NQ) nxt[row * XS + 16 * wave + j] = gemm_act(acc[mt][0][r], a.act) + cur[row * XS + 16 * wave + j];
            }
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
    }
    KF_STAMP(6);
    // ---- output_proj (row q of the weight for query q), eight lanes per query, then the tail ----
    {
        const int q = tid >> 3, part = tid & 7;
        float s = 0.f;
        if (q < NQ) {
            const float4* hr = reinterpret_cast<const float4*>(cur + q * XS + 16 * part);
            const float4* wr = reinterpret_cast<const float4*>(a.tail.wout + q * HID + 16 * part);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 hv = hr[k], wv = wr[k];
                s = fmaf(hv.x, wv.x, s); s = fmaf(hv.y, wv.y, s); s = fmaf(hv.z, wv.z, s); s = fmaf(hv.w, wv.w, s);
            }
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (q < NQ && part == 0) ys[q] = s + a.tail.bout[q];
    }
    __syncthreads();
    const float z = tid < NQ ? ys[tid] : 0.f;
    __syncthreads();
    kmm_tail_dev(a.tail, b, tid, z, ys);
#ifdef KM_KMMF_STAMP
    KF_STAMP(7);
    if (lane == 0 && blockIdx.x < 1024) {
        unsigned long long* o = km_kmmf_stamps + 2048 * 4 * 16 + ((size_t)blockIdx.x * 8 + wave) * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = st_acc[i];
    }
#endif
}


// ---------------------------------------------------------------------------------------------------------
// legacy_encoder_kernel: SimplifiedKoeMorphModel's audio_encoder (Linear(80, 256) ReLU Linear(256, 256) ReLU,
// simplified_model.py:44-51) and the key / value projections of its attention (:136-141) for 32 token rows per 256-thread
// workgroup: the two hidden activations never leave LDS, the four weights are MFMA operands straight from L2 (blob lgf_enc).
// Replaces four NT GEMM launches over (B Tm, 256) activations.  K, V (rows, 256) row-major for legacy_attention_kernel.
// ---------------------------------------------------------------------------------------------------------
// melmax != null: `mel` is the front end's POWER-mel (rows = (window, frame), Tm frames per window) and the dB / log conversion
// against the window's maximum happens while a tile is staged (the same log_one() as mel_log_kernel: that launch and the
// memset of the maxima behind it are gone from km_legacy_forward; legacy_attention_kernel, the next launch, re-zeroes them).
#ifndef KM_LEGACY_ENC_ROWS
#define KM_LEGACY_ENC_ROWS 32       /* token rows per workgroup: 32 (four waves) or 64 (eight waves: two row blocks that fetch the same weight fragments) */
#endif
#ifndef KM_LEGACY_ENC_CT
#define KM_LEGACY_ENC_CT 4          /* column tiles of 16 per wave: 4 (four waves per row block) or 2 (eight waves: twice the waves per SIMD) */
#endif
constexpr int LROWS = KM_LEGACY_ENC_ROWS, LCT = KM_LEGACY_ENC_CT, LCW = 16 / LCT, LNTH = 64 * LCW * (LROWS / 32), LIMG = LROWS * XS;
__global__ __launch_bounds__(LNTH) void legacy_encoder_kernel(const float* __restrict__ mel, const float* __restrict__ blob, int64_t rows,
                                                            float* __restrict__ Kp, float* __restrict__ Vp,
                                                            const unsigned* __restrict__ melmax, int Tm, LogParams lp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all % LCW, rblk = wave_all / LCW;             // column block of 16 LCT, row block of 32
    float* A0 = smem;
    float* B0 = smem + LIMG;
    float* A = A0 + rblk * 32 * XS;
    float* Bi = B0 + rblk * 32 * XS;
    const int g = lane >> 4, j = lane & 15;
    const int64_t r0w = (int64_t)blockIdx.x * LROWS;                     // first row of the workgroup
    const int64_t r0 = r0w + 32 * rblk;                                  // first row of this wave's row block
    const int col0 = 16 * LCT * wave;
    for (int i = tid; i < LROWS * (LG_MEL / 4); i += LNTH) {
        const int r = i / (LG_MEL / 4), c4 = i - r * (LG_MEL / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0w + r < rows) {
            v = *reinterpret_cast<const float4*>(mel + (r0w + r) * LG_MEL + 4 * c4);
            if (melmax) {
                float ref_db, floor_db;
                log_window_consts(lp, __uint_as_float(melmax[(r0w + r) / Tm]), ref_db, floor_db);
                v = make_float4(log_one(lp, v.x, ref_db, floor_db), log_one(lp, v.y, ref_db, floor_db), log_one(lp, v.z, ref_db, floor_db),
                                log_one(lp, v.w, ref_db, floor_db));
            }
        }
        *reinterpret_cast<float4*>(A0 + r * XS + 4 * c4) = v;
    }
    __syncthreads();
    auto layer = [&](const float* src, float* dst, int64_t w_off, int64_t b_off, int kbs) {     // dst = relu(src W^T + b)
        f32x4 acc[2][LCT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < LCT; ++nt) acc[mt][nt] = f32x4{0, 0, 0, 0};
        mm_cols<2, LCT, KM_LEGACY_ENC_PIN>(acc, src, blob + w_off, LCT * wave, kbs, 0, kbs, lane);
#pragma unroll
        for (int nt = 0; nt < LCT; ++nt) {
            const float bb = blob[b_off + col0 + 16 * nt + j];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[mt][nt][r] + bb;
                    dst[(16 * mt + 4 * g + r) * XS + col0 + 16 * nt + j] = v < 0.f ? 0.f : v;
                }
        }
    };
    layer(A, Bi, LG_W0, LG_B0, LG_MEL / 16);
    __syncthreads();
    layer(Bi, A, LG_W3, LG_B3, KBD);
    __syncthreads();
    auto project = [&](int64_t w_off, int64_t b_off, float* out) {                                // out rows = A W^T + b
        f32x4 acc[2][LCT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < LCT; ++nt) acc[mt][nt] = f32x4{0, 0, 0, 0};
        mm_cols<2, LCT, KM_LEGACY_ENC_PIN>(acc, A, blob + w_off, LCT * wave, KBD, 0, KBD, lane);
#pragma unroll
        for (int nt = 0; nt < LCT; ++nt) {
            const float bb = blob[b_off + col0 + 16 * nt + j];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = r0 + 16 * mt + 4 * g + r;
                    if (row < rows) out[row * D + col0 + 16 * nt + j] = acc[mt][nt][r] + bb;
                }
        }
    };
    project(LG_WK, LG_BK, Kp);
    project(LG_WV, LG_BV, Vp);
}


// ---------------------------------------------------------------------------------------------------------
// legacy_tail_kernel: SimplifiedKoeMorphModel after its attention, one 512-thread workgroup per window: out_proj, the decoder
// (Linear ReLU Linear ReLU Linear Sigmoid, simplified_model.py:63-72) on the 52 query rows resident in LDS and the mean over those
// rows (:144-147).  Replaces four NT GEMM launches + a row kernel.
// ---------------------------------------------------------------------------------------------------------
// window b of the launch; smem: the workgroup's dynamic LDS (2 x 64 x XS floats)
__device__ __forceinline__ void legacy_tail_body(const float* __restrict__ O, const float* __restrict__ blob, float* __restrict__ out, int64_t b,
                                                 float* smem) {
    float* A = smem;                 // [64][264] images; rows 52 .. 63 hold whatever the stores below leave (rows are independent)
    float* Bi = smem + 64 * XS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    for (int i = tid; i < 64 * 64; i += NTH) {
        const int q = i >> 6, c4 = i & 63;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < NQ) v = *reinterpret_cast<const float4*>(O + (b * NQ + q) * D + 4 * c4);
        *reinterpret_cast<float4*>(A + q * XS + 4 * c4) = v;
    }
    __syncthreads();
    {   // A1 = O Wo^T + bo: wave = 32 columns
        f32x4 acc[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { acc[mt][0] = f32x4{0, 0, 0, 0}; acc[mt][1] = f32x4{0, 0, 0, 0}; }
        mm_cols<4, 2, KM_KMMF_DEC_PIN>(acc, A, blob + LT_WO, 2 * wave, KBD, 0, KBD, lane);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float bb = blob[LT_BO + 32 * wave + 16 * nt + j];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Bi[(16 * mt + 4 * g + r) * XS + 32 * wave + 16 * nt + j] = acc[mt][nt][r] + bb;
        }
    }
    __syncthreads();
    auto hidden = [&](const float* src, float* dst, int64_t w_off, int64_t b_off, int kbs) {     // dst = relu(src W^T + b), 128 wide: wave = 16 columns
        f32x4 acc[4][1];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][0] = f32x4{0, 0, 0, 0};
        mm_cols<4, 1, KM_KMMF_DEC_PIN>(acc, src, blob + w_off, wave, kbs, 0, kbs, lane);
        const float bb = blob[b_off + 16 * wave + j];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[mt][0][r] + bb;
                dst[(16 * mt + 4 * g + r) * XS + 16 * wave + j] = v < 0.f ? 0.f : v;
            }
    };
    hidden(Bi, A, LT_W0, LT_B0, KBD);
    __syncthreads();
    hidden(A, Bi, LT_W3, LT_B3, HID / 16);
    __syncthreads();
    if (wave < 4) {      // the 52 outputs (padded to 64 columns): waves 0 - 3 own 16 columns each; mean of sigmoid over the 52 query rows
        f32x4 acc[4][1];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][0] = f32x4{0, 0, 0, 0};
        mm_cols<4, 1, KM_KMMF_DEC_PIN>(acc, Bi, blob + LT_W6, wave, HID / 16, 0, HID / 16, lane);
        const int col = 16 * wave + j;
        const float bb = blob[LT_B6 + col];
        float s = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = acc[mt][0][r] + bb;
                if (16 * mt + 4 * g + r < NQ) s += 1.0f / (1.0f + expf(-z));
            }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (g == 0 && col < NQ) out[b * NQ + col] = s / (float)NQ;
    }
}

__global__ __launch_bounds__(512) void legacy_tail_kernel(const float* __restrict__ O, const float* __restrict__ blob, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    legacy_tail_body(O, blob, out, (int64_t)blockIdx.x, smem);
}

// legacy_attn_tail_kernel: the attention of a window -- its 8 heads on the workgroup's 8 waves (legacy_attention_body: no LDS, no barrier) --
// and then its tail (legacy_tail_body), one launch: both were one-window-per-workgroup shaped already (the attention as two 4-wave
// workgroups per window), so the boundary between them only drained and refilled the chip.  O_b goes through memory, written and
// read by the same workgroup.
__global__ __launch_bounds__(512) void legacy_attn_tail_kernel(const float* __restrict__ Qs, const float* __restrict__ Kp,
                                                              const float* __restrict__ Vp, float* __restrict__ O, int Tm,
                                                              unsigned* __restrict__ zero_max, const float* __restrict__ blob,
                                                              float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t b = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    legacy_attention_body(Qs, Kp, Vp, O, b * 8 + wave, Tm, 8, NQ, zero_max);
    __syncthreads();      // the window's 52 x 256 attention output is in memory for every wave of this workgroup
    legacy_tail_body(O, blob, out, b, smem);
}

}  // namespace kf

static const float* dvp(Context* c, const char* name) { return c->packed.at(name).dev; }

bool koemorph_fused_ok(Context* c, int64_t B, int64_t T, const float* mel, const float* emo) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return c->kmm_fused && !c->opt.kmm_no_fuse && T >= 1 && T <= kmmf::TMAX && B >= 1 && al(mel) && al(emo);
}

// both streams of DualStreamEncoder: xm, xe (B, T, 256)
int launch_kmmf_encoder(Context* c, const float* mel, const float* emo, int64_t B, int64_t T, const unsigned char* kvalid, float* xm,
                        float* xe, void* stream) {
    static PerDeviceOnce once;
    if (once.first(c->device)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&kf::kmmf_encoder_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kf::ENC_LDS_FLOATS * 4));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&kf::kmmf_encoder_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kf::ENC_LDS_FLOATS * 4));
    }
    kf::EncArgs a{};
    a.in0 = mel; a.in1 = emo; a.in_dim0 = c->kmm.mel_dim; a.in_dim1 = c->kmm.emotion_dim;
    a.blob = dvp(c, "kmf_enc"); a.stream_floats = kmmf::enc_stream_floats(c->kmm.num_encoder_layers); a.layers = c->kmm.num_encoder_layers;
    a.kvalid = kvalid; a.out0 = xm; a.out1 = xe; a.B = (int)B; a.T = (int)T;
    while ((1 << a.slot_log2) < T) ++a.slot_log2;
    const int wpw = kf::EROWS >> a.slot_log2;
    a.groups = (int)((B + wpw - 1) / wpw);
    const dim3 grid((unsigned)(2 * a.groups));
    if (a.slot_log2 < 5) hipLaunchKernelGGL(kf::kmmf_encoder_kernel<true>, grid, dim3(kf::ENTH), kf::ENC_LDS_FLOATS * 4, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(kf::kmmf_encoder_kernel<false>, grid, dim3(kf::ENTH), kf::ENC_LDS_FLOATS * 4, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}


// everything after the encoders: average, queries, cross-attention layers, decoder, tail
int launch_kmmf_decode(Context* c, const float* xm, const float* xe, int64_t B, int64_t T, const unsigned char* kvalid, const float* prev,
                       float* attn, const KmmTail& tail, void* stream) {
    static PerDeviceOnce once;
    if (once.first(c->device))
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&kf::kmmf_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kf::DEC_LDS_FLOATS * 4));
    const km_koemorph_config& k = c->kmm;
    kf::DecArgs a{};
    a.xm = xm; a.xe = xe; a.emb = dvp(c, "query_embeddings.query_embeddings");
    a.cw0 = dvp(c, "query_embeddings.conditioning_net.0.weight"); a.cb0 = dvp(c, "query_embeddings.conditioning_net.0.bias");
    a.cw3 = dvp(c, "query_embeddings.conditioning_net.3.weight"); a.cb3 = dvp(c, "query_embeddings.conditioning_net.3.bias");
    a.prev = prev; a.cross = dvp(c, "kmf_cross"); a.cross_layers = k.num_attention_layers;
    a.dec = dvp(c, "kmf_dec"); a.dec_layers = k.decoder_layers;
    a.act = k.decoder_activation == 0 ? 1 : (k.decoder_activation == 1 ? 2 : (k.decoder_activation == 2 ? 3 : 4));
    a.kvalid = kvalid; a.causal = k.causal; a.window = k.window_size; a.attn = attn; a.B = (int)B; a.T = (int)T;
    a.tail = tail;
    hipLaunchKernelGGL(kf::kmmf_decode_kernel, dim3((unsigned)B), dim3(kf::NTH), kf::DEC_LDS_FLOATS * 4, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}


// SimplifiedKoeMorphModel: K, V (rows, 256) of the attention from the mel rows (rows, 80), one launch
int launch_legacy_encoder_fused(Context* c, const float* mel, int64_t rows, float* Kp, float* Vp, void* stream, const unsigned* melmax,
                                int Tm, const LogParams* lp) {
    static PerDeviceOnce once;
    if (once.first(c->device))
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&kf::legacy_encoder_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    2 * kf::LIMG * 4));
    hipLaunchKernelGGL(kf::legacy_encoder_kernel, dim3((unsigned)((rows + kf::LROWS - 1) / kf::LROWS)), dim3(kf::LNTH), 2 * kf::LIMG * 4,
                       (hipStream_t)stream, mel, dvp(c, "lgf_enc"), rows, Kp, Vp, melmax, Tm, lp ? LogParams(*lp) : LogParams{});
    HIP_TRY(hipGetLastError());
    return KM_OK;
}


// SimplifiedKoeMorphModel: attention output O (B, 52, 256) -> out (B, 52), one launch
int launch_legacy_tail_fused(Context* c, const float* O, int64_t B, float* out, void* stream) {
    static PerDeviceOnce once;
    if (once.first(c->device))
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&kf::legacy_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    2 * 64 * kf::XS * 4));
    hipLaunchKernelGGL(kf::legacy_tail_kernel, dim3((unsigned)B), dim3(kf::NTH), 2 * 64 * kf::XS * 4, (hipStream_t)stream, O, dvp(c, "lgf_tail"), out);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

// SimplifiedKoeMorphModel: attention (8 heads of 32, 52 queries) + everything behind it, one launch
int launch_legacy_attn_tail_fused(Context* c, const float* Kp, const float* Vp, float* O, int64_t B, int Tm, unsigned* zero_max, float* out,
                                  void* stream) {
    static PerDeviceOnce once;
    if (once.first(c->device))
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&kf::legacy_attn_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    2 * 64 * kf::XS * 4));
    hipLaunchKernelGGL(kf::legacy_attn_tail_kernel, dim3((unsigned)B), dim3(kf::NTH), 2 * 64 * kf::XS * 4, (hipStream_t)stream, dvp(c, "l_q"), Kp, Vp, O,
                       Tm, zero_max, dvp(c, "lgf_tail"), out);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

}  // namespace km

#ifdef KM_KMMF_STAMP
extern "C" __attribute__((visibility("default"))) int km_debug_kmmf_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(km::kf::km_kmmf_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif
