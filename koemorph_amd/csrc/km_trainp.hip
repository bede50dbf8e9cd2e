// Phased training step of the dual-stream core for gfx950 (BASELINE config C3).
//
// Same arithmetic as km_train.hip (unfolded forward with saved activations, KoeMorphLoss tail, backward into ONE flat
// gradient bucket), restructured for the shape the reference trains at: 8 windows per GPU.  There every product is a
// handful of 64 x 64 tiles, so the launch-per-op chain (~70 launches over two streams) was bound by launches and by the
// load -> LDS -> MFMA latency of each tiny kernel, not by arithmetic (round 1: 0.69 ms, 1 % of the MFMA peak).  Here the
// step is a PROGRAM of ~19 phases; a phase is ONE launch (phase_kernel) that runs every operation whose inputs are ready
// -- GEMM tiles, LayerNorm / softmax rows, column reductions -- side by side on different workgroups.  No side stream,
// no events, no atomics: every reduction is a fixed-order two-stage sum, so a step is bit-reproducible and capturable.
//
// Critical path shortened by algebra that leaves every gradient intact:
//   * out_proj -> mel_output_proj -> decoder[0] have nothing between them (dual_stream_attention.py:231, :248, :150), so the
//     forward needs only H = relu(A Wf^T + bf) with Wf = W1 Wmo Wo (computed per step, beside the encoder).  O1 = A Wo^T and
//     O2 = O1 Wmo^T are still produced -- the weight gradients dWmo = dO2^T O1, dW1 = dH^T O2 need them -- but beside the
//     chain, not in it; likewise dA = dH Wf replaces three dependent products in the backward chain.
//   * training-mode dropout (p = 0.1 in both nn.MultiheadAttention modules and the decoder, :106, :115, :153): masks are
//     drawn per step by a Philox4x32-10 generator into byte buffers (or supplied by the caller: parity tests feed the
//     masks of the reference-generated fixtures), applied in the softmax / GEMM epilogues.  With dropout the 24
//     expression rows of a window are no longer identical (each (head, query) keeps or drops its single attention
//     weight), so the emotion stream runs on 24 rows per window here.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstring>
#include <vector>

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

#include "km_gemm_dev.h"
#include "km_gemm_dma_dev.h"
#include "km_train_tail.h"
#include "km_train_attn_dev.h"

// ---- operations of a phase ------------------------------------------------------------------------------------
enum OpKind : int {
    OP_GEMM = 0, OP_ZERO, OP_FILL, OP_MASKGEN, OP_PACKX, OP_LOGPACK, OP_DBCONV, OP_COLSUM, OP_LN_FWD, OP_LN_BWD, OP_ATTN_FWD, OP_ATTN_BWD, OP_RELU_OUTER, OP_REDUCE,
    OP_EMO_EXPAND, OP_EMO_REDUCE, OP_REDUCE4, OP_PADROWS, OP_LNAPPLY
};


struct Op {
    int kind;
    int gx, gy;            // OP_GEMM: tiles along N and M (blocks = gx * gy * batch)
    int bm;                // OP_GEMM: tile rows (32 when the product would leave most of the chip idle with 64)
    signed char va, vb;    // OP_GEMM: 16-byte loads are legal for operand A / B
    signed char ln, pad_;  //          k > 0: the tiles leave LayerNorm parts for their output rows in Phase::ln[k - 1].stats; k < 0: operand A is read through the LayerNorm of Phase::ln[-k - 1] (LnXform, km_device.h)
    signed char dma, ns;   // OP_GEMM: the product runs on the LDS-DMA tile (km_gemm_dma_dev.h) with a ring of ns stages
    signed char ma, mb;    //          its operand modes (0 k-contiguous, 1 row-contiguous)
    union {
        GemmArgs g;
        ElemArgs e;
    };
};

constexpr int kMaxOps = 17;     // 8 + 17 * 4 + 17 * sizeof(Op) stays under the 4 KB kernel-argument limit
static_assert(sizeof(GemmArgs) <= 200, "Phase has to stay under the 4 KB kernel-argument limit");
struct Phase {
    int n_ops;
    int block_end[kMaxOps];
    Op ops[kMaxOps];
    DbXform xf;            // operand transform of the products with Op::dma == 2 (the channel encoder on the front end's packed rows)
    LnXform ln[2];         // LayerNorm-by-the-reader slots (Op::ln): mel rows, emotion rows
};
static_assert(sizeof(Phase) <= 4096, "Phase is passed by value: 4 KB of kernel arguments");

// Philox4x32-10 (Salmon et al. 2011): counter (c0..c3), key (k0, k1) -> 4 x 32 random bits
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ int drop_dz_index(int map, int64_t r, int& b_out) {
    // row of an activation buffer -> (window, blendshape index): map 1 = (window, mouth slot), 2 = (window, expression slot)
    if (map == 1) { const int64_t b = r / 28; const int q = (int)(r - b * 28); b_out = (int)b; return q < 27 ? 14 + q : 51; }
    const int64_t b = r / 24; const int q = (int)(r - b * 24); b_out = (int)b; return q < 14 ? q : q + 27;
}

// global -> LDS staging of n elements, element i read from src(i) and written to dst(i): the loads of eight elements per
// thread are issued before the first LDS store.  The LDS image is reached through a generic pointer here, so the compiler
// must assume a store may alias the next load and would otherwise serialise load -> store -> load (a full memory round trip
// per element: 20 - 30 of them made the attention blocks the long poles of their phases).
template <typename Src, typename Dst>
__device__ __forceinline__ void stage_to_lds(int n, Src src, Dst dst) {
    for (int i0 = threadIdx.x; i0 < n; i0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + 256 * u; v[u] = i < n ? src(i) : 0.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + 256 * u; if (i < n) dst(i, v[u]); }
    }
}

// vb / tid: block and thread index of the operation's 256-thread block
// LayerNorm of one row by one wave.  NP = d / 64 values per lane held in registers: every load of the row is in flight at once
// (as three loops over the row with a run-time trip count the first one was d / 64 dependent L2 round trips); NP = 0: any d,
// the loops.  Lane l owns elements l, l + 64, ... and adds them in that order either way: the same bits.
template <int NP>
__device__ __forceinline__ void ln_fwd_row(const ElemArgs& a, int64_t row, int lane) {
    const int d = a.i0;
    const float* p = a.p0 + row * d;
    constexpr int NR = NP > 0 ? NP : 1;
    float x[NR], gm[NR], bt[NR];
    float s = 0.f;
    if constexpr (NP > 0) {
#pragma unroll
        for (int u = 0; u < NP; ++u) { x[u] = p[lane + 64 * u]; gm[u] = a.p1[lane + 64 * u]; bt[u] = a.p2[lane + 64 * u]; }
#pragma unroll
        for (int u = 0; u < NP; ++u) s += x[u];
    } else {
        for (int i = lane; i < d; i += 64) s += p[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / d;
    float v = 0.f;
    if constexpr (NP > 0) {
#pragma unroll
        for (int u = 0; u < NP; ++u) { const float t = x[u] - mean; v += t * t; }
    } else {
        for (int i = lane; i < d; i += 64) { const float t = p[i] - mean; v += t * t; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const float rstd = 1.0f / sqrtf(v / d + 1e-5f);
    if constexpr (NP > 0) {
#pragma unroll
        for (int u = 0; u < NP; ++u) a.q0[row * d + lane + 64 * u] = (x[u] - mean) * rstd * gm[u] + bt[u];
    } else {
        for (int i = lane; i < d; i += 64) a.q0[row * d + i] = (p[i] - mean) * rstd * a.p1[i] + a.p2[i];
    }
    if (lane == 0) { a.q1[row] = mean; a.q2[row] = rstd; }
}

template <int NP>
__device__ __forceinline__ void ln_bwd_row(const ElemArgs& a, int64_t row, int lane) {
    const int d = a.i0;
    const float mu = a.p3[row], rs = a.p4[row];
    constexpr int NR = NP > 0 ? NP : 1;
    float dy[NR], xh[NR], gm[NR];
    float s1 = 0.f, s2 = 0.f;
    if constexpr (NP > 0) {
#pragma unroll
        for (int u = 0; u < NP; ++u) { dy[u] = a.p0[row * d + lane + 64 * u]; xh[u] = a.p1[row * d + lane + 64 * u]; gm[u] = a.p2[lane + 64 * u]; }
        if (a.n1) {         // dy arrives as TWO partial products (the K halves of dY = dKV Wkv, n1 floats apart: P10 at small batches): their sum
                            // is the gradient, and it is written back (q2) for the phase-12 column sum
            float dyb[NR];
#pragma unroll
            for (int u = 0; u < NP; ++u) dyb[u] = a.p0[a.n1 + row * d + lane + 64 * u];
#pragma unroll
            for (int u = 0; u < NP; ++u) { dy[u] += dyb[u]; a.q2[row * d + lane + 64 * u] = dy[u]; }
        }
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            xh[u] = (xh[u] - mu) * rs;
            const float dxh = dy[u] * gm[u];
            s1 += dxh; s2 += dxh * xh[u];
        }
    } else {
        for (int i = lane; i < d; i += 64) {
            const float x = (a.p1[row * d + i] - mu) * rs, dxh = a.p0[row * d + i] * a.p2[i];
            s1 += dxh; s2 += dxh * x;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    s1 /= d; s2 /= d;
    if constexpr (NP > 0) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const float dxh = dy[u] * gm[u];
            a.q0[row * d + lane + 64 * u] = rs * (dxh - s1 - xh[u] * s2);
            a.q1[row * d + lane + 64 * u] = dy[u] * xh[u];                     // column sums of this image = d gamma (a GEMM with a ones vector)
        }
    } else {
        for (int i = lane; i < d; i += 64) {
            const float dyv = a.p0[row * d + i];
            const float x = (a.p1[row * d + i] - mu) * rs, dxh = dyv * a.p2[i];
            a.q0[row * d + i] = rs * (dxh - s1 - x * s2);
            a.q1[row * d + i] = dyv * x;
        }
    }
}

// p[0] + p[stride] + ... in index order, eight loads in flight at a time (as a plain loop this is one dependent L2 round trip per
// term: the 64 per-window parts of dQ took 17.5 us at 64 windows, more than any product of their phase)
__device__ __forceinline__ float sum_rows_in_order(const float* p, int64_t stride, int n) {
    float s = 0.f;
    int y = 0;
    for (; y + 8 <= n; y += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = p[(int64_t)(y + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; y < n; ++y) s += p[(int64_t)y * stride];
    return s;
}

template <bool ATTN>
__device__ __forceinline__ void op_elem(const Op& op, int vb, int tid, float* smem) {
    const ElemArgs& a = op.e;
    const int lane = tid & 63, wv = tid >> 6;
    switch (op.kind) {
    case OP_ZERO: {         // 4096 floats per workgroup (four 16-byte stores per thread, each a contiguous 4 KB run of the workgroup)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = ((int64_t)vb * 1024 + u * 256 + tid) * 4;
            if (i + 3 < a.n0) *reinterpret_cast<float4*>(a.q0 + i) = make_float4(0.f, 0.f, 0.f, 0.f);
            else for (int64_t k = i; k < a.n0; ++k) a.q0[k] = 0.f;
        }
        break;
    }
    case OP_FILL: {
        const int64_t i = (int64_t)vb * 256 + tid;
        if (i < a.n0) a.q0[i] = a.f0;
        break;
    }
    case OP_MASKGEN: {      // mask_out[n0] bytes: keep (1) with probability 1 - p; counter = (byte index / 4, region, step)
        const int64_t i4 = (int64_t)vb * 256 + tid;
        if (i4 * 4 >= a.n0) break;
        const unsigned step = a.p0 ? (unsigned)reinterpret_cast<const int*>(a.p0)[0] : 0u;
        unsigned r[4];
        philox4x32_10((unsigned)i4, (unsigned)(i4 >> 32), (unsigned)a.i0, step, a.u0, a.u1, r);
        const unsigned thr = (unsigned)a.i1;                   // p * 2^32: keep <=> r >= thr
        unsigned char k4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) k4[k] = r[k] >= thr ? 1 : 0;
        if (i4 * 4 + 3 < a.n0) *reinterpret_cast<uchar4*>(a.mask_out + i4 * 4) = make_uchar4(k4[0], k4[1], k4[2], k4[3]);
        else for (int k = 0; i4 * 4 + k < a.n0; ++k) a.mask_out[i4 * 4 + k] = k4[k];
        break;
    }
    case OP_PACKX: {        // xt (B, NK, KP) <- mel (B, t_in, NK) rows [0, tv), zeros to T, 3 short rows, zeros to KP -- TRANSPOSED:
        // channel rows of KP frames (see train_forward_backward_phased: both products that read it then run on the LDS-DMA tile)
        const int64_t i = (int64_t)vb * 256 + tid;           // one float4 of channels each
        const int nk4 = a.i0 / 4, KP = a.i1, T = a.i2, t_in = a.i3;
        if (i >= a.n0 * KP * nk4) break;
        const int64_t row = i / nk4; const int c4 = (int)(i - row * nk4);
        const int64_t b = row / KP; const int t = (int)(row - b * KP);
        const int tv = t_in < T ? t_in : T;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < tv) v = reinterpret_cast<const float4*>(a.p0 + (b * t_in + t) * a.i0)[c4];
        else if (t >= T && t < T + 3) v = reinterpret_cast<const float4*>(a.p1 + (b * 3 + (t - T)) * a.i0)[c4];
        float* o = a.q0 + (b * a.i0 + 4 * c4) * KP + t;
        o[0] = v.x; o[KP] = v.y; o[2 * KP] = v.z; o[3 * KP] = v.w;
        break;
    }
    case OP_LOGPACK: {      // xt (B, NK, KP) <- the front end's power-mel (B, F, NK) + window maxima: dB / log conversion and row
        // packing (T long frames, the last 3 frames, zeros to KP) -- mel_log_packed_kernel as an op of phase 0, one launch less
        const int64_t i = (int64_t)vb * 256 + tid;           // one value each, as the stand-alone kernel (same log_one())
        const int NKk = a.i0, KP = a.i1, T = a.i2, F = a.i3;
        if (i >= a.n0 * KP * NKk) break;
        const int64_t b = i / ((int64_t)KP * NKk);
        const int j = (int)(i - b * KP * NKk), r = j / NKk, m = j - r * NKk;
        float ref_db, floor_db;
        log_window_consts(a.lp, __uint_as_float(reinterpret_cast<const unsigned*>(a.p1)[b]), ref_db, floor_db);
        int f = -1;
        if (r < T) f = r < F ? r : -1;
        else if (r < T + 3) {
            const int q = r - T;
            if (F >= 3) f = F - 3 + q; else if (q < F) f = q;
        }
        a.q0[(b * NKk + m) * KP + r] = f >= 0 ? log_one(a.lp, a.p0[(b * F + f) * NKk + m], ref_db, floor_db) : 0.f;
        break;
    }
    case OP_DBCONV: {       // q0 (B, NK, KP) <- p0: the front end's 10 log10(power) rows finished into dB features (db_finish with the
        // window's reference p1[b]; columns >= T + 3 zero) for the readers of the backward pass; the channel encoder beside this
        // operation converts its own fragments (DbXform).  Four values per thread.
        // i3 workgroups per window, 16 values per thread: the window (hence its reference) is workgroup-uniform and all index
        // arithmetic is 32-bit (as one flat 64-bit index with a division per value this operation took 3.5 us at 64 windows)
        const unsigned NKk = a.i0, KP = a.i1, T = a.i2, kp4 = KP >> 2, n4 = NKk * kp4;
        const unsigned b = (unsigned)vb / (unsigned)a.i3, e0 = ((unsigned)vb - b * a.i3) * 1024u + tid;
        const float4* src = reinterpret_cast<const float4*>(a.p0) + (int64_t)b * n4;
        float4* dst = reinterpret_cast<float4*>(a.q0) + (int64_t)b * n4;
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = src[e0 + u * 256 < n4 ? e0 + u * 256 : 0];
        float ref_db, floor_db;
        log_window_consts(a.lp, __uint_as_float(reinterpret_cast<const unsigned*>(a.p1)[b]), ref_db, floor_db);
        const float c1 = db_fast_c1(a.lp, ref_db);       // the channel encoder's own transform, bit for bit
        auto one = [&](float x, unsigned rr) { return rr < T + 3 ? db_finish_fast(x, a.lp.db_scale, c1) : 0.f; };
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned e = e0 + u * 256;
            if (e >= n4) break;
            const unsigned r = (e % kp4) * 4;
            dst[e] = make_float4(one(v[u].x, r), one(v[u].y, r + 1), one(v[u].z, r + 2), one(v[u].w, r + 3));
        }
        break;
    }
    case OP_COLSUM: {       // q0[chunk][c] (+)= sum over the chunk's rows r of w[r] m[r][c]: 16 columns x 16 row groups per workgroup, eight
        // loads in flight per thread, a fixed-order sum over the row groups.  The bias / LayerNorm-parameter / w2 gradients used to be
        // products with a ones (or per-row weight) vector on the matrix pipe: 32-row tiles for ONE useful row, 8.6 % of P10's MFMA
        // time at 64 windows
        const int n = a.i0, ncg = (n + 15) >> 4, S = a.i2;
        const int chunk = vb / ncg, cg = vb - chunk * ncg;
        const int c = 16 * cg + (tid & 15), rg = tid >> 4;
        const int64_t rows = a.n0, rs = a.i1;
        const int64_t rpc = (rows + S - 1) / S, lo = chunk * rpc, hi = lo + rpc < rows ? lo + rpc : rows;
        const int cc = c < n ? c : n - 1;
        float s = 0.f;
        for (int64_t r = lo + rg; r < hi; r += 128) {
            float t[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t rr = r + 16 * u < hi ? r + 16 * u : hi - 1;
                t[u] = a.p0[rr * rs + cc];
                w[u] = a.p1 ? a.p1[rr] : 1.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += r + 16 * u < hi ? t[u] * w[u] : 0.f;
        }
        smem[tid] = s;
        __syncthreads();
        if (rg == 0 && c < n) {
            float tot = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) tot += smem[16 * g + (tid & 15)];
            float* o = a.q0 + (int64_t)chunk * n + c;
            *o = a.i3 ? *o + tot : tot;
        }
        break;
    }
    case OP_PADROWS: {      // q0 (n0 rows x i1) <- p0 (n0 rows x i0), zeros beyond column i0: the channel encoder weight with rows of KP
        const int64_t i = (int64_t)vb * 256 + tid;
        if (i >= a.n0 * a.i1) break;
        const int64_t r = i / a.i1; const int t = (int)(i - r * a.i1);
        a.q0[i] = t < a.i0 ? a.p0[r * a.i0 + t] : 0.f;
        break;
    }
    case OP_LNAPPLY: {      // y = LayerNorm(x) from the producer's per-row parts (p3: float2 [row][i1]), statistics saved; one wave per row.
        // The readers that normalise on the fly (gemm_tile_dma_dev XFA == 2) use the same ln_combine / ln_apply: the same bits.
        const int64_t row = (int64_t)vb * 4 + wv;
        const int d = a.i0;
        if (row >= a.n0) break;
        float rs, nmurs, mean;
        ln_combine(reinterpret_cast<const float2*>(a.p3) + row * a.i1, a.i1, a.f0, rs, nmurs, &mean);
        for (int i = lane; i < d; i += 64) a.q0[row * d + i] = ln_apply(a.p0[row * d + i], rs, nmurs, a.p1[i], a.p2[i]);
        if (lane == 0) { a.q1[row] = mean; a.q2[row] = rs; }
        break;
    }
    case OP_LN_FWD: {       // y = LayerNorm(x), statistics saved; one wave per row
        const int64_t row = (int64_t)vb * 4 + wv;
        const int d = a.i0;
        if (row >= a.n0) break;
        if (d == 256) ln_fwd_row<4>(a, row, lane); else if (d == 512) ln_fwd_row<8>(a, row, lane);
        else if (d == 64) ln_fwd_row<1>(a, row, lane); else ln_fwd_row<0>(a, row, lane);
        break;
    }
    case OP_LN_BWD: {       // dx = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)), dxhat = dy gamma; out of place
        const int64_t row = (int64_t)vb * 4 + wv;
        const int d = a.i0;
        if (row >= a.n0) break;
        if (d == 256) ln_bwd_row<4>(a, row, lane); else if (d == 512) ln_bwd_row<8>(a, row, lane);
        else if (d == 64) ln_bwd_row<1>(a, row, lane); else ln_bwd_row<0>(a, row, lane);
        break;
    }
    case OP_RELU_OUTER: {   // dHpre[r][m] = g[r] w2[m] scale [H[r][m] > 0]; H is post-ReLU, post-dropout (H > 0 <=> kept and active)
        const int64_t i = (int64_t)vb * 256 + tid;
        const int n = a.i0;
        if (i >= a.n0 * n) break;
        const int64_t r = i / n; const int m = (int)(i - r * n);
        a.q0[i] = a.p1[i] > 0.f ? a.p0[r] * a.p2[m] * a.f0 : 0.f;
        break;
    }
    case OP_REDUCE: {       // out[c] (+)= sum_y part[y][c], fixed order
        const int64_t c0 = (int64_t)vb * 256 + tid;
        if (c0 >= a.n0) break;
        a.q0[c0] = (a.i1 ? a.q0[c0] : 0.f) + sum_rows_in_order(a.p0 + c0, a.n1, a.i0);
        break;
    }
    case OP_REDUCE4: {      // the workgroup's blocks are dealt to the (up to four) reductions in order
        const int b0 = (a.rn0 + 255) / 256, b1 = b0 + (a.rn1 + 255) / 256, b2 = b1 + (a.rn2 + 255) / 256;
        const int j = vb < b0 ? 0 : (vb < b1 ? 1 : (vb < b2 ? 2 : 3));                                  // workgroup-uniform
        const int base = j == 0 ? 0 : (j == 1 ? b0 : (j == 2 ? b1 : b2));
        const float* part = j == 0 ? a.p0 : (j == 1 ? a.p1 : (j == 2 ? a.p2 : a.p3));
        float* out = j == 0 ? a.q0 : (j == 1 ? a.q1 : (j == 2 ? a.q2 : a.q3));
        const int n = j == 0 ? a.rn0 : (j == 1 ? a.rn1 : (j == 2 ? a.rn2 : a.rn3));
        const int S = j == 0 ? a.rS0 : (j == 1 ? a.rS1 : (j == 2 ? a.rS2 : a.rS3));
        const int acc = (a.racc >> j) & 1;
        const int64_t c0 = (int64_t)(vb - base) * 256 + tid;
        if (c0 >= n) break;
        out[c0] = (acc ? out[c0] : 0.f) + sum_rows_in_order(part + c0, n, S);
        break;
    }
    case OP_EMO_EXPAND: {   // Ae[(b, q), c] = Ve[b, c] * keep[b, c / hd, q] / (1 - p): the one-key attention of the emotion stream
        // a workgroup = 256 columns of one (window, query) row: the row is workgroup-uniform (two 64-bit divisions per element before)
        const int d = a.i0, hd = a.i1, H = d / hd, bpr = (d + 255) >> 8;
        const int r = vb / bpr, cidx = (vb - r * bpr) * 256 + tid;
        if (cidx >= d) break;
        const int b = r / 24, q = r - b * 24;
        const float v = a.p0[(int64_t)b * d + cidx];
        a.q0[(int64_t)r * d + cidx] = a.mask ? (a.mask[((int64_t)b * H + cidx / hd) * 24 + q] ? v * a.f0 : 0.f) : v;
        break;
    }
    case OP_EMO_REDUCE: {   // dVe[b, c] = sum_q dAe[(b, q), c] * keep / (1 - p)
        const int64_t i = (int64_t)vb * 256 + tid;
        const int d = a.i0, hd = a.i1, H = d / hd;
        if (i >= a.n0 * d) break;
        const int64_t b = i / d; const int cidx = (int)(i - b * d);
        float gq[24];
        unsigned char kq[24];
#pragma unroll
        for (int q = 0; q < 24; ++q) { gq[q] = a.p0[(b * 24 + q) * d + cidx]; kq[q] = 1; }
        if (a.mask) {
#pragma unroll
            for (int q = 0; q < 24; ++q) kq[q] = a.mask[(b * H + cidx / hd) * 24 + q];
        }
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 24; ++q) s += a.mask ? (kq[q] ? gq[q] * a.f0 : 0.f) : gq[q];      // the same order as the loop it replaces
        a.q0[i] = s;
        break;
    }
    case OP_ATTN_FWD:
        if constexpr (!ATTN) break;
        else if (a.i1 == 32 && a.u1 == 0 && a.i2 > 64 && a.i2 <= 80) attn_fwd_dma32_dev<5>(a, vb, smem);      // 80 mel channels; u1 != 0: option train_attn_regs (the register-staged block)
        else if (a.i1 == 32) attn_fwd_mfma_dev<32>(a, vb, smem); else if (a.i1 == 64) attn_fwd_mfma_dev<64>(a, vb, smem); else attn_fwd_mfma_dev<16>(a, vb, smem);
        break;
    case OP_ATTN_BWD:
        if constexpr (!ATTN) break;
        else if (a.i1 == 32 && a.u1 == 0 && a.i2 > 64 && a.i2 <= 80) attn_bwd_dma32_dev<5>(a, vb, smem);
        else if (a.i1 == 32) attn_bwd_mfma_dev<32>(a, vb, smem); else if (a.i1 == 64) attn_bwd_mfma_dev<64>(a, vb, smem); else attn_bwd_mfma_dev<16>(a, vb, smem);
        break;
    default: break;
    }
}

// Occupancy is the lever of these phases: a workgroup is four waves (one per SIMD) that stop at a barrier every k-step, so what
// hides a wave's LDS and L2 round trips is the number of OTHER workgroups on the CU.  Two instantiations: phases without an
// attention block run with 2 k-tiles of operand loads in flight per thread (4 before) under a 96-register cap = 5 workgroups
// per CU (the 160 KB of LDS allow exactly five 32 KB tile buffers); the two attention phases need more registers and LDS.
// Measured at 8 / 64 windows per step: 3 waves per SIMD, depth 4 (round-3 start) 0.2122 / 0.444 ms; 4 waves (27 registers
// spilled) 0.2057 / 0.428; 4 waves, depth 2 (no spills) 0.2022 / 0.4055; 5 waves, depth 2 0.1984 / 0.3998.
#ifndef KM_TRAINP_D
#define KM_TRAINP_D 2          /* k-tiles of operand loads in flight per thread (gemm_tile_dev) */
#endif
#ifndef KM_TRAINP_WAVES
#define KM_TRAINP_WAVES 4      /* waves per SIMD the phases without attention blocks are compiled for */
#endif
#ifndef KM_TRAINP_XCD_REMAP
#define KM_TRAINP_XCD_REMAP 1
#endif
#ifndef KM_TRAINP_WAVES_ATTN
#define KM_TRAINP_WAVES_ATTN 2
#endif
// The 4 KB Phase is never touched as a by-value object: its words are read through the kernel-argument segment pointer
// (constant address space, uniform offsets: scalar loads).
// Indexing the by-value argument at run time works only as long as the compiler manages to lower it to such loads itself --
// beyond some amount of inlined code (round 3: both attention block families; round 4: the LDS-DMA tiles) it copies the whole
// argument into scratch memory instead, 4 KB per thread, and every phase runs several times slower.
typedef const __attribute__((address_space(4))) unsigned* KernargWords;
static_assert(sizeof(Op) % 4 == 0 && offsetof(Phase, ops) % 8 == 0, "operations are addressed in words, pointers inside stay 8-byte aligned");

template <bool ATTN>
__global__ __launch_bounds__(256, ATTN ? KM_TRAINP_WAVES_ATTN : KM_TRAINP_WAVES) void phase_kernel(Phase p_arg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // the host sizes it for the phase's largest operation
    (void)p_arg;
    KernargWords ka = (KernargWords)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int W_NOPS = offsetof(Phase, n_ops) / 4, W_END = offsetof(Phase, block_end) / 4,
                  W_OPS = offsetof(Phase, ops) / 4, W_OP = sizeof(Op) / 4;
    const int vb = blockIdx.x, n_ops = (int)ka[W_NOPS];
    int i = 0;
    while (i + 1 < n_ops && vb >= (int)ka[W_END + i]) ++i;            // workgroup-uniform
    const int start = i ? (int)ka[W_END + i - 1] : 0;
    int local = vb - start;
    // a generic pointer derived from the kernel-argument segment pointer: the address-space inference pass turns every access
    // through it back into a constant-address-space (scalar) load, and there is no local object the compiler could spill
    const Op& op = *(const Op*)(ka + W_OPS + i * W_OP);
    if (op.kind == OP_GEMM) {
#if KM_TRAINP_XCD_REMAP
        {   // workgroups are dealt round-robin over the 8 XCDs (observed, MI355X_MICROARCH.md "Workgroup dispatch"): blocks with the
            // same id mod 8 share an L2.  Give each such group a CONTIGUOUS run of the product's tiles (column tiles of one row
            // block, the tiles of one K chunk of a split product) so that the operands they share are fetched into one L2, once.
            // Bijective for any tile count; which workgroup computes which tile changes, the results do not.
            const int n = (int)ka[W_END + i] - start, q8 = n >> 3, r8 = n & 7, x = local & 7;
            local = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (local >> 3);
        }
#endif
        const int per = op.gx * op.gy;
        const int bz = local / per, t = local - bz * per;
        const int tx = t % op.gx, ty = t / op.gx;
        const LnXform* lnx = op.ln ? (const LnXform*)(ka + offsetof(Phase, ln) / 4) + ((op.ln < 0 ? -op.ln : op.ln) - 1) : nullptr;
        float* stats = op.ln > 0 ? const_cast<float*>(lnx->stats) : nullptr;
        if (op.dma == 2) {       // the channel encoder on the front end's 10 log10(power) rows (host: operand modes (0, 0))
            const DbXform* xf = (const DbXform*)(ka + offsetof(Phase, xf) / 4);
            if (op.bm == 32 && op.ns == 8) gemm_tile_dma_dev<32, 8, 0, 0, 1>(op.g, tx, ty, bz, smem, xf, nullptr, stats);
            else if (op.bm == 32) gemm_tile_dma_dev<32, KM_DMA_NS32, 0, 0, 1>(op.g, tx, ty, bz, smem, xf, nullptr, stats);
            else gemm_tile_dma_dev<64, KM_DMA_NS64, 0, 0, 1>(op.g, tx, ty, bz, smem, xf, nullptr, stats);
        } else if (op.dma && op.ln < 0) {       // a reader of LayerNorm rows (host: operand modes (0, 0))
            if (op.bm == 32 && op.ns == 8) gemm_tile_dma_dev<32, 8, 0, 0, 2>(op.g, tx, ty, bz, smem, nullptr, lnx);
            else if (op.bm == 32) gemm_tile_dma_dev<32, KM_DMA_NS32, 0, 0, 2>(op.g, tx, ty, bz, smem, nullptr, lnx);
            else gemm_tile_dma_dev<64, KM_DMA_NS64, 0, 0, 2>(op.g, tx, ty, bz, smem, nullptr, lnx);
        } else if (op.dma) {
            if (op.bm == 32 && op.ns == 8) {
                if (op.ma == 0 && op.mb == 0) gemm_tile_dma_dev<32, 8, 0, 0>(op.g, tx, ty, bz, smem, nullptr, nullptr, stats);
                else if (op.ma == 0) gemm_tile_dma_dev<32, 8, 0, 1>(op.g, tx, ty, bz, smem);
                else gemm_tile_dma_dev<32, 8, 1, 1>(op.g, tx, ty, bz, smem);
            } else if (op.bm == 32) {
                if (op.ma == 0 && op.mb == 0) gemm_tile_dma_dev<32, KM_DMA_NS32, 0, 0>(op.g, tx, ty, bz, smem, nullptr, nullptr, stats);
                else if (op.ma == 0) gemm_tile_dma_dev<32, KM_DMA_NS32, 0, 1>(op.g, tx, ty, bz, smem);
                else gemm_tile_dma_dev<32, KM_DMA_NS32, 1, 1>(op.g, tx, ty, bz, smem);
            } else {
                if (op.ma == 0 && op.mb == 0) gemm_tile_dma_dev<64, KM_DMA_NS64, 0, 0>(op.g, tx, ty, bz, smem, nullptr, nullptr, stats);
                else if (op.ma == 0) gemm_tile_dma_dev<64, KM_DMA_NS64, 0, 1>(op.g, tx, ty, bz, smem);
                else gemm_tile_dma_dev<64, KM_DMA_NS64, 1, 1>(op.g, tx, ty, bz, smem);
            }
        } else
        // operand B of the channel encoder (rows of 259 floats) is the one product without 16-byte rows
        if (op.bm == 32) {
            if (op.va && op.vb) gemm_tile_dev<32, KM_TRAINP_D, true, true>(op.g, tx, ty, bz, smem);
            else if (op.va) gemm_tile_dev<32, KM_TRAINP_D, true, false>(op.g, tx, ty, bz, smem);
            else gemm_tile_dev<32, KM_TRAINP_D, false, false>(op.g, tx, ty, bz, smem);
        } else {
            if (op.va && op.vb) gemm_tile_dev<64, KM_TRAINP_D, true, true>(op.g, tx, ty, bz, smem);
            else if (op.va) gemm_tile_dev<64, KM_TRAINP_D, true, false>(op.g, tx, ty, bz, smem);
            else gemm_tile_dev<64, KM_TRAINP_D, false, false>(op.g, tx, ty, bz, smem);
        }
    } else {
        op_elem<ATTN>(op, local, (int)threadIdx.x, smem);
    }
}

// 1024 threads per workgroup: its 16 waves compute the decoder logits of the workgroup's windows, then the loss tail proper
__global__ __launch_bounds__(1024) void trainp_tail_kernel(TailArgs a) { train_tail_dev<16>(a); }
// one window per 256-thread workgroup: hidden rows, logits and dL/dz stay in LDS / registers from the logits to dH
__global__ __launch_bounds__(256) void trainp_tail_window_kernel(TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    train_tail_window_dev(a, smem);
}
constexpr int kTailMaxGroups = 1024;      // rows of Context::trp_tail_part

// ---- host: building and launching the program ----------------------------------------------------------------
// Split-K: a gradient product over all rows of the batch (K = 80 B, 28 B or 24 B) with an output of a few tiles is a few
// long serial chains on a few CUs -- at 64 windows K = 5120 is 160 k-steps on 32 of 256 CUs and sets the duration of its
// phase.  Such a product is cut along K into S batch entries that write S partial outputs; the partials are summed in a
// fixed order (OP_REDUCE4) by the NEXT phase, which is early enough: nothing but the optimizer reads a parameter gradient.
// Products of up to kSplitMinK stay whole.  Round 4: 512 (1024 before) with chains of ~512 rows -- on the LDS-DMA tile a k-step is
// short enough that the K = 640 products of the 8-window step are worth cutting in two (0.1339 -> 0.1322 ms), and the finer
// cut helps at 64 windows too (0.2906 -> 0.2828 ms).
constexpr int kSplitMinK = 512;
struct PendingReduce { const float* part; float* out; int n, S, acc; };

struct Program {
    Phase cur{};
    int nb[kMaxOps] = {};                // blocks of each operation in units of 256-thread blocks (products: per output batch entry)
    int gbatch[kMaxOps] = {};            // products: output batch entries
    int blocks = 0;
    int launches = 0;
    bool use_dma = true, any_dma = false, dma_bm64 = false;      // LDS-DMA tiles (option train_no_dma switches them off)
    int alone_max = 0;                   // see launch_cur: phases of up to this many workgroups get 8-stage rings (option train_alone_max; 0: never)
    int rc = KM_OK;
    size_t lds = 0;
    bool has_attn = false;        // the phase holds an attention block: phase_kernel<true>
    float* scratch = nullptr;            // bump allocator for the partials of a step
    int64_t scratch_left = 0;
    std::vector<PendingReduce> pend_prev, pend_cur;
    bool allow_split = true;
    DbXform xf{};                 // copied into every phase's arguments (Phase::xf)
    LnXform ln[2] = {};           // ... (Phase::ln)
    const float* leaf_lo = nullptr; const float* leaf_hi = nullptr;      // the gradient bucket: outputs inside it may be split along K
    // S for a product of K rows with `tiles` output tiles, or 1
    int min_k = kSplitMinK, chain = 512;
    int bm32_below = 192;                // products with fewer 64-row tiles than this run on 32-row tiles
    bool op_per_launch = false; hipStream_t dbg_stream = nullptr;
    int split_factor(int64_t K, int tiles) const {
        if (K <= min_k || tiles >= 128) return 1;
        const int want = (int)((K + chain - 1) / chain);
        int best = 1;
        for (int S = 2; S <= 16; ++S)
            if (K % S == 0 && K / S >= 128 && tiles * S <= 512 && std::abs(S - want) < std::abs(best - want)) best = S;
        return best;
    }
    // op_lds: dynamic LDS the operation needs (the LDS-DMA tiles' rings are sized in launch_cur)
    void add(const Op& op, int nblocks, size_t op_lds = 0) {
        if (nblocks <= 0) return;
        // a full phase (the 4 KB kernel argument holds kMaxOps operations) is launched and continued in a second launch: at large
        // batches the reductions of the previous phase's split products join the phase's own operations
        if (cur.n_ops >= kMaxOps && !op_per_launch) { if (int r = launch_cur(dbg_stream)) { rc = r; return; } }
        if (cur.n_ops >= kMaxOps) { rc = fail(KM_ERR_UNSUPPORTED, "training program: more than %d operations in one phase", kMaxOps); return; }
        need_lds(op_lds);
        if (op.kind == OP_ATTN_FWD || op.kind == OP_ATTN_BWD) has_attn = true;
        if (op.kind == OP_GEMM && op.dma) { if (op.bm == 64) dma_bm64 = true; else any_dma = true; }
        cur.ops[cur.n_ops] = op;
        nb[cur.n_ops] = nblocks; gbatch[cur.n_ops] = 1;
        blocks += nblocks;
        cur.block_end[cur.n_ops] = blocks;
        ++cur.n_ops;
        if (op_per_launch) { if (int r = launch_cur(dbg_stream)) rc = r; }      // timing aid (option train_op_per_launch): every operation is its own launch, in program order
    }
    // out (n) (+)= sum_r w[r] m[r][.] (w null: ones) as OP_COLSUM; long sums into the gradient bucket are cut into row chunks whose
    // partial rows the next phase adds up (the split-K machinery of the products)
    void colsum(const float* m, int64_t rows, int64_t rs, int64_t n, float* out, int accumulate, const float* w) {
        const bool leaf = out >= leaf_lo && out < leaf_hi;
        int S = 1;
        if (allow_split && leaf && rows > min_k) { S = (int)((rows + chain / 2 - 1) / (chain / 2)); S = S > 16 ? 16 : S; }
        bool conflict = false;
        for (const auto& r : pend_prev) conflict = conflict || r.out == out;
        for (const auto& r : pend_cur) conflict = conflict || r.out == out;
        if (conflict && !(allow_split && leaf)) { rc = fail(KM_ERR_UNSUPPORTED, "training program: a column sum writes an output with a pending reduction"); return; }
        ElemArgs e{};
        e.p0 = m; e.p1 = w; e.q0 = out; e.n0 = rows; e.i0 = (int)n; e.i1 = (int)rs; e.i2 = S; e.i3 = accumulate;
        if (S > 1 || conflict) {
            if ((int64_t)S * n + 4 > scratch_left) {
                if (conflict) { rc = fail(KM_ERR_WORKSPACE, "training program: no room for the partial of a column sum whose output has a pending reduction"); return; }
                S = 1; e.i2 = 1;
            } else {
                float* part = scratch;
                scratch += ((int64_t)S * n + 3) / 4 * 4; scratch_left -= ((int64_t)S * n + 3) / 4 * 4;
                pend_cur.push_back({part, out, (int)n, S, accumulate ? 1 : 0});
                e.q0 = part; e.i3 = 0;
            }
        }
        // behind the phase's products in dispatch order: these short, latency-bound workgroups hold a tile slot each (the phase's
        // LDS size is every workgroup's) -- in the middle of P10 they made the phase 3 us LONGER at 64 windows, at its end they fill the
        // drain of the last tiles
        deferred.push_back({e, (int)(((n + 15) / 16) * S)});
    }
    struct DeferredElem { ElemArgs e; int nblocks; };
    std::vector<DeferredElem> deferred;
    void flush_deferred() {
        for (const auto& d : deferred) elem(OP_COLSUM, d.e, d.nblocks, 256 * sizeof(float));
        deferred.clear();
    }
    // ln: k > 0 the tiles leave LayerNorm parts for their output rows in slot k - 1 (Phase::ln), k < 0 operand A is read through the
    // LayerNorm of slot -k - 1
    void gemm(const GemmArgs& g_in, int batch, bool xfa = false, int ln_slot = 0) {
        GemmArgs g = g_in;
        const int tiles0 = ((g.N + 63) / 64) * ((g.M + 31) / 32);
        // only a product whose output lies in the gradient bucket may be cut: nothing but the optimizer (and the next phase's
        // reduction) reads it.  An activation of the backward chain -- dY = dKV Wkv has K = 2 d_model = 1024 at d_model 512 -- is
        // read by the very next phase and must be whole when its own phase ends.
        const bool leaf = g.C >= leaf_lo && g.C < leaf_hi;
        const bool splittable = allow_split && leaf && batch == 1 && g.kb_count == 1 && g.bias_mode == 0 && g.relu == 0 && !g.drop && g.alpha == 1.f &&
                                g.c_rs == g.N && (g.beta == 0.f || g.beta == 1.f);
        const int S = splittable ? split_factor(g.K, tiles0) : 1;
        // an output some pending reduction (of this or the previous phase) still has to write must not be touched directly:
        // the product then goes through a partial of its own (S = 1) and is added behind that reduction
        bool conflict = false;
        for (const auto& r : pend_prev) conflict = conflict || r.out == g.C;
        for (const auto& r : pend_cur) conflict = conflict || r.out == g.C;
        if (conflict && !splittable) { rc = fail(KM_ERR_UNSUPPORTED, "training program: a product writes an output with a pending reduction"); return; }
        // a conflicting product MUST go through a partial: written directly it would race with the reduction still to come
        if (conflict && (int64_t)S * g.M * g.N + 4 > scratch_left) {
            rc = fail(KM_ERR_WORKSPACE, "training program: no room for the partial of a product whose output has a pending reduction");
            return;
        }
        if ((S > 1 || conflict) && (int64_t)S * g.M * g.N + 4 <= scratch_left) {
            float* part = scratch;
            const int64_t n = (int64_t)g.M * g.N;
            scratch += ((int64_t)S * n + 3) / 4 * 4; scratch_left -= ((int64_t)S * n + 3) / 4 * 4;
            pend_cur.push_back({part, g.C, (int)n, S, g.beta == 1.f ? 1 : 0});
            const int64_t chunk = g.K / S;
            g.K = (int)chunk; g.beta = 0.f; g.C = part;
            g.a_bs1 = chunk * g.a_cs; g.b_bs1 = chunk * g.b_rs; g.c_bs1 = n; g.a_bs2 = g.b_bs2 = g.c_bs2 = 0; g.batch2 = 1;
            batch = S;
        }
        Op op{};
        op.kind = OP_GEMM; op.g = g;
        op.gx = (g.N + 63) / 64; op.gy = (g.M + 63) / 64; op.bm = 64;
        if (g.M <= 32 || op.gx * op.gy * batch < bm32_below) { op.bm = 32; op.gy = (g.M + 31) / 32; }   // finer tiles: more CUs, half the chain
        if ((g.a_cs != 1 && g.a_rs != 1) || (g.b_rs != 1 && g.b_cs != 1)) {
            rc = fail(KM_ERR_UNSUPPORTED, "training program: an operand is contiguous neither along k nor along its rows");
            return;
        }
        if (gemm_operand_extent(g.M, g.a_rs, g.K, g.a_cs, g.kb_count, g.a_kbs) >= ((int64_t)1 << 29) ||
            gemm_operand_extent(g.N, g.b_cs, g.K, g.b_rs, g.kb_count, g.b_kbs) >= ((int64_t)1 << 29)) {
            rc = fail(KM_ERR_UNSUPPORTED, "training program: operand larger than the 2 GiB a buffer descriptor addresses");
            return;
        }
        { int ma = 0, mb = 0; op.dma = (use_dma && gemm_dma_ok(g, &ma, &mb)) ? 1 : 0; op.ma = (signed char)ma; op.mb = (signed char)mb; }
        if (ln_slot) {
            if (!(op.dma && op.ma == 0 && op.mb == 0) || S > 1 || g.N % 32 != 0 || (ln_slot > 0 && (g.c_rs != g.N || (batch > 1 && g.c_bs1 != (int64_t)g.M * g.N)))) {
                rc = fail(KM_ERR_UNSUPPORTED, "training program: LayerNorm by the reader needs the LDS-DMA tile with k-contiguous operands and dense output rows");
                return;
            }
            op.ln = (signed char)ln_slot;
        }
        if (xfa) {
            if (!(op.dma && op.ma == 0 && op.mb == 0)) {
                rc = fail(KM_ERR_UNSUPPORTED, "training program: the operand transform needs the LDS-DMA tile with k-contiguous operands");
                return;
            }
            op.dma = 2;
        }
        op.va = gemm_operand_vec(g.A, g.a_rs, g.a_cs, g.a_bs1, g.a_bs2, g.a_kbs) ? 1 : 0;
        op.vb = gemm_operand_vec(g.B, g.b_cs, g.b_rs, g.b_bs1, g.b_bs2, g.b_kbs) ? 1 : 0;
        if (!op.va) op.vb = 0;                   // three instantiations: (vec, vec), (vec, scalar), (scalar, scalar)
        add(op, op.gx * op.gy * batch, (size_t)ggd::lds_floats(op.bm) * sizeof(float));
        if (!rc && !op_per_launch) gbatch[cur.n_ops - 1] = batch;
    }
    void need_lds(size_t bytes) { if (bytes > lds) lds = bytes; }
    void elem(int kind, const ElemArgs& e, int64_t nblocks, size_t op_lds = 0) {
        Op op{};
        op.kind = kind; op.e = e;
        add(op, (int)nblocks, op_lds);
    }
    void add_pending_reduces() {          // the partials written by the previous phase, four reductions per operation
        for (size_t i = 0; i < pend_prev.size(); i += 4) {
            ElemArgs e{};
            const float** ps[4] = {&e.p0, &e.p1, &e.p2, &e.p3};
            float** qs[4] = {&e.q0, &e.q1, &e.q2, &e.q3};
            int* ns[4] = {&e.rn0, &e.rn1, &e.rn2, &e.rn3};
            int* Ss[4] = {&e.rS0, &e.rS1, &e.rS2, &e.rS3};
            int nblocks = 0;
            for (size_t j = 0; j < 4 && i + j < pend_prev.size(); ++j) {
                const PendingReduce& r = pend_prev[i + j];
                *ps[j] = r.part; *qs[j] = r.out; *ns[j] = r.n; *Ss[j] = r.S; e.racc |= (r.acc ? 1 : 0) << j;
                nblocks += (r.n + 255) / 256;
            }
            elem(OP_REDUCE4, e, nblocks);
        }
        pend_prev.clear();
    }
    // launch what the phase holds so far (its operations are independent of each other, so a phase may be cut anywhere)
    int launch_cur(hipStream_t st) {
        if (rc) return rc;
        if (cur.n_ops > 0) {
            ++launches;
            // ring depth of the LDS-DMA tiles.  The 8-stage ring (the whole K = 256 of a small product requested at entry, 96 KB: one
            // workgroup per CU) paid while the small phases held nothing but products (0.181 -> 0.176 ms at 8 windows); with the
            // row / element operations that have since moved beside them it loses to the 4-stage ring with three workgroups per CU
            // (0.1238 -> 0.1215 ms; train_alone_max 256 brings it back): a phase of at most alone_max workgroups has the chip to itself (one workgroup
            // per CU: deep rings); beyond that LDS is occupancy
            const bool alone = blocks <= alone_max;
            for (int i = 0; i < cur.n_ops; ++i) {
                Op& o = cur.ops[i];
                if (o.kind != OP_GEMM || !o.dma) continue;
                o.ns = (signed char)gdma::ring_stages(o.bm, alone);
                need_lds((size_t)(gdma::lds_floats(o.bm, o.ns) + (o.ln < 0 ? 2 * o.g.K : 0)) * sizeof(float));
            }
            cur.xf = xf; cur.ln[0] = ln[0]; cur.ln[1] = ln[1];
            if (has_attn) hipLaunchKernelGGL(phase_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, st, cur);
            else hipLaunchKernelGGL(phase_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, st, cur);
            HIP_TRY(hipGetLastError());
        }
        cur = Phase{};
        blocks = 0;
        lds = 0;
        has_attn = false; any_dma = false; dma_bm64 = false;
        return KM_OK;
    }
    int end_phase(hipStream_t st) {
        flush_deferred();
        add_pending_reduces();
        pend_prev.swap(pend_cur);
        return launch_cur(st);
    }
};

static GemmArgs G(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float* C, int64_t c_rs,
                  int64_t M, int64_t N, int64_t K) {
    GemmArgs g{};
    g.alpha = 1.f; g.batch2 = 1; g.kb_count = 1;
    g.A = A; g.a_rs = a_rs; g.a_cs = a_cs; g.B = B; g.b_rs = b_rs; g.b_cs = b_cs; g.C = C; g.c_rs = c_rs;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    return g;
}
// C (rows x N) = A (rows x K) W^T (+ bias), W stored (N x K) like nn.Linear
static GemmArgs NT(const float* A, int64_t a_rs, const float* W, int64_t K, float* C, int64_t c_rs, int64_t rows, int64_t N,
                   const float* bias, int relu) {
    GemmArgs g = G(A, a_rs, 1, W, 1, K, C, c_rs, rows, N, K);
    g.bias = bias; g.bias_mode = bias ? 1 : 0; g.relu = relu;
    return g;
}
// C (rows x N) = A (rows x K) W, W stored (K x N)
static GemmArgs NN(const float* A, int64_t a_rs, const float* W, int64_t w_rs, float* C, int64_t c_rs, int64_t rows, int64_t N, int64_t K) {
    return G(A, a_rs, 1, W, w_rs, 1, C, c_rs, rows, N, K);
}
// C (M x N) = A^T B with A (rows x M), B (rows x N)
static GemmArgs TN(const float* A, int64_t a_rs, const float* B, int64_t b_rs, float* C, int64_t c_rs, int64_t M, int64_t N, int64_t rows) {
    return G(A, 1, a_rs, B, b_rs, 1, C, c_rs, M, N, rows);
}

static int split_rows(int64_t rows) {
    int S = (int)((rows + 511) / 512);
    return S < 1 ? 1 : (S > 32 ? 32 : S);
}

// frames per channel row of the packed encoder input: T + 3 rounded up to the LDS-DMA tile's k step (zeros beyond T + 3)
int64_t trainp_kp(Context* c) { return (c->KT + 31) / 32 * 32; }

int64_t trainp_mask_bytes(Context* c) { return (int64_t)c->H * 28 * c->NK + (int64_t)c->H * 24 + 52 * (int64_t)c->DH; }

// floats per window + a fixed part (returned through *fixed)
int64_t trainp_act_floats(Context* c, int64_t* fixed) {
    const int64_t d = c->d, H = c->H, NKk = c->NK, DH = c->DH, KP = trainp_kp(c);
    const int64_t per = KP * NKk + 2 * NKk * d + 2 * NKk + 2 * NKk * d /* KV */ + 3 * H * 28 * NKk /* P, Pd, dP */ + H * 28 * NKk /* dS */ +
                        3 * 28 * d + 28 * DH + 3 * 24 * d + 24 * DH + 64 /* z */ + (28 + 24) * DH /* dH */ + 6 * (28 + 24) * d / 2 /* dA dO2 dO1 x2 */ +
                        2 * NKk * d /* dKV */ + 3 * NKk * d /* dY (two K halves) dY0 */ + 8 * d + 8 + 5 * 52 + 4 + d * c->KT /* dWce partial */ + 28 * d /* dQ partial */ +
                        (trainp_mask_bytes(c) + 3) / 4 + 64 + 2 * (NKk + 1) * (d / 32 + 1) /* LayerNorm parts */;
    if (fixed) *fixed = 2 * 28 * d + 2 * d * d + 2 * DH * d + 2 * d + 2 * DH + 64 /* folds */ + d * KP /* channel encoder weight, rows of KP */ + 32 * (4 * d + 2 * DH + 4 * d) /* split partials */ + 1024;
    return per;
}

struct MaskSet { unsigned char *mel, *emo, *dec; };
static MaskSet mask_ptrs(Context* c, int64_t B) {
    unsigned char* base = reinterpret_cast<unsigned char*>(c->trp_masks);
    MaskSet m;
    m.mel = base;
    m.emo = m.mel + (size_t)c->tr_windows * c->H * 28 * c->NK;
    m.dec = m.emo + (((size_t)c->tr_windows * c->H * 24 + 15) / 16) * 16;
    (void)B;
    return m;
}

int trainp_mask_sizes(Context* c, int64_t B, int64_t* mel, int64_t* emo, int64_t* dec) {
    *mel = B * c->H * 28 * c->NK; *emo = B * c->H * 24; *dec = B * 52 * c->DH;
    return KM_OK;
}

int trainp_copy_masks(Context* c, int64_t B, unsigned char* mel, unsigned char* emo, unsigned char* dec, int to_device, void* stream) {
    int64_t nm, ne, nd;
    trainp_mask_sizes(c, B, &nm, &ne, &nd);
    const MaskSet m = mask_ptrs(c, B);
    hipStream_t st = (hipStream_t)stream;
    if (to_device) {
        HIP_TRY(hipMemcpyAsync(m.mel, mel, (size_t)nm, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(m.emo, emo, (size_t)ne, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(m.dec, dec, (size_t)nd, hipMemcpyHostToDevice, st));
    } else {
        HIP_TRY(hipMemcpyAsync(mel, m.mel, (size_t)nm, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(emo, m.emo, (size_t)ne, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(dec, m.dec, (size_t)nd, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return KM_OK;
}

int64_t trainp_mask_alloc_bytes(Context* c) {
    return (int64_t)c->tr_windows * (c->H * 28 * c->NK + c->H * 24 + 52 * c->DH) + 64;
}

#define RUN(expr) do { if (int rc_ = (expr)) return rc_; } while (0)

// xp_dev: packed encoder input (B, KP, NK) when the caller (the from-audio step) produced it, else null and it is packed
// here from mel / mel_short.
int train_forward_backward_phased(Context* c, const float* mel, int64_t B, int64_t T_in, const float* mel_short, const float* xp_dev,
                                  const TrainAudioSrc* asrc,
                                  const float* emo, const float* target, float mse_w, float l1_w, float* flat_grad, float* loss_dev,
                                  float* out_dev, float* ema_state, int ema_first, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int64_t d = c->d, H = c->H, hd = c->hd, T = c->T, KT = c->KT, DH = c->DH, NKk = c->NK, ED = c->ED;
    const int64_t KP = trainp_kp(c);
    const int64_t R = B * NKk, Rm = B * 28, Re = B * 24;
    if (NKk % 4 != 0 || NKk > 16 * kAttnMaxKT)
        return fail(KM_ERR_UNSUPPORTED, "phased training step needs num_mel_channels <= %d, a multiple of 4 (got %lld)", 16 * kAttnMaxKT, (long long)NKk);
    c->tr_alpha_live = ema_state != nullptr && !ema_first;
    auto P = [&](const char* k) -> const float* { return c->tr_params + c->tr_offset.at(k); };
    auto Gd = [&](const char* k) -> float* { return flat_grad + c->tr_offset.at(k); };
    float* w = c->trp_act;
    auto take = [&](int64_t n) { float* p = w; w += (n + 3) / 4 * 4; return p; };
    float* xp = take(c->tr_windows * KP * NKk);            // first: km_train_step_audio writes the packed input here
    float* x2 = take(c->tr_windows * KP * NKk);            // ... as 10 log10(power) when the front end packs (asrc->packed): the finished features
    // fixed part
    float* Qb = take(28 * d); float* dQb = take(28 * d);
    float* T1m = take(d * d); float* T1e = take(d * d); float* Wfm = take(DH * d); float* Wfe = take(DH * d);
    float* t1m = take(d); float* t1e = take(d); float* bfm = take(DH); float* bfe = take(DH);
    float* ones = take(c->tr_windows * NKk);
    float* WceP = c->trp_wcep;                             // channel encoder weight with rows padded to KP (zeros), kept beside the parameters (PaddedCopy, km_train.hip)
    // per-window part
    float* Y0 = take(R * d); float* Y = take(R * d); float* mu = take(R); float* rs = take(R);
    float* KV = take(R * 2 * d);
    float* Pm = take(B * H * 28 * NKk);
    float* A = take(Rm * d); float* O1 = take(Rm * d); float* O2 = take(Rm * d); float* H1 = take(Rm * DH);
    float* Ae = take(Re * d); float* Oe1 = take(Re * d); float* Oe2 = take(Re * d); float* He = take(Re * DH);
    float* zrows = take(Rm + Re); float* grow = take(Rm + Re);
    float* dH1 = take(Rm * DH); float* dHe = take(Re * DH);
    float* dA = take(Rm * d); float* dO2 = take(Rm * d); float* dO1 = take(Rm * d);
    float* dAe = take(Re * d); float* dOe2 = take(Re * d); float* dOe1 = take(Re * d);
    float* dKV = take(R * 2 * d); float* dY = take(2 * R * d) /* two K halves at small batches */; float* dY0 = take(R * d); float* Tm = take(R * d); float* Te = take(B * d);
    float* E0 = take(B * d); float* E = take(B * d); float* Ve = take(B * d); float* dVe = take(B * d); float* dE = take(B * d); float* dE0 = take(B * d);
    float* emu = take(B); float* ers = take(B);
    float* bs = take(B * 52); float* outb = take(B * 52); float* dz = take(B * 52); float* tfac = take(B * 52); float* txp = take(B * 52);
    float* dQ_part = take(B * 28 * d);
    float* statsY = take(2 * R * (d / 32 + 1)); float* statsE = take(2 * B * (d / 32 + 1));      // LayerNorm parts of Y0 / E0 rows (float2 [row][d / 32])
    if ((w - c->trp_act) > c->trp_act_floats) return fail(KM_ERR_WORKSPACE, "phased training workspace too small (internal)");

    const float p_drop = c->tr_dropout_p;
    const bool drop = p_drop > 0.f;
    const float keep_scale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
    const MaskSet ms = mask_ptrs(c, B);
    const unsigned char* m_mel = drop ? ms.mel : nullptr;
    const unsigned char* m_emo = drop ? ms.emo : nullptr;
    const unsigned char* m_dec = drop ? ms.dec : nullptr;

    const float* inw = P("mel_attention.in_proj_weight"); const float* inb = P("mel_attention.in_proj_bias");
    const float* einw = P("emotion_attention.in_proj_weight"); const float* einb = P("emotion_attention.in_proj_bias");
    const float* Wo = P("mel_attention.out_proj.weight"); const float* bo = P("mel_attention.out_proj.bias");
    const float* Wmo = P("mel_output_proj.weight"); const float* bmo = P("mel_output_proj.bias");
    const float* Woe = P("emotion_attention.out_proj.weight"); const float* boe = P("emotion_attention.out_proj.bias");
    const float* Weo = P("emotion_output_proj.weight"); const float* beo = P("emotion_output_proj.bias");
    const float* W1 = P("blendshape_decoder.0.weight"); const float* b1 = P("blendshape_decoder.0.bias");
    const float* w2 = P("blendshape_decoder.3.weight"); const float* b2 = P("blendshape_decoder.3.bias");
    const float scale = 1.0f / std::sqrt((float)hd);
    float* gin_w = Gd("mel_attention.in_proj_weight"); float* gin_b = Gd("mel_attention.in_proj_bias");
    float* gein_w = Gd("emotion_attention.in_proj_weight"); float* gein_b = Gd("emotion_attention.in_proj_bias");

    static PerDeviceOnce once;
    if (once.first(c->device))
    {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&phase_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&phase_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const size_t attn_lds = (size_t)attn_mfma_lds_floats((int)hd, (int)NKk) * sizeof(float);
    if (attn_lds > 160 * 1024 || (hd != 16 && hd != 32 && hd != 64))
        return fail(KM_ERR_UNSUPPORTED, "the fused training attention blocks are built for heads of 16, 32 or 64 columns (got %lld)", (long long)hd);

    Program pg;
    pg.scratch = c->trp_split; pg.scratch_left = c->trp_split_floats;
    pg.allow_split = !c->opt.train_no_split;
    pg.leaf_lo = flat_grad; pg.leaf_hi = flat_grad + c->tr_nparams;
    if (c->opt.train_bm32_below > 0) pg.bm32_below = c->opt.train_bm32_below;
    pg.op_per_launch = c->opt.train_op_per_launch != 0; pg.dbg_stream = st;
    pg.use_dma = !c->opt.train_no_dma;
    if (c->opt.train_split_min_k > 0) { pg.min_k = c->opt.train_split_min_k; pg.chain = c->opt.train_split_min_k; }
    if (c->opt.train_alone_max > 0) pg.alone_max = c->opt.train_alone_max;
    auto blocks4 = [](int64_t rows) { return (rows + 3) / 4; };            // one wave per row, 4 rows per block
    auto blocks256 = [](int64_t n) { return (n + 255) / 256; };
    // column sums as products with a vector: out (1 x n) (+)= w^T (1 x rows) M (rows x n); w = ones, or per-row loss gradients
    const bool colsum_valu = !c->opt.train_colsum_gemm;
    auto colsum = [&](const float* m, int64_t rows, int64_t rs_, int64_t n, float* out, int accumulate, const float* wvec) {
        if (colsum_valu) { pg.colsum(m, rows, rs_, n, out, accumulate, wvec == ones ? nullptr : wvec); return; }
        GemmArgs g = G(wvec, 0, 1, m, rs_, 1, out, n, 1, n, rows);
        g.beta = accumulate ? 1.f : 0.f;
        pg.gemm(g, 1);
    };
    auto reduce = [&](const float* part, int S, int64_t n, int64_t stride, float* out, int accumulate) {
        ElemArgs e{};
        e.p0 = part; e.i0 = S; e.n0 = n; e.n1 = stride; e.q0 = out; e.i1 = accumulate;
        pg.elem(OP_REDUCE, e, blocks256(n));
    };
    unsigned scale_bits;
    std::memcpy(&scale_bits, &scale, sizeof(scale_bits));

    // ================= P0: what the channel encoder waits for and nothing else: packed input, padded weight =================
    // (the gradient bucket, the ones vector and the dropout masks are first read in P4 / P8: they are made in P3 / P4, behind
    // those phases' own work, where they cost nothing -- P0 7.6 -> 5 us at 8 windows)
    if (asrc && asrc->packed) {
        // the front end wrote 10 log10(power) into the packed rows itself (MelArgs::pack_*); the channel encoder finishes the
        // conversion on its operand fragments (DbXform).  With the padded weight kept beside the parameters (PaddedCopy) phase 0
        // is empty: one launch less per step
    } else if (asrc) {                         // from audio: the power-mel of the front end -> packed log-mel rows (read by P1)
        ElemArgs e{};
        e.p0 = asrc->melpow; e.p1 = reinterpret_cast<const float*>(asrc->melmax); e.q0 = const_cast<float*>(xp_dev); e.n0 = B;
        e.i0 = (int)NKk; e.i1 = (int)KP; e.i2 = (int)T; e.i3 = asrc->n_frames; e.lp = *asrc->lp;
        pg.elem(OP_LOGPACK, e, blocks256(B * KP * NKk));
    } else {
        ElemArgs e{};
        e.p0 = mel; e.p1 = mel_short; e.q0 = xp; e.n0 = B; e.i0 = (int)NKk; e.i1 = (int)KP; e.i2 = (int)T; e.i3 = (int)T_in;
        pg.elem(OP_PACKX, e, blocks256(B * KP * (NKk / 4)));
    }
    const float* X = xp_dev ? xp_dev : xp;
    const bool xf = asrc && asrc->packed;
    if (xf) pg.xf = DbXform{*asrc->lp, asrc->melmax};
    // LayerNorm by the reader (option train_no_ln_fuse switches back to the LayerNorm phase): every product involved on the LDS-DMA tile
    // -- and a small batch: the reader pays two vector instructions per operand value, two LDS reads per k block and a pooling
    // prologue per tile, which the launch it saves outweighs up to ~40 windows (8 windows 0.1288-0.1301 -> 0.125 ms, 24: 0.1647 ->
    // 0.1606, 32: 0.1717 -> 0.1683, 48: 0.2079 -> 0.2105, 64: 0.229 -> 0.238: there the K / V product is bound by the matrix pipe and the
    // LayerNorm phase is cheap beside it)
    const int ln_fuse_rows = c->opt.train_ln_fuse_rows > 0 ? c->opt.train_ln_fuse_rows : 3200;
    const bool fuse_ln = pg.use_dma && !c->opt.train_no_ln_fuse && d % 32 == 0 && ED % 32 == 0 && R <= ln_fuse_rows;
    if (fuse_ln) {
        pg.ln[0] = LnXform{statsY, P("mel_norm.weight"), P("mel_norm.bias"), (int)(d / 32), 1e-5f};
        pg.ln[1] = LnXform{statsE, P("emotion_norm.weight"), P("emotion_norm.bias"), (int)(d / 32), 1e-5f};
    }
    RUN(pg.end_phase(st));
    // ================= P1: channel encoder; beside it the input-independent products (Q, T1 = Wmo Wo) and E0 =================
    // (round 4: they used to sit in P0, whose K = 256 tile chains made that phase as long as a product phase for nothing the
    // channel encoder waits for; the emotion chain and the folds run one phase later than before, still ahead of their readers)
    if (asrc && !asrc->packed) {       // the window maxima were read in P0: clean slots for the next front-end launch (no memset)
        ElemArgs z{};
        z.q0 = reinterpret_cast<float*>(c->ws_melmax); z.n0 = c->ws_windows;
        pg.elem(OP_ZERO, z, (c->ws_windows + 4095) / 4096);
        c->melmax_dirty = false;
    }
    {   // Y0[b] (NK x d) = XT_b (NK x KP) WceP^T + b: both operands k-contiguous with K = KP a multiple of 32 (zeros beyond KT on
        // both sides): the LDS-DMA tile.  Round 3 read X (B, KP, NK) as a row-contiguous operand against Wce's rows of 259
        // floats on the register tile (scalar loads): 12 us of the 8-window step
        GemmArgs g = NT(X, KP, WceP, KP, Y0, d, NKk, d, P("mel_channel_encoder.bias"), 0);
        g.a_bs1 = KP * NKk; g.c_bs1 = NKk * d;
        pg.gemm(g, (int)B, xf, fuse_ln ? 1 : 0);       // xf: X holds 10 log10(power), finished on the fragments; fuse_ln: LayerNorm parts of Y0's rows
    }
    pg.gemm(NT(P("mouth_queries"), d, inw, d, Qb, d, 28, d, inb, 0), 1);                                  // Q = mq Wq^T + bq
    pg.gemm(NN(Wmo, d, Wo, d, T1m, d, d, d, d), 1);                                                        // T1 = Wmo Wo
    pg.gemm(NN(Weo, d, Woe, d, T1e, d, d, d, d), 1);
    { GemmArgs g = G(Wmo, d, 1, bo, 1, 0, t1m, 1, d, 1, d); g.bias = bmo; g.bias_mode = 2; pg.gemm(g, 1); }   // t1 = Wmo bo + bmo
    { GemmArgs g = G(Weo, d, 1, boe, 1, 0, t1e, 1, d, 1, d); g.bias = beo; g.bias_mode = 2; pg.gemm(g, 1); }
    pg.gemm(NT(emo, ED, P("emotion_encoder.weight"), ED, E0, d, B, d, P("emotion_encoder.bias"), 0), 1, false, fuse_ln ? 2 : 0);
    RUN(pg.end_phase(st));
    // The folds Wf = W1 T1 (operands of P5) and the finished features x2 (operand of P12): with fuse_ln they ride in P3
    auto folds = [&]() {
        pg.gemm(NN(W1, d, T1m, d, Wfm, d, DH, d, d), 1);                                                        // Wf = W1 T1
        pg.gemm(NN(W1, d, T1e, d, Wfe, d, DH, d, d), 1);
        { GemmArgs g = G(W1, d, 1, t1m, 1, 0, bfm, 1, DH, 1, d); g.bias = b1; g.bias_mode = 2; pg.gemm(g, 1); }     // bf = W1 t1 + b1
        { GemmArgs g = G(W1, d, 1, t1e, 1, 0, bfe, 1, DH, 1, d); g.bias = b1; g.bias_mode = 2; pg.gemm(g, 1); }
    };
    auto dbconv = [&]() {
        if (!xf) return;
        // the packed rows once more, finished into x2 for the channel encoder's weight gradient (phase 12).  Not beside the channel
        // encoder in phase 1: its workgroups held tile slots (the phase's LDS size) while they streamed -- +3.2 us at 64 windows
        ElemArgs e{};
        e.p0 = X; e.p1 = reinterpret_cast<const float*>(asrc->melmax); e.q0 = x2; e.n0 = B; e.i0 = (int)NKk; e.i1 = (int)KP; e.i2 = (int)T;
        e.lp = *asrc->lp;
        e.i3 = (int)((NKk * (KP / 4) + 1023) / 1024);
        pg.elem(OP_DBCONV, e, B * e.i3);
    };
    if (!fuse_ln) {
    // ================= P2: LayerNorm (both streams); folds Wf = W1 T1 =================
    // (the folds first: their K = d tile chains are the longest thing in the phase and should not wait in the dispatch order behind
    // the hundreds of row workgroups of the LayerNorm and the conversion)
    folds();
    {
        ElemArgs e{};
        e.p0 = Y0; e.q0 = Y; e.n0 = R; e.i0 = (int)d; e.p1 = P("mel_norm.weight"); e.p2 = P("mel_norm.bias"); e.q1 = mu; e.q2 = rs;
        pg.elem(OP_LN_FWD, e, blocks4(R));
    }
    {
        ElemArgs e{};
        e.p0 = E0; e.q0 = E; e.n0 = B; e.i0 = (int)d; e.p1 = P("emotion_norm.weight"); e.p2 = P("emotion_norm.bias"); e.q1 = emu; e.q2 = ers;
        pg.elem(OP_LN_FWD, e, blocks4(B));
    }
    dbconv();
    RUN(pg.end_phase(st));
    }
    // ================= P3: [K | V]; emotion value projection =================
    auto zero_maxima = [&]() {          // the window maxima were read by DbXform and OP_DBCONV: clean slots for the next front-end launch
        if (!xf) return;
        ElemArgs z{};
        z.q0 = reinterpret_cast<float*>(c->ws_melmax); z.n0 = c->ws_windows;
        pg.elem(OP_ZERO, z, (c->ws_windows + 4095) / 4096);
        c->melmax_dirty = false;
    };
    if (!fuse_ln) zero_maxima();          // (with fuse_ln OP_DBCONV reads them in THIS phase: they are cleaned in P4)
    // fuse_ln (round 4): there is no LayerNorm phase.  The tiles of Y0 and E0 (P1) left per-row parts of the statistics; the two
    // products here read Y0 / E0 and normalise their operand fragments on the way to the MFMAs (LnXform), OP_LNAPPLY beside them
    // writes Y, E, mu, rstd for the backward pass with the same two functions (ln_combine, ln_apply: the same bits)
    pg.gemm(NT(fuse_ln ? Y0 : Y, d, inw + d * d, d, KV, 2 * d, R, 2 * d, inb + d, 0), 1, false, fuse_ln ? -1 : 0);
    pg.gemm(NT(fuse_ln ? E0 : E, d, einw + 2 * d * d, d, Ve, d, B, d, einb + 2 * d, 0), 1, false, fuse_ln ? -2 : 0);
    if (fuse_ln) {
        folds();
        ElemArgs e{};
        e.p0 = Y0; e.p1 = P("mel_norm.weight"); e.p2 = P("mel_norm.bias"); e.p3 = statsY; e.q0 = Y; e.q1 = mu; e.q2 = rs;
        e.n0 = R; e.i0 = (int)d; e.i1 = (int)(d / 32); e.f0 = 1e-5f;
        pg.elem(OP_LNAPPLY, e, blocks4(R));
        ElemArgs f{};
        f.p0 = E0; f.p1 = P("emotion_norm.weight"); f.p2 = P("emotion_norm.bias"); f.p3 = statsE; f.q0 = E; f.q1 = emu; f.q2 = ers;
        f.n0 = B; f.i0 = (int)d; f.i1 = (int)(d / 32); f.f0 = 1e-5f;
        pg.elem(OP_LNAPPLY, f, blocks4(B));
        dbconv();
    }
    // the dropout masks of the step (first read in P4), behind the phase's own work in dispatch order
    if (drop && c->tr_dropout_mode == 0) {
        int64_t nm, ne, nd;
        trainp_mask_sizes(c, B, &nm, &ne, &nd);
        const double t32 = (double)p_drop * 4294967296.0;
        const unsigned thr = (unsigned)(t32 > 4294967295.0 ? 4294967295.0 : t32);
        unsigned char* regions[3] = {ms.mel, ms.emo, ms.dec};
        const int64_t sizes[3] = {nm, ne, nd};
        for (int r = 0; r < 3; ++r) {
            ElemArgs e{};
            e.mask_out = regions[r]; e.n0 = sizes[r]; e.i0 = r; e.i1 = (int)thr; e.u0 = (unsigned)c->tr_dropout_seed;
            e.u1 = (unsigned)(c->tr_dropout_seed >> 32); e.p0 = reinterpret_cast<const float*>(c->trp_drop_ctr);
            pg.elem(OP_MASKGEN, e, (sizes[r] + 1023) / 1024);
        }
    }
    RUN(pg.end_phase(st));
    // ================= P4: attention (scores, softmax, dropout, P V) per (window, head); emotion attention (one key: weight 1,
    // dropped or kept per head and query) =================
    {
        ElemArgs e{};
        e.p0 = Qb; e.p1 = KV; e.q0 = Pm; e.q1 = A; e.i0 = (int)d; e.i1 = (int)hd; e.i2 = (int)NKk; e.i3 = (int)H;
        e.mask = m_mel; e.f0 = keep_scale; e.u0 = scale_bits; e.u1 = c->opt.train_attn_regs ? 1u : 0u;
        {
            size_t fl = (size_t)attn_mfma_fwd_lds_floats((int)hd, (int)NKk);
            if (hd == 32 && !c->opt.train_attn_regs && NKk > 64 && NKk <= 80) fl = kAttnDmaFwdLdsFloats;
            pg.elem(OP_ATTN_FWD, e, B * H, fl * sizeof(float));
        }
    }
    {
        ElemArgs e{};
        e.p0 = Ve; e.q0 = Ae; e.n0 = B; e.i0 = (int)d; e.i1 = (int)hd; e.mask = m_emo; e.f0 = keep_scale;
        pg.elem(OP_EMO_EXPAND, e, Re * ((d + 255) / 256));
    }
    if (fuse_ln) zero_maxima();
    // clean gradient bucket and the ones vector (first touched in P8): they ride under the attention blocks
    {
        ElemArgs e{};
        e.q0 = flat_grad; e.n0 = c->tr_nparams;
        pg.elem(OP_ZERO, e, (c->tr_nparams + 4095) / 4096);
        ElemArgs f{};
        f.q0 = ones; f.n0 = R; f.f0 = 1.0f;
        pg.elem(OP_FILL, f, blocks256(R));
    }
    RUN(pg.end_phase(st));
    // ================= P5: decoder hidden through the fold (+ ReLU + dropout), both streams; O1, Oe1 beside it =================
    { GemmArgs g = NT(A, d, Wfm, d, H1, DH, Rm, DH, bfm, 1); g.drop = m_dec; g.drop_scale = keep_scale; g.drop_map = 1; pg.gemm(g, 1); }
    { GemmArgs g = NT(Ae, d, Wfe, d, He, DH, Re, DH, bfe, 1); g.drop = m_dec; g.drop_scale = keep_scale; g.drop_map = 2; pg.gemm(g, 1); }
    pg.gemm(NT(A, d, Wo, d, O1, d, Rm, d, bo, 0), 1);
    pg.gemm(NT(Ae, d, Woe, d, Oe1, d, Re, d, boe, 0), 1);
    RUN(pg.end_phase(st));
    // ================= P6: decoder output layer + loss tail; it also writes the hidden layer's gradients dH =================
    {
        TailArgs t{};
        t.zrows = zrows; t.zrows_out = zrows; t.grow = grow; t.h1 = H1; t.he = He; t.w2 = w2; t.b2 = b2;
        t.mel_w = P("mel_weights"); t.emo_w = P("emotion_weights"); t.temperature = c->cfg.temperature; t.target = target;
        t.bs = bs; t.out = outb; t.dz = dz; t.ema_state = ema_state; t.ema_first = ema_first; t.alpha_p = P("smoothing_alpha");
        t.mse_w = mse_w; t.l1_w = l1_w; t.lc = c->tr_loss_cfg; t.fac = tfac; t.xp = txp; t.loss = loss_dev;
        t.d_melw = Gd("mel_weights"); t.d_emow = Gd("emotion_weights"); t.d_alpha = Gd("smoothing_alpha");
        t.B = (int)B; t.DH = (int)DH; t.expr_rows = 24; t.audio_energy = c->tr_loss_cfg.audio_energy_dev; t.out2 = out_dev;
        t.d_b2 = Gd("blendshape_decoder.3.bias"); t.drop_ctr = (drop && c->tr_dropout_mode == 0) ? c->trp_drop_ctr : nullptr;
        t.part = c->trp_tail_part; t.ctr = c->trp_tail_ctr;
        t.dh1 = dH1; t.dhe = dHe; t.keep_scale = keep_scale;
        // one workgroup per window, at most 32 (round 4, with the hidden layer's gradient in the tail: 8 workgroups 0.1628 ms per
        // 8-window step, 4: 0.1653, 2: 0.1717); the audio-visual term couples the whole batch: one
        const bool av = c->tr_loss_cfg.perceptual_weight > 0.f && c->tr_loss_cfg.audio_energy_dev;
        if (!av && c->opt.train_tail_groups <= 0 && B <= kTailMaxGroups && DH % 4 == 0 && DH <= 128) {
            // the window-resident tail (train_tail_window_dev): 17 -> ~11 us per 8-window step
            hipLaunchKernelGGL(trainp_tail_window_kernel, dim3((unsigned)B), dim3(256), (size_t)train_tail_window_lds_floats((int)DH) * sizeof(float),
                               st, t);
        } else {
            int groups = av ? 1 : (int)B;
            if (c->opt.train_tail_groups > 0 && !av) groups = c->opt.train_tail_groups;
            groups = groups < 1 ? 1 : (groups > 32 ? 32 : groups);
            hipLaunchKernelGGL(trainp_tail_kernel, dim3((unsigned)groups), dim3(1024), 0, st, t);
        }
        HIP_TRY(hipGetLastError());
    }
    // ================= P8: input gradients through the fold; O2, Oe2 (needed by the decoder[0] gradients); dw2, db1 =================
    // (P7 -- the outer product dH = g w2 [H > 0] and O2 -- is gone: the tail writes dH, O2 rides here)
    pg.gemm(NN(dH1, DH, Wfm, d, dA, d, Rm, d, DH), 1);
    pg.gemm(NN(dHe, DH, Wfe, d, dAe, d, Re, d, DH), 1);
    pg.gemm(NN(dH1, DH, W1, d, dO2, d, Rm, d, DH), 1);
    pg.gemm(NN(dHe, DH, W1, d, dOe2, d, Re, d, DH), 1);
    pg.gemm(NT(O1, d, Wmo, d, O2, d, Rm, d, bmo, 0), 1);
    pg.gemm(NT(Oe1, d, Weo, d, Oe2, d, Re, d, beo, 0), 1);
    colsum(H1, Rm, DH, DH, Gd("blendshape_decoder.3.weight"), 0, grow);                                     // dw2 = sum_r g[r] H1[r]
    colsum(dH1, Rm, DH, DH, Gd("blendshape_decoder.0.bias"), 0, ones);
    RUN(pg.end_phase(st));
    // ================= P9: attention backward per (window, head); emotion value gradient; output projections; decoder[0] weight =================
    // Dispatch order: with fewer attention blocks than CUs (8 windows) they go first -- they are the phase's longest workgroups; from
    // 256 blocks on they would take every LDS slot of the chip (53 KB each, three per CU) and the products would run BEHIND them instead
    // of beside them: products first (64 windows: 0.2311 -> 0.2292 ms; 8 windows the other way round: 0.1289 against 0.1296)
    const bool p9_products_first = B * H >= 256;
    auto p9_products = [&]() {
        pg.gemm(NN(dO2, d, Wmo, d, dO1, d, Rm, d, d), 1);
        pg.gemm(NN(dOe2, d, Weo, d, dOe1, d, Re, d, d), 1);
        pg.gemm(TN(dH1, DH, O2, d, Gd("blendshape_decoder.0.weight"), d, DH, d, Rm), 1);
    };
    if (p9_products_first) p9_products();
    {
        ElemArgs e{};
        e.p0 = Qb; e.p1 = KV; e.p2 = Pm; e.p3 = dA; e.q0 = dKV; e.q1 = dQ_part; e.i0 = (int)d; e.i1 = (int)hd; e.i2 = (int)NKk; e.i3 = (int)H;
        e.mask = m_mel; e.f0 = keep_scale; e.u0 = scale_bits; e.u1 = c->opt.train_attn_regs ? 1u : 0u;
        const bool dma_blk = hd == 32 && !c->opt.train_attn_regs && NKk > 64 && NKk <= 80;
        pg.elem(OP_ATTN_BWD, e, B * H, dma_blk ? (size_t)kAttnDmaBwdLdsFloats * sizeof(float) : attn_lds);
    }
    {
        ElemArgs e{};
        e.p0 = dAe; e.q0 = dVe; e.n0 = B; e.i0 = (int)d; e.i1 = (int)hd; e.mask = m_emo; e.f0 = keep_scale;
        pg.elem(OP_EMO_REDUCE, e, blocks256(B * d));
    }
    if (!p9_products_first) p9_products();
    colsum(He, Re, DH, DH, Gd("blendshape_decoder.3.weight"), 1, grow + Rm);
    RUN(pg.end_phase(st));
    // ================= P10: dY; in_proj [K | V] gradients; dQ; out_proj gradients; emotion value projection backward =================
    // dY = dKV Wkv has the longest tile chain of the phase (K = 2 d: 16 steps at d_model 256).  At small batches -- fewer tiles than CUs --
    // it runs as TWO products over the K halves (a batch of two, 8 steps each) whose sum the reader takes: OP_LN_BWD of P11 adds the halves
    // and writes the sum back for P12's column sum
    const bool split_dy = ((R / 32) * (d / 64) * 2 <= 256 || c->opt.train_no_dy_split == 2) && (d == 64 || d == 256 || d == 512) && c->opt.train_no_dy_split != 1;
    if (split_dy) {
        GemmArgs g = NN(dKV, 2 * d, inw + d * d, d, dY, d, R, d, d);
        g.a_bs1 = d; g.b_bs1 = d * d; g.c_bs1 = R * d;
        pg.gemm(g, 2);
    } else {
        pg.gemm(NN(dKV, 2 * d, inw + d * d, d, dY, d, R, d, 2 * d), 1);
    }
    pg.gemm(TN(dKV, 2 * d, Y, d, gin_w + d * d, d, 2 * d, d, R), 1);
    colsum(dKV, R, 2 * d, 2 * d, gin_b + d, 0, ones);
    reduce(dQ_part, (int)B, 28 * d, 28 * d, dQb, 0);
    pg.gemm(TN(dO1, d, A, d, Gd("mel_attention.out_proj.weight"), d, d, d, Rm), 1);
    pg.gemm(TN(dOe1, d, Ae, d, Gd("emotion_attention.out_proj.weight"), d, d, d, Re), 1);
    colsum(dO1, Rm, d, d, Gd("mel_attention.out_proj.bias"), 0, ones);
    colsum(dOe1, Re, d, d, Gd("emotion_attention.out_proj.bias"), 0, ones);
    pg.gemm(NN(dVe, d, einw + 2 * d * d, d, dE, d, B, d, d), 1);
    // (the parameter gradients of the output projections and of decoder[0]'s emotion rows need nothing of P9: they sit here (behind the products the next phase waits for), where
    // a CU takes five tile workgroups -- the attention blocks' 64 KB of LDS leave room for two in P9)
    pg.gemm(TN(dO2, d, O1, d, Gd("mel_output_proj.weight"), d, d, d, Rm), 1);
    pg.gemm(TN(dOe2, d, Oe1, d, Gd("emotion_output_proj.weight"), d, d, d, Re), 1);
    colsum(dO2, Rm, d, d, Gd("mel_output_proj.bias"), 0, ones);
    colsum(dOe2, Re, d, d, Gd("emotion_output_proj.bias"), 0, ones);
    { GemmArgs g = TN(dHe, DH, Oe2, d, Gd("blendshape_decoder.0.weight"), d, DH, d, Re); g.beta = 1.f; pg.gemm(g, 1); }
    colsum(dHe, Re, DH, DH, Gd("blendshape_decoder.0.bias"), 1, ones);
    RUN(pg.end_phase(st));
    // ================= P11: LayerNorm backward (both streams); query-side gradients (+ the partial sums of P10's products) =========
    {
        ElemArgs b{};
        b.p0 = dY; b.p1 = Y0; b.p2 = P("mel_norm.weight"); b.p3 = mu; b.p4 = rs; b.q0 = dY0; b.q1 = Tm; b.n0 = R; b.i0 = (int)d;
        if (split_dy) { b.n1 = R * d; b.q2 = dY; }
        pg.elem(OP_LN_BWD, b, blocks4(R));
        ElemArgs e{};
        e.p0 = dE; e.p1 = E0; e.p2 = P("emotion_norm.weight"); e.p3 = emu; e.p4 = ers; e.q0 = dE0; e.q1 = Te; e.n0 = B; e.i0 = (int)d;
        pg.elem(OP_LN_BWD, e, blocks4(B));
    }
    // nothing of this phase may be deferred to a reduction in P12: the "early" event below declares its gradients final
    // (with train_split_min_k < 256 or d_model > 1024 the K = d products here would otherwise split)
    const bool allow_split_saved = pg.allow_split;
    pg.allow_split = false;
    // the emotion value projection's parameter gradients (K = B: never split): moved here from P10, whose 17 operations plus the
    // reductions of P9's split products (large batches) overflowed the phase's argument block into a second launch (5.8 us at 64
    // windows)
    pg.gemm(TN(dVe, d, E, d, gein_w + 2 * d * d, d, d, d, B), 1);                                           // only the V third of in_proj
    colsum(dVe, B, d, d, gein_b + 2 * d, 0, ones);
    pg.gemm(NN(dQb, d, inw, d, Gd("mouth_queries"), d, 28, d, d), 1);
    pg.gemm(TN(dQb, d, P("mouth_queries"), d, gin_w, d, d, d, 28), 1);       // query rows of in_proj: dWq = dQ^T mq, dbq = column sums of dQ
    colsum(dQb, 28, d, d, gin_b, 0, ones);
    pg.allow_split = allow_split_saved;
    RUN(pg.end_phase(st));
    if (!pg.pend_prev.empty()) return fail(KM_ERR_UNSUPPORTED, "training program: a reduction is still pending behind phase 11 (internal)");
    // everything but the "late" group of the bucket (km_train_init) is final: a side stream may start its all-reduce
    HIP_TRY(hipEventRecord((hipEvent_t)c->tr_ev[0], st));
    c->tr_early_recorded = true;
    // ================= P12: channel encoder gradients; LayerNorm parameters; emotion encoder =================
    // dWce (d x KT) = dY0^T XT over ALL rows (window, channel) of the batch: with the transposed input it is ONE product with a
    // uniform k stride (round 3: a contraction batch over the windows on the register tile, 19 us at 8 windows); split along K
    // like every other gradient product when the batch is large (partials summed by P13)
    // (P12 is the last phase with work of its own: a product cut here needs a launch just for its reduction, which pays only for
    // the long chains of large batches)
    const int min_k_saved = pg.min_k;
    if (pg.min_k < 1024) pg.min_k = 1024;
    pg.gemm(TN(dY0, d, xf ? x2 : X, KP, Gd("mel_channel_encoder.weight"), KT, d, KT, R), 1);
    colsum(dY0, R, d, d, Gd("mel_channel_encoder.bias"), 0, ones);
    colsum(Tm, R, d, d, Gd("mel_norm.weight"), 0, ones);
    colsum(dY, R, d, d, Gd("mel_norm.bias"), 0, ones);
    colsum(Te, B, d, d, Gd("emotion_norm.weight"), 0, ones);
    colsum(dE, B, d, d, Gd("emotion_norm.bias"), 0, ones);
    pg.gemm(TN(dE0, d, emo, ED, Gd("emotion_encoder.weight"), ED, d, ED, B), 1);
    colsum(dE0, B, d, d, Gd("emotion_encoder.bias"), 0, ones);
    pg.min_k = min_k_saved;
    RUN(pg.end_phase(st));
    // ================= P13: the partial sums of P12's split products (large batches) =================
    RUN(pg.end_phase(st));
    return KM_OK;
}

// per-window audio energy for the audio-visual loss term: mean over T of the L2 norm over D (losses.py:352-358)
__global__ __launch_bounds__(64) void audio_energy_kernel(const float* __restrict__ f, int T, int D, float* __restrict__ out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) {
        const float* row = f + ((int64_t)b * T + t) * D;
        float s = 0.f;
        for (int i = lane; i < D; i += 64) s += row[i] * row[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        acc += sqrtf(s);
    }
    if (lane == 0) out[b] = acc / (float)T;
}

int launch_audio_energy(const float* feats, int64_t B, int64_t T, int64_t D, float* out, void* stream) {
    hipLaunchKernelGGL(audio_energy_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, feats, (int)T, (int)D, out);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

}  // namespace km
