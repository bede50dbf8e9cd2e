// Attention core of the KoeMorph hot path for gfx950 (MI355X): one fused kernel per window.
//
// Replaces DualStreamCrossAttention.forward (reference src/model/dual_stream_attention.py:162-280)
// in eval mode for the production configuration d_model=256, mel_sequence_length=256, 8 heads.
//
// One 512-thread workgroup (8 waves, 2 per SIMD) owns one window and keeps every intermediate
// on chip; all contractions run on the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32
// (bitwise a k-ordered fmaf chain), so the result differs from the PyTorch CPU forward only by
// summation order.
//
//   phase 0  X (264 x 80, [t][c]) <- mel rows, zero pad / truncate to T, + 3 short-term rows (:189-208)
//   phase 1  Y = X^T Wce^T + b   (80 x 256, K = 259)            wave w owns output columns 32w..32w+31
//            LayerNorm(eps 1e-5) two-pass, cross-wave row statistics through LDS (:211-212)
//            Y -> LDS [80][264] (aliases X)
//   phase 2  per head h = wave:  S^T = Y Qk_h^T  (80 keys x 32 query slots, K = 256), softmax over keys
//   phase 3                      V_h = Y Wv_h^T  (80 x 32, K = 256)       -- same sweep over Y as phase 2
//   phase 4                      O_h^T = V_h^T P_h^T  (32 x 32, K = 80)   -- both operands ARE the phase 2/3
//            accumulators: the C/D layout of the 16x16x4 MFMA (col = lane&15, row = 4*(lane>>4)+reg)
//            is its own A/B operand layout when the contraction runs over the row index, so the
//            attention weights never leave registers.
//   phase 5  O -> LDS [32 q][256], hidden^T = Wf^T O^T (128 x 32, K = 256), ReLU, dot w2, sigmoid,
//            stream weights, clamp, optional EMA (:231-270 folded on the host, km_host.cpp)
//
// LDS operand images: X rows are 80 floats (320 B = 16 banks mod 32, so two consecutive rows fill the
// 32 banks of a ds_read_b32 half-wave); Y / O rows are 264 floats (stride = 8 mod 64 dwords) which makes
// every 16-lane group of the ds_read_b128 A-fragment reads hit 16 distinct 16-byte slots.
// The contraction index inside a 16-wide k block is permuted (lane group g reads k = 16kb+4g..+3 as one
// b128) -- legal because the weights are packed on the host with the same permutation.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "km_context.h"
#include "km_device.h"

namespace km {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define KM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(KM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace fused {
constexpr int D = 256, T = 256, NK = 80, NQ = 28, DH = 128;
constexpr int KT = 259, KTP = 264, YS = 264, NW = 8, NT = 512;
constexpr int KP = 33;               // k-step pairs of the encoder GEMM (66 steps of 4 >= 259)
constexpr int KB = 16;               // 16-wide k blocks of the d=256 contractions
constexpr int R1_FLOATS = KTP * NK;  // 21120: X [264][80]  ==  Y [80][264]
constexpr int R2_FLOATS = 2 * NW * NK + 2 * NK;   // 1440: cross-wave reduction scratch: one [80 rows][8 waves] image of partials and one [80] row of totals per LayerNorm pass
constexpr int LDS_BYTES = (R1_FLOATS + R2_FLOATS) * 4;
// split-bf16 variant (NP pieces per fp32 value): Y lives in LDS as NP planes [80][YSB] of bf16
constexpr int YSB = 264;                            // bf16 per plane row: 132 dwords = 4 (mod 64), conflict-free b128 fragments
constexpr int PLANE_FLOATS = NK * YSB / 2;          // 10560
// ... and X as NP planes [80 channels][XSB frames] (transposed while it is staged): 288 = 9 k blocks of 32 >= 259 frames,
// 296 bf16 = 148 dwords = 20 (mod 64): the 16 rows of a b128 beat still hit 64 different banks
constexpr int XSB = 296, XKB = 9;
constexpr int XPLANE_FLOATS = NK * XSB / 2;         // 11840
constexpr int r1_floats(int np) { return np * XPLANE_FLOATS > R1_FLOATS ? np * XPLANE_FLOATS : R1_FLOATS; }
constexpr int lds_bytes(int np) { return (r1_floats(np) + R2_FLOATS) * 4; }
}  // namespace fused

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));

struct CoreArgs {
    const float* mel;        // (B, t_in, 80)
    const float* mel_short;  // (B, 3, 80)
    const float* zemo;       // (B) emotion-stream logit
    int t_in;
    const float *wce_p, *bce, *ln_g, *ln_b, *qk_p, *wv_p, *wf_p, *bf, *w2, *b2, *wsum;
    const void* qkv_s;       // split-bf16 image of qk_p / wv_p (NP > 0 variants only)
    const void* wce_s;       // ... and of wce_p
    float* out;    // (B, 52)
    float* raw;    // (B, 52) or null
    float* attn;   // (B, 28, 80) or null (ATTN variant only)
    float* state;  // (B, 52) EMA state or null
    int first;
    float alpha;
    // FUSE_DB variant: the front end's power-mel is converted to log-mel while it is staged into LDS
    const float* melpow;  // (B, n_frames, 80) power-mel
    unsigned* melmax;     // (B) window maxima (float bits); the entry is re-zeroed for the next call
    int n_frames;
    LogParams lp;
    // sequence mode: one emotion logit per clip, shared by its windows: zemo[(win0 + b) / zemo_div]
    int64_t win0;
    int zemo_div;
    // streaming mode: frames kept by the truncate / repeat-last policy (255 of 256, mel_sliding_window.py:300-307),
    // per-stream readiness (ring full) and per-stream 'EMA started' flags
    int n_use;
    // shared-frame sequence mode: frames 1 .. n_frames-2 of window (clip, i) are rows i*seq_stride + fr of the clip's
    // power-mel image, frames 0 and n_frames-1 (zero-padded at the window boundary) come from seq_edge (2 rows/window)
    const float* seq_pow;    // (clips, seq_nfc, 80) or null
    const float* seq_edge;   // (clips * seq_n, 2, 80)
    int seq_nfc, seq_stride, seq_n;
    const unsigned char* ready;   // (B) or null: windows with ready[b] == 0 are skipped entirely
    unsigned char* started;       // (B) or null: first = !started[b], then started[b] = 1
};

// blendshape index -> mouth query slot (MOUTH_INDICES = 14..40, 51; dual_stream_attention.py:14-45)
// or -1 for the 24 expression rows
__device__ __forceinline__ int mouth_slot_of(int i) { return (i >= 14 && i <= 40) ? i - 14 : (i == 51 ? 27 : -1); }

// NP > 0 (experimental, opt-in: KM_CORE_SPLIT = 3 or 6 terms): phases 2+3 on the bf16 matrix pipe with every
// fp32 operand split into NP bf16 pieces and the products of weight >= 2^-16 (NP 2: three) or >= 2^-24 (NP 3: six)
// accumulated in fp32.  Everything else is unchanged.
template <bool ATTN, bool FUSE_DB, int NP = 0>
__global__ __launch_bounds__(512) void core_fused_kernel(CoreArgs a) {
    using namespace fused;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* R1 = smem;
    float* R2 = smem + r1_floats(NP);

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    if constexpr (FUSE_DB) {
        if (a.ready && !a.ready[b]) return;     // stream still filling its ring (workgroup-uniform)
    }

    // Operands that do not depend on the window are requested before anything waits: the first two k pairs of the
    // encoder weight, the bias and the LayerNorm affine of this lane's two columns (each was an exposed L2 round trip
    // at its point of use).
    const int n0 = 32 * wave + j;      // this lane's two columns are n0 and n0 + 16
    float4 bw0 = make_float4(0.f, 0.f, 0.f, 0.f), bw1 = bw0;
    if constexpr (NP == 0) {
        const float4* wp0 = reinterpret_cast<const float4*>(a.wce_p) + (size_t)wave * KP * 64 + lane;
        bw0 = wp0[0];
    }
    const float bb0 = a.bce[n0], bb1 = a.bce[n0 + 16];
    const float g0 = a.ln_g[n0], g1 = a.ln_g[n0 + 16], be0 = a.ln_b[n0], be1 = a.ln_b[n0 + 16];
    // ... and so are the decoder's per-unit bias / output weight of phase 5 and the tail's scalars
    float bf4[4], w24[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { bf4[r] = a.bf[16 * wave + 4 * g + r]; w24[r] = a.w2[16 * wave + 4 * g + r]; }
    const float tail_b2 = a.b2[0], tail_ws = tid < 52 ? a.wsum[tid] : 0.f;

    // ---- phase 0: X -> LDS, [t][c] exactly as the caller's (t_in, 80) rows --------------------
    if constexpr (FUSE_DB) {
        // rows come from the power-mel workspace; dB / log conversion on the fly (bit-identical to
        // mel_log_kernel: same log_one()), long rows = first min(F, T) frames, short rows = last 3 frames
        // (simplified_dual_stream_model.py:199-214)
        float ref_db, floor_db;
        log_window_consts(a.lp, __uint_as_float(a.melmax[b]), ref_db, floor_db);
        const int F = a.n_frames;          // frames the front end computed (the dB reference spans all of them)
        const int U = a.n_use;             // rows kept: truncated, or padded by repeating the last frame
        const int tv = U < T ? U : T;
        const float4* src = reinterpret_cast<const float4*>(a.melpow + (int64_t)b * F * NK);
        const float4 *seq_rows = nullptr, *seq_e = nullptr;
        if (a.seq_pow) {
            const int64_t gw = a.win0 + b, clip = gw / a.seq_n, wi = gw - clip * a.seq_n;
            seq_rows = reinterpret_cast<const float4*>(a.seq_pow + (clip * a.seq_nfc + wi * a.seq_stride) * NK);
            seq_e = reinterpret_cast<const float4*>(a.seq_edge + gw * 2 * NK);
        }
        auto row4 = [&](int fr) -> const float4* {     // the 20 float4 of power-mel row fr of this window
            if (!seq_rows) return src + fr * 20;
            return fr == 0 ? seq_e : (fr == F - 1 ? seq_e + 20 : seq_rows + fr * 20);
        };
        float4* dst = reinterpret_cast<float4*>(R1);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (NP > 0) {
            // Split variant: X goes to LDS transposed, as NP bf16 planes [80 channels][XSB frames].  A thread takes EIGHT
            // consecutive frames of one float4 of channels (loads still coalesced along the 20 float4 of a row), converts,
            // splits, and writes the eight frames of each channel and piece as ONE 16-byte store; frames 259 .. 287 (k
            // padding) come out of the same loop as zeros.  36 blocks of 8 frames x 20 = 720 units over 512 threads.
            auto src_row = [&](int t) -> int {          // frame feeding encoder input row t, or -1 (zero row)
                if (t < T) return t < tv ? (t < F ? t : F - 1) : -1;
                if (t >= T + 3) return -1;
                const int r = t - T;
                int fr = -1;
                if (U >= 3) fr = U - 3 + r; else if (r < U) fr = r;
                return fr >= F ? F - 1 : fr;
            };
            auto stage_planes = [&](auto mode) {
                constexpr int MODE = decltype(mode)::value;
#pragma unroll 1
                for (int unit = tid; unit < XKB * 4 * 20; unit += NT) {
                    const int t8 = unit / 20, c4 = unit - t8 * 20;
                    float4 xv8[8];
                    unsigned okm = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int fr = src_row(8 * t8 + q);
                        okm |= (fr >= 0 ? 1u : 0u) << q;
                        xv8[q] = fr >= 0 ? row4(fr)[c4] : z4;
                    }
                    float xe[4][8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float4 v = (okm >> q) & 1 ? log_four_t<MODE>(a.lp, xv8[q], ref_db, floor_db) : z4;
                        xe[0][q] = v.x; xe[1][q] = v.y; xe[2][q] = v.z; xe[3][q] = v.w;
                    }
                    unsigned short* xp = reinterpret_cast<unsigned short*>(R1) + (4 * c4) * XSB + 8 * t8;
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            bf16x8 pk;
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const __bf16 pbf = (__bf16)xe[e][q];
                                pk[q] = pbf;
                                xe[e][q] -= (float)pbf;
                            }
                            *reinterpret_cast<bf16x8*>(xp + pc * (NK * XSB) + e * XSB) = pk;
                        }
                }
            };
            if (a.lp.log_mode == KM_LOG_LN_EPS) stage_planes(std::integral_constant<int, KM_LOG_LN_EPS>{});
            else stage_planes(std::integral_constant<int, KM_LOG_DB_MAX>{});
        } else {
        // all T*20/NT = 10 row loads of a thread (and the short-row load) are in flight before the first conversion
        constexpr int NLD = T * 20 / NT;
        static_assert(NLD * NT == T * 20, "phase 0 assumes T*20 is a multiple of the workgroup size");
        float4 xv[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = tid + NT * u;
            int fr = i / 20;
            const int c4 = i - fr * 20;
            fr = fr < F ? fr : F - 1;
            xv[u] = i < tv * 20 ? row4(fr)[c4] : z4;
        }
        float4 sv = z4;
        bool s_ok = false;
        if (tid < 60) {
            const int r = tid / 20;
            int fr = -1;
            if (U >= 3) fr = U - 3 + r; else if (r < U) fr = r;
            if (fr >= F) fr = F - 1;
            s_ok = fr >= 0;
            if (s_ok) sv = row4(fr)[tid - r * 20];
        }
        auto convert = [&](auto mode) {          // the conversion mode is uniform: one branch, not one per value
            constexpr int MODE = decltype(mode)::value;
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int i = tid + NT * u;
                dst[i] = i < tv * 20 ? log_four_t<MODE>(a.lp, xv[u], ref_db, floor_db) : z4;
            }
            if (tid < 60) dst[T * 20 + tid] = s_ok ? log_four_t<MODE>(a.lp, sv, ref_db, floor_db) : z4;
        };
        if (a.lp.log_mode == KM_LOG_LN_EPS) convert(std::integral_constant<int, KM_LOG_LN_EPS>{});
        else convert(std::integral_constant<int, KM_LOG_DB_MAX>{});
        if (tid >= 64 && tid < 64 + (KTP - KT) * 20) dst[KT * 20 + tid - 64] = z4;
        }
    } else {
        const int tv = a.t_in < T ? a.t_in : T;   // rows beyond T are truncated (:200-202)
        const float4* src = reinterpret_cast<const float4*>(a.mel + (int64_t)b * a.t_in * NK);
        float4* dst = reinterpret_cast<float4*>(R1);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = tid; i < tv * 20; i += NT) dst[i] = src[i];
        for (int i = tv * 20 + tid; i < T * 20; i += NT) dst[i] = z4;                       // zero pad (:193-199)
        const float4* s4 = reinterpret_cast<const float4*>(a.mel_short + (int64_t)b * 3 * NK);
        if (tid < 60) dst[T * 20 + tid] = s4[tid];                                           // 3 short rows (:205-208)
        if (tid >= 64 && tid < 64 + (KTP - KT) * 20) dst[KT * 20 + tid - 64] = z4;          // k padding rows
    }
    __syncthreads();
    if constexpr (FUSE_DB) {
        if (tid == 0) a.melmax[b] = 0u;     // every thread has read it: hand a clean slot to the next front-end launch
    }

    // ---- phase 1: channel encoder GEMM -----------------------------------------------------
    f32x4 acc[5][2];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) { acc[mt][0] = f32x4{0, 0, 0, 0}; acc[mt][1] = f32x4{0, 0, 0, 0}; }
    if constexpr (NP > 0) {
        // [wave][k block of 32][column tile][piece][lane] x 16 bytes, one k block prefetched; A = b128 fragments of the X planes
        const u32x4c* wp = reinterpret_cast<const u32x4c*>(a.wce_s) + (size_t)wave * XKB * 2 * NP * 64 + lane;
        const unsigned short* Xp = reinterpret_cast<const unsigned short*>(R1) + j * XSB + 8 * g;
        u32x4c bw[2][NP];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) bw[t][pc] = wp[(t * NP + pc) * 64];
        for (int kb = 0; kb < XKB; ++kb) {
            const int kn = kb + 1 < XKB ? kb + 1 : kb;
            u32x4c bn[2][NP];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) bn[t][pc] = wp[((kn * 2 + t) * NP + pc) * 64];
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
                bf16x8 ap[NP];
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    ap[pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4c*>(Xp + pc * (NK * XSB) + 16 * mt * XSB + 32 * kb));
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 c4 = acc[mt][t];
                    if constexpr (NP == 3) {
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], __builtin_bit_cast(bf16x8, bw[t][1]), c4, 0, 0, 0);
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], __builtin_bit_cast(bf16x8, bw[t][2]), c4, 0, 0, 0);
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[2], __builtin_bit_cast(bf16x8, bw[t][0]), c4, 0, 0, 0);
                    }
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], __builtin_bit_cast(bf16x8, bw[t][1]), c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], __builtin_bit_cast(bf16x8, bw[t][0]), c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], __builtin_bit_cast(bf16x8, bw[t][0]), c4, 0, 0, 0);
                    acc[mt][t] = c4;
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) bw[t][pc] = bn[t][pc];
        }
    } else {
        const float4* wp = reinterpret_cast<const float4*>(a.wce_p) + (size_t)wave * KP * 64 + lane;
        float4 bw = bw0;
        for (int kp = 0; kp < KP; ++kp) {
            const float4 bn = wp[(size_t)(kp + 1 < KP ? kp + 1 : kp) * 64];   // prefetch next pair
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) {
                const float* xr = R1 + (4 * (2 * kp + ds) + g) * NK + j;
                const float b0 = ds ? bw.z : bw.x, b1 = ds ? bw.w : bw.y;
                float av[5];
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) av[mt] = xr[16 * mt];
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) {
                    acc[mt][0] = KM_MFMA(av[mt], b0, acc[mt][0]);
                    acc[mt][1] = KM_MFMA(av[mt], b1, acc[mt][1]);
                }
            }
            bw = bn;
        }
    }
    // the first k block of the phase 2+3 operand images is requested here, ahead of LayerNorm and its two barriers
    f32x4 q0 = f32x4{0, 0, 0, 0}, q1 = q0, v0 = q0, v1 = q0;
    if constexpr (NP == 0) {
        const f32x4* qp0 = reinterpret_cast<const f32x4*>(a.qk_p) + (size_t)wave * KB * 2 * 64 + lane;
        const f32x4* vp0 = reinterpret_cast<const f32x4*>(a.wv_p) + (size_t)wave * KB * 2 * 64 + lane;
        q0 = qp0[0]; q1 = qp0[64]; v0 = vp0[0]; v1 = vp0[64];
    }
    // bias
    {
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[mt][0][r] += bb0; acc[mt][1][r] += bb1; }
    }
    // LayerNorm over the 256 columns of each of the 80 rows: two-pass (mean, then centred squares)
    float mean[5][4], rstd[5][4];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float part[5][4];
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v;
                if (pass == 0) {
                    v = acc[mt][0][r] + acc[mt][1][r];
                } else {
                    const float d0 = acc[mt][0][r] - mean[mt][r], d1 = acc[mt][1][r] - mean[mt][r];
                    v = d0 * d0 + d1 * d1;
                }
                part[mt][r] = row16_sum(v);
            }
        // partial sums of this wave's 32 columns -> [row][wave]; the two passes use separate images, so the only
        // barrier a pass needs is the one between its writes and its reads
        float* P = R2 + pass * (NW * NK);
        if (j == 0) {
#pragma unroll
            for (int mt = 0; mt < 5; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) P[(16 * mt + 4 * g + r) * NW + wave] = part[mt][r];
        }
        __syncthreads();   // pass 0: also fences every wave's last read of X
        // row totals: wave w adds up the 8 partials (waves in index order, as before) of rows 10 w .. 10 w + 9, one row per
        // lane, then every lane fetches its 20 totals with five 16-byte reads -- a second barrier, but 5 instead of 40
        // ds_read_b128 per lane and pass
        float* Tt = R2 + 2 * NW * NK + pass * NK;
        if (lane < NK / NW) {
            const int row = wave * (NK / NW) + lane;
            const f32x4* pr = reinterpret_cast<const f32x4*>(P + row * NW);
            const f32x4 lo = pr[0], hi = pr[1];
            float s = 0.f;
            s += lo[0]; s += lo[1]; s += lo[2]; s += lo[3];
            s += hi[0]; s += hi[1]; s += hi[2]; s += hi[3];
            Tt[row] = s;
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(Tt + 16 * mt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (pass == 0) mean[mt][r] = t4[r] * (1.0f / D);
                else rstd[mt][r] = __builtin_amdgcn_rsqf(t4[r] * (1.0f / D) + 1e-5f);     // v_rsq_f32 (1 ulp)
            }
        }
    }
    {
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y0 = (acc[mt][0][r] - mean[mt][r]) * rstd[mt][r] * g0 + be0;
                const float y1 = (acc[mt][1][r] - mean[mt][r]) * rstd[mt][r] * g1 + be1;
                if constexpr (NP == 0) {
                    float* yr = R1 + (16 * mt + 4 * g + r) * YS + n0;
                    yr[0] = y0;
                    yr[16] = y1;
                } else {       // NP bf16 pieces per value, one plane per piece
                    unsigned short* yp = reinterpret_cast<unsigned short*>(R1) + (16 * mt + 4 * g + r) * YSB + n0;
                    float r0 = y0, r1 = y1;
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc) {
                        const __bf16 p0 = (__bf16)r0, p1 = (__bf16)r1;
                        yp[pc * (NK * YSB)] = __builtin_bit_cast(unsigned short, p0);
                        yp[pc * (NK * YSB) + 16] = __builtin_bit_cast(unsigned short, p1);
                        r0 -= (float)p0;
                        r1 -= (float)p1;
                    }
                }
            }
    }
    __syncthreads();

    // ---- phases 2+3: S^T = Y Qk_h^T and V_h = Y Wv_h^T in one sweep over Y -------------------
    f32x4 S[5][2], V[5][2];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) {
        S[mt][0] = f32x4{0, 0, 0, 0}; S[mt][1] = f32x4{0, 0, 0, 0};
        V[mt][0] = f32x4{0, 0, 0, 0}; V[mt][1] = f32x4{0, 0, 0, 0};
    }
    if constexpr (NP > 0) {
        // [head = wave][k block of 32][tile: Qk 0, Qk 1, Wv 0, Wv 1][piece][lane] x 16 bytes, one k block prefetched
        const u32x4c* wp = reinterpret_cast<const u32x4c*>(a.qkv_s) + (size_t)wave * 8 * 4 * NP * 64 + lane;
        const unsigned short* Yp = reinterpret_cast<const unsigned short*>(R1) + j * YSB + 8 * g;
        u32x4c bw[4][NP];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) bw[t][pc] = wp[(t * NP + pc) * 64];
        for (int kb = 0; kb < 8; ++kb) {
            const int kn = kb + 1 < 8 ? kb + 1 : kb;
            u32x4c bn[4][NP];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) bn[t][pc] = wp[((kn * 4 + t) * NP + pc) * 64];
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
                bf16x8 ap[NP];
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    ap[pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4c*>(Yp + pc * (NK * YSB) + 16 * mt * YSB + 32 * kb));
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f32x4 c4 = t < 2 ? S[mt][t] : V[mt][t - 2];
                    // smallest terms first: (1,1) (0,2) (2,0) | (0,1) (1,0) | (0,0)
                    if constexpr (NP == 3) {
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], __builtin_bit_cast(bf16x8, bw[t][1]), c4, 0, 0, 0);
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], __builtin_bit_cast(bf16x8, bw[t][2]), c4, 0, 0, 0);
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[2], __builtin_bit_cast(bf16x8, bw[t][0]), c4, 0, 0, 0);
                    }
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], __builtin_bit_cast(bf16x8, bw[t][1]), c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], __builtin_bit_cast(bf16x8, bw[t][0]), c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], __builtin_bit_cast(bf16x8, bw[t][0]), c4, 0, 0, 0);
                    if (t < 2) S[mt][t] = c4; else V[mt][t - 2] = c4;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) bw[t][pc] = bn[t][pc];
        }
    } else {
        const f32x4* qp = reinterpret_cast<const f32x4*>(a.qk_p) + (size_t)wave * KB * 2 * 64 + lane;
        const f32x4* vp = reinterpret_cast<const f32x4*>(a.wv_p) + (size_t)wave * KB * 2 * 64 + lane;
        for (int kb = 0; kb < KB; ++kb) {
            const int kn = kb + 1 < KB ? kb + 1 : kb;
            const f32x4 q0n = qp[(size_t)kn * 128], q1n = qp[(size_t)kn * 128 + 64];
            const f32x4 v0n = vp[(size_t)kn * 128], v1n = vp[(size_t)kn * 128 + 64];
            f32x4 ya[5];
#pragma unroll
            for (int mt = 0; mt < 5; ++mt)
                ya[mt] = *reinterpret_cast<const f32x4*>(R1 + (16 * mt + j) * YS + 16 * kb + 4 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) {
                    const float av = ya[mt][s];
                    S[mt][0] = KM_MFMA(av, q0[s], S[mt][0]);
                    S[mt][1] = KM_MFMA(av, q1[s], S[mt][1]);
                    V[mt][0] = KM_MFMA(av, v0[s], V[mt][0]);
                    V[mt][1] = KM_MFMA(av, v1[s], V[mt][1]);
                }
            q0 = q0n; q1 = q1n; v0 = v0n; v1 = v1n;
        }
    }
    // softmax over the 80 keys of each query column: 20 values in-lane, then across the 4 lane groups
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float m = S[0][qt][0];
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, S[mt][qt][r]);
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.f;
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f((S[mt][qt][r] - m) * 1.44269504088896341f);   // v_exp_f32: arguments <= 0
                S[mt][qt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = __builtin_amdgcn_rcpf(sum);                                  // v_rcp_f32 (1 ulp), sum in [1, 80]
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[mt][qt][r] *= inv;
    }

    // ---- phase 4: O_h^T (32 d x 32 q) = V_h^T P_h^T, operands straight from the accumulators ----
    f32x4 O[2][2];
    O[0][0] = f32x4{0, 0, 0, 0}; O[0][1] = f32x4{0, 0, 0, 0};
    O[1][0] = f32x4{0, 0, 0, 0}; O[1][1] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int mt = 0; mt < 5; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            O[0][0] = KM_MFMA(V[mt][0][r], S[mt][0][r], O[0][0]);
            O[0][1] = KM_MFMA(V[mt][0][r], S[mt][1][r], O[0][1]);
            O[1][0] = KM_MFMA(V[mt][1][r], S[mt][0][r], O[1][0]);
            O[1][1] = KM_MFMA(V[mt][1][r], S[mt][1][r], O[1][1]);
        }

    __syncthreads();   // every wave is done reading Y; R1 becomes O [32 q][264]
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
            *reinterpret_cast<f32x4*>(R1 + (16 * qt + j) * YS + 32 * wave + 16 * dt + 4 * g) = O[dt][qt];

    if constexpr (ATTN) {
        // head-averaged attention weights (:225-230, average_attn_weights=True), summed in head order
        float* Pa = R1 + 32 * YS;   // [28][80] behind the O image
        for (int h = 0; h < NW; ++h) {
            if (wave == h) {
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    const int q = 16 * qt + j;
                    if (q < NQ) {
#pragma unroll
                        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float* p = Pa + q * NK + 16 * mt + 4 * g + r;
                                *p = (h == 0 ? 0.f : *p) + S[mt][qt][r];
                            }
                    }
                }
            }
            __syncthreads();
        }
        float* ao = a.attn + (int64_t)b * NQ * NK;
        for (int i = tid; i < NQ * NK; i += NT) ao[i] = Pa[i] * (1.0f / NW);
    } else {
        __syncthreads();
    }

    // ---- phase 5: hidden^T (128 x 32 q) = Wf^T O^T, wave w owns hidden units 16w..16w+15 -------
    f32x4 Z[2];
    Z[0] = f32x4{0, 0, 0, 0}; Z[1] = f32x4{0, 0, 0, 0};
    {
        const f32x4* fp = reinterpret_cast<const f32x4*>(a.wf_p) + (size_t)wave * KB * 64 + lane;
#pragma unroll 4
        for (int kb = 0; kb < KB; ++kb) {
            const f32x4 wa = fp[(size_t)kb * 64];
            const f32x4 o0 = *reinterpret_cast<const f32x4*>(R1 + j * YS + 16 * kb + 4 * g);
            const f32x4 o1 = *reinterpret_cast<const f32x4*>(R1 + (16 + j) * YS + 16 * kb + 4 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                Z[0] = KM_MFMA(wa[s], o0[s], Z[0]);
                Z[1] = KM_MFMA(wa[s], o1[s], Z[1]);
            }
        }
    }
    {
        float zp[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float bfv = bf4[r], w2v = w24[r];
            zp[0] += fmaxf(Z[0][r] + bfv, 0.f) * w2v;
            zp[1] += fmaxf(Z[1][r] + bfv, 0.f) * w2v;
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            zp[qt] += __shfl_xor(zp[qt], 16);
            zp[qt] += __shfl_xor(zp[qt], 32);
        }
        if (g == 0) { R2[wave * 32 + j] = zp[0]; R2[wave * 32 + 16 + j] = zp[1]; }
    }
    __syncthreads();
    if (tid < 52) {
        const int slot = mouth_slot_of(tid);
        float z;
        if (slot >= 0) {
            z = tail_b2;
#pragma unroll
            for (int w = 0; w < NW; ++w) z += R2[w * 32 + slot];
        } else {
            z = a.zemo[(a.win0 + b) / a.zemo_div];
        }
        const float bs = 1.0f / (1.0f + expf(-z));                      // nn.Sigmoid (:155)
        float val = fminf(fmaxf(tail_ws * bs, 0.f), 1.f);               // stream weights + clamp (:264-270)
        if (a.raw) a.raw[(int64_t)b * 52 + tid] = bs;
        if (a.state) {                                                  // EMA (simplified_dual_stream_model.py:357-366)
            float* st = a.state + (int64_t)b * 52 + tid;
            const bool first = a.started ? !a.started[b] : (a.first != 0);
            if (!first) val = a.alpha * val + (1.0f - a.alpha) * (*st);
            *st = val;
        }
        a.out[(int64_t)b * 52 + tid] = val;
    }
    if constexpr (FUSE_DB) {
        // all 52 threads above read started[b] before this barrier-free point?  No: order it explicitly.
        __syncthreads();
        if (tid == 0 && a.started) a.started[b] = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// Emotion stream: z_e[b] = w2 . relu(LN(Wee emo + bee) We2 + be2) + b2   (one scalar per window: the
// 24 expression queries all attend to the single eGeMAPS token, dual_stream_attention.py:234-240).
// 0.2 MFLOP / window after folding; the kernel is LATENCY bound (two dependent weight sweeps), so it
// is built for memory-level parallelism: 1024 threads per 4 windows, the contraction index split 4-way
// (layer 1) / 8-way (layer 2) across thread groups with 16 independent coalesced loads in flight per
// thread, partial sums combined through LDS in a fixed order (deterministic).  Generic in (ED, d, DH).
// ---------------------------------------------------------------------------------------------
constexpr int EWPB = 4;      // windows per workgroup
constexpr int ENT = 1024;    // threads per workgroup
constexpr int EKQ1 = 4;      // k-split of layer 1 (256 output columns x 4)
constexpr int EKQ2 = 8;      // k-split of layer 2 (128 output columns x 8)

__global__ __launch_bounds__(1024) void emotion_kernel(const float* __restrict__ emo, int64_t B, int ED, int d, int DH,
                                                       const float* __restrict__ wee_t, const float* __restrict__ bee,
                                                       const float* __restrict__ lg, const float* __restrict__ lb,
                                                       const float* __restrict__ we2, const float* __restrict__ be2,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       float* __restrict__ zemo) {
    extern __shared__ __attribute__((aligned(16))) float es[];
    float* emo_s = es;                        // [EWPB][ED]
    float* e1 = emo_s + EWPB * ED;            // [EWPB][d]
    float* part = e1 + EWPB * d;              // [max(EKQ1*EWPB*d, EKQ2*EWPB*DH)]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * EWPB;
    const int nb = (int)((B - b0) < EWPB ? (B - b0) : EWPB);
    for (int i = tid; i < EWPB * ED; i += ENT) {
        const int w = i / ED;
        emo_s[i] = w < nb ? emo[b0 * ED + i] : 0.f;
    }
    __syncthreads();
    {   // layer 1 partial sums: thread (column n, k-quarter kq)
        const int kq = tid >> 8, kchunk = (ED + EKQ1 - 1) / EKQ1;
        const int k0 = kq * kchunk, k1 = (k0 + kchunk) < ED ? (k0 + kchunk) : ED;
        for (int n = tid & 255; n < d; n += 256) {
            float acc[EWPB];
#pragma unroll
            for (int w = 0; w < EWPB; ++w) acc[w] = 0.f;
            const float* wp = wee_t + n;
            int k = k0;
            for (; k + 16 <= k1; k += 16) {       // 16 independent coalesced loads in flight, then the FMAs
                float wv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = wp[(size_t)(k + u) * d];
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int w = 0; w < EWPB; ++w) acc[w] = fmaf(emo_s[w * ED + k + u], wv[u], acc[w]);
            }
            for (; k < k1; ++k) {
                const float wv = wp[(size_t)k * d];
#pragma unroll
                for (int w = 0; w < EWPB; ++w) acc[w] = fmaf(emo_s[w * ED + k], wv, acc[w]);
            }
#pragma unroll
            for (int w = 0; w < EWPB; ++w) part[(kq * EWPB + w) * d + n] = acc[w];
        }
    }
    __syncthreads();
    for (int i = tid; i < EWPB * d; i += ENT) {
        const int w = i / d, n = i - w * d;
        float s = bee[n];
#pragma unroll
        for (int kq = 0; kq < EKQ1; ++kq) s += part[(kq * EWPB + w) * d + n];
        e1[i] = s;
    }
    __syncthreads();
    if (wave < EWPB) {   // LayerNorm (eps 1e-5), one wave per window, two-pass
        const int w = wave;
        float s = 0.f;
        for (int n = lane; n < d; n += 64) s += e1[w * d + n];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / d;
        float v = 0.f;
        for (int n = lane; n < d; n += 64) { const float t = e1[w * d + n] - mean; v += t * t; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        const float rstd = 1.0f / sqrtf(v / d + 1e-5f);
        for (int n = lane; n < d; n += 64) e1[w * d + n] = (e1[w * d + n] - mean) * rstd * lg[n] + lb[n];
    }
    __syncthreads();
    {   // layer 2 partial sums: thread (hidden unit m, k-eighth kq)
        const int kq = tid >> 7, kchunk = (d + EKQ2 - 1) / EKQ2;
        const int k0 = kq * kchunk, k1 = (k0 + kchunk) < d ? (k0 + kchunk) : d;
        for (int m = tid & 127; m < DH; m += 128) {
            float acc[EWPB];
#pragma unroll
            for (int w = 0; w < EWPB; ++w) acc[w] = 0.f;
            const float* wp = we2 + m;
            int n = k0;
            for (; n + 16 <= k1; n += 16) {
                float wv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = wp[(size_t)(n + u) * DH];
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int w = 0; w < EWPB; ++w) acc[w] = fmaf(e1[w * d + n + u], wv[u], acc[w]);
            }
            for (; n < k1; ++n) {
                const float wv = wp[(size_t)n * DH];
#pragma unroll
                for (int w = 0; w < EWPB; ++w) acc[w] = fmaf(e1[w * d + n], wv, acc[w]);
            }
#pragma unroll
            for (int w = 0; w < EWPB; ++w) part[(kq * EWPB + w) * DH + m] = acc[w];
        }
    }
    __syncthreads();
    if (wave < EWPB) {   // ReLU, dot with w2, one wave per window
        const int w = wave;
        float s = 0.f;
        for (int m = lane; m < DH; m += 64) {
            float h = be2[m];
#pragma unroll
            for (int kq = 0; kq < EKQ2; ++kq) h += part[(kq * EWPB + w) * DH + m];
            s += fmaxf(h, 0.f) * w2[m];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0 && w < nb) zemo[b0 + w] = s + b2[0];
    }
}

// Production shape (emotion_dim <= 256, d_model 256): every weight this thread will ever need is requested
// before the first use -- 64 + 32 independent coalesced loads in flight per thread -- because the weights are
// L2-cold on every step (the front end streams 140 MB through the L2s in between) and the kernel is pure
// memory latency: one round trip instead of six dependent ones.
__global__ __launch_bounds__(1024) void emotion_kernel_d256(const float* __restrict__ emo, int64_t B, int ED,
                                                            const float* __restrict__ wee_t, const float* __restrict__ bee,
                                                            const float* __restrict__ lg, const float* __restrict__ lb,
                                                            const float* __restrict__ we2, const float* __restrict__ be2,
                                                            const float* __restrict__ w2, const float* __restrict__ b2,
                                                            float* __restrict__ zemo) {
    constexpr int d = 256, DH = 128, EDP = 256;
    __shared__ __attribute__((aligned(16))) float emo_s[EWPB * EDP];
    __shared__ __attribute__((aligned(16))) float e1[EWPB * d];
    __shared__ __attribute__((aligned(16))) float part[EKQ1 * EWPB * d];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * EWPB;
    const int nb = (int)((B - b0) < EWPB ? (B - b0) : EWPB);
    const int n = tid & 255, kq = tid >> 8;      // layer 1: column n, k in [64 kq, 64 kq + 64)
    const int m = tid & 127, kq2 = tid >> 7;     // layer 2: hidden unit m, n in [32 kq2, 32 kq2 + 32)
    float w1[64], w2r[32];
#pragma unroll
    for (int u = 0; u < 64; ++u) {
        const int k = 64 * kq + u;
        w1[u] = k < ED ? wee_t[(size_t)k * d + n] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) w2r[u] = we2[(size_t)(32 * kq2 + u) * DH + m];
    for (int i = tid; i < EWPB * EDP; i += ENT) {
        const int w = i >> 8, k = i & 255;
        emo_s[i] = (w < nb && k < ED) ? emo[(b0 + w) * ED + k] : 0.f;
    }
    __syncthreads();
    {
        float acc[EWPB];
#pragma unroll
        for (int w = 0; w < EWPB; ++w) acc[w] = 0.f;
#pragma unroll
        for (int u = 0; u < 64; u += 4)
#pragma unroll
            for (int w = 0; w < EWPB; ++w) {
                const float4 e = *reinterpret_cast<const float4*>(emo_s + w * EDP + 64 * kq + u);
                acc[w] = fmaf(e.x, w1[u], acc[w]);
                acc[w] = fmaf(e.y, w1[u + 1], acc[w]);
                acc[w] = fmaf(e.z, w1[u + 2], acc[w]);
                acc[w] = fmaf(e.w, w1[u + 3], acc[w]);
            }
#pragma unroll
        for (int w = 0; w < EWPB; ++w) part[(kq * EWPB + w) * d + n] = acc[w];
    }
    __syncthreads();
    {
        const int w = tid >> 8;                   // EWPB * d == ENT: one element per thread
        float s = bee[n];
#pragma unroll
        for (int q = 0; q < EKQ1; ++q) s += part[(q * EWPB + w) * d + n];
        e1[tid] = s;
    }
    __syncthreads();
    if (wave < EWPB) {   // LayerNorm (eps 1e-5), one wave per window, two-pass
        const int w = wave;
        float x[4], s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = e1[w * d + lane + 64 * i]; s += x[i]; }
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
            e1[w * d + c] = (x[i] - mean) * rstd * lg[c] + lb[c];
        }
    }
    __syncthreads();
    {
        float acc[EWPB];
#pragma unroll
        for (int w = 0; w < EWPB; ++w) acc[w] = 0.f;
#pragma unroll
        for (int u = 0; u < 32; u += 4)
#pragma unroll
            for (int w = 0; w < EWPB; ++w) {
                const float4 e = *reinterpret_cast<const float4*>(e1 + w * d + 32 * kq2 + u);
                acc[w] = fmaf(e.x, w2r[u], acc[w]);
                acc[w] = fmaf(e.y, w2r[u + 1], acc[w]);
                acc[w] = fmaf(e.z, w2r[u + 2], acc[w]);
                acc[w] = fmaf(e.w, w2r[u + 3], acc[w]);
            }
#pragma unroll
        for (int w = 0; w < EWPB; ++w) part[(kq2 * EWPB + w) * DH + m] = acc[w];
    }
    __syncthreads();
    if (wave < EWPB) {   // ReLU, dot with w2, one wave per window
        const int w = wave;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mm = lane + 64 * i;
            float h = be2[mm];
#pragma unroll
            for (int q = 0; q < EKQ2; ++q) h += part[(q * EWPB + w) * DH + mm];
            s += fmaxf(h, 0.f) * w2[mm];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0 && w < nb) zemo[b0 + w] = s + b2[0];
    }
}

__global__ void smooth_kernel(float* __restrict__ x, float* __restrict__ state, int64_t n, int first, float alpha) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    if (!first) v = alpha * v + (1.0f - alpha) * state[i];
    state[i] = v;
    x[i] = v;
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static const float* dv(Context* c, const char* name) { return c->packed.at(name).dev; }

int launch_emotion(Context* c, const float* emo, int64_t B, float* zemo, void* stream) {
    const size_t p1 = (size_t)EKQ1 * EWPB * c->d, p2 = (size_t)EKQ2 * EWPB * c->DH;
    const size_t lds = ((size_t)EWPB * (c->ED + c->d) + (p1 > p2 ? p1 : p2)) * sizeof(float);
    if (lds > 160 * 1024) return fail(KM_ERR_UNSUPPORTED, "emotion_dim/d_model too large for the emotion kernel");
    static PerDeviceOnce once;
    if (once.first(c->device))
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&emotion_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const unsigned grid = (unsigned)((B + EWPB - 1) / EWPB);
    if (c->d == 256 && c->DH == 128 && c->ED <= 256) {
        hipLaunchKernelGGL(emotion_kernel_d256, dim3(grid), dim3(ENT), 0, (hipStream_t)stream, emo, B, c->ED,
                           dv(c, "wee_t"), dv(c, "bee"), dv(c, "eln_g"), dv(c, "eln_b"), dv(c, "we2"), dv(c, "be2"),
                           dv(c, "w2"), dv(c, "b2"), zemo);
        HIP_TRY(hipGetLastError());
        return KM_OK;
    }
    hipLaunchKernelGGL(emotion_kernel, dim3(grid), dim3(ENT), lds, (hipStream_t)stream, emo, B, c->ED, c->d, c->DH,
                       dv(c, "wee_t"), dv(c, "bee"), dv(c, "eln_g"), dv(c, "eln_b"), dv(c, "we2"), dv(c, "be2"),
                       dv(c, "w2"), dv(c, "b2"), zemo);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

static int core_attrs(Context* c) {
    static PerDeviceOnce once;
    if (once.first(c->device)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&core_fused_kernel<false, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, fused::LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&core_fused_kernel<true, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, fused::LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&core_fused_kernel<false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, fused::LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&core_fused_kernel<false, true, 2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, fused::lds_bytes(2)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&core_fused_kernel<false, true, 3>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, fused::lds_bytes(3)));
    }
    return KM_OK;
}

static void core_weights(Context* c, CoreArgs& a) {
    a.wce_p = dv(c, "wce_p"); a.bce = dv(c, "bce"); a.ln_g = dv(c, "ln_g"); a.ln_b = dv(c, "ln_b");
    a.qk_p = dv(c, "qk_p"); a.wv_p = dv(c, "wv_p"); a.wf_p = dv(c, "wf_p"); a.bf = dv(c, "bf");
    a.w2 = dv(c, "w2"); a.b2 = dv(c, "b2"); a.wsum = dv(c, "wsum");
    a.alpha = c->alpha;
}

int launch_core_fused(Context* c, const float* mel, int64_t B, int64_t T_in, const float* mel_short,
                      const float* zemo, float* out, float* raw, float* attn, float* state, int first,
                      void* stream) {
    if (int rc = core_attrs(c)) return rc;
    CoreArgs a{};
    core_weights(c, a);
    a.mel = mel; a.mel_short = mel_short; a.zemo = zemo; a.t_in = (int)T_in;
    a.out = out; a.raw = raw; a.attn = attn; a.state = state; a.first = first; a.win0 = 0; a.zemo_div = 1;
    if (attn)
        hipLaunchKernelGGL((core_fused_kernel<true, false>), dim3((unsigned)B), dim3(fused::NT), fused::LDS_BYTES, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((core_fused_kernel<false, false>), dim3((unsigned)B), dim3(fused::NT), fused::LDS_BYTES, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

LogParams plan_log_params(MelPlan* p);

// per-window maximum for the shared-frame sequence mode: max over the window's rows of the per-frame maxima
// (one wave per window; values are float bits of non-negative numbers, so unsigned max == float max)
__global__ __launch_bounds__(64) void seq_window_max_kernel(const unsigned* __restrict__ fmax, const unsigned* __restrict__ emax,
                                                            unsigned* __restrict__ melmax, int64_t win0, int nfc, int stride,
                                                            int n_per_clip, int n_frames) {
    const int64_t gw = win0 + blockIdx.x, clip = gw / n_per_clip, wi = gw - clip * n_per_clip;
    const unsigned* f = fmax + clip * nfc + wi * stride;
    unsigned m = 0;
    for (int r = 1 + threadIdx.x; r < n_frames - 1; r += 64) m = max(m, f[r]);
    if (threadIdx.x < 2) m = max(m, emax[gw * 2 + threadIdx.x]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if (threadIdx.x == 0) melmax[blockIdx.x] = m;
}

int launch_seq_window_max(Context* c, const unsigned* fmax, const unsigned* emax, int64_t nw, int64_t win0, int nfc, int stride,
                          int n_per_clip, int n_frames, void* stream) {
    hipLaunchKernelGGL(seq_window_max_kernel, dim3((unsigned)nw), dim3(64), 0, (hipStream_t)stream, fmax, emax, c->ws_melmax,
                       win0, nfc, stride, n_per_clip, n_frames);
    HIP_TRY(hipGetLastError());
    c->melmax_dirty = false;     // every slot the following core launch reads was just written (and is re-zeroed by it)
    return KM_OK;
}

int launch_core_fused_db(Context* c, MelPlan* p, int64_t B, int64_t n_frames, const float* zemo, float* out,
                         float* state, int first, void* stream, int64_t win0, int zemo_div, int64_t n_use,
                         const unsigned char* ready, unsigned char* started, const SeqCore* seq) {
    if (int rc = core_attrs(c)) return rc;
    CoreArgs a{};
    core_weights(c, a);
    a.zemo = zemo; a.t_in = (int)n_frames;
    a.out = out; a.state = state; a.first = first; a.win0 = win0; a.zemo_div = zemo_div > 0 ? zemo_div : 1;
    a.melpow = c->ws_melpow; a.melmax = c->ws_melmax; a.n_frames = (int)n_frames; a.lp = plan_log_params(p);
    a.n_use = (int)(n_use > 0 ? n_use : n_frames); a.ready = ready; a.started = started;
    if (seq) { a.seq_pow = seq->pow; a.seq_edge = seq->edge; a.seq_nfc = seq->nfc; a.seq_stride = seq->stride; a.seq_n = seq->n_per_clip; }
    // experimental, off by default: option core_split = 3 / 6 runs phases 1 and 2+3 as split-bf16 products
    const int terms = c->opt.core_split;
    if (terms == 3) {
        a.qkv_s = dv(c, "qkv_s2"); a.wce_s = dv(c, "wce_s2");
        hipLaunchKernelGGL((core_fused_kernel<false, true, 2>), dim3((unsigned)B), dim3(fused::NT), fused::lds_bytes(2), (hipStream_t)stream, a);
    } else if (terms == 6) {
        a.qkv_s = dv(c, "qkv_s3"); a.wce_s = dv(c, "wce_s3");
        hipLaunchKernelGGL((core_fused_kernel<false, true, 3>), dim3((unsigned)B), dim3(fused::NT), fused::lds_bytes(3), (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL((core_fused_kernel<false, true>), dim3((unsigned)B), dim3(fused::NT), fused::LDS_BYTES, (hipStream_t)stream, a);
    }
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

// EMA along the frame axis of (B, N, 52), state reset at the start of every clip
// (sequential_dual_stream_model.py:99,136 + simplified_dual_stream_model.py:357-366).  First-order linear
// recurrence: one thread per (clip, coefficient) walks the N frames.
__global__ void ema_scan_kernel(float* __restrict__ x, int64_t B, int64_t N, int nb, float alpha) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nb) return;
    const int64_t b = i / nb;
    const int cidx = (int)(i - b * nb);
    float* p = x + b * N * nb + cidx;
    float prev = p[0];                                   // first call returns its input unchanged
    for (int64_t t = 1; t < N; ++t) {
        const float v = alpha * p[t * nb] + (1.0f - alpha) * prev;
        p[t * nb] = v;
        prev = v;
    }
}

int launch_ema_scan(Context* c, float* x, int64_t B, int64_t N, void* stream) {
    const int64_t n = B * c->NB;
    hipLaunchKernelGGL(ema_scan_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (hipStream_t)stream, x, B, N,
                       c->NB, c->alpha);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

int launch_smooth(Context* c, float* x, float* state, int64_t B, int first, void* stream) {
    const int64_t n = B * c->NB;
    hipLaunchKernelGGL(smooth_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, state, n,
                       first, c->alpha);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

}  // namespace km
