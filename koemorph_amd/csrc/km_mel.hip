// Log-mel front end of the KoeMorph hot path for gfx950 (MI355X).
//
// Replaces the reference's three CPU front ends:
//   librosa.feature.melspectrogram + power_to_db   src/model/simplified_dual_stream_model.py:188-200
//                                                  src/features/mel_sliding_window.py:280-307
//   torchaudio.transforms.MelSpectrogram + log     src/features/stft.py:84-140
//
// Kernel 1  mel_power_kernel<NFFT>: grid (frame chunks, windows).  A 256-thread workgroup owns 16
//   consecutive frames of one window.  Each wave transforms TWO real frames at once as one complex
//   NFFT-point FFT (z = frame_a + i frame_b), fully wave-private:
//     pass 1  radix-R0 (R0 = NFFT/64: 16 or 8) butterflies in registers on samples n = lane + 64 i
//             (global loads are unit-stride across lanes whatever the hop / alignment),
//     pass 2  radix-8, pass 3 radix-8, with two transposes through a wave-private LDS buffer
//             (row strides 72 / 9 complex keep the ds_read_b64 / ds_write_b64 half-waves on distinct banks),
//   the two spectra are separated with the conjugate-symmetry identities, |.|^2 goes to an LDS
//   [frame][bin] image with an odd row stride, and the sparse triangular mel filters (<= 2 filters per
//   bin, stored CSR per filter) are applied with lanes = frames so every LDS read is conflict free.
//   Output: power-mel (B, F, n_mels) + per-window max via atomicMax on the float bits (values >= 0).
// Kernel 2  mel_log_kernel: dB against the per-window max, top_db clip and affine (librosa
//   power_to_db(ref=np.max) is a whole-window reduction, so it cannot be fused into kernel 1), or
//   log(x + eps); applies the truncate / repeat-last-frame output policy; emits the last 3 frames.
//
// Everything is fp32.  librosa runs the rFFT in float64 and rounds to complex64; the fp32 FFT here
// differs by ~1e-7 of the frame's peak amplitude, visible only in bins > 60 dB below the peak.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "km_context.h"
#include "km_device.h"

namespace km {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(KM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace mel {
constexpr int FPB = 16;          // frames per workgroup
constexpr int WAVES = 4;
constexpr int FFT_BUF = 1152;    // complex elements per wave-private LDS buffer (16*72 = 128*9)
}  // namespace mel

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// multiply by W_R^idx = exp(-2 pi i idx / R) for the constant indices of an unrolled radix-R DIF
template <int R>
__device__ __forceinline__ float2 twiddle_const(float2 d, int idx) {
    // idx * 16 / R in sixteenths of a turn; after full unrolling every branch folds away
    const int s = idx * (16 / R);
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, C2 = 0.70710678118654752f;
    switch (s) {
        case 0: return d;
        case 1: return make_float2(d.x * C1 + d.y * S1, d.y * C1 - d.x * S1);
        case 2: return make_float2((d.x + d.y) * C2, (d.y - d.x) * C2);
        case 3: return make_float2(d.x * S1 + d.y * C1, d.y * S1 - d.x * C1);
        case 4: return make_float2(d.y, -d.x);
        case 5: return make_float2(d.y * C1 - d.x * S1, -d.x * C1 - d.y * S1);
        case 6: return make_float2((d.y - d.x) * C2, -(d.x + d.y) * C2);
        default: return make_float2(-d.x * C1 + d.y * S1, -d.y * C1 - d.x * S1);   // 7
    }
}

__host__ __device__ constexpr int bitrev(int i, int bits) {
    int r = 0;
    for (int b = 0; b < bits; ++b) r |= ((i >> b) & 1) << (bits - 1 - b);
    return r;
}

// In-register radix-2 DIF FFT of R points (R = 8 or 16).  Output X[bitrev(i)] is left in v[i].
template <int R>
__device__ __forceinline__ void dif_fft(float2 (&v)[R]) {
#pragma unroll
    for (int h = R / 2; h >= 1; h >>= 1) {
#pragma unroll
        for (int base = 0; base < R; base += 2 * h) {
#pragma unroll
            for (int i = 0; i < h; ++i) {
                const float2 a = v[base + i], b = v[base + i + h];
                v[base + i] = cadd(a, b);
                v[base + i + h] = twiddle_const<R>(csub(a, b), i * (R / (2 * h)));
            }
        }
    }
}

struct MelArgs {
    const float* audio;   // (B, L)  -- or (clips, clip_len) in sequence mode
    int64_t L;            // samples per window (zero padded past the end of the clip)
    // sequence mode (SequentialDualStreamModel, sequential_dual_stream_model.py:101-117): window w of the launch
    // is global window g = win0 + w, clip g / wins_per_clip, offset (g % wins_per_clip) * win_step in that clip;
    // a plain batch is wins_per_clip = 1, clip_len = L, win_step = 0
    int64_t clip_len, win_step;
    int64_t win0;
    int wins_per_clip;
    // ring mode (km_stream_*): window b is the ring of stream b read in chronological order starting at
    // ring_start[b] (mod L); streams whose ring is not full yet (ready[b] == 0) are skipped
    const int* ring_start;
    const unsigned char* ready;
    int n_frames;         // frames computed per window = 1 + L / hop
    int hop;
    int pad_mode;
    int n_mels;
    const float* window;  // NFFT
    const float2* twiddle;  // NFFT: W_N^q
    const int* fb_start;
    const int* fb_count;
    const int* fb_offset;
    const float* fb_weight;
    int fb_nnz;           // number of stored filter weights
    const int* fbg_gid;   // grouped image of the filters (mel_power_rp_kernel; MelPlan::fbg_*)
    const int* fbg_desc;
    const float* fbg_weight;
    int fbg_nw;           // floats in fbg_weight
    float* melpow;        // (B, n_frames, n_mels)
    unsigned* melmax;     // (B) float bits, zero-initialised
    EmoArgs emo;          // optional: the window's emotion logit, computed by the last workgroup of each window
    // shared-frame sequence mode (km_sequence_forward): output row r of a window is STFT frame r * frame_mul, and the
    // maxima go per row into frame_max[(window, row)] instead of per window into melmax
    int frame_mul;
    unsigned* frame_max;  // (B, n_frames) float bits, zero-initialised, or null
    // mel_power_rp_kernel hands a window's 16-frame chunks (beyond the first two of each workgroup) to its workgroups in the
    // order they ask for them: chunk_ctr[window] counts the requests, is zero on entry and is put back to zero by the last
    // request of the launch
    unsigned* chunk_ctr;
    // training from audio (km_train_step_audio; PACK instantiation of mel_power_rp_kernel): instead of melpow the kernel writes
    // 10 log10(max(amin, power)) straight into the channel encoder's packed input xt (B, n_mels, KP) -- row m of a window: T long
    // frames, then the last three computed frames (dual_stream_attention.py:189-211) -- and the rest of librosa.power_to_db, which
    // needs the WINDOW's maximum (simplified_dual_stream_model.py:199-200), is applied by the readers: the channel encoder's tile
    // on its operand fragments, OP_DBCONV beside it for the backward pass (km_trainp.hip).  No conversion launch in between.
    float* pack_xt;
    int pack_T, pack_KP;
    float pack_amin;
};

// raw samples of the frame pair (fa, fa+1) into z[i] = (x_a[lane + 64 i], x_b[lane + 64 i]); wave-uniform fa
template <int NFFT, bool RING>
__device__ __forceinline__ void load_pair(const MelArgs& a, const float* __restrict__ x, int64_t Lv, int rs, int fa,
                                          int lane, float2 (&z)[NFFT / 64]) {
    constexpr int R0 = NFFT / 64;
    if (fa >= a.n_frames) {
#pragma unroll
        for (int i = 0; i < R0; ++i) z[i] = make_float2(0.f, 0.f);
        return;
    }
    const int64_t p0 = (int64_t)fa * a.hop - NFFT / 2;
    const bool have_b = fa + 1 < a.n_frames;
    if constexpr (RING) {                                     // ring: logical sample q lives at (rs + q) mod L
        const int Li = (int)Lv;
#pragma unroll
        for (int i = 0; i < R0; ++i) {
            int qa = (int)p0 + lane + 64 * i, qb = qa + a.hop;
            float va = 0.f, vb = 0.f;
            if (a.pad_mode == KM_PAD_REFLECT) {
                qa = qa < 0 ? -qa : (qa >= Li ? 2 * (Li - 1) - qa : qa);
                qb = qb < 0 ? -qb : (qb >= Li ? 2 * (Li - 1) - qb : qb);
                int ia = rs + qa; ia -= ia >= Li ? Li : 0;
                int ib = rs + qb; ib -= ib >= Li ? Li : 0;
                va = x[ia];
                vb = have_b ? x[ib] : 0.f;
            } else {
                int ia = rs + qa; ia -= ia >= Li ? Li : 0;
                int ib = rs + qb; ib -= ib >= Li ? Li : 0;
                if (qa >= 0 && qa < Li) va = x[ia];
                if (have_b && qb >= 0 && qb < Li) vb = x[ib];
            }
            z[i] = make_float2(va, vb);
        }
        return;
    }
    if (p0 >= 0 && p0 + a.hop + NFFT <= Lv && have_b) {      // interior pair: no padding, 32-bit offsets
        const float* xa = x + p0 + lane;
        const float* xb = xa + a.hop;
#pragma unroll
        for (int i = 0; i < R0; ++i) z[i] = make_float2(xa[64 * i], xb[64 * i]);
        return;
    }
#pragma unroll
    for (int i = 0; i < R0; ++i) {
        int64_t qa = p0 + lane + 64 * i, qb = qa + a.hop;
        float va, vb;
        if (a.pad_mode == KM_PAD_REFLECT) {                    // np.pad(mode='reflect')
            qa = qa < 0 ? -qa : (qa >= Lv ? 2 * (Lv - 1) - qa : qa);
            qb = qb < 0 ? -qb : (qb >= Lv ? 2 * (Lv - 1) - qb : qb);
            va = x[qa];
            vb = have_b ? x[qb] : 0.f;
        } else {                                               // zero padding
            va = (qa >= 0 && qa < Lv) ? x[qa] : 0.f;
            vb = (have_b && qb >= 0 && qb < Lv) ? x[qb] : 0.f;
        }
        z[i] = make_float2(va, vb);
    }
}

// RING = true is the streaming instantiation (device rings read with modular addressing).  It is a separate
// instantiation on purpose: as a run-time branch its address arithmetic pushed the batch kernel from 222 to 310
// registers (1 wave per SIMD, 1.7x slower).
template <int NFFT, bool RING>
__global__ __launch_bounds__(256) void mel_power_kernel(MelArgs a) {
    using namespace mel;
    constexpr int R0 = NFFT / 64;          // 16 or 8
    constexpr int G = R0 / 8;              // radix-8 groups per lane in passes 2 and 3
    constexpr int LOG_R0 = R0 == 16 ? 4 : 3;
    constexpr int NFS = NFFT / 2 + 2;      // power row stride: 2 mod 32 dwords (see the mel stage)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* fbuf = reinterpret_cast<float2*>(smem);                     // [WAVES][FFT_BUF]
    float* pw = smem + WAVES * FFT_BUF * 2;                              // [FPB][NFS]
    float* outs = pw + FPB * NFS;                                        // [FPB][n_mels]
    float* redmax = outs + FPB * a.n_mels;                               // [WAVES]
    int* fbs = reinterpret_cast<int*>(redmax + WAVES);                   // [n_mels] first bin
    int* fbc = fbs + a.n_mels;                                           // [n_mels] tap count
    int* fbo = fbc + a.n_mels;                                           // [n_mels] offset into fbw
    float* fbw = reinterpret_cast<float*>(fbo + a.n_mels);               // [fb_nnz] triangle weights

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int64_t gw = a.win0 + b;
    const int64_t clip = gw / a.wins_per_clip, woff = (gw - clip * a.wins_per_clip) * a.win_step;
    const float* x = a.audio + clip * a.clip_len + woff;
    const int64_t Lv = (a.clip_len - woff) < a.L ? (a.clip_len - woff) : a.L;   // samples that exist; the rest reads as zero
    if (a.ready && !a.ready[b]) return;                        // workgroup-uniform: stream still filling
    const int rs = RING ? a.ring_start[b] : -1;
    float2* buf = fbuf + wave * FFT_BUF;
    const int n_chunks = (a.n_frames + FPB - 1) / FPB;

    // the sparse filterbank lives in LDS for the lifetime of the (persistent) workgroup; first use is
    // behind the chunk loop's first __syncthreads()
    for (int i = tid; i < a.n_mels; i += 256) { fbs[i] = a.fb_start[i]; fbc[i] = a.fb_count[i]; fbo[i] = a.fb_offset[i]; }
    for (int i = tid; i < a.fb_nnz; i += 256) fbw[i] = a.fb_weight[i];

    // lane-constant operands, hoisted out of the frame loop
    float win[R0];
    float2 tw1[R0];                        // pass-1 twiddles W_N^(lane * k0), k0 = bitrev(i)
#pragma unroll
    for (int i = 0; i < R0; ++i) {
        win[i] = a.window[lane + 64 * i];
        tw1[i] = a.twiddle[(lane * bitrev(i, LOG_R0)) & (NFFT - 1)];
    }
    const int n2 = lane & 7, k0l = lane >> 3;
    float2 tw2[8];                         // pass-2 twiddles W_64^(n2 * k1) = W_N^(R0 * n2 * k1), k1 = bitrev(i)
#pragma unroll
    for (int i = 0; i < 8; ++i) tw2[i] = a.twiddle[(R0 * n2 * bitrev(i, 3)) & (NFFT - 1)];

    float vmax = 0.f;
    float2 zn[R0];                          // software prefetch: raw samples of the NEXT pair of this wave
    int chunk = blockIdx.x;
    load_pair<NFFT, RING>(a, x, Lv, rs, chunk * FPB + 2 * wave, lane, zn);
    for (; chunk < n_chunks; chunk += gridDim.x) {
        const int f0 = chunk * FPB;
#pragma unroll 1
        for (int pi = 0; pi < FPB / 2 / WAVES; ++pi) {
            const int p = wave + WAVES * pi;
            const int fa = f0 + 2 * p;
            float2 z[R0];
#pragma unroll
            for (int i = 0; i < R0; ++i) z[i] = make_float2(zn[i].x * win[i], zn[i].y * win[i]);
            {   // issue the next pair's loads now; they complete under this pair's FFT
                const int fn = (pi + 1 < FPB / 2 / WAVES) ? fa + 2 * WAVES
                                                          : (chunk + (int)gridDim.x) * FPB + 2 * wave;
                load_pair<NFFT, RING>(a, x, Lv, rs, (pi + 1 < FPB / 2 / WAVES || chunk + (int)gridDim.x < n_chunks) ? fn : a.n_frames,
                                lane, zn);
            }
            if (fa < a.n_frames) {   // wave-uniform
                // ---- pass 1: radix-R0 over n0 (n = 64 n0 + m, m = lane) ---------------------------------
                dif_fft<R0>(z);
#pragma unroll
                for (int i = 0; i < R0; ++i) {
                    const int k0 = bitrev(i, LOG_R0);
                    buf[k0 * 72 + lane] = k0 == 0 ? z[i] : cmul(z[i], tw1[i]);
                }
                __builtin_amdgcn_wave_barrier();
                // ---- pass 2: radix-8 over n1 (m = 8 n1 + n2); this lane: n2 = lane&7, k0 = (lane>>3) + 8u ----
                float2 y[G][8];
#pragma unroll
                for (int u = 0; u < G; ++u) {
#pragma unroll
                    for (int n1 = 0; n1 < 8; ++n1) y[u][n1] = buf[(k0l + 8 * u) * 72 + 8 * n1 + n2];
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    dif_fft<8>(y[u]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int k1 = bitrev(i, 3);
                        // transposed image indexed by c = k0 + R0 k1: row of 8, column XOR-swizzled by (c>>2)&7 so
                        // that both this write (16-lane groups) and the pass-3 read (32-lane halves) are conflict free
                        const int c = k0l + 8 * u + R0 * k1;
                        buf[c * 8 + (n2 ^ ((c >> 2) & 7))] = k1 == 0 ? y[u][i] : cmul(y[u][i], tw2[i]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // ---- pass 3: radix-8 over n2; this lane: c = lane + 64 v; output k = c + 8 R0 k2 ---------
#pragma unroll
                for (int v = 0; v < G; ++v) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int c = lane + 64 * v;
                        y[v][q] = buf[c * 8 + (q ^ ((c >> 2) & 7))];
                    }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int v = 0; v < G; ++v) {
                    dif_fft<8>(y[v]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) buf[lane + 64 * v + 8 * R0 * bitrev(i, 3)] = y[v][i];
                }
                __builtin_amdgcn_wave_barrier();
                // ---- separate the two real spectra, |.|^2 ----------------------------------------------
                // A[k] = (Z[k] + conj Z[N-k]) / 2,  B[k] = (Z[k] - conj Z[N-k]) / 2i
                float* pwa = pw + (2 * p) * NFS;
                float* pwb = pwa + NFS;
#pragma unroll
                for (int i = 0; i <= NFFT / 128; ++i) {
                    const int k = lane + 64 * i;
                    if (k <= NFFT / 2) {
                        const float2 zk = buf[k], zc = buf[(NFFT - k) & (NFFT - 1)];
                        const float ar = zk.x + zc.x, ai = zk.y - zc.y, br = zk.y + zc.y, bi = zk.x - zc.x;
                        pwa[k] = 0.25f * (ar * ar + ai * ai);
                        pwb[k] = 0.25f * (br * br + bi * bi);
                    }
                }
            }
        }
        __syncthreads();

        // ---- sparse triangular mel filters: 16 lanes = 16 frames of one filter, the two 16-lane groups of
        // a half-wave take the even / odd taps of the SAME filter; with a row stride of 2 (mod 32) dwords the
        // 32 lanes of every ds_read_b32 hit 32 distinct banks.
        {
            const int fl = tid & 15, par = (tid >> 4) & 1;
            const bool fvalid = f0 + fl < a.n_frames;
            for (int m = tid >> 5; m < a.n_mels; m += 8) {
                const int st = fbs[m], cnt = fbc[m];
                const float* wt = fbw + fbo[m];
                const float* pr = pw + fl * NFS + st;
                float acc = 0.f;
                if (fvalid) {
#pragma unroll 4
                    for (int i = par; i < cnt; i += 2) acc = fmaf(pr[i], wt[i], acc);
                }
                acc += __shfl_xor(acc, 16);
                if (par == 0) outs[fl * a.n_mels + m] = acc;
                vmax = fmaxf(vmax, acc);
            }
        }
        __syncthreads();
        const int nf_here = (a.n_frames - f0) < FPB ? (a.n_frames - f0) : FPB;
        float* dst = a.melpow + ((int64_t)b * a.n_frames + f0) * a.n_mels;
        for (int i = tid; i < nf_here * a.n_mels; i += 256) dst[i] = outs[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if (lane == 0) redmax[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
        const float mx = fmaxf(fmaxf(redmax[0], redmax[1]), fmaxf(redmax[2], redmax[3]));
        atomicMax(a.melmax + b, __float_as_uint(mx));
    }
}

// ---------------------------------------------------------------------------------------------------------
// mel_power_rp_kernel: the 1024-point front end with ONE real frame per wave, transformed as a 512-point complex
// FFT of the even/odd-packed samples (z[m] = x[2m] + i x[2m+1]) followed by the real-input post-processing
//     X[k] = E + W_N^k O,  X[N/2-k] = conj(E - W_N^k O),  E = (Z[k] + conj Z[M-k]) / 2,  O = (Z[k] - conj Z[M-k]) / 2i.
// Same radix 8 x 8 x 8 passes and LDS exchange images as mel_power_kernel<512,...>, but half the live data per wave:
// ~120 registers instead of ~240, so 512-thread workgroups run at 4 waves per SIMD (2 workgroups of 80 KB LDS per
// CU).  The two-frames-per-wave kernel above is latency bound (T ~ 41 us + 136 us / waves-per-SIMD at the C2 shape).
// ---------------------------------------------------------------------------------------------------------
// KM_MEL_SKIP (A/B builds only, tools/micro/mel_variants.sh; 0 in the library): bit 0 no mel filter stage, bit 1 no LDS
// exchanges between the FFT passes (data stays in registers: wrong spectra, same arithmetic), bit 2 no FFT arithmetic,
// bit 3 no audio loads, bit 4 no emotion rider
#ifndef KM_MEL_SKIP
#define KM_MEL_SKIP 0
#endif
// KM_MEL_STAMP (timing builds only, tools/micro/mel_stamp.py): every wave sums the shader-clock cycles it spends in each part
// of the frame loop (s_memtime at the part boundaries; the stamps drain the wave's LDS queue, so the parts are measured with
// less overlap than they run with) into km_mel_stamps[(workgroup, wave)][part], read back through km_debug_mel_stamps.
#ifdef KM_MEL_STAMP
#define KM_ASM asm volatile      /* keeps the arithmetic blocks between their stamps */
__device__ unsigned long long km_mel_stamps[1024 * 8 * 24];
#define KM_STAMP(i)                                                  \
    do {                                                             \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        st_acc[i] += (unsigned)(now_ - st_last);                     \
        st_last = now_;                                              \
    } while (0)
#else
#define KM_ASM asm
#define KM_STAMP(i) do {} while (0)
#endif
namespace melrp {
constexpr int WAVES = kMelRpWaves, NT = 64 * WAVES, FPB = 16, NC = 512, FFT_BUF = 576, NGW = kMelRpGroups;
constexpr int NFS = kMelRpRow;   // power-row stride in dwords: 4 (mod 64), see the mel stage
static_assert(WAVES == 8 && NFS % 64 == 4 && NFS >= 528, "mel stage layout");

// Complex arithmetic on register pairs, one packed instruction (v_pk_add/mul/fma_f32) per complex operation.
// Multiplications by -i, conjugations and the cross terms of a complex product are half-selects (op_sel / op_sel_hi)
// and per-half sign flips (neg_lo / neg_hi) of the packed operands.  The compiler only folds splats and whole-vector
// negations into those modifiers -- a swap becomes v_pk_mov_b32 and a one-sided negation v_xor_b32, ~100 extra
// instructions per frame -- so the three arithmetic blocks of a frame are written as inline assembly.  gfx950 needs
// one wait state between a packed-fp32 result and a VALU instruction that reads it (the compiler pads its own code
// with s_nop 0 for this): inside a block no instruction reads the result of its predecessor, and each block begins
// and ends with s_nop 0 because the hazard recogniser does not look through inline assembly.
// Measured on gfx950: a packed fp32 instruction costs ~1.2x a scalar one and produces two results
// (tools/micro/valu_rate.hip).
typedef float v2f __attribute__((ext_vector_type(2)));

#define KM_ADD(D, A, B) "v_pk_add_f32 %[" #D "], %[" #A "], %[" #B "]\n\t"
#define KM_SUB(D, A, B) "v_pk_add_f32 %[" #D "], %[" #A "], %[" #B "] neg_lo:[0,1] neg_hi:[0,1]\n\t"
// D = A + (-i) B = (A.x + B.y, A.y - B.x);  D = A - (-i) B
#define KM_ADD_NI(D, A, B) "v_pk_add_f32 %[" #D "], %[" #A "], %[" #B "] op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
#define KM_SUB_NI(D, A, B) "v_pk_add_f32 %[" #D "], %[" #A "], %[" #B "] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
// D = (-i) S - S = (S.y - S.x, -S.x - S.y)
#define KM_ROT3(D, S) "v_pk_add_f32 %[" #D "], %[" #S "], %[" #S "] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[1,1]\n\t"
// complex product in two steps: T = A.xx * W;  D = (-A.y, A.y) * W.yx + T
#define KM_CMUL1(T, A, W) "v_pk_mul_f32 %[" #T "], %[" #A "], %[" #W "] op_sel_hi:[0,1]\n\t"
#define KM_CMUL2(D, A, W, T) \
    "v_pk_fma_f32 %[" #D "], %[" #A "], %[" #W "], %[" #T "] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"

// radix-8 DIF of v[0..7] (26 packed instructions); output X[bitrev(i)] in v[i]
__device__ __forceinline__ void dif8(v2f (&v)[8]) {
    constexpr float C2 = 0.70710678118654752f;
    const v2f c = {C2, C2};
    v2f r0 = v[0], r1 = v[1], r2 = v[2], r3 = v[3], r4 = v[4], r5 = v[5], r6 = v[6], r7 = v[7], t;
    KM_ASM("s_nop 0\n\t"
        // stage 1: s_i = v_i + v_{i+4}, d_i = v_i - v_{i+4};  u1 = d1 + (-i) d1, u3 = (-i) d3 - d3 (their 1/sqrt2 is applied in stage 3)
        KM_ADD(t, r0, r4) KM_SUB(r4, r0, r4)          // s0 = t,  d0 = r4
        KM_ADD(r0, r1, r5) KM_SUB(r5, r1, r5)         // s1 = r0, d1 = r5
        KM_ADD(r1, r2, r6) KM_SUB(r6, r2, r6)         // s2 = r1, d2 = r6
        KM_ADD(r2, r3, r7) KM_SUB(r7, r3, r7)         // s3 = r2, d3 = r7
        KM_ADD_NI(r3, r5, r5)                         // u1 = r3
        KM_ROT3(r5, r7)                               // u3 = r5
        // stage 2
        KM_ADD(r7, t, r1) KM_SUB(r1, t, r1)           // a0 = r7, b0 = r1
        KM_ADD(t, r0, r2) KM_SUB(r2, r0, r2)          // a1 = t,  g  = r2 (s1 - s3)
        KM_ADD_NI(r0, r4, r6) KM_SUB_NI(r6, r4, r6)   // c0 = r0 = d0 + (-i) d2, e0 = r6
        KM_ADD(r4, r3, r5) KM_SUB(r5, r3, r5)         // w = r4 = u1 + u3, q = r5 = u1 - u3
        // stage 3
        KM_ADD(r3, r7, t) KM_SUB(r7, r7, t)           // X0 = r3, X4 = r7
        KM_ADD_NI(t, r1, r2) KM_SUB_NI(r2, r1, r2)    // X2 = t,  X6 = r2
        "v_pk_fma_f32 %[r1], %[r4], %[c], %[r0]\n\t"                                   // X1 = c0 + C w
        "v_pk_fma_f32 %[r4], %[r4], %[c], %[r0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"     // X5 = c0 - C w
        "v_pk_fma_f32 %[r0], %[r5], %[c], %[r6] op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]\n\t"   // X3 = e0 + C (-i) q
        "v_pk_fma_f32 %[r5], %[r5], %[c], %[r6] op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"   // X7 = e0 - C (-i) q
        "s_nop 0"
        : [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2), [r3] "+v"(r3), [r4] "+v"(r4), [r5] "+v"(r5), [r6] "+v"(r6),
          [r7] "+v"(r7), [t] "=&v"(t)
        : [c] "s"(c));
    v[0] = r3; v[1] = r7; v[2] = t; v[3] = r2; v[4] = r1; v[5] = r4; v[6] = r0; v[7] = r5;
}

// v[i] *= w[i] for i = 1..7 (14 packed instructions)
__device__ __forceinline__ void cmul7(v2f (&v)[8], const v2f (&w)[8]) {
    v2f ta, tb;
    KM_ASM("s_nop 0\n\t"
        KM_CMUL1(ta, v1, w1) KM_CMUL1(tb, v2, w2)
        KM_CMUL2(v1, v1, w1, ta) KM_CMUL1(ta, v3, w3)
        KM_CMUL2(v2, v2, w2, tb) KM_CMUL1(tb, v4, w4)
        KM_CMUL2(v3, v3, w3, ta) KM_CMUL1(ta, v5, w5)
        KM_CMUL2(v4, v4, w4, tb) KM_CMUL1(tb, v6, w6)
        KM_CMUL2(v5, v5, w5, ta) KM_CMUL1(ta, v7, w7)
        KM_CMUL2(v6, v6, w6, tb)
        KM_CMUL2(v7, v7, w7, ta)
        "s_nop 0"
        : [v1] "+v"(v[1]), [v2] "+v"(v[2]), [v3] "+v"(v[3]), [v4] "+v"(v[4]), [v5] "+v"(v[5]), [v6] "+v"(v[6]),
          [v7] "+v"(v[7]), [ta] "=&v"(ta), [tb] "=&v"(tb)
        : [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]), [w6] "v"(w[6]), [w7] "v"(w[7]));
}

// Real-input post-processing of four bin pairs: zk_i = Z[k_i], zc_i = Z[512 - k_i], w_i = W_1024^(k_i).
//   2E = zk + conj zc,  2O = -i (zk - conj zc),  2X[k] = 2E + w 2O,  2X[512-k] = conj(2E - w 2O)
// Result in zk_i: (4 |X[k_i]|^2, 4 |X[512 - k_i]|^2).  32 packed instructions.
#define KM_POST4(OP) OP(0) OP(1) OP(2) OP(3)
#define KM_P1(i) "v_pk_add_f32 %[t" #i "], %[k" #i "], %[c" #i "] neg_hi:[0,1]\n\t"                                   /* 2E */
#define KM_P2(i) "v_pk_add_f32 %[c" #i "], %[k" #i "], %[c" #i "] op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"     /* 2O */
#define KM_P3(i) "v_pk_mul_f32 %[k" #i "], %[c" #i "], %[w" #i "] op_sel_hi:[0,1]\n\t"
#define KM_P4(i) "v_pk_fma_f32 %[c" #i "], %[c" #i "], %[w" #i "], %[k" #i "] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"  /* w 2O */
#define KM_P5(i) "v_pk_add_f32 %[k" #i "], %[t" #i "], %[c" #i "] op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n\t"     /* Re 2X[k], Re 2X[512-k] */
#define KM_P6(i) "v_pk_add_f32 %[c" #i "], %[t" #i "], %[c" #i "] op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]\n\t"     /* Im */
#define KM_P7(i) "v_pk_mul_f32 %[t" #i "], %[k" #i "], %[k" #i "]\n\t"
#define KM_P8(i) "v_pk_fma_f32 %[k" #i "], %[c" #i "], %[c" #i "], %[t" #i "]\n\t"
__device__ __forceinline__ void post4(v2f (&zk)[4], v2f (&zc)[4], const v2f (&w)[4]) {
    v2f t0, t1, t2, t3;
    KM_ASM("s_nop 0\n\t"
        KM_POST4(KM_P1) KM_POST4(KM_P2) KM_POST4(KM_P3) KM_POST4(KM_P4)
        KM_POST4(KM_P5) KM_POST4(KM_P6) KM_POST4(KM_P7) KM_POST4(KM_P8)
        "s_nop 0"
        : [k0] "+v"(zk[0]), [k1] "+v"(zk[1]), [k2] "+v"(zk[2]), [k3] "+v"(zk[3]),
          [c0] "+v"(zc[0]), [c1] "+v"(zc[1]), [c2] "+v"(zc[2]), [c3] "+v"(zc[3]),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
        : [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]));
}
// Eight 8-byte LDS reads at a constant byte stride from one address register, drained before returning.  Written as
// assembly because the compiler pairs neighbouring reads into ds_read2_b64, which the LDS serves at half the rate of two
// ds_read_b64 (8 vs 2 x 2 array cycles per wave, /opt/skills/guides/MI355X_MICROARCH.md LDS table) and banks mod 32.
__device__ __forceinline__ unsigned lds_offset(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
template <int S>
__device__ __forceinline__ void lds_read8(unsigned addr, v2f (&y)[8]) {
    asm volatile("ds_read_b64 %0, %8 offset:%9\n\t"
                 "ds_read_b64 %1, %8 offset:%10\n\t"
                 "ds_read_b64 %2, %8 offset:%11\n\t"
                 "ds_read_b64 %3, %8 offset:%12\n\t"
                 "ds_read_b64 %4, %8 offset:%13\n\t"
                 "ds_read_b64 %5, %8 offset:%14\n\t"
                 "ds_read_b64 %6, %8 offset:%15\n\t"
                 "ds_read_b64 %7, %8 offset:%16\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]), "=&v"(y[4]), "=&v"(y[5]), "=&v"(y[6]), "=&v"(y[7])
                 : "v"(addr), "i"(0 * S), "i"(1 * S), "i"(2 * S), "i"(3 * S), "i"(4 * S), "i"(5 * S), "i"(6 * S), "i"(7 * S)
                 : "memory");
}
// The matching stores: register i (digit bitrev(i)) goes to base + S * bitrev(i) bytes.  Assembly for the same reason as the
// reads: the compiler pairs the stores into ds_write2_b64.
template <int S>
__device__ __forceinline__ void lds_write8_bitrev(unsigned addr, const v2f (&y)[8]) {
    asm volatile("ds_write_b64 %8, %0 offset:%9\n\t"
                 "ds_write_b64 %8, %1 offset:%10\n\t"
                 "ds_write_b64 %8, %2 offset:%11\n\t"
                 "ds_write_b64 %8, %3 offset:%12\n\t"
                 "ds_write_b64 %8, %4 offset:%13\n\t"
                 "ds_write_b64 %8, %5 offset:%14\n\t"
                 "ds_write_b64 %8, %6 offset:%15\n\t"
                 "ds_write_b64 %8, %7 offset:%16"
                 :
                 : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(addr),
                   "i"(0 * S), "i"(4 * S), "i"(2 * S), "i"(6 * S), "i"(1 * S), "i"(5 * S), "i"(3 * S), "i"(7 * S)
                 : "memory");
}
__device__ __forceinline__ v2f ld2(const float2* p) { const float2 t = *p; return v2f{t.x, t.y}; }

// ---- the exchanges between the FFT passes as REGISTER exchanges (KM_MEL_XCHG bit 0: the one after pass 1, bit 1: the one
// after pass 2; a cleared bit keeps that exchange in the wave's LDS buffer) -------------------------------------------------
// An exchange is a transposition of the 8 x 8 grid (register index, three lane bits), done as three stages; stage (j, b)
// swaps bit j of the register index with lane bit b: for every register pair (r, r | 1 << j) the element of r in the
// lanes whose bit b is set trades places with the element of r | 1 << j in lane ^ (1 << b).
//   b = 5: v_permlane32_swap_b32 (one instruction per register pair and dword), b = 4: v_permlane16_swap_b32,
//   b <= 3: two v_cndmask_b32_dpp per pair and dword -- row_ror:8, row_shr:4 / row_shl:4, quad_perm -- each lane keeps one
//   element of the pair and takes the other from its partner.
// 8 + 8 + 16 vector instructions for lane bits 5:3, 48 for lane bits 2:0, no LDS instruction and no dependent LDS round trip
// (8 ds_write_b64 + 8 ds_read_b64 per exchange).  The arithmetic is untouched, so the spectra are bit-identical; what changes
// is WHICH lane ends up with which bins (fft_lane_bin below).
#ifndef KM_MEL_XCHG
#define KM_MEL_XCHG 1
#endif
#ifndef KM_MEL_REMAP
#define KM_MEL_REMAP 1
#endif
#ifndef KM_MEL_FLAT
#define KM_MEL_FLAT 0
#endif
#ifndef KM_MEL_PRIO
#define KM_MEL_PRIO 3     /* wave priority during the mel stage (0: leave it alone) */
#endif
#ifndef KM_MEL_W2
#define KM_MEL_W2 0     /* 1: let the compiler pair the exchange stores into ds_write2_b64 */
#endif
__device__ __forceinline__ void swap_lane32(v2f& a, v2f& b) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = v2f{__uint_as_float(r0[0]), __uint_as_float(r1[0])}; b = v2f{__uint_as_float(r0[1]), __uint_as_float(r1[1])};
}
__device__ __forceinline__ void swap_lane16(v2f& a, v2f& b) {
    const auto r0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = v2f{__uint_as_float(r0[0]), __uint_as_float(r1[0])}; b = v2f{__uint_as_float(r0[1]), __uint_as_float(r1[1])};
}
// stage (., B) on two register pairs (a0, b0), (a1, b1) = four dword pairs (a: register bit clear, b: set).  The lanes whose
// bit B is clear keep a and take the partner's a as their b; the others keep b and take the partner's b as their a.
//   B = 3, 2: the lanes that receive are whole DPP banks (groups of four lanes), so a write mask does the selection -- three
//     v_mov_b32_dpp per dword pair: t = a[partner] (row_ror:8 / row_shl:4), a = b[partner] into the banks with bit B set
//     (bank_mask 0xc / 0xa), b = t into the others (bank_mask 0x3 / 0x5).
//   B = 1, 0: the partner is inside the quad: two v_mov_b32_dpp (quad_perm) and two v_cndmask_b32_e64 with the lane mask in
//     an SGPR pair.  NOT v_cndmask_b32_dpp: the VOP2 encoding of v_cndmask_b32 (mask in vcc) issues at 16-24 cycles per
//     instruction on gfx950 against 4.5-4.9 for the VOP3 encoding (tools/micro/xlane_rate.hip; a DPP move is 4.4, a
//     v_permlane*_swap 9.0, a packed fp32 instruction 4.9).
// `s_nop 1`: a DPP read needs two wait states after a vector write of its source and the hazard recogniser does not look into
// inline assembly; inside a block every DPP source was written at least three instructions earlier.
#define KM_XM " row_mask:0xf bank_mask:"
#define KM_XBANK(CT, MT, CA, MA, MB)                                      \
    "s_nop 1\n\t"                                                        \
    "v_mov_b32_dpp %[t0], %[a0] " CT KM_XM MT "\n\t"                      \
    "v_mov_b32_dpp %[t1], %[a1] " CT KM_XM MT "\n\t"                      \
    "v_mov_b32_dpp %[t2], %[a2] " CT KM_XM MT "\n\t"                      \
    "v_mov_b32_dpp %[t3], %[a3] " CT KM_XM MT "\n\t"                      \
    "v_mov_b32_dpp %[a0], %[b0] " CA KM_XM MA "\n\t"                      \
    "v_mov_b32_dpp %[a1], %[b1] " CA KM_XM MA "\n\t"                      \
    "v_mov_b32_dpp %[a2], %[b2] " CA KM_XM MA "\n\t"                      \
    "v_mov_b32_dpp %[a3], %[b3] " CA KM_XM MA "\n\t"                      \
    "v_mov_b32_dpp %[b0], %[t0] quad_perm:[0,1,2,3]" KM_XM MB "\n\t"      \
    "v_mov_b32_dpp %[b1], %[t1] quad_perm:[0,1,2,3]" KM_XM MB "\n\t"      \
    "v_mov_b32_dpp %[b2], %[t2] quad_perm:[0,1,2,3]" KM_XM MB "\n\t"      \
    "v_mov_b32_dpp %[b3], %[t3] quad_perm:[0,1,2,3]" KM_XM MB "\n\t"      \
    "s_nop 0"
#define KM_XBANK_OPERANDS                                                                                              \
    : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [a0] "+v"(p0), [a1] "+v"(p1), [a2] "+v"(p2), [a3] "+v"(p3), \
      [b0] "+v"(q0), [b1] "+v"(q1), [b2] "+v"(q2), [b3] "+v"(q3)
#define KM_XQUAD(QP)                                                       \
    "s_nop 1\n\t"                                                          \
    "v_mov_b32_dpp %[t0], %[a0] " QP KM_XM "0xf\n\t"                        \
    "v_mov_b32_dpp %[t1], %[a1] " QP KM_XM "0xf\n\t"                        \
    "v_mov_b32_dpp %[u0], %[b0] " QP KM_XM "0xf\n\t"                        \
    "v_mov_b32_dpp %[u1], %[b1] " QP KM_XM "0xf\n\t"                        \
    "v_mov_b32_dpp %[t2], %[a2] " QP KM_XM "0xf\n\t"                        \
    "v_mov_b32_dpp %[t3], %[a3] " QP KM_XM "0xf\n\t"                        \
    "v_mov_b32_dpp %[u2], %[b2] " QP KM_XM "0xf\n\t"                        \
    "v_mov_b32_dpp %[u3], %[b3] " QP KM_XM "0xf\n\t"                        \
    "v_cndmask_b32_e64 %[a0], %[a0], %[u0], %[mk]\n\t"                      \
    "v_cndmask_b32_e64 %[a1], %[a1], %[u1], %[mk]\n\t"                      \
    "v_cndmask_b32_e64 %[b0], %[t0], %[b0], %[mk]\n\t"                      \
    "v_cndmask_b32_e64 %[b1], %[t1], %[b1], %[mk]\n\t"                      \
    "v_cndmask_b32_e64 %[a2], %[a2], %[u2], %[mk]\n\t"                      \
    "v_cndmask_b32_e64 %[a3], %[a3], %[u3], %[mk]\n\t"                      \
    "v_cndmask_b32_e64 %[b2], %[t2], %[b2], %[mk]\n\t"                      \
    "v_cndmask_b32_e64 %[b3], %[t3], %[b3], %[mk]\n\t"                      \
    "s_nop 0"
#define KM_XQUAD_OPERANDS                                                                                              \
    : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [u0] "=&v"(u0), [u1] "=&v"(u1), [u2] "=&v"(u2), [u3] "=&v"(u3), \
      [a0] "+v"(p0), [a1] "+v"(p1), [a2] "+v"(p2), [a3] "+v"(p3), [b0] "+v"(q0), [b1] "+v"(q1), [b2] "+v"(q2), [b3] "+v"(q3)       \
    : [mk] "s"(mk)
template <int B>
__device__ __forceinline__ void swap_lane_dpp(v2f& a0, v2f& b0, v2f& a1, v2f& b1) {
    float p0 = a0.x, p1 = a0.y, p2 = a1.x, p3 = a1.y, q0 = b0.x, q1 = b0.y, q2 = b1.x, q3 = b1.y, t0, t1, t2, t3;
    if constexpr (B == 3) KM_ASM(KM_XBANK("row_ror:8", "0xf", "row_ror:8", "0xc", "0x3") KM_XBANK_OPERANDS);
    else if constexpr (B == 2) KM_ASM(KM_XBANK("row_shl:4", "0x5", "row_shr:4", "0xa", "0x5") KM_XBANK_OPERANDS);
    else {
        constexpr unsigned long long mk = B == 1 ? 0xCCCCCCCCCCCCCCCCull : 0xAAAAAAAAAAAAAAAAull;
        float u0, u1, u2, u3;
        if constexpr (B == 1) KM_ASM(KM_XQUAD("quad_perm:[2,3,0,1]") KM_XQUAD_OPERANDS);
        else KM_ASM(KM_XQUAD("quad_perm:[1,0,3,2]") KM_XQUAD_OPERANDS);
    }
    a0 = v2f{p0, p1}; a1 = v2f{p2, p3}; b0 = v2f{q0, q1}; b1 = v2f{q2, q3};
}
// register index bits (2, 1, 0) <-> lane bits (5, 4, 3)
__device__ __forceinline__ void exchange_hi(v2f (&y)[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) swap_lane32(y[r], y[r + 4]);
#pragma unroll
    for (int r = 0; r < 8; r += 4) { swap_lane16(y[r], y[r + 2]); swap_lane16(y[r + 1], y[r + 3]); }
    swap_lane_dpp<3>(y[0], y[1], y[2], y[3]);
    swap_lane_dpp<3>(y[4], y[5], y[6], y[7]);
}
// register index bits (2, 1, 0) <-> lane bits (2, 1, 0)
__device__ __forceinline__ void exchange_lo(v2f (&y)[8]) {
    swap_lane_dpp<2>(y[0], y[4], y[1], y[5]);
    swap_lane_dpp<2>(y[2], y[6], y[3], y[7]);
    swap_lane_dpp<1>(y[0], y[2], y[1], y[3]);
    swap_lane_dpp<1>(y[4], y[6], y[5], y[7]);
    swap_lane_dpp<0>(y[0], y[1], y[2], y[3]);
    swap_lane_dpp<0>(y[4], y[5], y[6], y[7]);
}
// Where the spectrum ends up: after pass 3 register i of a lane holds Z[c + 64 bitrev(i)], c = fft_lane_bin(lane).  An
// exchange through LDS delivers its digit in natural order, a register exchange bit-reversed (register i of the pass before
// held digit bitrev(i), and register bit j went to lane bit j [+ 3]).
__host__ __device__ constexpr int fft_lane_k0(int lane) { return (KM_MEL_XCHG & 1) ? bitrev(lane >> 3, 3) : lane >> 3; }
__host__ __device__ constexpr int fft_lane_bin(int lane) {
    return (KM_MEL_XCHG & 2) ? fft_lane_k0(lane) + 8 * bitrev(lane & 7, 3) : lane;
}
__host__ __device__ constexpr int fft_bin_lane(int c) {
    return (KM_MEL_XCHG & 2) ? 8 * ((KM_MEL_XCHG & 1) ? bitrev(c & 7, 3) : (c & 7)) + bitrev(c >> 3, 3) : c;
}
}  // namespace melrp

// Interior frames (the common case) are 8 unit-stride 8-byte loads per lane from one base pointer.  Returns false --
// a wave-uniform decision -- when the frame touches the padding or straddles the wrap point of a ring.
template <bool RING>
__device__ __forceinline__ bool load_frame_rp_fast(const float* __restrict__ x, int Li, int rs, int hop, int f, int lane,
                                                   melrp::v2f (&z)[8]) {
    const int lo = f * hop - 512;
    if (lo < 0 || lo + 1024 > Li) return false;
    int start = lo;
    if constexpr (RING) {
        start += rs;
        start -= start >= Li ? Li : 0;
        if (start + 1024 > Li) return false;
    }
    const float* xp = x + start + 2 * lane;
    // (round 4, measured and dropped on the rotating, HBM-resident inputs: one dword per 128-byte line of the frame AFTER next touched a
    // frame early -- step 0.1397 against 0.1334-0.1353 ms, tools/ab_c2.sh; the next frame requested at the START of the current one instead of
    // behind its first pass (no spills, but 16 more registers live through pass 1) -- 0.1354-0.1361 against 0.1337-0.1348; and non-temporal sample loads, so that the audio stream would not displace the power-mel rows the core
    // reads next -- front end 63.6 -> 70 us, core 79.3 -> 82.6 us on rotating inputs)
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = melrp::v2f{xp[128 * i], xp[128 * i + 1]};
    return true;
}

// Edge frames (2 of 257 in the batch shape): zero / reflect padding and ring wrap-around per sample.  Deliberately
// NOT inlined and staged through the wave's LDS buffer: inlined, the compiler if-converts this path into the hot
// block (predicated integer arithmetic executed for every frame: ~500 extra VALU instructions per frame).
__device__ __noinline__ void load_frame_rp_slow(const float* __restrict__ x, int Li, int rs, int ring, int hop, int refl, int f,
                                                int lane, float2* __restrict__ buf) {
    const int p0 = f * hop - 512 + 2 * lane;
    for (int i = 0; i < 8; ++i) {
        float v[2];
        for (int e = 0; e < 2; ++e) {
            int q = p0 + 128 * i + e;
            bool ok = true;
            if (refl) q = q < 0 ? -q : (q >= Li ? 2 * (Li - 1) - q : q);      // np.pad(mode='reflect')
            else ok = q >= 0 && q < Li;                                       // zero padding
            if (ring) { q += rs; q -= q >= Li ? Li : 0; }                     // logical sample q lives at (rs + q) mod L
            v[e] = ok ? x[ok ? q : 0] : 0.f;
        }
        buf[lane + 64 * i] = make_float2(v[0], v[1]);
    }
}

// Mel stage lane map: the four 16-lane groups in which the LDS serves a ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31}, and the same + 32) are the four filter slots; a group's lanes in ascending order are frames 0..15.
__device__ __forceinline__ int mel_slot_of_lane(int lane) {
    const int m = lane & 31;
    const bool g0 = m < 4 || (m >= 12 && m < 16) || (m >= 20 && m < 28);
    return 2 * (lane >> 5) + (g0 ? 0 : 1);
}
__device__ __forceinline__ int mel_frame_of_lane(int lane) {
    const int m = lane & 31;
    if (m < 4) return m;               // group 0: 0-3
    if (m < 12) return m - 4;          // group 1: 4-11 -> 0-7
    if (m < 16) return m - 8;          // group 0: 12-15 -> 4-7
    if (m < 20) return m - 8;          // group 1: 16-19 -> 8-11
    if (m < 28) return m - 12;         // group 0: 20-27 -> 8-15
    return m - 16;                     // group 1: 28-31 -> 12-15
}
__device__ __forceinline__ int mel_lane_of(int slot, int frame) {
    int m;
    if (slot & 1) m = frame < 8 ? frame + 4 : (frame < 12 ? frame + 8 : frame + 16);
    else m = frame < 4 ? frame : (frame < 8 ? frame + 8 : frame + 12);
    return 32 * (slot >> 1) + m;
}

template <bool RING, bool PACK = false>
__global__ __launch_bounds__(512, 4) void mel_power_rp_kernel(MelArgs a) {
    using namespace melrp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* fbuf = reinterpret_cast<float2*>(smem);                     // [WAVES][FFT_BUF]
    float* pw = smem + WAVES * FFT_BUF * 2;                              // [FPB][NFS]
    float* redmax = pw + FPB * NFS;                                      // [WAVES]
    int* sched = reinterpret_cast<int*>(redmax + WAVES);                 // [0]: the chunk index handed to this workgroup last ([1..3] pad)
    float* fbw = redmax + WAVES + 4;                                     // [fbg_nw] filter taps x 1/4, four per step

#ifdef KM_MEL_STAMP
    const unsigned long long st_entry = __builtin_readcyclecounter(), st_entry_rt = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup -> (window, chunk phase).  KM_MEL_REMAP: the linear workgroup id counts windows fastest, so the workgroups of
    // one window are gridDim.y ids apart instead of adjacent (tools/micro: which workgroups share a CU / an XCD's L2).
#if KM_MEL_REMAP
    const unsigned lin_ = blockIdx.x + gridDim.x * blockIdx.y;
    const int b = (int)(lin_ % gridDim.y), bx = (int)(lin_ / gridDim.y);
#else
    const int b = blockIdx.y, bx = (int)blockIdx.x;
#endif
    const int64_t gw = a.win0 + b;
    const int64_t clip = gw / a.wins_per_clip, woff = (gw - clip * a.wins_per_clip) * a.win_step;
    const float* x = a.audio + clip * a.clip_len + woff;
    const int Lv = (int)((a.clip_len - woff) < a.L ? (a.clip_len - woff) : a.L);
    if (a.ready && !a.ready[b]) return;                        // workgroup-uniform: stream still filling
    const int rs = RING ? a.ring_start[b] : -1;
    v2f* buf = reinterpret_cast<v2f*>(fbuf + wave * FFT_BUF);
    const int n_chunks = (a.n_frames + FPB - 1) / FPB;
    const unsigned rd2 = lds_offset(buf + (lane >> 3) * 72 + (lane & 7)), rd3 = lds_offset(buf + lane * 9);
    const unsigned wr3 = lds_offset(buf + fft_lane_k0(lane) * 9 + (lane & 7));
    const int cbin = fft_lane_bin(lane);                       // this lane finishes the bins cbin + 64 i and 512 - cbin - 64 i

    // Software prefetch: the samples of this wave's NEXT frame are requested as soon as pass 1 has left the registers
    // (they complete under passes 2-3 and the post-processing; across a chunk boundary under the mel stage).  Without
    // it every workgroup stalls on HBM latency after each barrier: the kernel is latency bound, not VALU bound.  The first
    // frame is requested here, ahead of everything else the prologue loads (tables through L2; these come from HBM).
    v2f zn[8];
    bool zn_ok = false;                                                  // wave-uniform: zn holds the next frame
    const int fmul = a.frame_mul;
    auto first_frame = [&]() {
        if (KM_MEL_SKIP & 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) zn[i] = v2f{0.001f * lane, 0.5f + i};
            zn_ok = true;
        } else if (bx * FPB + wave < a.n_frames) zn_ok = load_frame_rp_fast<RING>(x, Lv, rs, a.hop, (bx * FPB + wave) * fmul, lane, zn);
    };
    // The emotion stream of this window (0.2 MFLOP, latency bound, independent of the audio) rides in the last workgroup of
    // the window: no separate launch, and the chunk hand-out below lets that workgroup take correspondingly fewer chunks.
    // (It requests its first frame after the rider: the rider's weight loads need the registers.)
    if (!(KM_MEL_SKIP & 16) && a.emo.emo && bx == (int)gridDim.x - 1) {
        if (a.emo.d == 256 && a.emo.DH == 128) emotion_window_d256(a.emo, gw, pw);
        else emotion_window_generic(a.emo, EmoShape{a.emo.d, a.emo.DH}, gw, pw);
        first_frame();
    } else {
        first_frame();
    }

    // Chunks: the first two of a workgroup are fixed (bx and bx + gridDim.x: their first frames can be requested at once),
    // the others are handed out on request, so the workgroups of a window finish within one chunk of each other whatever
    // else they did (the rider above) or shared their CU with.  A workgroup always knows its current and its next chunk (the
    // next one's first frames are prefetched under the current one) and asks for the one after while it works on the
    // current one: request k of the launch returns chunk 2 gridDim.x + k.  Every chunk is worked on exactly once and each
    // costs one request, so a window sees n_chunks requests per launch; the last one puts the counter back to zero.
    const unsigned ctr_last = (unsigned)n_chunks - 1u;
    for (int i = tid; i < a.fbg_nw; i += NT) fbw[i] = a.fbg_weight[i];
    for (int i = tid; i < FPB * (NFS - 513); i += NT) pw[(i / (NFS - 513)) * NFS + 513 + i % (NFS - 513)] = 0.f;   // row padding the mel steps may read

    // lane-constant operands (56 registers)
    const int n2 = lane & 7, k0 = fft_lane_k0(lane);
    v2f win2[8], tw1[8], tw2[8], twp[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        win2[i] = v2f{a.window[2 * (lane + 64 * i)], a.window[2 * (lane + 64 * i) + 1]};
        tw1[i] = ld2(a.twiddle + ((2 * lane * bitrev(i, 3)) & 1023));   // W_512^(lane k0)
        tw2[i] = ld2(a.twiddle + ((16 * n2 * bitrev(i, 3)) & 1023));    // W_64^(n2 k1)
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) twp[i] = ld2(a.twiddle + cbin + 64 * i);   // W_1024^k, k = cbin + 64 i

    // filter groups of this wave: lane = 16 x (filter in group) + frame; group slot i of the wave holds group my_gid[i]
    // (wave-uniform, -1 = none) and this lane's descriptor first bin / 4 | steps << 8 | tap offset << 16 (steps equal
    // within a group).  The descriptors are re-read (L1) ahead of every chunk's barrier rather than held in registers
    // across the FFT, which has none to spare.
    int my_gid[NGW];
#pragma unroll
    for (int i = 0; i < NGW; ++i) my_gid[i] = __builtin_amdgcn_readfirstlane(a.fbg_gid[wave * NGW + i]);
    const int* my_desc_p = a.fbg_desc + wave * NGW * 4 + mel_slot_of_lane(lane);
    // Use every loop-invariant operand once before the loop: the waits for their loads are placed here, not (with
    // conservative counts that would also drain the sample prefetch) at their first use inside the loop.
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(win2[i]), "v"(tw1[i]), "v"(tw2[i]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(twp[i]));
#pragma unroll
    for (int i = 0; i < NGW; ++i) asm volatile("" ::"s"(my_gid[i]));

    float vmax = 0.f;
    int chunk = bx, chunk_next = bx + (int)gridDim.x;
#if defined(KM_MEL_DELAY)          /* timing experiment: hold the odd workgroups of a window back by KM_MEL_DELAY x 64 cycles */
    if (bx & 1) { for (int i = 0; i < KM_MEL_DELAY; i += 16) __builtin_amdgcn_s_sleep(16); }
#endif
#ifdef KM_MEL_STAMP
    unsigned st_acc[16] = {0};
    unsigned long long st_last = __builtin_readcyclecounter();
    const unsigned long long st_first = st_last;
#endif
    while (chunk < n_chunks) {
        unsigned req = 0;                                                // the chunk after next: requested here, looked at before the barrier
        if (tid == 0) req = atomicAdd(a.chunk_ctr + b, 1u);
        const int f0 = chunk * FPB;
#pragma unroll 1
        for (int pi = 0; pi < FPB / WAVES; ++pi) {
            const int fl = wave + WAVES * pi;
            const int f = f0 + fl;
            if (f < a.n_frames) {   // wave-uniform
                if (!zn_ok) {       // edge frame (padding / ring wrap): 2 of 257 in the batch shape
                    load_frame_rp_slow(x, Lv, rs, RING ? 1 : 0, a.hop, a.pad_mode == KM_PAD_REFLECT ? 1 : 0, f * fmul, lane,
                                       reinterpret_cast<float2*>(buf));
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i = 0; i < 8; ++i) zn[i] = buf[lane + 64 * i];
                    __builtin_amdgcn_wave_barrier();
                }
                KM_STAMP(0);                                              // loop overhead, edge frames, waiting for the samples
                v2f z[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) z[i] = zn[i] * win2[i];
                // ---- pass 1: radix-8 over n0 (m = 64 n0 + lane) ----
                if (!(KM_MEL_SKIP & 4)) { dif8(z); cmul7(z, tw1); }
                KM_STAMP(1);
                v2f y[8];
                if ((KM_MEL_XCHG & 1) && !(KM_MEL_SKIP & 2)) {              // register i (digit k0 = bitrev(i)) <-> lane bits 5:3 (n1)
                    exchange_hi(z);
#pragma unroll
                    for (int i = 0; i < 8; ++i) y[i] = z[i];
                } else if (!(KM_MEL_SKIP & 2)) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) buf[bitrev(i, 3) * 72 + lane] = z[i];
                    __builtin_amdgcn_wave_barrier();
                }
                {
                    const int fnext = pi + 1 < FPB / WAVES ? f + WAVES : chunk_next * FPB + wave;
                    if (!(KM_MEL_SKIP & 8)) zn_ok = fnext < a.n_frames && load_frame_rp_fast<RING>(x, Lv, rs, a.hop, fnext * fmul, lane, zn);
                }
                // ---- pass 2: radix-8 over n1 (lane = 8 n1 + n2); this lane: n2 = lane & 7, digit k0 = fft_lane_k0(lane) ----
                if (!(KM_MEL_XCHG & 1) && !(KM_MEL_SKIP & 2)) {
                    lds_read8<64>(rd2, y);                                // y[n1] = buf[k0 * 72 + 8 n1 + n2]
                    __builtin_amdgcn_wave_barrier();
                } else if (KM_MEL_SKIP & 2) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) y[i] = z[i];
                }
                KM_STAMP(2);                                              // exchange 1 (+ issuing the next frame's loads)
                if (!(KM_MEL_SKIP & 4)) { dif8(y); cmul7(y, tw2); }
                KM_STAMP(3);
                // ---- pass 3: radix-8 over n2; afterwards register i holds Z[cbin + 64 bitrev(i)] ----
                if ((KM_MEL_XCHG & 2) && !(KM_MEL_SKIP & 2)) {              // register i (digit k1 = bitrev(i)) <-> lane bits 2:0 (n2)
                    exchange_lo(y);
                } else if (!(KM_MEL_SKIP & 2)) {
                    // transposed image indexed by c = k0 + 8 k1, row stride 9 (padding keeps every address base + immediate;
                    // the residual 2-way conflict on 3 of 32 slots is cheaper than per-access swizzle arithmetic)
#if KM_MEL_W2
#pragma unroll
                    for (int i = 0; i < 8; ++i) buf[(k0 + 8 * bitrev(i, 3)) * 9 + n2] = y[i];
#else
                    lds_write8_bitrev<8 * 9 * 8>(wr3, y);                 // buf[(k0 + 8 bitrev(i)) * 9 + n2] = y[i]
#endif
                    __builtin_amdgcn_wave_barrier();
                    lds_read8<8>(rd3, y);                                 // y[q] = buf[lane * 9 + q]
                    __builtin_amdgcn_wave_barrier();
                }
                KM_STAMP(4);                                              // exchange 2
                if (!(KM_MEL_SKIP & 4)) dif8(y);
                KM_STAMP(5);
                // ---- real-input post-processing + |.|^2, two bins (k and 512 - k) per packed instruction ----
                // y[i] = Z[c + 64 bitrev(i)], c = cbin.  The bins k = c + 64 i, i = 0..3, this lane finishes are its own registers
                // 0, 4, 2, 6; their partners Z[512 - k] = Z[(64 - c) + 64 (7 - i)] are registers 7, 3, 5, 1 of the lane that
                // holds 64 - c: one fixed lane permutation of four complex registers (8 ds_bpermute_b32, no LDS memory) instead of
                // a third exchange through the buffer (8 ds_write_b64 + 8 ds_read_b64).  c = 0 (lane 0) pairs with itself, one
                // register further: Z[512 - 64 i] = Z[64 ((8 - i) & 7)] = its registers 0, 7, 3, 5.
                float* pwr = pw + fl * NFS;
                v2f zk[4], zc[4];
                v2f zmid = y[1];                                          // Z[256] in lane 0 (k = 256 pairs with itself)
                if (!(KM_MEL_SKIP & 2)) {
                    const int src = fft_bin_lane((64 - cbin) & 63) << 2;
                    auto perm = [&](v2f v) {
                        return v2f{__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.x))),
                                   __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.y)))};
                    };
                    const v2f p7 = perm(y[7]), p3 = perm(y[3]), p5 = perm(y[5]), p1 = perm(y[1]);
                    const bool l0 = lane == 0;
                    zk[0] = y[0]; zk[1] = y[4]; zk[2] = y[2]; zk[3] = y[6];
                    zc[0] = l0 ? y[0] : p7; zc[1] = l0 ? y[7] : p3; zc[2] = l0 ? y[3] : p5; zc[3] = l0 ? y[5] : p1;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { zk[i] = y[i]; zc[i] = y[i + 4]; }
                }
                KM_STAMP(6);                                              // partner permutation
                if (!(KM_MEL_SKIP & 4)) post4(zk, zc, twp);                                       // the 1/4 of |X|^2 lives in the filter weights
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pwr[cbin + 64 * i] = zk[i].x;
                    pwr[2 * 256 - cbin - 64 * i] = zk[i].y;
                }
                if (lane == 0) pwr[256] = 4.0f * (zmid.x * zmid.x + zmid.y * zmid.y);      // k = 256 pairs with itself
                KM_STAMP(7);                                              // post-processing + power stores
            }
        }
        KM_STAMP(8);
        int my_desc[NGW];
#pragma unroll
        for (int i = 0; i < NGW; ++i) my_desc[i] = my_desc_p[4 * i];
        if (tid == 0) {
            if (req == ctr_last) a.chunk_ctr[b] = 0u;
            sched[0] = (int)(req + 2u * gridDim.x);
        }
        __syncthreads();
        KM_STAMP(9);                                                      // waiting at the barrier ahead of the mel stage
        // The stage is a chain of LDS round trips with two vector instructions per step: at top priority those are issued
        // ahead of the FFT arithmetic of the sibling workgroup's waves on the same SIMD instead of queueing behind it
        // (same-run A/B: front end 57.6 -> 55.7 us on one box, 53.5 -> 51.5 on another; priority 1 does the same)
        if (KM_MEL_PRIO) __builtin_amdgcn_s_setprio(KM_MEL_PRIO);
        {   // Sparse triangular mel filters.  A lane owns one (frame, filter) pair: lane = 16 x (filter in its group of
            // four) + frame, and walks the filter four bins a step: one ds_read_b128 of powers and one 16-byte load of taps
            // (the 16 frame lanes of a filter share the address; the taps come through the vector cache, not LDS: the
            // stage is bound by LDS bandwidth, and the power reads alone are 6 KB per frame here against 16 KB for powers
            // + taps in 16-bin trips with one filter per wave -- 24 of the kernel's 77 us at the C2 shape).  Row stride
            // 580 = 4 (mod 64) dwords: the 16 lanes of a b128 beat (16 frames, one filter) cover all 64 banks -- the beats
            // of ds_read_b128 are NOT contiguous lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the same + 32:
            // MI355X_MICROARCH.md, LDS table), so (filter slot, frame) follow that grouping (mel_slot_of_lane): with
            // lane = 16 slot + frame every beat mixed two filters and SQ_LDS_BANK_CONFLICT rose from 4.9 to 8.9 % of the
            // busy cycles.  No cross-lane reduction, no per-filter descriptor traffic.  Results go straight to HBM (the 80 filters of a
            // frame are written by 8 waves within one chunk, L2 merges the lines).
            const int slot = mel_slot_of_lane(lane), flm = mel_frame_of_lane(lane);
            const bool fvalid = f0 + flm < a.n_frames;
            const float4* prow = reinterpret_cast<const float4*>(pw + flm * NFS);
            const float4* wbase = reinterpret_cast<const float4*>(fbw);
            float* dst = a.melpow + ((int64_t)b * a.n_frames + f0 + flm) * a.n_mels + slot;
            // PACK: filter m of frame f is element (m, f) of the window's packed rows -- the 16 frame lanes of a filter write 64
            // contiguous bytes -- and the last three computed frames are stored a second time behind the T long ones
            const int fr = f0 + flm, qs = a.n_frames >= 3 ? fr - (a.n_frames - 3) : fr;
            const bool st_long = fvalid && fr < a.pack_T, st_short = fvalid && qs >= 0 && qs < 3;
            float* dst_long = nullptr; float* dst_short = nullptr;
            if constexpr (PACK) {
                dst_long = a.pack_xt + ((int64_t)b * a.n_mels + slot) * a.pack_KP + fr;
                dst_short = a.pack_xt + ((int64_t)b * a.n_mels + slot) * a.pack_KP + a.pack_T + qs;
            }
            auto store_mel = [&](int gid, float r) {
                if (4 * gid + slot >= a.n_mels) return;
                if constexpr (PACK) {
                    const float db = db10(r, a.pack_amin);
                    if (st_long) dst_long[(int64_t)4 * gid * a.pack_KP] = db;
                    if (st_short) dst_short[(int64_t)4 * gid * a.pack_KP] = db;
                } else {
                    if (fvalid) dst[4 * gid] = r;
                }
            };
            float fmx = 0.f;
#if KM_MEL_FLAT
            // One chain of steps over all groups of the wave: the operands of a step are requested one step ahead, and the
            // last step of a group requests the first step of the next group (not, as before, its own operands again:
            // ~2.5 of a wave's ~14.5 steps per chunk were such re-reads, and every group started with an exposed LDS round trip).
            auto group_pr = [&](int i) { return prow + (my_desc[i] & 255); };
            auto group_wt = [&](int i) { return wbase + ((unsigned)my_desc[i] >> 16); };
            float4 wv = make_float4(0.f, 0.f, 0.f, 0.f), pv = wv;
            if (!(KM_MEL_SKIP & 1) && my_gid[0] >= 0) { wv = group_wt(0)[0]; pv = group_pr(0)[0]; }
#pragma unroll
            for (int i = 0; i < ((KM_MEL_SKIP & 1) ? 0 : NGW); ++i) {
                if (my_gid[i] < 0) break;                                              // wave-uniform
                const int steps = __builtin_amdgcn_readfirstlane((my_desc[i] >> 8) & 255);
                const float4* pr = group_pr(i);
                const float4* wt = group_wt(i);
                v2f acc = {0.f, 0.f};
                for (int t = 1; t < steps; ++t) {
                    const float4 wn = wt[t], pn = pr[t];
                    acc = __builtin_elementwise_fma(v2f{pv.x, pv.y}, v2f{wv.x, wv.y}, acc);
                    acc = __builtin_elementwise_fma(v2f{pv.z, pv.w}, v2f{wv.z, wv.w}, acc);
                    pv = pn; wv = wn;
                }
                float4 wn = wv, pn = pv;
                if (i + 1 < NGW && my_gid[i + 1 < NGW ? i + 1 : i] >= 0) { wn = group_wt(i + 1 < NGW ? i + 1 : i)[0]; pn = group_pr(i + 1 < NGW ? i + 1 : i)[0]; }
                acc = __builtin_elementwise_fma(v2f{pv.x, pv.y}, v2f{wv.x, wv.y}, acc);
                acc = __builtin_elementwise_fma(v2f{pv.z, pv.w}, v2f{wv.z, wv.w}, acc);
                pv = pn; wv = wn;
                float r = acc.x + acc.y;
                r = fvalid ? r : 0.f;
                asm("v_max_f32 %0, %0, %1" : "+v"(vmax) : "v"(r));
                asm("v_max_f32 %0, %0, %1" : "+v"(fmx) : "v"(r));
                store_mel(my_gid[i], r);
            }
#else
#pragma unroll
            for (int i = 0; i < ((KM_MEL_SKIP & 1) ? 0 : NGW); ++i) {
                if (my_gid[i] < 0) break;                                              // wave-uniform
                const int desc = my_desc[i];
                const int steps = __builtin_amdgcn_readfirstlane((desc >> 8) & 255);
                const float4* pr = prow + (desc & 255);
                const float4* wt = wbase + ((unsigned)desc >> 16);
                v2f acc = {0.f, 0.f};
                float4 wv = wt[0], pv = pr[0];
                for (int t = 1; t < steps; ++t) {                                      // the next step's operands ahead of the FMAs
                    const float4 wn = wt[t], pn = pr[t];
                    acc = __builtin_elementwise_fma(v2f{pv.x, pv.y}, v2f{wv.x, wv.y}, acc);
                    acc = __builtin_elementwise_fma(v2f{pv.z, pv.w}, v2f{wv.z, wv.w}, acc);
                    pv = pn; wv = wn;
                }
                acc = __builtin_elementwise_fma(v2f{pv.x, pv.y}, v2f{wv.x, wv.y}, acc);
                acc = __builtin_elementwise_fma(v2f{pv.z, pv.w}, v2f{wv.z, wv.w}, acc);
                float r = acc.x + acc.y;
                r = fvalid ? r : 0.f;
                asm("v_max_f32 %0, %0, %1" : "+v"(vmax) : "v"(r));
                asm("v_max_f32 %0, %0, %1" : "+v"(fmx) : "v"(r));
                store_mel(my_gid[i], r);
            }
#endif
            if (a.frame_max) {                                             // this wave's filters of frame f0 + flm
                float fall = fmx;
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) fall = fmaxf(fall, __shfl(fmx, mel_lane_of(s2, flm)));
                fmx = fall;
                if (slot == 0 && fvalid) atomicMax(a.frame_max + (int64_t)b * a.n_frames + f0 + flm, __float_as_uint(fmx));
            }
        }
        if (KM_MEL_PRIO) __builtin_amdgcn_s_setprio(0);
        KM_STAMP(10);                                                     // mel stage
        chunk = chunk_next;
        chunk_next = __builtin_amdgcn_readfirstlane(sched[0]);           // written ahead of the barrier that opened the mel stage
        __syncthreads();                                                 // the power rows are free again
        KM_STAMP(11);
    }
#ifdef KM_MEL_STAMP
    if (lane == 0) {
        unsigned long long* o = km_mel_stamps + ((size_t)(b * gridDim.x + bx) * WAVES + wave) * 24;
        if (b * gridDim.x + bx < 1024) {
            for (int i = 0; i < 12; ++i) o[i] = st_acc[i];
            o[12] = st_last - st_first;
            o[13] = st_entry; o[14] = st_first; o[15] = st_last; o[16] = st_entry_rt; o[17] = __builtin_amdgcn_s_memrealtime();
            o[18] = __builtin_readcyclecounter();
        }
    }
#endif
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if (lane == 0) redmax[wave] = vmax;
    __syncthreads();
    if (tid == 0 && !a.frame_max) {
        float mx = redmax[0];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) mx = fmaxf(mx, redmax[w]);
        atomicMax(a.melmax + b, __float_as_uint(mx));
    }
}

struct LogArgs {
    const float* melpow;     // (B, n_frames, n_mels)
    const unsigned* melmax;  // (B)
    int n_frames, out_frames, n_mels;
    LogParams lp;
    float* mel_long;         // (B, out_frames, n_mels)
    float* mel_short;        // (B, 3, n_mels) or null: last three COMPUTED frames
};

__global__ __launch_bounds__(256) void mel_log_kernel(LogArgs a) {
    const int b = blockIdx.y;
    float ref_db, floor_db;
    log_window_consts(a.lp, __uint_as_float(a.melmax[b]), ref_db, floor_db);
    const float* src = a.melpow + (int64_t)b * a.n_frames * a.n_mels;
    const int total = a.out_frames * a.n_mels;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < total) {
        int f = i / a.n_mels;
        const int m = i - f * a.n_mels;
        if (f >= a.n_frames) f = a.n_frames - 1;            // repeat the last frame (stft.py:136-140)
        a.mel_long[(int64_t)b * total + i] = log_one(a.lp, src[f * a.n_mels + m], ref_db, floor_db);
    }
    if (a.mel_short && blockIdx.x == 0) {
        for (int s = threadIdx.x; s < 3 * a.n_mels; s += 256) {
            const int r = s / a.n_mels, m = s - r * a.n_mels;
            float v = 0.f;
            // simplified_dual_stream_model.py:206-214: last 3 frames; fewer than 3 -> first rows, zero rest
            if (a.n_frames >= 3) v = log_one(a.lp, src[(a.n_frames - 3 + r) * a.n_mels + m], ref_db, floor_db);
            else if (r < a.n_frames) v = log_one(a.lp, src[r * a.n_mels + m], ref_db, floor_db);
            a.mel_short[(int64_t)b * 3 * a.n_mels + s] = v;
        }
    }
}

// Log conversion straight into the packed encoder input of the generic core (km_generic.hip encoder_tn_kernel):
// xp (B, KP, n_mels) with rows 0 .. T-1 = the first min(n_frames, T) frames (zero rows if the window is shorter,
// dual_stream_attention.py:193-202), rows T .. T+2 = the last three computed frames
// (simplified_dual_stream_model.py:205-214), rows up to KP zero.  One pass replaces mel_log_kernel + the K=3 GEMM.
__global__ __launch_bounds__(256) void mel_log_packed_kernel(LogArgs a, float* __restrict__ xp, int T, int KP) {
    const int b = blockIdx.y;
    float ref_db, floor_db;
    log_window_consts(a.lp, __uint_as_float(a.melmax[b]), ref_db, floor_db);
    const float* src = a.melpow + (int64_t)b * a.n_frames * a.n_mels;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= KP * a.n_mels) return;
    const int r = i / a.n_mels, m = i - r * a.n_mels;
    int f = -1;
    if (r < T) f = r < a.n_frames ? r : -1;
    else if (r < T + 3) {
        const int q = r - T;
        if (a.n_frames >= 3) f = a.n_frames - 3 + q; else if (q < a.n_frames) f = q;
    }
    xp[(int64_t)b * KP * a.n_mels + i] = f >= 0 ? log_one(a.lp, src[f * a.n_mels + m], ref_db, floor_db) : 0.f;
}

// ---------------------------------------------------------------------------------------------
template <typename Tv>
static int upload(Tv** dst, const std::vector<Tv>& src) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(dst), src.size() * sizeof(Tv)));
    HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(Tv), hipMemcpyHostToDevice));
    return KM_OK;
}

int upload_mel_plan(MelPlan* p) {
    if (p->uploaded) return KM_OK;
    if (int rc = upload(&p->d_window, p->window)) return rc;
    if (int rc = upload(&p->d_twiddle, p->twiddle)) return rc;
    if (int rc = upload(&p->d_fb_start, p->fb_start)) return rc;
    if (int rc = upload(&p->d_fb_count, p->fb_count)) return rc;
    if (int rc = upload(&p->d_fb_offset, p->fb_offset)) return rc;
    if (int rc = upload(&p->d_fb_weight, p->fb_weight)) return rc;
    if (!p->fbg_gid.empty()) {
        if (int rc = upload(&p->d_fbg_gid, p->fbg_gid)) return rc;
        if (int rc = upload(&p->d_fbg_desc, p->fbg_desc)) return rc;
        if (int rc = upload(&p->d_fbg_weight, p->fbg_weight)) return rc;
    }
    p->uploaded = true;
    return KM_OK;
}

void free_mel_plan(MelPlan* p) {
    if (p->uploaded) {
        (void)hipFree(p->d_window); (void)hipFree(p->d_twiddle); (void)hipFree(p->d_fb_start);
        (void)hipFree(p->d_fb_count); (void)hipFree(p->d_fb_offset); (void)hipFree(p->d_fb_weight);
        (void)hipFree(p->d_fbg_gid); (void)hipFree(p->d_fbg_desc); (void)hipFree(p->d_fbg_weight);
    }
    delete p;
}

static size_t melrp_lds_bytes(int nnz4) {
    return (size_t)(melrp::WAVES * melrp::FFT_BUF * 2 + melrp::FPB * melrp::NFS + melrp::WAVES + 4 + nnz4) * sizeof(float);
}

static size_t mel_lds_bytes(int nfft, int n_mels = 128, int nnz = 0) {
    if (nnz == 0) nnz = 2 * (nfft / 2 + 1) + 4 * 128;    // upper bound: every bin feeds at most two triangles, rows padded to 4
    return (size_t)(mel::WAVES * mel::FFT_BUF * 2 + mel::FPB * (nfft / 2 + 2) + mel::FPB * n_mels + mel::WAVES +
                    3 * n_mels + nnz) * sizeof(float);
}

static LogParams log_params(const km_mel_config& m) {
    LogParams lp;
    lp.log_mode = m.log_mode; lp.amin = m.amin; lp.top_db = m.top_db; lp.db_add = m.db_add;
    lp.db_scale = m.db_scale; lp.log_eps = m.log_eps;
    return lp;
}

// One chunk-request counter per window of a launch (MelArgs::chunk_ctr).  The kernel leaves them at zero, so the buffer is
// zeroed once, when it is (re)allocated; km_reserve sizes it, a larger launch grows it (not inside a stream capture).
int ensure_chunk_counters(Context* c, int64_t windows, void* stream) {
    if (windows <= c->ws_chunkctr_cap) return KM_OK;
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(KM_ERR_WORKSPACE, "front end: %lld windows exceed the reserved %lld during a stream capture: call km_reserve first",
                    (long long)windows, (long long)c->ws_chunkctr_cap);
    if (c->ws_chunkctr) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(c->ws_chunkctr)); c->ws_chunkctr = nullptr; c->ws_chunkctr_cap = 0; }
    const int64_t cap_n = windows < 4096 ? 4096 : windows;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ws_chunkctr), (size_t)cap_n * sizeof(unsigned)));
    HIP_TRY(hipMemsetAsync(c->ws_chunkctr, 0, (size_t)cap_n * sizeof(unsigned), st));
    HIP_TRY(hipStreamSynchronize(st));            // the caller's later launches may be on another stream
    c->ws_chunkctr_cap = cap_n;
    return KM_OK;
}

// power-mel (B, n_frames, n_mels) + per-window max into the workspace
// true when launch_mel_power can compute the per-window emotion logits inside the front-end kernel
bool mel_fuses_emotion(Context* c, MelPlan* p) {
    const bool use_rp = !c->opt.mel_two_frame && !c->opt.emotion_separate;
    auto pow2 = [](int v) { return v >= 64 && v <= 1024 && (v & (v - 1)) == 0; };
    static const bool generic_ok = std::getenv("KM_EMOTION_GENERIC_SEPARATE") == nullptr;
    const bool d256 = c->d == 256 && c->DH == 128;
    return use_rp && p->cfg.n_fft == 1024 && c->kind == 0 && c->ED <= 256 && (d256 || (generic_ok && pow2(c->d) && pow2(c->DH)));
}

// true when launch_mel_power can write the training step's packed input itself (MelPack): the 1024-point kernel with a grouped
// filter image, librosa's dB conversion with top_db == db_add (the floor maps to exactly 0), and at least T frames (every slot
// of a row is then written by the launch: short clips take the conversion operation of phase 0)
bool mel_packs(Context* c, MelPlan* p, int64_t n_frames, int64_t T) {
    const km_mel_config& m = p->cfg;
    return m.n_fft == 1024 && !c->opt.mel_two_frame && p->d_fbg_gid != nullptr && m.log_mode == KM_LOG_DB_MAX && m.top_db == m.db_add &&
           n_frames >= T && n_frames >= 3;
}

int launch_mel_power(Context* c, MelPlan* p, const float* audio, int64_t B, int64_t L, void* stream,
                     int64_t clip_len, int64_t win_step, int64_t win0, int wins_per_clip, const int* ring_start,
                     const unsigned char* ready, const float* emotion, float* zemo, const SeqFrames* seq, const MelPack* pack) {
    const km_mel_config& m = p->cfg;
    const int64_t n_frames = seq ? seq->n_rows : 1 + L / m.hop_length;
    if (m.pad_mode == KM_PAD_REFLECT && L <= m.n_fft / 2)
        return fail(KM_ERR_INVALID_ARG, "reflect padding needs more than n_fft/2 = %d samples (got %lld)", m.n_fft / 2, (long long)L);
    if (!seq && (B > c->ws_windows || n_frames > c->ws_frames))
        return fail(KM_ERR_WORKSPACE, "workspace holds %lld windows x %lld frames, need %lld x %lld: call km_reserve",
                    (long long)c->ws_windows, (long long)c->ws_frames, (long long)B, (long long)n_frames);
    if (!seq && m.n_mels > c->ws_mels)
        return fail(KM_ERR_WORKSPACE, "workspace rows hold %d mel bins, this plan has %d: call km_reserve", c->ws_mels, m.n_mels);
    if (seq && !(m.n_fft == 1024 && !c->opt.mel_two_frame))
        return fail(KM_ERR_UNSUPPORTED, "shared-frame sequence mode needs the 1024-point front end");
    if (!p->uploaded) return fail(KM_ERR_NOT_FINALIZED, "mel plan not uploaded (km_finalize / km_reserve first)");
    static PerDeviceOnce once;
    if (once.first(c->device)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mel_power_kernel<1024, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)mel_lds_bytes(1024)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mel_power_kernel<512, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)mel_lds_bytes(512)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mel_power_kernel<1024, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)mel_lds_bytes(1024)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mel_power_kernel<512, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)mel_lds_bytes(512)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mel_power_rp_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mel_power_rp_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mel_power_rp_kernel<false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipStream_t st = (hipStream_t)stream;
    if (int rc = ensure_chunk_counters(c, B, stream)) return rc;
    // ws_melmax is all-zero on entry: the fused core, the generic encoder and phase 1 of the training program re-zero the
    // entries they consume, and the stand-alone log kernels below are followed by a memset of theirs -- so a step recorded
    // into a hipGraph needs no memset node and stays correct whatever ran between two replays.  The flag only covers a
    // freshly (re)allocated workspace.
    if (!seq && c->melmax_dirty) {
        HIP_TRY(hipMemsetAsync(c->ws_melmax, 0, (size_t)c->ws_windows * sizeof(unsigned), st));
        c->melmax_dirty = false;
    }
    MelArgs a;
    a.audio = audio; a.L = L; a.clip_len = clip_len > 0 ? clip_len : L; a.win_step = win_step; a.win0 = win0;
    a.wins_per_clip = wins_per_clip > 0 ? wins_per_clip : 1; a.n_frames = (int)n_frames;
    a.ring_start = ring_start; a.ready = ready; a.hop = m.hop_length; a.pad_mode = m.pad_mode;
    a.n_mels = m.n_mels; a.window = p->d_window; a.twiddle = reinterpret_cast<const float2*>(p->d_twiddle);
    a.fb_start = p->d_fb_start; a.fb_count = p->d_fb_count; a.fb_offset = p->d_fb_offset; a.fb_weight = p->d_fb_weight;
    a.fb_nnz = (int)p->fb_weight.size();
    a.fbg_gid = p->d_fbg_gid; a.fbg_desc = p->d_fbg_desc; a.fbg_weight = p->d_fbg_weight; a.fbg_nw = (int)p->fbg_weight.size();
    a.melpow = c->ws_melpow; a.melmax = c->ws_melmax;
    a.frame_mul = 1; a.frame_max = nullptr; a.chunk_ctr = c->ws_chunkctr;
    a.pack_xt = nullptr; a.pack_T = a.pack_KP = 0; a.pack_amin = m.amin;
    if (pack) {
        if (!mel_packs(c, p, n_frames, pack->T) || seq || ring_start || wins_per_clip > 1 || pack->KP < pack->T + 3)
            return fail(KM_ERR_UNSUPPORTED, "launch_mel_power: packed output requested for an unsupported configuration");
        a.pack_xt = pack->xt; a.pack_T = pack->T; a.pack_KP = pack->KP;
    }
    if (seq) { a.melpow = seq->pow; a.frame_max = seq->fmax; a.frame_mul = seq->frame_mul; a.melmax = nullptr; }
    if (emotion) {
        if (!mel_fuses_emotion(c, p) || !zemo || wins_per_clip > 1)
            return fail(KM_ERR_UNSUPPORTED, "launch_mel_power: emotion fusion requested for an unsupported configuration");
        auto dvp = [&](const char* name) { return (const float*)c->packed.at(name).dev; };
        a.emo.emo = emotion; a.emo.ED = c->ED; a.emo.zemo = zemo;
        a.emo.d = c->d; a.emo.DH = c->DH;
        a.emo.wee_t = (c->d == 256 && c->DH == 128) ? dvp("wee_t256") : dvp("wee_t"); a.emo.bee = dvp("bee"); a.emo.lg = dvp("eln_g"); a.emo.lb = dvp("eln_b");
        a.emo.we2 = dvp("we2"); a.emo.be2 = dvp("be2"); a.emo.w2 = dvp("w2"); a.emo.b2 = dvp("b2");
    }
    // persistent over frame chunks: each workgroup walks chunks blockIdx.x, +gridDim.x, ... of its window with the
    // next pair's samples prefetched; two workgroups are resident per CU (LDS), so aim at 512 in total
    const int n_chunks = (int)((n_frames + mel::FPB - 1) / mel::FPB);
    int per_window = (int)((512 + B / 2) / B);
    if (per_window > n_chunks) per_window = n_chunks;
    if (per_window < 1) per_window = 1;
    const dim3 grid((unsigned)per_window, (unsigned)B);
    const bool use_rp = !c->opt.mel_two_frame;   // A/B switch: the two-frames-per-wave kernel
    if (m.n_fft == 1024 && use_rp && p->d_fbg_gid) {     // (more than 128 filters have no grouped image: the kernel below)
        const size_t ldsrp = melrp_lds_bytes(a.fbg_nw);      // 74 KB: two workgroups per CU
        if (pack) hipLaunchKernelGGL((mel_power_rp_kernel<false, true>), grid, dim3(melrp::NT), ldsrp, st, a);
        else if (!ring_start) hipLaunchKernelGGL((mel_power_rp_kernel<false>), grid, dim3(melrp::NT), ldsrp, st, a);
        else hipLaunchKernelGGL((mel_power_rp_kernel<true>), grid, dim3(melrp::NT), ldsrp, st, a);
        HIP_TRY(hipGetLastError());
        return KM_OK;
    }
    const size_t lds = mel_lds_bytes(m.n_fft, m.n_mels, a.fb_nnz);
    if (m.n_fft == 1024 && !ring_start) hipLaunchKernelGGL((mel_power_kernel<1024, false>), grid, dim3(256), lds, st, a);
    else if (m.n_fft == 1024) hipLaunchKernelGGL((mel_power_kernel<1024, true>), grid, dim3(256), lds, st, a);
    else if (!ring_start) hipLaunchKernelGGL((mel_power_kernel<512, false>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((mel_power_kernel<512, true>), grid, dim3(256), lds, st, a);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

// front end for the generic core: power-mel, then the packed log-mel image (B, KP, n_mels)
int launch_mel_packed(Context* c, MelPlan* p, const float* audio, int64_t B, int64_t L, float* xp, int T, int KP, void* stream) {
    if (int rc = launch_mel_power(c, p, audio, B, L, stream)) return rc;
    const km_mel_config& m = p->cfg;
    LogArgs g;
    g.melpow = c->ws_melpow; g.melmax = c->ws_melmax; g.n_frames = (int)(1 + L / m.hop_length);
    g.out_frames = g.n_frames; g.n_mels = m.n_mels; g.lp = log_params(m); g.mel_long = nullptr; g.mel_short = nullptr;
    const dim3 grid((unsigned)((KP * m.n_mels + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(mel_log_packed_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, xp, T, KP);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(c->ws_melmax, 0, (size_t)B * sizeof(unsigned), (hipStream_t)stream));   // leave the maxima as found
    return KM_OK;
}

int launch_mel(Context* c, MelPlan* p, const float* audio, int64_t B, int64_t L, int64_t out_frames,
               float* mel_long, float* mel_short, void* stream, int64_t clip_len, int64_t win_step, int64_t win0,
               int wins_per_clip) {
    if (int rc = launch_mel_power(c, p, audio, B, L, stream, clip_len, win_step, win0, wins_per_clip)) return rc;
    const km_mel_config& m = p->cfg;
    const int64_t n_frames = 1 + L / m.hop_length;
    LogArgs g;
    g.melpow = c->ws_melpow; g.melmax = c->ws_melmax; g.n_frames = (int)n_frames;
    g.out_frames = (int)(out_frames > 0 ? out_frames : n_frames); g.n_mels = m.n_mels; g.lp = log_params(m);
    g.mel_long = mel_long; g.mel_short = mel_short;
    const dim3 grid2((unsigned)((g.out_frames * m.n_mels + 255) / 256), (unsigned)B);
    if (grid2.x > 0) {
        hipLaunchKernelGGL(mel_log_kernel, grid2, dim3(256), 0, (hipStream_t)stream, g);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemsetAsync(c->ws_melmax, 0, (size_t)B * sizeof(unsigned), (hipStream_t)stream));       // leave the maxima as found
    return KM_OK;
}

// MelAudioBuffer.add_audio_frame for every stream at once (mel_sliding_window.py:70-116): the incoming frame
// is zero-padded / truncated to the ring's own hop (532 by default: the reference derives it from
// update_interval = 0.0333 s), written at the write pointer with wrap-around, and the ring is marked full once
// frames_added * hop >= ring_len.
__global__ void ring_push_kernel(float* __restrict__ ring, int* __restrict__ wptr, int* __restrict__ frames,
                                 unsigned char* __restrict__ ready, const float* __restrict__ samples,
                                 int n_in, int hop, int ring_len) {
    const int s = blockIdx.x;
    const int w0 = wptr[s];
    float* r = ring + (int64_t)s * ring_len;
    const float* in = samples + (int64_t)s * n_in;
    for (int i = threadIdx.x; i < hop; i += blockDim.x) {
        int p = w0 + i;
        p -= p >= ring_len ? ring_len : 0;
        r[p] = i < n_in ? in[i] : 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int w = w0 + hop;
        w -= w >= ring_len ? ring_len : 0;
        wptr[s] = w;
        const int f = frames[s] + 1;
        frames[s] = f;
        if ((int64_t)f * hop >= ring_len) ready[s] = 1;
    }
}

int launch_ring_push(Context* c, const float* samples, int64_t n_per_stream, void* stream) {
    hipLaunchKernelGGL(ring_push_kernel, dim3((unsigned)c->n_streams), dim3(256), 0, (hipStream_t)stream, c->ring,
                       c->ring_wptr, c->ring_frames, c->ring_ready, samples, (int)n_per_stream, c->ring_hop,
                       (int)c->ring_len);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

LogParams plan_log_params(MelPlan* p) { return log_params(p->cfg); }

#ifdef KM_MEL_STAMP
extern "C" __attribute__((visibility("default"))) int km_debug_mel_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(km_mel_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif

}  // namespace km
