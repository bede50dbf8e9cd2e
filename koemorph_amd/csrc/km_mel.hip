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
    float* melpow;        // (B, n_frames, n_mels)
    unsigned* melmax;     // (B) float bits, zero-initialised
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
namespace melrp {
constexpr int WAVES = 8, NT = 512, FPB = 16, NC = 512, FFT_BUF = 576, NFS = 514;
}

// Interior frames (the common case) are 8 unit-stride 8-byte loads per lane from one base pointer.  Returns false --
// a wave-uniform decision -- when the frame touches the padding or straddles the wrap point of a ring.
template <bool RING>
__device__ __forceinline__ bool load_frame_rp_fast(const float* __restrict__ x, int Li, int rs, int hop, int f, int lane,
                                                   float2 (&z)[8]) {
    const int lo = f * hop - 512;
    if (lo < 0 || lo + 1024 > Li) return false;
    int start = lo;
    if constexpr (RING) {
        start += rs;
        start -= start >= Li ? Li : 0;
        if (start + 1024 > Li) return false;
    }
    const float* xp = x + start + 2 * lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = make_float2(xp[128 * i], xp[128 * i + 1]);
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

template <bool RING>
__global__ __launch_bounds__(512, 4) void mel_power_rp_kernel(MelArgs a) {
    using namespace melrp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* fbuf = reinterpret_cast<float2*>(smem);                     // [WAVES][FFT_BUF]
    float* pw = smem + WAVES * FFT_BUF * 2;                              // [FPB][NFS]
    float* outs = pw + FPB * NFS;                                        // [FPB][n_mels]
    float* redmax = outs + FPB * a.n_mels;                               // [WAVES]
    int* fbs = reinterpret_cast<int*>(redmax + WAVES);
    int* fbc = fbs + a.n_mels;
    int* fbo = fbc + a.n_mels;
    float* fbw = reinterpret_cast<float*>(fbo + a.n_mels);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int64_t gw = a.win0 + b;
    const int64_t clip = gw / a.wins_per_clip, woff = (gw - clip * a.wins_per_clip) * a.win_step;
    const float* x = a.audio + clip * a.clip_len + woff;
    const int Lv = (int)((a.clip_len - woff) < a.L ? (a.clip_len - woff) : a.L);
    if (a.ready && !a.ready[b]) return;                        // workgroup-uniform: stream still filling
    const int rs = RING ? a.ring_start[b] : -1;
    float2* buf = fbuf + wave * FFT_BUF;
    const int n_chunks = (a.n_frames + FPB - 1) / FPB;

    for (int i = tid; i < a.n_mels; i += NT) { fbs[i] = a.fb_start[i]; fbc[i] = a.fb_count[i]; fbo[i] = a.fb_offset[i]; }
    for (int i = tid; i < a.fb_nnz; i += NT) fbw[i] = 0.25f * a.fb_weight[i];

    // lane-constant operands (56 registers)
    const int n2 = lane & 7, k0 = lane >> 3;
    float2 win2[8], tw1[8], tw2[8], twp[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        win2[i] = make_float2(a.window[2 * (lane + 64 * i)], a.window[2 * (lane + 64 * i) + 1]);
        tw1[i] = a.twiddle[(2 * lane * bitrev(i, 3)) & 1023];           // W_512^(lane k0)
        tw2[i] = a.twiddle[(16 * n2 * bitrev(i, 3)) & 1023];            // W_64^(n2 k1)
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) twp[i] = a.twiddle[lane + 64 * i];      // W_1024^k, k = lane + 64 i

    float vmax = 0.f;
    for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int f0 = chunk * FPB;
#pragma unroll 1
        for (int pi = 0; pi < FPB / WAVES; ++pi) {
            const int fl = wave + WAVES * pi;
            const int f = f0 + fl;
            if (f < a.n_frames) {   // wave-uniform
                float2 z[8];
                if (!load_frame_rp_fast<RING>(x, Lv, rs, a.hop, f, lane, z)) {      // wave-uniform
                    load_frame_rp_slow(x, Lv, rs, RING ? 1 : 0, a.hop, a.pad_mode == KM_PAD_REFLECT ? 1 : 0, f, lane, buf);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i = 0; i < 8; ++i) z[i] = buf[lane + 64 * i];
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) z[i] = make_float2(z[i].x * win2[i].x, z[i].y * win2[i].y);
                // ---- pass 1: radix-8 over n0 (m = 64 n0 + lane) ----
                dif_fft<8>(z);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kk = bitrev(i, 3);
                    buf[kk * 72 + lane] = kk == 0 ? z[i] : cmul(z[i], tw1[i]);
                }
                __builtin_amdgcn_wave_barrier();
                // ---- pass 2: radix-8 over n1 (lane = 8 n1 + n2); this lane: n2 = lane & 7, k0 = lane >> 3 ----
                float2 y[8];
#pragma unroll
                for (int n1 = 0; n1 < 8; ++n1) y[n1] = buf[k0 * 72 + 8 * n1 + n2];
                __builtin_amdgcn_wave_barrier();
                dif_fft<8>(y);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k1 = bitrev(i, 3);
                    // transposed image indexed by c = k0 + 8 k1, row stride 9 (padding keeps every address base + immediate;
                    // the residual 2-way conflict on 3 of 32 slots is cheaper than per-access swizzle arithmetic)
                    buf[(k0 + 8 * k1) * 9 + n2] = k1 == 0 ? y[i] : cmul(y[i], tw2[i]);
                }
                __builtin_amdgcn_wave_barrier();
                // ---- pass 3: radix-8 over n2; this lane: c = lane; output k = c + 64 k2 ----
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = buf[lane * 9 + q];
                __builtin_amdgcn_wave_barrier();
                dif_fft<8>(y);
#pragma unroll
                for (int i = 0; i < 8; ++i) buf[lane + 64 * bitrev(i, 3)] = y[i];
                __builtin_amdgcn_wave_barrier();
                // ---- real-input post-processing + |.|^2 ----
                float* pwr = pw + fl * NFS;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = lane + 64 * i;
                    const float2 zk = buf[k], zc = buf[(NC - k) & (NC - 1)];
                    const float2 E = make_float2(zk.x + zc.x, zk.y - zc.y);            // 2 E
                    const float2 O = make_float2(zk.y + zc.y, zc.x - zk.x);            // 2 O
                    const float2 Tw = cmul(O, twp[i]);                                  // 2 W^k O
                    const float pr = E.x + Tw.x, pi2 = E.y + Tw.y, mr = E.x - Tw.x, mi = E.y - Tw.y;
                    pwr[k] = pr * pr + pi2 * pi2;                 // 4 |X[k]|^2: the 1/4 lives in the filter weights (exact)
                    pwr[2 * 256 - k] = mr * mr + mi * mi;
                }
                if (lane == 0) {                                                        // k = 256 pairs with itself
                    const float2 zk = buf[256];
                    pwr[256] = 4.0f * (zk.x * zk.x + zk.y * zk.y);
                }
            }
        }
        __syncthreads();
        {   // sparse triangular mel filters, lanes = frames (see mel_power_kernel)
            const int fl = tid & 15, par = (tid >> 4) & 1;
            const bool fvalid = f0 + fl < a.n_frames;
            for (int m = tid >> 5; m < a.n_mels; m += NT / 32) {
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
        for (int i = tid; i < nf_here * a.n_mels; i += NT) dst[i] = outs[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if (lane == 0) redmax[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
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
    p->uploaded = true;
    return KM_OK;
}

void free_mel_plan(MelPlan* p) {
    if (p->uploaded) {
        (void)hipFree(p->d_window); (void)hipFree(p->d_twiddle); (void)hipFree(p->d_fb_start);
        (void)hipFree(p->d_fb_count); (void)hipFree(p->d_fb_offset); (void)hipFree(p->d_fb_weight);
    }
    delete p;
}

static size_t melrp_lds_bytes(int n_mels = 128, int nnz = 0) {
    if (nnz == 0) nnz = 2 * 513 + 128;
    return (size_t)(melrp::WAVES * melrp::FFT_BUF * 2 + melrp::FPB * melrp::NFS + melrp::FPB * n_mels + melrp::WAVES +
                    3 * n_mels + nnz) * sizeof(float);
}

static size_t mel_lds_bytes(int nfft, int n_mels = 128, int nnz = 0) {
    if (nnz == 0) nnz = 2 * (nfft / 2 + 1) + 128;    // upper bound: every bin feeds at most two triangles
    return (size_t)(mel::WAVES * mel::FFT_BUF * 2 + mel::FPB * (nfft / 2 + 2) + mel::FPB * n_mels + mel::WAVES +
                    3 * n_mels + nnz) * sizeof(float);
}

static LogParams log_params(const km_mel_config& m) {
    LogParams lp;
    lp.log_mode = m.log_mode; lp.amin = m.amin; lp.top_db = m.top_db; lp.db_add = m.db_add;
    lp.db_scale = m.db_scale; lp.log_eps = m.log_eps;
    return lp;
}

// power-mel (B, n_frames, n_mels) + per-window max into the workspace
int launch_mel_power(Context* c, MelPlan* p, const float* audio, int64_t B, int64_t L, void* stream,
                     int64_t clip_len, int64_t win_step, int64_t win0, int wins_per_clip, const int* ring_start,
                     const unsigned char* ready) {
    const km_mel_config& m = p->cfg;
    const int64_t n_frames = 1 + L / m.hop_length;
    if (m.pad_mode == KM_PAD_REFLECT && L <= m.n_fft / 2)
        return fail(KM_ERR_INVALID_ARG, "reflect padding needs more than n_fft/2 = %d samples (got %lld)", m.n_fft / 2, (long long)L);
    if (B > c->ws_windows || n_frames > c->ws_frames)
        return fail(KM_ERR_WORKSPACE, "workspace holds %lld windows x %lld frames, need %lld x %lld: call km_reserve",
                    (long long)c->ws_windows, (long long)c->ws_frames, (long long)B, (long long)n_frames);
    if (!p->uploaded) return fail(KM_ERR_NOT_FINALIZED, "mel plan not uploaded (km_finalize / km_reserve first)");
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mel_power_kernel<1024, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)mel_lds_bytes(1024)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mel_power_kernel<512, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)mel_lds_bytes(512)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mel_power_kernel<1024, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)mel_lds_bytes(1024)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mel_power_kernel<512, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)mel_lds_bytes(512)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mel_power_rp_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)melrp_lds_bytes()));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mel_power_rp_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)melrp_lds_bytes()));
        attr_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
    // ws_melmax is all-zero on entry unless a previous non-fused call left its maxima behind
    // (the fused core kernel re-zeroes the entry it consumes)
    if (c->melmax_dirty) {
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
    a.melpow = c->ws_melpow; a.melmax = c->ws_melmax;
    // persistent over frame chunks: each workgroup walks chunks blockIdx.x, +gridDim.x, ... of its window with the
    // next pair's samples prefetched; two workgroups are resident per CU (LDS), so aim at 512 in total
    const int n_chunks = (int)((n_frames + mel::FPB - 1) / mel::FPB);
    int per_window = (int)((512 + B / 2) / B);
    if (per_window > n_chunks) per_window = n_chunks;
    if (per_window < 1) per_window = 1;
    const dim3 grid((unsigned)per_window, (unsigned)B);
    static const bool use_rp = std::getenv("KM_MEL_TWO_FRAME") == nullptr;   // A/B switch: the two-frames-per-wave kernel
    if (m.n_fft == 1024 && use_rp) {
        const size_t ldsrp = melrp_lds_bytes(m.n_mels, a.fb_nnz);
        if (!ring_start) hipLaunchKernelGGL((mel_power_rp_kernel<false>), grid, dim3(melrp::NT), ldsrp, st, a);
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

int launch_mel(Context* c, MelPlan* p, const float* audio, int64_t B, int64_t L, int64_t out_frames,
               float* mel_long, float* mel_short, void* stream, int64_t clip_len, int64_t win_step, int64_t win0,
               int wins_per_clip) {
    if (int rc = launch_mel_power(c, p, audio, B, L, stream, clip_len, win_step, win0, wins_per_clip)) return rc;
    const km_mel_config& m = p->cfg;
    const int64_t n_frames = 1 + L / m.hop_length;
    c->melmax_dirty = true;     // mel_log_kernel leaves the maxima in place
    LogArgs g;
    g.melpow = c->ws_melpow; g.melmax = c->ws_melmax; g.n_frames = (int)n_frames;
    g.out_frames = (int)(out_frames > 0 ? out_frames : n_frames); g.n_mels = m.n_mels; g.lp = log_params(m);
    g.mel_long = mel_long; g.mel_short = mel_short;
    const dim3 grid2((unsigned)((g.out_frames * m.n_mels + 255) / 256), (unsigned)B);
    if (grid2.x > 0) {
        hipLaunchKernelGGL(mel_log_kernel, grid2, dim3(256), 0, (hipStream_t)stream, g);
        HIP_TRY(hipGetLastError());
    }
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

}  // namespace km
