// eGeMAPSv02 functionals (88 values per audio window) on gfx950 -- SURVEY.md section 8 row f-4, second half.
//
// Replaces the openSMILE call of the reference's long-context emotion stream, OpenSMILEeGeMAPSExtractor._extract_features_from_audio
// (src/features/opensmile_extractor.py:427-439: peak normalisation, `self.smile.process_signal`, 88 functionals) for many
// 20 s windows at once.  At BASELINE's streaming configuration (1 024 speaker streams, one update per 0.3 s) that is 3 400
// windows of 320 000 samples per second -- hundreds of CPU cores of openSMILE.
//
// PARITY UNPINNED: openSMILE is a third-party package the reference neither vendors nor pins and that is not installed
// here.  The arithmetic below is the restatement in oracle/egemaps.py (the published GeMAPS / eGeMAPS parameter definitions,
// Eyben et al. 2016, with openSMILE 3.0's documented processing chain); the tests compare this file with that oracle and
// with known answers on synthetic signals.  Every constant is defined once, in EgmPlan (host, double) == oracle/egemaps.py.
//
//   egm_peak_kernel        per window: 1 / max |x|                                   (opensmile_extractor.py:431-433)
//   egm_frame_kernel       per 10 ms frame: 60 ms Gaussian + 20 ms Hamming frames as ONE 1024-point complex FFT in LDS,
//                          spectral descriptors, loudness, MFCC 1-4, sub-harmonic-summation pitch candidates, LPC formants
//   egm_viterbi_kernel     per window: pitch track over the candidates (sequential dynamic programme, 4 states)
//   egm_voiced_kernel      per voiced frame: HNR (autocorrelation), jitter / shimmer (pitch periods marked in the
//                          waveform), harmonic differences and formant amplitudes
//   egm_functional_kernel  per window: 3-frame smoothing, means / normalised deviations / percentiles (bitonic sort in
//                          LDS) / slopes of rising and falling parts over voiced, unvoiced or all frames
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "km_context.h"

namespace km {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(KM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace egm {
constexpr int SR = 16000, HOP = 160, N60 = 960, N20 = 320, NFFT = 1024, NB = NFFT / 2 + 1, OFF20 = (N60 - N20) / 2;
constexpr int NBANDS = 26, LPC = 11, NCAND = 3, PPO = 48, NHARM = 15;
constexpr int REC = 36;                 // floats per frame record (layout below)
constexpr int MAXF = 2048;              // frames per window the functional kernel holds in LDS (20.5 s)
enum Rec { R_LOUD = 0, R_ALPHA, R_HAMM, R_SL0, R_SL1, R_FLUX, R_MFCC, R_RMS = 10, R_CF = 11, R_CS = 14, R_VOI = 17, R_F = 18, R_BW = 21,
           R_F0 = 24, R_JIT, R_SHIM, R_HNR, R_H1H2, R_H1A3, R_FAMP = 30 };
constexpr float VOICING_CUTOFF = 0.55f, RMS_FLOOR = 0.001f;   // on the autocorrelation measure (oracle/egemaps.py acf_strength)
}  // namespace egm

struct EgmPlan {
    // host tables (double precision arithmetic, stored as float), mirrored on the device
    std::vector<float> h;
    float* d = nullptr;
    // offsets into the table blob
    int o_g60, o_ham, o_tw, o_fb_w, o_fb_start, o_fb_count, o_fb_off, o_eql, o_dct, o_logi, o_logf, o_hshift, o_hweight, o_pre, o_cos,
        o_sl0, o_sl1, o_gg;
    int n_log, j0, j1, nb_lpc, b_alpha[4], b_hamm[3], b_sl0[2], b_sl1[2], fb_nnz;
};

static int push(std::vector<float>& h, const std::vector<double>& v) {
    const int o = (int)h.size();
    for (double x : v) h.push_back((float)x);
    while (h.size() % 4) h.push_back(0.f);
    return o;
}
static int pushi(std::vector<float>& h, const std::vector<int>& v) {
    const int o = (int)h.size();
    for (int x : v) { float f; std::memcpy(&f, &x, 4); h.push_back(f); }
    while (h.size() % 4) h.push_back(0.f);
    return o;
}

static EgmPlan* build_egm_plan() {
    using namespace egm;
    EgmPlan* p = new EgmPlan();
    const double PI = 3.14159265358979323846;
    std::vector<double> g60(N60), ham(N20), tw(NFFT);
    for (int n = 0; n < N60; ++n) { const double k = n - (N60 - 1) / 2.0; g60[n] = std::exp(-0.5 * std::pow(k / (0.4 * (N60 - 1) / 2.0), 2)); }
    for (int n = 0; n < N20; ++n) ham[n] = 0.54 - 0.46 * std::cos(2.0 * PI * n / (N20 - 1));
    for (int k = 0; k < NFFT / 2; ++k) { tw[2 * k] = std::cos(-2.0 * PI * k / NFFT); tw[2 * k + 1] = std::sin(-2.0 * PI * k / NFFT); }
    p->o_g60 = push(p->h, g60); p->o_ham = push(p->h, ham); p->o_tw = push(p->h, tw);
    // window autocorrelation for the HNR's window compensation: gg[lag] = sum_n g[n] g[n + lag]
    std::vector<double> gg(N60);
    for (int lag = 0; lag < N60; ++lag) { double s = 0; for (int n = 0; n + lag < N60; ++n) s += g60[n] * g60[n + lag]; gg[lag] = s; }
    p->o_gg = push(p->h, gg);
    // 26 triangular HTK-mel filters, 20 .. 8000 Hz, over the 513 bins (sparse)
    auto mel = [](double f) { return 1127.0 * std::log(1.0 + f / 700.0); };
    auto imel = [](double m) { return 700.0 * (std::exp(m / 1127.0) - 1.0); };
    std::vector<double> edges(NBANDS + 2), w, eql(NBANDS);
    std::vector<int> start(NBANDS), count(NBANDS), off(NBANDS);
    for (int j = 0; j < NBANDS + 2; ++j) edges[j] = imel(mel(20.0) + (mel(8000.0) - mel(20.0)) * j / (NBANDS + 1));
    for (int j = 0; j < NBANDS; ++j) {
        const double l = edges[j], c = edges[j + 1], r = edges[j + 2];
        int s = -1, n = 0;
        off[j] = (int)w.size();
        for (int k = 0; k < NB; ++k) {
            const double f = k * (double)SR / NFFT;
            const double v = std::fmax(0.0, std::fmin((f - l) / (c - l), (r - f) / (r - c)));
            if (v > 0) { if (s < 0) s = k; n = k - s + 1; }
        }
        start[j] = s < 0 ? 0 : s; count[j] = n;
        for (int k = start[j]; k < start[j] + n; ++k) {
            const double f = k * (double)SR / NFFT;
            w.push_back(std::fmax(0.0, std::fmin((f - l) / (c - l), (r - f) / (r - c))));
        }
        const double w2 = std::pow(2.0 * PI * c, 2);
        eql[j] = ((w2 + 56.8e6) * w2 * w2) / (std::pow(w2 + 6.3e6, 2) * (w2 + 0.38e9));
    }
    p->fb_nnz = (int)w.size();
    p->o_fb_w = push(p->h, w); p->o_fb_start = pushi(p->h, start); p->o_fb_count = pushi(p->h, count); p->o_fb_off = pushi(p->h, off);
    p->o_eql = push(p->h, eql);
    std::vector<double> dct(4 * NBANDS);
    for (int i = 1; i <= 4; ++i)
        for (int j = 0; j < NBANDS; ++j)
            dct[(i - 1) * NBANDS + j] = std::sqrt(2.0 / NBANDS) * std::cos(PI * i * (j + 0.5) / NBANDS) * (1.0 + 11.0 * std::sin(PI * i / 22.0));
    p->o_dct = push(p->h, dct);
    // log2-frequency axis: PPO points per octave from 25 Hz to Nyquist, linear interpolation from the FFT bins
    p->n_log = (int)std::floor(std::log2((SR / 2.0) / 25.0) * PPO) + 1;
    p->j0 = (int)std::ceil(std::log2(55.0 / 25.0) * PPO);
    p->j1 = (int)std::floor(std::log2(1000.0 / 25.0) * PPO) + 1;
    std::vector<int> logi(p->n_log); std::vector<double> logf(p->n_log);
    for (int j = 0; j < p->n_log; ++j) {
        const double f = 25.0 * std::pow(2.0, (double)j / PPO), pos = f / ((double)SR / NFFT);
        int i0 = (int)std::floor(pos);
        if (i0 >= NB - 1) i0 = NB - 2;
        logi[j] = i0; logf[j] = pos - i0;
    }
    p->o_logi = pushi(p->h, logi); p->o_logf = push(p->h, logf);
    std::vector<int> hs(NHARM); std::vector<double> hw(NHARM);
    for (int h = 0; h < NHARM; ++h) { hs[h] = (int)std::lround(PPO * std::log2((double)(h + 1))); hw[h] = std::pow(0.85, h); }
    p->o_hshift = pushi(p->h, hs); p->o_hweight = push(p->h, hw);
    // LPC on the spectrum below 5.5 kHz: pre-emphasis and trapezoid weights folded together, cosine table for 12 lags
    p->nb_lpc = (int)(5500.0 / ((double)SR / NFFT)) + 1;
    std::vector<double> pre(p->nb_lpc), cs((size_t)p->nb_lpc * (LPC + 1));
    for (int k = 0; k < p->nb_lpc; ++k) {
        const double wv = PI * k / (p->nb_lpc - 1);
        pre[k] = (1.0 + 0.97 * 0.97 - 2.0 * 0.97 * std::cos(wv)) * ((k == 0 || k == p->nb_lpc - 1) ? 0.5 : 1.0);
        for (int l = 0; l <= LPC; ++l) cs[(size_t)l * p->nb_lpc + k] = std::cos(wv * l);
    }
    p->o_pre = push(p->h, pre); p->o_cos = push(p->h, cs);
    // band edges as bin ranges [lo, hi): f >= lo_hz && f < hi_hz
    auto bin_ge = [](double hz) { return (int)std::ceil(hz / ((double)SR / NFFT) - 1e-9); };
    p->b_alpha[0] = bin_ge(50); p->b_alpha[1] = bin_ge(1000); p->b_alpha[2] = bin_ge(1000); p->b_alpha[3] = bin_ge(5000);
    p->b_hamm[0] = 0; p->b_hamm[1] = bin_ge(2000); p->b_hamm[2] = bin_ge(5000);
    auto regress = [&](double lo, double hi, int (&b)[2]) {
        b[0] = bin_ge(lo); b[1] = bin_ge(hi);
        std::vector<double> fx(b[1] - b[0]);
        double m = 0, ss = 0;
        for (int k = b[0]; k < b[1]; ++k) m += k * (double)SR / NFFT;
        m /= (b[1] - b[0]);
        for (int k = b[0]; k < b[1]; ++k) { fx[k - b[0]] = k * (double)SR / NFFT - m; ss += fx[k - b[0]] * fx[k - b[0]]; }
        for (auto& v : fx) v /= ss;
        return fx;
    };
    p->o_sl0 = push(p->h, regress(0, 500, p->b_sl0));
    p->o_sl1 = push(p->h, regress(500, 1500, p->b_sl1));
    return p;
}

struct EgmArgs {
    const float* audio; int64_t L; int nf;
    const float* scale;      // (B) peak normalisation factors, or null
    const float* tab;        // EgmPlan table blob
    float* rec;              // (B, nf, REC)
    int o_g60, o_ham, o_tw, o_fb_w, o_fb_start, o_fb_count, o_fb_off, o_eql, o_dct, o_logi, o_logf, o_hshift, o_hweight, o_pre, o_cos,
        o_sl0, o_sl1, o_gg;
    int n_log, j0, j1, nb_lpc, b_alpha[4], b_hamm[3], b_sl0[2], b_sl1[2];
};

__global__ __launch_bounds__(256) void egm_peak_kernel(const float* __restrict__ x, int64_t L, float* __restrict__ scale) {
    __shared__ float red[256];
    const float* p = x + (int64_t)blockIdx.x * L;
    float m = 0.f;
    for (int64_t i = threadIdx.x; i < L; i += 256) m = fmaxf(m, fabsf(p[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) scale[blockIdx.x] = red[0] > 0.f ? 1.0f / red[0] : 1.0f;
}

// block-wide reductions (256 threads = 4 waves), fixed order: an xor butterfly inside each wave, then the four wave results
// in wave order through 4 floats of LDS scratch -- two barriers instead of the ten of a 256-element LDS tree
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// in-place 1024-point complex FFT of buf (interleaved re, im) in LDS: bit-reversed load order is the caller's; radix-2 DIT
__device__ __forceinline__ void fft1024(float2* buf, const float2* tw) {
    const int tid = threadIdx.x;
    for (int s = 0; s < 10; ++s) {
        const int half = 1 << s;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = tid + 256 * r;
            const int j = i & (half - 1);
            const int base = ((i >> s) << (s + 1)) + j;
            const float2 w = tw[j << (9 - s)];
            const float2 a = buf[base], b = buf[base + half];
            const float2 bw = make_float2(b.x * w.x - b.y * w.y, b.x * w.y + b.y * w.x);
            buf[base] = make_float2(a.x + bw.x, a.y + bw.y);
            buf[base + half] = make_float2(a.x - bw.x, a.y - bw.y);
        }
        __syncthreads();
    }
}
__device__ __forceinline__ int bitrev10(int v) { return (int)(__brev((unsigned)v) >> 22); }

// window-compensated autocorrelation of the Gaussian-windowed frame sw[960] at lags lag0 .. lag0 + nl - 1 (nl <= 128), one lag
// per thread, into r[]: r = sum_n sw[n] sw[n + lag] / gg[lag] * gg[0]
__device__ __forceinline__ void acf_lags(const float* sw, const float* gg, int lag0, int nl, float* r) {
    using namespace egm;
    const int i = threadIdx.x;
    if (i < nl) {
        const int lag = lag0 + i;
        float acc = 0.f;
        for (int n = 0; n + lag < N60; ++n) acc = fmaf(sw[n], sw[n + lag], acc);
        r[i] = acc / fmaxf(gg[lag], 1e-12f) * gg[0];
    }
    __syncthreads();
}

// 60 ms Gaussian frame (real part) and, optionally, a 20 ms Hamming frame (imaginary part) -> 1024-point spectra.
// zb is loaded in bit-reversed order; after the FFT M60[k] = |(Z[k] + conj Z[N-k]) / 2|, M20[k] = |(Z[k] - conj Z[N-k]) / 2i|.
__device__ __forceinline__ void frame_spectra(const EgmArgs& a, const float* xw, float sc, int start60, int start20, bool with20,
                                              float2* zb, const float2* tw, float* M60, float* M20) {
    using namespace egm;
    const float* g60 = a.tab + a.o_g60; const float* ham = a.tab + a.o_ham;
    for (int n = threadIdx.x; n < NFFT; n += 256) {
        float re = 0.f, im = 0.f;
        if (n < N60) re = g60[n] * xw[start60 + n] * sc;
        if (with20 && n < N20) im = ham[n] * xw[start20 + n] * sc;
        zb[bitrev10(n)] = make_float2(re, im);
    }
    __syncthreads();
    fft1024(zb, tw);
    for (int k = threadIdx.x; k < NB; k += 256) {
        const float2 z = zb[k], zc = zb[(NFFT - k) & (NFFT - 1)];
        const float ar = 0.5f * (z.x + zc.x), ai = 0.5f * (z.y - zc.y);          // (Z[k] + conj Z[N-k]) / 2
        const float br = 0.5f * (z.y + zc.y), bi = -0.5f * (z.x - zc.x);         // (Z[k] - conj Z[N-k]) / (2i)
        M60[k] = sqrtf(ar * ar + ai * ai);
        if (M20) M20[k] = sqrtf(br * br + bi * bi);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256, 6) void egm_frame_kernel(EgmArgs a) {
    using namespace egm;
    __shared__ float2 zb[NFFT];
    __shared__ float2 tw[NFFT / 2];
    __shared__ float M60[NB + 3], M20[NB + 3], Mp[NB + 3], red[256], E[32], slog[512], shs[256], misc[64], sw[N60], rl[128];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* xw = a.audio + (int64_t)b * a.L;
    const float sc = a.scale ? a.scale[b] : 1.0f;
    float* rec = a.rec + ((int64_t)b * a.nf + t) * REC;
    for (int k = tid; k < NFFT / 2; k += 256) tw[k] = reinterpret_cast<const float2*>(a.tab + a.o_tw)[k];
    __syncthreads();
    // previous frame's 20 ms spectrum (for the spectral flux): the imaginary slot of a transform whose real slot is unused
    if (t > 0) {
        const float* ham = a.tab + a.o_ham;
        for (int n = tid; n < NFFT; n += 256) zb[bitrev10(n)] = make_float2(n < N20 ? ham[n] * xw[HOP * (t - 1) + OFF20 + n] * sc : 0.f, 0.f);
        __syncthreads();
        fft1024(zb, tw);
        for (int k = tid; k < NB; k += 256) Mp[k] = sqrtf(zb[k].x * zb[k].x + zb[k].y * zb[k].y);
        __syncthreads();
    }
    frame_spectra(a, xw, sc, HOP * t, HOP * t + OFF20, true, zb, tw, M60, M20);

    // ---- frame energy; the windowed frame stays in LDS for the voicing measure ----
    float e2 = 0.f;
    for (int n = tid; n < N60; n += 256) { const float v = xw[HOP * t + n] * sc; e2 += v * v; sw[n] = v * a.tab[a.o_g60 + n]; }
    const float rms = sqrtf(block_sum(e2, red) / N60);

    // ---- band powers -> loudness, MFCC 1-4 ----
    if (tid < NBANDS) {
        const int s0 = __float_as_int(a.tab[a.o_fb_start + tid]), cnt = __float_as_int(a.tab[a.o_fb_count + tid]);
        const float* w = a.tab + a.o_fb_w + __float_as_int(a.tab[a.o_fb_off + tid]);
        float acc = 0.f;
        for (int k = 0; k < cnt; ++k) acc = fmaf(M20[s0 + k] * M20[s0 + k], w[k], acc);
        E[tid] = acc;
    }
    __syncthreads();
    if (tid < 64) {          // wave 0, lane j = band j: the 26 powers / logarithms in parallel, sums by butterfly
        const bool on = tid < NBANDS;
        const float ej = on ? E[tid] : 0.f;
        const float loud = wave_sum(on ? powf(ej * a.tab[a.o_eql + tid], 0.33f) : 0.f);
        const float le = on ? logf(fmaxf(ej, 1e-8f)) : 0.f;
        float c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = wave_sum(on ? le * a.tab[a.o_dct + i * NBANDS + tid] : 0.f);
        if (tid == 0) {
            rec[R_LOUD] = loud;
#pragma unroll
            for (int i = 0; i < 4; ++i) rec[R_MFCC + i] = c[i];
            rec[R_RMS] = rms;
        }
    }
    // ---- alpha ratio, Hammarberg index, slopes, flux ----
    {
        float lo = 0.f, hi = 0.f, pl = 0.f, ph = 0.f, s0 = 0.f, s1 = 0.f, fl = 0.f;
        for (int k = tid; k < NB; k += 256) {
            const float pw = M20[k] * M20[k];
            if (k >= a.b_alpha[0] && k < a.b_alpha[1]) lo += pw;
            if (k >= a.b_alpha[2] && k < a.b_alpha[3]) hi += pw;
            if (k >= a.b_hamm[0] && k < a.b_hamm[1]) pl = fmaxf(pl, pw);
            if (k >= a.b_hamm[1] && k < a.b_hamm[2]) ph = fmaxf(ph, pw);
            const float ldb = 10.0f * log10f(pw + 1e-12f);
            if (k >= a.b_sl0[0] && k < a.b_sl0[1]) s0 = fmaf(ldb, a.tab[a.o_sl0 + k - a.b_sl0[0]], s0);
            if (k >= a.b_sl1[0] && k < a.b_sl1[1]) s1 = fmaf(ldb, a.tab[a.o_sl1 + k - a.b_sl1[0]], s1);
            if (t > 0) { const float dm = M20[k] - Mp[k]; fl += dm * dm; }
        }
        const float LO = block_sum(lo, red), HI = block_sum(hi, red), PL = block_max(pl, red), PH = block_max(ph, red);
        const float S0 = block_sum(s0, red), S1 = block_sum(s1, red), FL = block_sum(fl, red);
        if (tid == 0) {
            rec[R_ALPHA] = 10.0f * log10f((LO + 1e-12f) / (HI + 1e-12f));
            rec[R_HAMM] = 10.0f * log10f((PL + 1e-12f) / (PH + 1e-12f));
            rec[R_SL0] = S0; rec[R_SL1] = S1;
            rec[R_FLUX] = t > 0 ? sqrtf(FL / NB) : 0.f;
        }
    }
    // ---- sub-harmonic summation on the log2-frequency axis ----
    for (int j = tid; j < a.n_log; j += 256) {
        const int i0 = __float_as_int(a.tab[a.o_logi + j]); const float fr = a.tab[a.o_logf + j];
        slog[j] = M60[i0] * (1.0f - fr) + M60[i0 + 1] * fr;
    }
    __syncthreads();
    float sm[2] = {0.f, 0.f};
    for (int r = 0; r < 2; ++r) {
        const int j = tid + 256 * r;
        if (j < a.n_log) sm[r] = (j == 0 || j == a.n_log - 1) ? slog[j] : 0.25f * slog[j - 1] + 0.5f * slog[j] + 0.25f * slog[j + 1];
    }
    __syncthreads();
    for (int r = 0; r < 2; ++r) { const int j = tid + 256 * r; if (j < a.n_log) slog[j] = sm[r]; }
    __syncthreads();
    const int nshs = a.j1 - a.j0;
    float sv = 0.f;
    if (tid < nshs) {
        for (int h = 0; h < NHARM; ++h) {
            const int idx = a.j0 + tid + __float_as_int(a.tab[a.o_hshift + h]);
            if (idx < a.n_log) sv = fmaf(a.tab[a.o_hweight + h], slog[idx], sv);
        }
        shs[tid] = sv;
    }
    const float peak = block_max(tid < nshs ? sv : 0.f, red);
    const float mean = block_sum(tid < nshs ? sv : 0.f, red) / nshs;
    (void)mean;
    // greedy candidates: highest point (first index on ties), then blank +- 1/6 octave around it -- one block arg-max per
    // candidate (the value first, then the lowest index that reaches it) instead of thread 0 scanning the curve NCAND times
    {
        int kprev[NCAND];
#pragma unroll
        for (int c = 0; c < NCAND; ++c) {
            bool blanked = tid >= nshs;
#pragma unroll
            for (int q = 0; q < c; ++q) if (kprev[q] >= 0 && tid >= kprev[q] - PPO / 6 && tid <= kprev[q] + PPO / 6) blanked = true;
            const float v = blanked ? 0.f : sv;
            const float best = block_max(v, red);
            const float ki = block_max((best > 0.f && !blanked && v == best) ? (float)(nshs - tid) : 0.f, red);   // first index wins
            const int k = best > 0.f ? nshs - (int)ki : -1;
            kprev[c] = k;
            if (tid == 0) {
                if (k < 0) { rec[R_CF + c] = 0.f; rec[R_CS + c] = 0.f; misc[c] = -1000.f; }
                else {
                    misc[c] = (float)k;
                    float dlt = 0.f;
                    if (k > 0 && k < nshs - 1) {
                        const float y0 = shs[k - 1], y1 = shs[k], y2 = shs[k + 1], den = y0 - 2.f * y1 + y2;
                        dlt = den == 0.f ? 0.f : fminf(fmaxf(0.5f * (y0 - y2) / den, -0.5f), 0.5f);
                    }
                    rec[R_CF + c] = 25.0f * exp2f((a.j0 + k + dlt) / (float)PPO);
                    rec[R_CS + c] = shs[k] / peak;
                    if (c == 0) misc[4] = rec[R_CF];
                }
            }
        }
        if (tid == 0 && kprev[0] < 0) misc[4] = 0.f;
    }
    __syncthreads();
    // ---- voicing: normalised autocorrelation at the strongest candidate's lag (+- 10 %), only if it is a local maximum ----
    {
        const float cf0 = misc[4];
        float vo = 0.f;
        if (cf0 > 0.f) {                                   // uniform
            const float T0 = (float)SR / cf0;
            int lo = (int)floorf(0.9f * T0), hi = (int)ceilf(1.1f * T0);
            lo = lo < 1 ? 1 : lo; hi = hi > N60 - 2 ? N60 - 2 : hi;
            const int nl = hi - lo + 3;                    // lags lo - 1 .. hi + 1
            acf_lags(sw, a.tab + a.o_gg, lo - 1, nl < 128 ? nl : 128, rl);
            float r0p = 0.f;
            for (int n = tid; n < N60; n += 256) r0p += sw[n] * sw[n];
            const float r0 = block_sum(r0p, red);
            if (tid == 0 && r0 > 0.f) {
                const int ne = nl < 128 ? nl : 128;
                int k = 1;
                for (int q = 2; q < ne - 1; ++q) if (rl[q] > rl[k]) k = q;
                if (rl[k] >= rl[k - 1] && rl[k] >= rl[k + 1]) vo = fminf(fmaxf(rl[k] / r0, 0.f), 1.f);
            }
        }
        if (tid == 0) rec[R_VOI] = vo;
    }
    // ---- LPC formants from the 20 ms power spectrum below 5.5 kHz ----
    {
        float* R = misc + 8;       // [12]
        for (int l = 0; l <= LPC; ++l) {
            float acc = 0.f;
            for (int k = tid; k < a.nb_lpc; k += 256) acc = fmaf(M20[k] * M20[k] * a.tab[a.o_pre + k], a.tab[a.o_cos + l * a.nb_lpc + k], acc);
            const float v = block_sum(acc, red);
            if (tid == 0) R[l] = v;
        }
        __syncthreads();
        if (tid < 64) {     // wave 0: Levinson-Durbin redundantly in every lane (uniform), then ONE ROOT PER LANE
            float F[3] = {0.f, 0.f, 0.f}, BW[3] = {0.f, 0.f, 0.f};
            float* fsel = misc + 24;   // [LPC] candidate frequencies, [LPC] bandwidths behind them
            float* bsel = fsel + LPC;
            if (R[0] > 0.f) {           // uniform
                float ac[LPC + 1], tmp[LPC + 1];
                ac[0] = 1.f;
#pragma unroll
                for (int i = 1; i <= LPC; ++i) ac[i] = 0.f;
                float err = R[0];
                bool ok = true;
#pragma unroll
                for (int i = 1; i <= LPC; ++i) {                            // Levinson-Durbin
                    if (!ok) continue;
                    float acc = R[i];
#pragma unroll
                    for (int j = 1; j < i; ++j) acc = fmaf(ac[j], R[i - j], acc);
                    const float kk = -acc / err;
#pragma unroll
                    for (int j = 1; j < i; ++j) tmp[j] = ac[j] + kk * ac[i - j];
#pragma unroll
                    for (int j = 1; j < i; ++j) ac[j] = tmp[j];
                    ac[i] = kk;
                    err *= 1.0f - kk * kk;
                    if (err <= 0.f) ok = false;
                }
                // roots of z^11 + ac[1] z^10 + ... + ac[11] by Durand-Kerner (Weierstrass) iteration, lane i = root i, all
                // roots updated together from the previous iterate (the serial version updated them one after another on
                // thread 0 and was 85 % of this kernel's time); the partners' values come by shuffle
                const int li = tid < LPC ? tid : 0;
                const float ang = 2.0f * 3.14159265f * (li + 0.35f) / LPC;
                float zr = 0.9f * cosf(ang), zi = 0.9f * sinf(ang);
                for (int it = 0; it < 60; ++it) {
                    float pr = 1.f, pi = 0.f;                               // Horner: p(z_i)
#pragma unroll
                    for (int j = 1; j <= LPC; ++j) { const float nr = pr * zr - pi * zi + ac[j], ni = pr * zi + pi * zr; pr = nr; pi = ni; }
                    float qr = 1.f, qi = 0.f;                               // prod_{j != i} (z_i - z_j)
#pragma unroll
                    for (int j = 0; j < LPC; ++j) {
                        const float zjr = __shfl(zr, j), zji = __shfl(zi, j);
                        const float dr = zr - zjr, di = zi - zji;
                        const float nr = qr * dr - qi * di, ni = qr * di + qi * dr;
                        if (j != li) { qr = nr; qi = ni; }
                    }
                    const float den = qr * qr + qi * qi + 1e-30f;
                    const float cr = (pr * qr + pi * qi) / den, ci = (pi * qr - pr * qi) / den;
                    zr -= cr; zi -= ci;
                    const float change = wave_max(tid < LPC ? fabsf(cr) + fabsf(ci) : 0.f);
                    if (change < 1e-7f) break;                              // uniform
                }
                const float fs2 = 11000.0f;
                float fr = -1.f, bw = 0.f;
                if (tid < LPC && zi > 1e-6f) {
                    const float f = atan2f(zi, zr) * fs2 / (2.0f * 3.14159265f);
                    const float bq = -logf(fmaxf(sqrtf(zr * zr + zi * zi), 1e-12f)) * fs2 / 3.14159265f;
                    if (f > 90.f && f < 5400.f && bq < 1000.f) { fr = f; bw = bq; }
                }
                if (tid < LPC) { fsel[tid] = fr; bsel[tid] = bw; }
                __builtin_amdgcn_wave_barrier();
                if (tid == 0) {
                    for (int c = 0; c < 3; ++c) {                           // three lowest
                        int best = -1;
                        for (int i = 0; i < LPC; ++i) if (fsel[i] > 0.f && (best < 0 || fsel[i] < fsel[best])) best = i;
                        if (best < 0) break;
                        F[c] = fsel[best]; BW[c] = bsel[best]; fsel[best] = -1.f;
                    }
                }
            }
            if (tid == 0)
                for (int c = 0; c < 3; ++c) { rec[R_F + c] = F[c]; rec[R_BW + c] = BW[c]; }
        }
    }
}

// pitch track over the candidates: states 0..2 = candidate, 3 = unvoiced.  One thread per window walks the frames; the
// back pointers live in the record's Famp slots until egm_voiced_kernel overwrites them.
__global__ __launch_bounds__(256) void egm_viterbi_kernel(float* __restrict__ recs, int nf) {
    using namespace egm;
    // The recursion itself is sequential (thread 0), but its inputs are not: all threads first gather the seven numbers a
    // step needs -- voicing flag, log2 of the three candidate frequencies, their strengths -- into LDS, so that the walk
    // reads LDS instead of paying a global round trip per frame (2.7 ms -> 0.2 ms per call), and write the chosen F0 back
    // together at the end.  Same arithmetic in the same order as before.
    __shared__ float in[7 * MAXF];            // [0] voiced-ok, [1..3] log2 f (or -1: no candidate), [4..6] strength
    __shared__ unsigned char bp[MAXF];        // back pointers (two bits per state), then the chosen state
    float* rec = recs + (int64_t)blockIdx.x * nf * REC;
    for (int t = threadIdx.x; t < nf; t += 256) {
        const float* r = rec + (int64_t)t * REC;
        in[t] = (r[R_VOI] >= VOICING_CUTOFF && r[R_RMS] >= RMS_FLOOR) ? 1.f : 0.f;
        for (int c = 0; c < 3; ++c) {
            const float f = r[R_CF + c];
            in[(1 + c) * MAXF + t] = f > 0.f ? log2f(f) : -1.f;       // candidates are >= 25 Hz: log2 > 0
            in[(4 + c) * MAXF + t] = r[R_CS + c];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float w_local = 2.0f, w_vv = 10.0f, w_vuv = 10.0f / 8.0f, w_thr = 4.0f, INF = 1e30f;
        float cost[4], lf_prev[3] = {0.f, 0.f, 0.f};
        for (int t = 0; t < nf; ++t) {
            const bool vok = in[t] != 0.f;
            float loc[4], lf[3];
            for (int c = 0; c < 3; ++c) {
                const float l2 = in[(1 + c) * MAXF + t];
                const bool has = l2 >= 0.f;
                lf[c] = has ? l2 : 0.f;
                loc[c] = has ? w_local * (1.0f - in[(4 + c) * MAXF + t]) + (vok ? 0.f : w_thr) : INF;
            }
            loc[3] = vok ? w_thr : 0.f;
            float nc[4];
            int packed = 0;                                   // back pointers: two bits per state
            for (int s = 0; s < 4; ++s) {
                if (t == 0) { nc[s] = loc[s]; continue; }
                if (loc[s] >= INF) { nc[s] = INF; continue; }
                float best = INF; int arg = 0;
                for (int p = 0; p < 4; ++p) {
                    if (cost[p] >= INF) continue;
                    const float tr = (s < 3 && p < 3) ? w_vv * fabsf(lf[s] - lf_prev[p]) : ((s == 3 && p == 3) ? 0.f : w_vuv);
                    const float v = cost[p] + tr;
                    if (v < best) { best = v; arg = p; }
                }
                nc[s] = best + loc[s];
                packed |= arg << (2 * s);
            }
            bp[t] = (unsigned char)packed;
            for (int s = 0; s < 4; ++s) cost[s] = nc[s];
            for (int c = 0; c < 3; ++c) lf_prev[c] = lf[c];
        }
        int s = 0;
        for (int q = 1; q < 4; ++q) if (cost[q] < cost[s]) s = q;
        for (int t = nf - 1; t >= 0; --t) {
            const int packed = bp[t];
            bp[t] = (unsigned char)s;                         // the state chosen for frame t
            s = (packed >> (2 * s)) & 3;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nf; t += 256) {
        float* r = rec + (int64_t)t * REC;
        const int st = bp[t];
        r[R_F0] = st < 3 ? r[R_CF + st] : 0.f;
    }
}

__global__ __launch_bounds__(256) void egm_voiced_kernel(EgmArgs a) {
    using namespace egm;
    __shared__ float2 zb[NFFT];
    __shared__ float2 tw[NFFT / 2];
    __shared__ float M60[NB + 3], seg[N60], red[256], cc[128];
    __shared__ int redi[256], marks[64];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* xw = a.audio + (int64_t)b * a.L;
    const float sc = a.scale ? a.scale[b] : 1.0f;
    float* rec = a.rec + ((int64_t)b * a.nf + t) * REC;
    const float f0 = rec[R_F0];
    __syncthreads();                                     // everyone has read the record before thread 0 rewrites parts of it
    if (f0 <= 0.f) {
        if (tid == 0) {
            rec[R_JIT] = rec[R_SHIM] = rec[R_HNR] = rec[R_H1H2] = rec[R_H1A3] = 0.f;
            for (int c = 0; c < 3; ++c) { rec[R_FAMP + c] = 0.f; rec[R_F + c] = 0.f; rec[R_BW + c] = 0.f; }   // formants: voiced frames only
        }
        return;
    }
    for (int k = tid; k < NFFT / 2; k += 256) tw[k] = reinterpret_cast<const float2*>(a.tab + a.o_tw)[k];
    for (int n = tid; n < N60; n += 256) seg[n] = xw[HOP * t + n] * sc;
    __syncthreads();
    frame_spectra(a, xw, sc, HOP * t, 0, false, zb, tw, M60, nullptr);
    const float T0 = (float)SR / f0;
    int lo = (int)floorf(0.9f * T0), hi = (int)ceilf(1.1f * T0);
    if (hi > N60 - 2) hi = N60 - 2;
    const float* g60 = a.tab + a.o_g60; const float* gg = a.tab + a.o_gg;
    // ---- HNR from the window-compensated autocorrelation at the pitch lag ----
    float r0p = 0.f;
    for (int n = tid; n < N60; n += 256) { const float v = seg[n] * g60[n]; r0p += v * v; }
    const float r0 = block_sum(r0p, red);
    float best = 0.f;
    {
        float* swv = reinterpret_cast<float*>(zb);        // the FFT buffer is idle from here on (occupancy: LDS per workgroup)
        for (int n = tid; n < N60; n += 256) swv[n] = seg[n] * g60[n];
        __syncthreads();
        const int lag0 = lo > 1 ? lo : 1;
        const int nl = hi - lag0 + 1 < 128 ? hi - lag0 + 1 : 128;
        acf_lags(swv, gg, lag0, nl, cc);
        for (int q = 0; q < nl; ++q) best = fmaxf(best, cc[q]);
        __syncthreads();
    }
    // ---- pitch periods: one candidate lag per thread ----
    {
        float mv = -INFINITY; int mi = 0;
        for (int n = tid; n < N60; n += 256) if (seg[n] > mv) { mv = seg[n]; mi = n; }
        red[tid] = mv; redi[tid] = mi;
        __syncthreads();
        if (tid == 0) {
            float bv = red[0]; int bi = redi[0];
            for (int k = 1; k < 256; ++k) if (red[k] > bv || (red[k] == bv && redi[k] < bi)) { bv = red[k]; bi = redi[k]; }
            const int T = (int)roundf(T0);
            marks[0] = bi >= T ? bi % T : bi;
            marks[63] = 1;                                // number of marks
        }
        __syncthreads();
        const int lagA = lo > 2 ? lo : 2;
        for (int step = 0; step < 60; ++step) {
            const int pos = marks[marks[63] - 1];
            const int lag = lagA + tid;
            float c = -INFINITY;
            if (lag <= hi && pos + 2 * lag <= N60) {
                float ab = 0.f, aa = 0.f, bb = 0.f;
                for (int n = 0; n < lag; ++n) { const float u = seg[pos + n], v = seg[pos + lag + n]; ab = fmaf(u, v, ab); aa = fmaf(u, u, aa); bb = fmaf(v, v, bb); }
                c = ab / sqrtf(fmaxf(aa * bb, 1e-20f));
            }
            // strongest lag, first one on ties: block arg-max (value, then the lowest index reaching it) instead of a scan
            const int nl = hi - lagA + 1 < 128 ? hi - lagA + 1 : 128;
            const bool in = tid < nl;
            const float bc = block_max(in ? c : -INFINITY, red);
            const float kf = block_max((in && c == bc && bc > -INFINITY) ? (float)(128 - tid) : 0.f, red);
            if (tid == 0) {
                const int bl = kf > 0.f ? lagA + (128 - (int)kf) : 0;
                if (bl == 0 || bc < 0.5f || marks[63] >= 62) marks[62] = 0;          // stop
                else { marks[marks[63]] = pos + bl; marks[63] += 1; marks[62] = 1; }
            }
            __syncthreads();
            if (marks[62] == 0) break;
        }
    }
    if (tid == 0) {
        const float rr = fminf(fmaxf(best / fmaxf(r0, 1e-20f), 1e-6f), 1.0f - 1e-6f);
        rec[R_HNR] = 10.0f * log10f(rr / (1.0f - rr));
        const int nm = marks[63];
        float jit = 0.f, shim = 0.f;
        if (nm >= 3) {
            float psum = 0.f, dsum = 0.f, prev_amp = 0.f, ssum = 0.f;
            for (int i = 0; i + 1 < nm; ++i) {
                const float per = (float)(marks[i + 1] - marks[i]);
                psum += per;
                if (i > 0) dsum += fabsf(per - (float)(marks[i] - marks[i - 1]));
                float mx = -INFINITY, mn = INFINITY;
                for (int n = marks[i]; n < marks[i + 1]; ++n) { mx = fmaxf(mx, seg[n]); mn = fminf(mn, seg[n]); }
                const float amp = fmaxf(mx - mn, 1e-9f);
                if (i > 0) ssum += fabsf(20.0f * log10f(amp / prev_amp));
                prev_amp = amp;
            }
            jit = (dsum / (nm - 2)) / (psum / (nm - 1));
            shim = ssum / (nm - 2);
        }
        rec[R_JIT] = jit; rec[R_SHIM] = shim;
        // harmonic amplitudes: strongest bin within +- 20 % of f0 around the target frequency, in dB
        const float binw = (float)SR / NFFT;
        auto harm_db = [&](float freq) {
            int l = (int)floorf((freq - 0.2f * f0) / binw), h = (int)ceilf((freq + 0.2f * f0) / binw);
            l = l < 0 ? 0 : l; h = h > NB - 1 ? NB - 1 : h;
            float m = 0.f;
            for (int k = l; k <= h; ++k) m = fmaxf(m, M60[k]);
            return 20.0f * log10f(fmaxf(m, 1e-12f));
        };
        const float h1 = harm_db(f0), h2 = harm_db(2.0f * f0);
        rec[R_H1H2] = h1 - h2;
        float h1a3 = 0.f;
        for (int c = 0; c < 3; ++c) {
            const float F = rec[R_F + c];
            float amp = 0.f;
            if (F > 0.f) {
                const float av = harm_db(fmaxf(f0, rintf(F / f0) * f0));
                amp = av - h1;
                if (c == 2) h1a3 = h1 - av;
            }
            rec[R_FAMP + c] = amp;
        }
        rec[R_H1A3] = h1a3;
    }
}

// ---- functionals: one workgroup per window ---------------------------------------------------------------------------
struct FuncScratch {
    float* v;       // [MAXF] contour (smoothed)
    float* s;       // [MAXF] sort buffer
    float* red;     // [256]
};

__device__ __forceinline__ void bitonic_sort(float* s, int n2) {      // ascending, n2 a power of two, padding = +inf
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool up = (i & k) == 0;
                    const float x = s[i], y = s[ixj];
                    if ((x > y) == up) { s[i] = y; s[ixj] = x; }
                }
            }
            __syncthreads();
        }
}

__global__ __launch_bounds__(256) void egm_functional_kernel(const float* __restrict__ recs, int nf, float* __restrict__ out) {
    using namespace egm;
    __shared__ float v[MAXF], s[MAXF], f0s[MAXF], red[256], res[16];
    __shared__ int cnt_s[4];
    const int tid = threadIdx.x;
    const float* rec = recs + (int64_t)blockIdx.x * nf * REC;
    float* o = out + (int64_t)blockIdx.x * 88;
    auto raw = [&](int t, int field) { return rec[(int64_t)t * REC + field]; };
    // smoothed F0 (non-zero smoothing) defines the voiced frames
    for (int t = tid; t < nf; t += 256) {
        const float c = raw(t, R_F0);
        float acc = 0.f; int n = 0;
        if (c != 0.f)
            for (int q = (t > 0 ? t - 1 : 0); q <= (t + 1 < nf ? t + 1 : nf - 1); ++q) { const float x = raw(q, R_F0); if (x != 0.f) { acc += x; ++n; } }
        f0s[t] = n ? acc / n : 0.f;
    }
    __syncthreads();
    // load a contour: mode 0 plain 3-frame average, 1 non-zero average of the values masked to voiced frames, 2 semitone of f0s
    auto load = [&](int field, int mode) {
        for (int t = tid; t < nf; t += 256) {
            float r = 0.f;
            if (mode == 2) r = f0s[t] > 0.f ? 12.0f * log2f(f0s[t] / 27.5f) : 0.f;
            else if (mode == 0) {
                float acc = 0.f; int n = 0;
                for (int q = (t > 0 ? t - 1 : 0); q <= (t + 1 < nf ? t + 1 : nf - 1); ++q) { acc += raw(q, field); ++n; }
                r = acc / n;
            } else {
                const float c = f0s[t] > 0.f ? raw(t, field) : 0.f;
                if (c != 0.f) {
                    float acc = 0.f; int n = 0;
                    for (int q = (t > 0 ? t - 1 : 0); q <= (t + 1 < nf ? t + 1 : nf - 1); ++q) {
                        const float x = f0s[q] > 0.f ? raw(q, field) : 0.f;
                        if (x != 0.f) { acc += x; ++n; }
                    }
                    r = acc / n;
                }
            }
            v[t] = r;
        }
        __syncthreads();
    };
    // mean and normalised standard deviation over frames selected by sel: 0 all, 1 voiced, 2 unvoiced
    auto mean_sn = [&](int sel, float& m, float& sn) {
        float a0 = 0.f; int n0 = 0;
        for (int t = tid; t < nf; t += 256) { const bool vo = f0s[t] > 0.f; if (sel == 0 || (sel == 1) == vo) { a0 += v[t]; ++n0; } }
        const float sum = block_sum(a0, red);
        const int n = (int)block_sum((float)n0, red);
        m = n ? sum / n : 0.f;
        float a1 = 0.f;
        for (int t = tid; t < nf; t += 256) { const bool vo = f0s[t] > 0.f; if (sel == 0 || (sel == 1) == vo) { const float dd = v[t] - m; a1 += dd * dd; } }
        const float var = block_sum(a1, red);
        sn = (n && m != 0.f) ? sqrtf(var / n) / fabsf(m) : 0.f;
    };
    // the ten functionals of a contour over the selected frames -> o[base .. base + 9]
    auto ten = [&](int sel, int base) {
        float m, sn;
        mean_sn(sel, m, sn);
        // compact the selected values (order preserved) into s by thread 0 -- also the scan for slopes
        if (tid == 0) {
            int n = 0;
            for (int t = 0; t < nf; ++t) { const bool vo = f0s[t] > 0.f; if (sel == 0 || (sel == 1) == vo) s[n++] = v[t]; }
            cnt_s[0] = n;
            // slopes of the rising / falling parts (cut at local extrema)
            float rs = 0.f, rq = 0.f, fs = 0.f, fq = 0.f; int rn = 0, fn = 0;
            if (n >= 2) {
                int start = 0;
                for (int t = 1; t < n; ++t) {
                    const bool last = t == n - 1;
                    const bool turn = !last && ((s[t] - s[t - 1]) * (s[t + 1] - s[t]) < 0.f);
                    if (turn || last) {
                        const float dv = s[t] - s[start];
                        const float sl = dv / ((t - start) * (float)HOP / SR);
                        if (dv > 0.f) { rs += sl; rq += sl * sl; ++rn; } else if (dv < 0.f) { fs += sl; fq += sl * sl; ++fn; }
                        start = t;
                    }
                }
            }
            res[0] = rn ? rs / rn : 0.f; res[1] = rn ? sqrtf(fmaxf(rq / rn - (rs / rn) * (rs / rn), 0.f)) : 0.f;
            res[2] = fn ? fs / fn : 0.f; res[3] = fn ? sqrtf(fmaxf(fq / fn - (fs / fn) * (fs / fn), 0.f)) : 0.f;
        }
        __syncthreads();
        const int n = cnt_s[0];
        int n2 = 1;
        while (n2 < n) n2 <<= 1;
        for (int i = n + tid; i < n2; i += 256) s[i] = INFINITY;
        __syncthreads();
        if (n2 > 1) bitonic_sort(s, n2);
        if (tid == 0) {
            auto pct = [&](float p) {
                if (n == 0) return 0.f;
                const float pos = p * (n - 1);
                const int i = (int)floorf(pos); const float fr = pos - i;
                return i + 1 >= n ? s[i] : s[i] * (1.f - fr) + s[i + 1] * fr;
            };
            const float p20 = pct(0.2f), p50 = pct(0.5f), p80 = pct(0.8f);
            o[base] = m; o[base + 1] = sn; o[base + 2] = p20; o[base + 3] = p50; o[base + 4] = p80; o[base + 5] = p80 - p20;
            o[base + 6] = res[0]; o[base + 7] = res[1]; o[base + 8] = res[2]; o[base + 9] = res[3];
        }
        __syncthreads();
    };
    float m, sn;
    load(R_F0, 2); ten(1, 0);
    load(R_LOUD, 0); ten(0, 10);
    // loudness peaks (local maxima of the smoothed contour)
    {
        float pk = 0.f;
        for (int t = 1 + tid; t + 1 < nf; t += 256) if (v[t] > v[t - 1] && v[t] >= v[t + 1]) pk += 1.f;
        const float npk = block_sum(pk, red);
        if (tid == 0) o[81] = npk / (nf * (float)HOP / SR);
    }
    load(R_FLUX, 0);
    mean_sn(0, m, sn); if (tid == 0) { o[20] = m; o[21] = sn; }
    mean_sn(1, m, sn); if (tid == 0) { o[66] = m; o[67] = sn; }
    mean_sn(2, m, sn); if (tid == 0) o[80] = m;
    for (int i = 0; i < 4; ++i) {
        load(R_MFCC + i, 0);
        mean_sn(0, m, sn); if (tid == 0) { o[22 + 2 * i] = m; o[23 + 2 * i] = sn; }
        mean_sn(1, m, sn); if (tid == 0) { o[68 + 2 * i] = m; o[69 + 2 * i] = sn; }
    }
    {
        const int fields[14] = {R_JIT, R_SHIM, R_HNR, R_H1H2, R_H1A3, R_F, R_BW, R_FAMP, R_F + 1, R_BW + 1, R_FAMP + 1, R_F + 2, R_BW + 2, R_FAMP + 2};
        for (int i = 0; i < 14; ++i) {
            load(fields[i], 1);
            mean_sn(1, m, sn); if (tid == 0) { o[30 + 2 * i] = m; o[31 + 2 * i] = sn; }
        }
        const int spec[4] = {R_ALPHA, R_HAMM, R_SL0, R_SL1};
        for (int i = 0; i < 4; ++i) {
            load(spec[i], 0);
            mean_sn(1, m, sn); if (tid == 0) { o[58 + 2 * i] = m; o[59 + 2 * i] = sn; }
            mean_sn(2, m, sn); if (tid == 0) o[76 + i] = m;
        }
    }
    // voiced / unvoiced segments and the equivalent sound level
    {
        float e2 = 0.f;
        for (int t = tid; t < nf; t += 256) { const float r = raw(t, R_RMS); e2 += r * r; }
        const float es = block_sum(e2, red);
        if (tid == 0) {
            float vs = 0.f, vq = 0.f, us = 0.f, uq = 0.f; int vn = 0, un = 0, run = 0; bool cur = false;
            for (int t = 0; t <= nf; ++t) {
                const bool vo = t < nf && f0s[t] > 0.f;
                if (t == nf || (run && vo != cur)) {
                    if (run) { const float len = (float)run; if (cur) { vs += len; vq += len * len; ++vn; } else { us += len; uq += len * len; ++un; } }
                    run = 0;
                }
                if (t < nf) { cur = vo; ++run; }
            }
            const float dur = nf * (float)HOP / SR, fr = (float)HOP / SR;
            o[82] = vn / dur;
            o[83] = vn ? vs / vn * fr : 0.f;
            o[84] = vn ? sqrtf(fmaxf(vq / vn - (vs / vn) * (vs / vn), 0.f)) * fr : 0.f;
            o[85] = un ? us / un * fr : 0.f;
            o[86] = un ? sqrtf(fmaxf(uq / un - (us / un) * (us / un), 0.f)) * fr : 0.f;
            o[87] = 10.0f * log10f(fmaxf(es / nf, 1e-12f));
        }
    }
}

static EgmArgs egm_args(const EgmPlan* p, const float* audio, int64_t L, int nf, const float* scale, float* rec) {
    EgmArgs a{};
    a.audio = audio; a.L = L; a.nf = nf; a.scale = scale; a.tab = p->d; a.rec = rec;
    a.o_g60 = p->o_g60; a.o_ham = p->o_ham; a.o_tw = p->o_tw; a.o_fb_w = p->o_fb_w; a.o_fb_start = p->o_fb_start; a.o_fb_count = p->o_fb_count;
    a.o_fb_off = p->o_fb_off; a.o_eql = p->o_eql; a.o_dct = p->o_dct; a.o_logi = p->o_logi; a.o_logf = p->o_logf; a.o_hshift = p->o_hshift;
    a.o_hweight = p->o_hweight; a.o_pre = p->o_pre; a.o_cos = p->o_cos; a.o_sl0 = p->o_sl0; a.o_sl1 = p->o_sl1; a.o_gg = p->o_gg;
    a.n_log = p->n_log; a.j0 = p->j0; a.j1 = p->j1; a.nb_lpc = p->nb_lpc;
    for (int i = 0; i < 4; ++i) a.b_alpha[i] = p->b_alpha[i];
    for (int i = 0; i < 3; ++i) a.b_hamm[i] = p->b_hamm[i];
    for (int i = 0; i < 2; ++i) { a.b_sl0[i] = p->b_sl0[i]; a.b_sl1[i] = p->b_sl1[i]; }
    return a;
}

}  // namespace km

using namespace km;

extern "C" {

int km_egemaps_plan_create(void** plan_out) {
    if (!plan_out) return fail(KM_ERR_INVALID_ARG, "km_egemaps_plan_create: NULL argument");
    EgmPlan* p = build_egm_plan();
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->d), p->h.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(p->d, p->h.data(), p->h.size() * sizeof(float), hipMemcpyHostToDevice));
    *plan_out = p;
    return KM_OK;
}

int km_egemaps_plan_destroy(void* plan) {
    if (!plan) return KM_OK;
    EgmPlan* p = static_cast<EgmPlan*>(plan);
    if (p->d) (void)hipFree(p->d);
    delete p;
    return KM_OK;
}

int64_t km_egemaps_num_frames(int64_t L) { return L < egm::N60 ? 0 : (L - egm::N60) / egm::HOP + 1; }

int64_t km_egemaps_workspace_floats(int64_t B, int64_t L) {
    if (B <= 0 || L <= 0) return -1;
    return B * (km_egemaps_num_frames(L) * egm::REC + 4);
}

int km_egemaps_functionals(void* plan, const float* audio_dev, int64_t B, int64_t L, int32_t normalize, float* work_dev,
                           int64_t work_floats, float* out_dev, void* stream) {
    using namespace egm;
    if (!plan || !audio_dev || !work_dev || !out_dev || B <= 0 || L <= 0) return fail(KM_ERR_INVALID_ARG, "km_egemaps_functionals: bad argument");
    const int64_t nf = km_egemaps_num_frames(L);
    if (nf < 1) return fail(KM_ERR_INVALID_ARG, "km_egemaps_functionals: the window is shorter than one 60 ms frame (%lld samples)", (long long)L);
    if (nf > MAXF) return fail(KM_ERR_UNSUPPORTED, "km_egemaps_functionals: %lld frames per window, at most %d (20.5 s)", (long long)nf, MAXF);
    if (work_floats < km_egemaps_workspace_floats(B, L)) return fail(KM_ERR_WORKSPACE, "km_egemaps_functionals: workspace too small");
    if (B > 65535) return fail(KM_ERR_UNSUPPORTED, "km_egemaps_functionals: at most 65535 windows per call");
    EgmPlan* p = static_cast<EgmPlan*>(plan);
    hipStream_t st = (hipStream_t)stream;
    float* scale = work_dev;                       // (B), padded to 4 B floats
    float* rec = work_dev + 4 * B;
    if (normalize) {
        hipLaunchKernelGGL(egm_peak_kernel, dim3((unsigned)B), dim3(256), 0, st, audio_dev, L, scale);
        HIP_TRY(hipGetLastError());
    }
    const EgmArgs a = egm_args(p, audio_dev, L, (int)nf, normalize ? scale : nullptr, rec);
    hipLaunchKernelGGL(egm_frame_kernel, dim3((unsigned)nf, (unsigned)B), dim3(256), 0, st, a);
    hipLaunchKernelGGL(egm_viterbi_kernel, dim3((unsigned)B), dim3(256), 0, st, rec, (int)nf);
    hipLaunchKernelGGL(egm_voiced_kernel, dim3((unsigned)nf, (unsigned)B), dim3(256), 0, st, a);
    hipLaunchKernelGGL(egm_functional_kernel, dim3((unsigned)B), dim3(256), 0, st, rec, (int)nf, out_dev);
    HIP_TRY(hipGetLastError());
    return KM_OK;
}

// per-frame records of the most recent call on this workspace, for the tests: (B, nf, 36) floats
int km_egemaps_records(const float* work_dev, int64_t B, int64_t L, float* rec_host, void* stream) {
    if (!work_dev || !rec_host || B <= 0) return fail(KM_ERR_INVALID_ARG, "km_egemaps_records: bad argument");
    const int64_t nf = km_egemaps_num_frames(L);
    HIP_TRY(hipMemcpyAsync(rec_host, work_dev + 4 * B, (size_t)B * nf * egm::REC * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return KM_OK;
}

}  // extern "C"
