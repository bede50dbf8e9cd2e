"""Oracle: the 88 eGeMAPSv02 functionals, restated in numpy float64.  TEST INFRASTRUCTURE (see oracle/__init__.py).

PARITY UNPINNED.  The reference obtains these features from the third-party `opensmile` package
(/root/reference/src/features/opensmile_extractor.py:227-235 `opensmile.Smile(FeatureSet.eGeMAPSv02, FeatureLevel.Functionals)`,
:439 `process_signal`); the package is not installed here, the reference pins no version of it (it is not even listed in
pyproject.toml, EGEMAPS_SETUP.md:18) and ships no feature fixtures.  What follows restates the PUBLISHED parameter set --
F. Eyben et al., "The Geneva Minimalistic Acoustic Parameter Set (GeMAPS) for Voice Research and Affective Computing",
IEEE Trans. Affective Computing 7(2), 2016, sections 3.1-3.3 and the extended set of 3.4 -- with the processing chain the
openSMILE 3.0 eGeMAPSv02 configuration documents (20 ms Hamming / 60 ms Gaussian frames every 10 ms, sub-harmonic-summation
pitch with Viterbi smoothing, LPC formants on the spectrum below 5.5 kHz, 26-band auditory spectrum, 3-frame moving-average
smoothing, functionals over voiced / unvoiced / all frames).  Where the paper leaves a constant open the choice is written
next to it.  The order of the 88 outputs is openSMILE's (FEATURE_NAMES).  The reference itself treats the vector as opaque
(it feeds a randomly initialised Linear(264, 256), opensmile_extractor.py:575-590), so what matters downstream is that
training and inference see the SAME extractor; what is tested is GPU == this restatement, and known answers on synthetic
signals (a harmonic complex of known F0, a two-resonance vowel of known formants, known jitter).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

SR = 16000
HOP = 160                      # 10 ms
N60, N20, NFFT = 960, 320, 1024
OFF20 = (N60 - N20) // 2       # the 20 ms frame sits in the middle of the 60 ms frame
NBANDS = 26
LPC_ORDER, LPC_MAXF = 11, 5500.0
F0_MIN, F0_MAX = 55.0, 1000.0
PPO = 48                       # points per octave of the log-frequency axis of the pitch detector
N_HARM, SHS_COMPRESSION = 15, 0.85
N_CAND = 3
VOICING_CUTOFF, RMS_FLOOR = 0.55, 0.001    # on the autocorrelation measure of acf_strength() (openSMILE's 0.7 is on its own SHS measure)

FEATURE_NAMES: List[str] = (
    [f"F0semitoneFrom27.5Hz_sma3nz_{s}" for s in ("amean", "stddevNorm", "percentile20.0", "percentile50.0", "percentile80.0",
                                                   "pctlrange0-2", "meanRisingSlope", "stddevRisingSlope", "meanFallingSlope",
                                                   "stddevFallingSlope")] +
    [f"loudness_sma3_{s}" for s in ("amean", "stddevNorm", "percentile20.0", "percentile50.0", "percentile80.0", "pctlrange0-2",
                                     "meanRisingSlope", "stddevRisingSlope", "meanFallingSlope", "stddevFallingSlope")] +
    ["spectralFlux_sma3_amean", "spectralFlux_sma3_stddevNorm"] +
    [f"mfcc{i}_sma3_{s}" for i in (1, 2, 3, 4) for s in ("amean", "stddevNorm")] +
    [f"{n}_sma3nz_{s}" for n in ("jitterLocal", "shimmerLocaldB", "HNRdBACF", "logRelF0-H1-H2", "logRelF0-H1-A3",
                                  "F1frequency", "F1bandwidth", "F1amplitudeLogRelF0", "F2frequency", "F2bandwidth",
                                  "F2amplitudeLogRelF0", "F3frequency", "F3bandwidth", "F3amplitudeLogRelF0")
     for s in ("amean", "stddevNorm")] +
    [f"{n}V_sma3nz_{s}" for n in ("alphaRatio", "hammarbergIndex", "slope0-500", "slope500-1500", "spectralFlux",
                                   "mfcc1", "mfcc2", "mfcc3", "mfcc4") for s in ("amean", "stddevNorm")] +
    [f"{n}UV_sma3nz_amean" for n in ("alphaRatio", "hammarbergIndex", "slope0-500", "slope500-1500", "spectralFlux")] +
    ["loudnessPeaksPerSec", "VoicedSegmentsPerSec", "MeanVoicedSegmentLengthSec", "StddevVoicedSegmentLengthSec",
     "MeanUnvoicedSegmentLength", "StddevUnvoicedSegmentLength", "equivalentSoundLevel_dBp"])
assert len(FEATURE_NAMES) == 88


# ---- tables shared with the GPU plan (koemorph_amd/csrc/km_egemaps.hip gets them from km_host.cpp-style host code) ----
def gauss_window(n: int = N60, sigma: float = 0.4) -> np.ndarray:
    k = np.arange(n) - (n - 1) / 2.0
    return np.exp(-0.5 * (k / (sigma * (n - 1) / 2.0)) ** 2)


def hamming_window(n: int = N20) -> np.ndarray:
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(n) / (n - 1))


def mel_bands(nbands: int = NBANDS, lo: float = 20.0, hi: float = 8000.0) -> Tuple[np.ndarray, np.ndarray]:
    """Triangular HTK-mel filters over the 513 bins of the 1024-point spectrum, (nbands, 513), and the band centres in Hz."""
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    imel = lambda m: 700.0 * (np.exp(m / 1127.0) - 1.0)
    edges = imel(np.linspace(mel(lo), mel(hi), nbands + 2))
    f = np.arange(NFFT // 2 + 1) * (SR / NFFT)
    fb = np.zeros((nbands, f.size))
    for j in range(nbands):
        l, c, r = edges[j], edges[j + 1], edges[j + 2]
        fb[j] = np.clip(np.minimum((f - l) / (c - l), (r - f) / (r - c)), 0.0, None)
    return fb, edges[1:-1]


def equal_loudness(fc: np.ndarray) -> np.ndarray:
    """Hermansky's equal-loudness curve (PLP), evaluated at the band centres."""
    w2 = (2.0 * np.pi * fc) ** 2
    return ((w2 + 56.8e6) * w2 * w2) / ((w2 + 6.3e6) ** 2 * (w2 + 0.38e9))


def log_axis() -> Tuple[np.ndarray, int, int]:
    """Log2-frequency axis of the pitch detector: PPO points per octave from 25 Hz up to Nyquist; returns the frequencies and
    the index range [j0, j1) of the F0 search band."""
    n = int(np.floor(np.log2((SR / 2) / 25.0) * PPO)) + 1
    fj = 25.0 * 2.0 ** (np.arange(n) / PPO)
    j0 = int(np.ceil(np.log2(F0_MIN / 25.0) * PPO))
    j1 = int(np.floor(np.log2(F0_MAX / 25.0) * PPO)) + 1
    return fj, j0, j1


HARM_SHIFT = np.round(PPO * np.log2(np.arange(1, N_HARM + 1))).astype(int)
HARM_WEIGHT = SHS_COMPRESSION ** np.arange(N_HARM)


# ---- per-frame low-level descriptors -------------------------------------------------------------------------------
def frame_count(L: int) -> int:
    return 0 if L < N60 else (L - N60) // HOP + 1


def spectra(x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Magnitude spectra (NF, 513) of the 60 ms Gaussian frames and of the centred 20 ms Hamming frames, 1024-point FFT."""
    nf = frame_count(len(x))
    idx = HOP * np.arange(nf)[:, None]
    f60 = x[idx + np.arange(N60)[None, :]] * gauss_window()[None, :]
    f20 = x[idx + OFF20 + np.arange(N20)[None, :]] * hamming_window()[None, :]
    return np.abs(np.fft.rfft(f60, NFFT, axis=1)), np.abs(np.fft.rfft(f20, NFFT, axis=1))


def spectral_llds(M20: np.ndarray) -> Dict[str, np.ndarray]:
    """Loudness, alpha ratio, Hammarberg index, spectral slopes, spectral flux, MFCC 1-4 from the 20 ms magnitude spectra."""
    P = M20 ** 2
    f = np.arange(P.shape[1]) * (SR / NFFT)
    fb, fc = mel_bands()
    E = P @ fb.T                                                   # (NF, 26) band powers
    loud = ((E * equal_loudness(fc)[None, :]) ** 0.33).sum(axis=1)  # auditory spectrum: equal loudness, cube-root-like compression
    logE = np.log(np.maximum(E, 1e-8))
    i = np.arange(1, 5)[:, None]
    dct = np.sqrt(2.0 / NBANDS) * np.cos(np.pi * i * (np.arange(NBANDS)[None, :] + 0.5) / NBANDS)
    lift = 1.0 + 11.0 * np.sin(np.pi * np.arange(1, 5) / 22.0)    # cepstral lifter L = 22
    mfcc = (logE @ dct.T) * lift[None, :]
    band = lambda lo, hi: (f >= lo) & (f < hi)
    eps = 1e-12
    alpha = 10.0 * np.log10((P[:, band(50, 1000)].sum(1) + eps) / (P[:, band(1000, 5000)].sum(1) + eps))
    hamm = 10.0 * np.log10((P[:, band(0, 2000)].max(1) + eps) / (P[:, band(2000, 5000)].max(1) + eps))
    LdB = 10.0 * np.log10(P + eps)

    def slope(lo, hi):                                              # least-squares slope of dB power against Hz
        m = band(lo, hi)
        fx = f[m] - f[m].mean()
        return (LdB[:, m] * fx[None, :]).sum(1) / (fx ** 2).sum()
    flux = np.zeros(len(M20))
    if len(M20) > 1:
        flux[1:] = np.sqrt(((M20[1:] - M20[:-1]) ** 2).mean(axis=1))
    return {"loudness": loud, "alphaRatio": alpha, "hammarbergIndex": hamm, "slope0-500": slope(0, 500),
            "slope500-1500": slope(500, 1500), "spectralFlux": flux, "mfcc": mfcc}


def shs_candidates(M60: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Sub-harmonic summation on a log2-frequency axis: (NF, N_CAND) candidate frequencies, their scores in [0, 1] and the
    frame's voicing measure (peak height over the mean of the summation spectrum)."""
    fj, j0, j1 = log_axis()
    f = np.arange(M60.shape[1]) * (SR / NFFT)
    S = np.stack([np.interp(fj, f, m) for m in M60])                # linear interpolation onto the log axis
    S = np.concatenate([S[:, :1], 0.25 * S[:, :-2] + 0.5 * S[:, 1:-1] + 0.25 * S[:, 2:], S[:, -1:]], axis=1)   # 3-tap smoothing
    n = S.shape[1]
    shs = np.zeros((len(M60), j1 - j0))
    for h in range(N_HARM):
        idx = np.arange(j0, j1) + HARM_SHIFT[h]
        ok = idx < n
        shs[:, ok] += HARM_WEIGHT[h] * S[:, idx[ok]]
    cf = np.zeros((len(M60), N_CAND)); cs = np.zeros((len(M60), N_CAND)); vo = np.zeros(len(M60))
    for t, row in enumerate(shs):
        peak, mean = row.max(), row.mean()
        vo[t] = 0.0 if peak <= 0 else 1.0 - mean / peak
        r = row.copy()
        for c in range(N_CAND):                                     # greedy: highest point, then blank +- 1/6 octave around it
            k = int(np.argmax(r))
            if r[k] <= 0:
                break
            # parabolic refinement of the peak position on the log axis
            d = 0.0
            if 0 < k < len(row) - 1:
                a, b, cc = row[k - 1], row[k], row[k + 1]
                den = a - 2 * b + cc
                d = 0.0 if den == 0 else float(np.clip(0.5 * (a - cc) / den, -0.5, 0.5))
            cf[t, c] = 25.0 * 2.0 ** ((j0 + k + d) / PPO)
            cs[t, c] = row[k] / peak
            r[max(0, k - PPO // 6):k + PPO // 6 + 1] = 0.0
    return cf, cs, vo


def acf_strength(x: np.ndarray, cf0: np.ndarray) -> np.ndarray:
    """Voicing measure of a frame: the window-compensated normalised autocorrelation of the 60 ms Gaussian frame at the lag
    of the strongest pitch candidate (searched within +- 10 %), counted only if that maximum is a LOCAL one -- red noise has a
    high but monotonically decaying autocorrelation and must not pass for voiced."""
    g = gauss_window()
    gg0 = np.dot(g, g)
    out = np.zeros(len(cf0))
    for t, f in enumerate(cf0):
        if f <= 0:
            continue
        sw = x[HOP * t:HOP * t + N60] * g
        T0 = SR / f
        lo, hi = max(int(np.floor(0.9 * T0)), 1), min(int(np.ceil(1.1 * T0)), N60 - 2)
        r0 = np.dot(sw, sw)
        if r0 <= 0:
            continue
        r = np.array([np.dot(sw[:-lag], sw[lag:]) / max(np.dot(g[:-lag], g[lag:]), 1e-12) * gg0 for lag in range(lo - 1, hi + 2)])
        k = int(np.argmax(r[1:-1])) + 1                              # best lag inside [lo, hi]
        if r[k] >= r[k - 1] and r[k] >= r[k + 1]:
            out[t] = min(max(r[k] / r0, 0.0), 1.0)
    return out


def viterbi_f0(cf: np.ndarray, cs: np.ndarray, vo: np.ndarray, rms: np.ndarray) -> np.ndarray:
    """Smoothed F0 track: per frame one of the N_CAND candidates or 'unvoiced' (0).  Local cost favours strong candidates in
    voiced frames and the unvoiced state where the voicing measure is below the cutoff; transitions pay for octave jumps
    (w_vv per octave) and for switching voicing (w_vuv)."""
    w_local, w_vv, w_vuv, w_thr = 2.0, 10.0, 10.0 / 8.0, 4.0
    nf, S = len(vo), N_CAND + 1
    INF = 1e30
    cost = np.full((nf, S), INF); back = np.zeros((nf, S), int)
    voiced_ok = (vo >= VOICING_CUTOFF) & (rms >= RMS_FLOOR)
    loc = np.full((nf, S), INF)
    for c in range(N_CAND):
        good = cf[:, c] > 0
        loc[good, c] = w_local * (1.0 - cs[good, c]) + np.where(voiced_ok[good], 0.0, w_thr)
    loc[:, N_CAND] = np.where(voiced_ok, w_thr, 0.0)
    lf = np.where(cf > 0, np.log2(np.maximum(cf, 1e-9)), 0.0)
    cost[0] = loc[0]
    for t in range(1, nf):
        for s in range(S):
            if loc[t, s] >= INF:
                continue
            best, arg = INF, 0
            for p in range(S):
                if cost[t - 1, p] >= INF:
                    continue
                if s < N_CAND and p < N_CAND:
                    tr = w_vv * abs(lf[t, s] - lf[t - 1, p])
                elif s == N_CAND and p == N_CAND:
                    tr = 0.0
                else:
                    tr = w_vuv
                v = cost[t - 1, p] + tr
                if v < best:
                    best, arg = v, p
            cost[t, s] = best + loc[t, s]; back[t, s] = arg
    f0 = np.zeros(nf)
    s = int(np.argmin(cost[-1]))
    for t in range(nf - 1, -1, -1):
        f0[t] = cf[t, s] if s < N_CAND else 0.0
        s = back[t, s]
    return f0


def lpc_formants(M20: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Formant frequencies and bandwidths (NF, 3) in Hz: order-11 LPC of the spectrum below 5.5 kHz (the spectrum is cut there
    and treated as that of a signal sampled at 11 kHz, which is what resampling in the spectral domain does), autocorrelation
    method, roots of the prediction polynomial with 90 Hz < f < 5400 Hz and bandwidth < 1 kHz... sorted by frequency."""
    nb = int(LPC_MAXF / (SR / NFFT)) + 1                            # bins 0 .. 5500 Hz
    P = M20[:, :nb] ** 2
    # pre-emphasis in the spectral domain: |1 - 0.97 e^{-jw}|^2 at the 11 kHz rate
    w = np.pi * np.arange(nb) / (nb - 1)
    P = P * (1.0 + 0.97 ** 2 - 2.0 * 0.97 * np.cos(w))[None, :]
    lags = np.arange(LPC_ORDER + 1)
    wgt = np.ones(nb); wgt[0] = wgt[-1] = 0.5
    R = (P * wgt[None, :]) @ np.cos(np.outer(w, lags))              # autocorrelation by the cosine transform of the power spectrum
    fs2 = 2.0 * LPC_MAXF
    F = np.zeros((len(M20), 3)); BW = np.zeros((len(M20), 3))
    for t, r in enumerate(R):
        if r[0] <= 0:
            continue
        a = np.zeros(LPC_ORDER + 1); a[0] = 1.0; e = r[0]
        for i in range(1, LPC_ORDER + 1):                           # Levinson-Durbin
            k = -(r[i] + np.dot(a[1:i], r[i - 1:0:-1])) / e
            a[1:i + 1] = a[1:i + 1] + k * np.concatenate([a[i - 1:0:-1], [1.0]])
            e *= 1.0 - k * k
            if e <= 0:
                break
        roots = np.roots(a)
        roots = roots[np.imag(roots) > 1e-9]
        fr = np.angle(roots) * fs2 / (2.0 * np.pi)
        bw = -np.log(np.maximum(np.abs(roots), 1e-12)) * fs2 / np.pi
        ok = (fr > 90.0) & (fr < LPC_MAXF - 100.0) & (bw < 1000.0)
        order = np.argsort(fr[ok])[:3]
        F[t, :len(order)] = fr[ok][order]; BW[t, :len(order)] = bw[ok][order]
    return F, BW


def voiced_llds(x: np.ndarray, M60: np.ndarray, f0: np.ndarray, F: np.ndarray) -> Dict[str, np.ndarray]:
    """What needs the final F0: jitter, shimmer (pitch periods marked in the waveform), HNR from the autocorrelation, harmonic
    differences H1-H2 / H1-A3 and formant amplitudes relative to H1 from the 60 ms spectrum.  Zero in unvoiced frames."""
    nf = len(f0)
    out = {k: np.zeros(nf) for k in ("jitterLocal", "shimmerLocaldB", "HNRdBACF", "H1-H2", "H1-A3")}
    out["Famp"] = np.zeros((nf, 3))
    binw = SR / NFFT
    g = gauss_window()

    def harm_db(m, freq):                                          # dB amplitude of the strongest bin within +- 20 % of f0 around freq
        lo = int(max(0, np.floor((freq - 0.2 * f0t) / binw))); hi = int(min(len(m) - 1, np.ceil((freq + 0.2 * f0t) / binw)))
        return 20.0 * np.log10(max(m[lo:hi + 1].max(), 1e-12))
    for t in range(nf):
        f0t = f0[t]
        if f0t <= 0:
            continue
        seg = x[HOP * t:HOP * t + N60]
        T0 = SR / f0t
        # --- HNR: normalised autocorrelation of the windowed frame at the pitch lag (searched +- 10 %), window-compensated
        sw = seg * g
        lo, hi = int(np.floor(0.9 * T0)), int(np.ceil(1.1 * T0))
        hi = min(hi, N60 - 2)
        r0 = np.dot(sw, sw)
        rw0 = np.dot(g, g)
        best = 0.0
        for lag in range(max(lo, 1), hi + 1):
            r = np.dot(sw[:-lag], sw[lag:]) / max(np.dot(g[:-lag], g[lag:]), 1e-12) * rw0
            best = max(best, r)
        rr = min(max(best / max(r0, 1e-20), 1e-6), 1.0 - 1e-6)
        out["HNRdBACF"][t] = 10.0 * np.log10(rr / (1.0 - rr))
        # --- pitch periods: start at the largest sample, step by the lag in [0.9 T0, 1.1 T0] that maximises the cross-correlation
        #     of the next period with the current one
        p0 = int(np.argmax(seg))
        T = int(round(T0))
        marks = [p0 % T if p0 >= T else p0]
        pos = marks[0]
        while True:
            bestc, bestl = -np.inf, 0
            for lag in range(max(lo, 2), hi + 1):
                if pos + 2 * lag > N60:
                    break
                a = seg[pos:pos + lag]; b = seg[pos + lag:pos + 2 * lag]
                c = np.dot(a, b) / np.sqrt(max(np.dot(a, a) * np.dot(b, b), 1e-20))
                if c > bestc:
                    bestc, bestl = c, lag
            if bestl == 0 or bestc < 0.5:
                break
            pos += bestl
            marks.append(pos)
        if len(marks) >= 3:
            per = np.diff(marks).astype(float)
            out["jitterLocal"][t] = np.abs(np.diff(per)).mean() / per.mean()
            amp = np.array([seg[marks[i]:marks[i + 1]].max() - seg[marks[i]:marks[i + 1]].min() for i in range(len(marks) - 1)])
            amp = np.maximum(amp, 1e-9)
            out["shimmerLocaldB"][t] = np.abs(20.0 * np.log10(amp[1:] / amp[:-1])).mean()
        # --- harmonics
        m = M60[t]
        h1, h2 = harm_db(m, f0t), harm_db(m, 2 * f0t)
        out["H1-H2"][t] = h1 - h2
        for i in range(3):
            if F[t, i] > 0:
                a = harm_db(m, max(f0t, round(F[t, i] / f0t) * f0t))
                out["Famp"][t, i] = a - h1
                if i == 2:
                    out["H1-A3"][t] = h1 - a
    return out


# ---- functionals -------------------------------------------------------------------------------------------------------
def sma3(v: np.ndarray, nz: bool) -> np.ndarray:
    """3-frame moving average; `nz`: zeros (unvoiced frames) neither enter a neighbour's average nor get filled in."""
    out = np.zeros_like(v, dtype=float)
    n = len(v)
    for t in range(n):
        lo, hi = max(0, t - 1), min(n, t + 2)
        w = v[lo:hi]
        if nz:
            if v[t] == 0:
                continue
            w = w[w != 0]
        out[t] = w.mean()
    return out


def percentile(sorted_v: np.ndarray, p: float) -> float:
    if len(sorted_v) == 0:
        return 0.0
    pos = p * (len(sorted_v) - 1)
    i = int(np.floor(pos)); fr = pos - i
    return float(sorted_v[i] if i + 1 >= len(sorted_v) else sorted_v[i] * (1 - fr) + sorted_v[i + 1] * fr)


def mean_std_norm(v: np.ndarray) -> Tuple[float, float]:
    if len(v) == 0:
        return 0.0, 0.0
    m = float(v.mean()); s = float(np.sqrt(((v - m) ** 2).mean()))
    return m, (0.0 if m == 0 else s / abs(m))


def slopes(v: np.ndarray) -> Tuple[float, float, float, float]:
    """Mean / standard deviation of the slopes (per second) of the rising and of the falling parts of a contour: the contour is
    cut at its local extrema, a part's slope is its height over its duration."""
    n = len(v)
    rise, fall = [], []
    if n >= 2:
        start = 0
        for t in range(1, n):
            last = t == n - 1
            turn = (not last) and ((v[t] - v[t - 1]) * (v[t + 1] - v[t]) < 0)
            if turn or last:
                dv = v[t] - v[start]
                if dv > 0:
                    rise.append(dv / ((t - start) * HOP / SR))
                elif dv < 0:
                    fall.append(dv / ((t - start) * HOP / SR))
                start = t
    ms = lambda a: (0.0, 0.0) if len(a) == 0 else (float(np.mean(a)), float(np.std(a)))
    return (*ms(rise), *ms(fall))


def ten_functionals(v: np.ndarray) -> List[float]:
    m, sn = mean_std_norm(v)
    s = np.sort(v)
    p20, p50, p80 = percentile(s, 0.2), percentile(s, 0.5), percentile(s, 0.8)
    return [m, sn, p20, p50, p80, p80 - p20, *slopes(v)]


def segments(mask: np.ndarray) -> np.ndarray:
    """Lengths (frames) of the runs of True."""
    out, run = [], 0
    for b in mask:
        if b:
            run += 1
        elif run:
            out.append(run); run = 0
    if run:
        out.append(run)
    return np.array(out, float)


def llds(x: np.ndarray) -> Dict[str, np.ndarray]:
    x = np.asarray(x, np.float64)
    M60, M20 = spectra(x)
    nf = len(M60)
    sp = spectral_llds(M20)
    idx = HOP * np.arange(nf)[:, None] + np.arange(N60)[None, :]
    rms = np.sqrt((x[idx] ** 2).mean(axis=1))
    cf, cs, _ = shs_candidates(M60)
    vo = acf_strength(x, cf[:, 0])
    f0 = viterbi_f0(cf, cs, vo, rms)
    F, BW = lpc_formants(M20)
    F = F * (f0 > 0)[:, None]; BW = BW * (f0 > 0)[:, None]          # formants are kept in voiced frames only
    vl = voiced_llds(x, M60, f0, F)
    return {"f0": f0, "rms": rms, "F": F, "BW": BW, "cand_f": cf, "cand_s": cs, "voicing": vo, **sp, **vl}


def functionals(x: np.ndarray) -> np.ndarray:
    """The 88 eGeMAPSv02 functionals of one audio window (FEATURE_NAMES order)."""
    d = llds(x)
    nf = len(d["f0"])
    out: List[float] = []
    if nf == 0:
        return np.zeros(88, np.float32)
    f0 = sma3(d["f0"], True)
    voiced = f0 > 0
    semitone = np.where(voiced, 12.0 * np.log2(np.maximum(f0, 1e-9) / 27.5), 0.0)
    out += ten_functionals(semitone[voiced])
    loud = sma3(d["loudness"], False)
    out += ten_functionals(loud)
    flux = sma3(d["spectralFlux"], False)
    out += list(mean_std_norm(flux))
    mf = np.stack([sma3(d["mfcc"][:, i], False) for i in range(4)], axis=1)
    for i in range(4):
        out += list(mean_std_norm(mf[:, i]))
    nzs = [d["jitterLocal"], d["shimmerLocaldB"], d["HNRdBACF"], d["H1-H2"], d["H1-A3"]]
    for i in range(3):
        nzs += [d["F"][:, i], d["BW"][:, i], d["Famp"][:, i]]
    for v in nzs:                                                   # voiced frames only
        out += list(mean_std_norm(sma3(np.where(voiced, v, 0.0), True)[voiced]))
    spec = [sma3(d[k], False) for k in ("alphaRatio", "hammarbergIndex", "slope0-500", "slope500-1500")] + [flux]
    for v in spec + [mf[:, i] for i in range(4)]:
        out += list(mean_std_norm(v[voiced]))
    for v in spec:
        out.append(float(v[~voiced].mean()) if (~voiced).any() else 0.0)
    dur = nf * HOP / SR
    peaks = int(((loud[1:-1] > loud[:-2]) & (loud[1:-1] >= loud[2:])).sum()) if nf > 2 else 0
    vs, us = segments(voiced), segments(~voiced)
    out.append(peaks / dur)
    out.append(len(vs) / dur)
    out.append(float(vs.mean() * HOP / SR) if len(vs) else 0.0)
    out.append(float(vs.std() * HOP / SR) if len(vs) else 0.0)
    out.append(float(us.mean() * HOP / SR) if len(us) else 0.0)
    out.append(float(us.std() * HOP / SR) if len(us) else 0.0)
    out.append(float(10.0 * np.log10(max((d["rms"] ** 2).mean(), 1e-12))))
    assert len(out) == 88
    return np.asarray(out, np.float32)


def normalise(audio: np.ndarray) -> np.ndarray:
    """opensmile_extractor.py:431-433: peak normalisation to [-1, 1] before the extractor."""
    audio = np.asarray(audio, np.float32)
    m = np.max(np.abs(audio)) if audio.size else 0.0
    return audio / m if m > 0 else audio
