"""Oracle: the three log-mel front ends of the reference, restated in numpy.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: librosa and torchaudio are
third-party dependencies of the reference (pyproject.toml:12-24, lower bounds
``librosa>=0.10.0``, ``torchaudio>=2.0.0``, no lock file) and are not installed in this
image; the reference ships no mel fixtures.  The functions below restate the published
algorithms of ``librosa.feature.melspectrogram`` / ``librosa.power_to_db`` /
``librosa.filters.mel`` (librosa 0.10) and ``torchaudio.transforms.MelSpectrogram``
(torchaudio 2.x) at the reference's call sites:

  (A) batch / training front end     src/model/simplified_dual_stream_model.py:184-214
  (B) real-time sliding window       src/features/mel_sliding_window.py:280-307
  (C) rt.py front end (torchaudio)   src/features/stft.py:84-140

``tests/test_oracle_mel.py`` cross-checks the filterbanks and the dB conversion against
``transformers.audio_utils`` (pure numpy, present in the image).

Precision model: librosa multiplies the float32 frames by a float64 window, so the rFFT
runs in float64 and is then stored as complex64; |.|^2, the mel product and the dB
conversion run in float32.  ``precision="ref"`` mirrors that; ``precision="f64"`` keeps
everything in float64 (used to bound rounding noise).
"""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


# ---------------------------------------------------------------------------
# mel scales and filterbanks
# ---------------------------------------------------------------------------
def hz_to_mel_slaney(f):
    """librosa.hz_to_mel(htk=False): linear below 1 kHz, log above."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def mel_filterbank_librosa(sr: int, n_fft: int, n_mels: int = 80, fmin: float = 0.0,
                           fmax: Optional[float] = None, htk: bool = False,
                           norm: Optional[str] = "slaney") -> np.ndarray:
    """librosa.filters.mel -> (n_mels, 1 + n_fft//2) float32."""
    if fmax is None:
        fmax = sr / 2.0
    n_freq = 1 + n_fft // 2
    weights = np.zeros((n_mels, n_freq), dtype=np.float32)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    to_mel, to_hz = (hz_to_mel_htk, mel_to_hz_htk) if htk else (hz_to_mel_slaney, mel_to_hz_slaney)
    mel_f = to_hz(np.linspace(to_mel(fmin), to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    if norm == "slaney":
        enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
        weights *= enorm[:, np.newaxis]
    return weights


def mel_filterbank_torchaudio(n_freqs: int, f_min: float, f_max: float, n_mels: int,
                              sample_rate: int, norm: Optional[str] = None,
                              mel_scale: str = "htk") -> np.ndarray:
    """torchaudio.functional.melscale_fbanks -> (n_freqs, n_mels) float32."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    to_mel, to_hz = (hz_to_mel_htk, mel_to_hz_htk) if mel_scale == "htk" else (hz_to_mel_slaney, mel_to_hz_slaney)
    m_pts = np.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2)
    f_pts = to_hz(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    if norm == "slaney":
        fb = fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels]))[None, :]
    return fb.astype(np.float32)


# ---------------------------------------------------------------------------
# STFT power
# ---------------------------------------------------------------------------
def hann_periodic(n: int) -> np.ndarray:
    """scipy.signal.get_window('hann', n, fftbins=True) == torch.hann_window(n, periodic=True)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)


def num_frames(length: int, n_fft: int, hop: int, center: bool = True) -> int:
    return 1 + (length + (2 * (n_fft // 2) if center else 0) - n_fft) // hop


def stft_power(y: np.ndarray, n_fft: int, hop: int, win_length: Optional[int] = None,
               center: bool = True, pad_mode: str = "constant", precision: str = "ref",
               window_norm: bool = False) -> np.ndarray:
    """|STFT|^2 -> (n_frames, 1+n_fft//2).  window_norm=True divides the complex STFT by
    sqrt(sum(w^2)) (torchaudio Spectrogram(normalized=True) == "window")."""
    y = np.asarray(y)
    win_length = win_length or n_fft
    w = hann_periodic(win_length)
    if win_length < n_fft:                              # librosa.util.pad_center
        lp = (n_fft - win_length) // 2
        w = np.pad(w, (lp, n_fft - win_length - lp))
    if center:
        mode = {"constant": "constant", "reflect": "reflect"}[pad_mode]
        y = np.pad(y, n_fft // 2, mode=mode)
    n_frames = 1 + (len(y) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = y[idx].astype(np.float64) * w[None, :]     # float32 frame x float64 window
    spec = np.fft.rfft(frames, axis=1)
    if window_norm:
        spec = spec / np.sqrt(np.sum(w * w))
    if precision == "ref":
        spec = spec.astype(np.complex64)
        mag = np.abs(spec)                              # float32
        return (mag * mag).astype(np.float32)
    return (spec.real ** 2 + spec.imag ** 2)


def power_to_db(S: np.ndarray, ref_max: bool = True, amin: float = 1e-10,
                top_db: Optional[float] = 80.0) -> np.ndarray:
    """librosa.power_to_db(S, ref=np.max) on one spectrogram (global max, then clip)."""
    S = np.asarray(S)
    ref_value = np.max(S) if ref_max else 1.0
    ten = S.dtype.type(10.0)
    log_spec = ten * np.log10(np.maximum(S.dtype.type(amin), S))
    log_spec = log_spec - ten * np.log10(np.maximum(S.dtype.type(amin), ref_value))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - S.dtype.type(top_db))
    return log_spec


# ---------------------------------------------------------------------------
# (A) batch front end -- simplified_dual_stream_model.py:184-214
# ---------------------------------------------------------------------------
def mel_batch_window(y: np.ndarray, sample_rate: int = 16000, n_fft: int = 1024,
                     hop: int = 533, n_mels: int = 80, fmin: float = 80.0, fmax: float = 8000.0,
                     precision: str = "ref") -> Tuple[np.ndarray, np.ndarray]:
    """One window: librosa.feature.melspectrogram (defaults: hann, center, pad_mode
    'constant', power 2, slaney) :188-196 -> power_to_db(ref=np.max) :199 -> (x+80)/80 :200
    -> transpose :202 -> (long (T_mel,80), short = last 3 frames (3,80)) :206-214."""
    fb = mel_filterbank_librosa(sample_rate, n_fft, n_mels, fmin, fmax)
    P = stft_power(y, n_fft, hop, center=True, pad_mode="constant", precision=precision)
    if precision == "ref":
        mel = (P.astype(np.float32) @ fb.T.astype(np.float32)).astype(np.float32)   # (T,80)
    else:
        mel = P @ fb.T.astype(np.float64)
    db = power_to_db(mel)
    db = (db + db.dtype.type(80)) / db.dtype.type(80)
    long = db
    if long.shape[0] >= 3:
        short = long[-3:]
    else:
        short = np.zeros((3, n_mels), dtype=long.dtype)
        short[:long.shape[0]] = long
    return long, short


def mel_batch(audio: np.ndarray, **kw) -> Tuple[np.ndarray, np.ndarray]:
    """extract_mel_features for (B,L) audio -> ((B,T_mel,80), (B,3,80)) float32
    (:217-229; all rows have equal length so the zero-padding to max length is a no-op)."""
    longs, shorts = zip(*(mel_batch_window(a, **kw) for a in audio))
    return np.stack(longs).astype(np.float32), np.stack(shorts).astype(np.float32)


# ---------------------------------------------------------------------------
# (B) real-time sliding-window front end -- mel_sliding_window.py:280-307
# ---------------------------------------------------------------------------
def mel_sliding_window(audio_window: np.ndarray, sample_rate: int = 16000, n_fft: int = 512,
                       hop: int = 532, n_mels: int = 80, fmin: float = 80.0,
                       fmax: Optional[float] = None, context_window: float = 8.5,
                       update_interval: float = 0.0333, precision: str = "ref") -> np.ndarray:
    """librosa melspectrogram with pad_mode='reflect' :280-292 -> power_to_db(ref=np.max)
    :295 (NO (x+80)/80 here) -> (T,80) -> truncate to / pad-with-last-frame to
    int(context_window/update_interval) frames :300-307.  float32, values in [-80, 0]."""
    fmax = fmax or sample_rate // 2
    fb = mel_filterbank_librosa(sample_rate, n_fft, n_mels, fmin, fmax)
    P = stft_power(audio_window, n_fft, hop, center=True, pad_mode="reflect", precision=precision)
    mel = (P @ fb.T.astype(P.dtype)).astype(P.dtype)
    db = power_to_db(mel)
    expected = int(context_window / update_interval)
    if db.shape[0] > expected:
        db = db[:expected]
    elif db.shape[0] < expected:
        db = np.vstack([db, np.tile(db[-1:], (expected - db.shape[0], 1))])
    return db.astype(np.float32)


# ---------------------------------------------------------------------------
# (C) rt.py front end -- src/features/stft.py:84-140 (torchaudio MelSpectrogram)
# ---------------------------------------------------------------------------
def mel_torchaudio(waveform: np.ndarray, sample_rate: int = 16000, target_fps: float = 30.0,
                   n_fft: int = 512, n_mels: int = 80, f_min: float = 80.0,
                   f_max: Optional[float] = None, eps: float = 1e-8,
                   precision: str = "ref") -> np.ndarray:
    """MelSpectrogramExtractor.forward: Spectrogram(power 2, normalized "window", center,
    reflect pad, periodic hann) -> HTK fbank (norm None) -> log(mel + eps) :123 ->
    (B,T,80) :126 -> truncate / pad-with-last-frame to int(L/sr*fps) frames :130-140."""
    wav = np.atleast_2d(np.asarray(waveform))
    f_max = f_max or sample_rate // 2
    hop = int(sample_rate / target_fps)
    fb = mel_filterbank_torchaudio(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate)
    outs = []
    for y in wav:
        P = stft_power(y, n_fft, hop, center=True, pad_mode="reflect", precision=precision,
                       window_norm=True)
        mel = (P @ fb.astype(P.dtype)).astype(P.dtype)
        log_mel = np.log(mel + mel.dtype.type(eps))
        expected = int(len(y) / sample_rate * target_fps)
        cur = log_mel.shape[0]
        if cur > expected:
            log_mel = log_mel[:expected]
        elif cur < expected:
            log_mel = np.concatenate([log_mel, np.repeat(log_mel[-1:], expected - cur, axis=0)])
        outs.append(log_mel)
    return np.stack(outs).astype(np.float32)
