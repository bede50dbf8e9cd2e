"""The branch-free erf of km_gemm.h (km_erff: the GELU of every GEMM epilogue) restated in numpy fp32: <= 1.5 ulp from math.erf
for every rounding v_exp_f32 may take (its result, one ulp up, one ulp down)."""
import math

import numpy as np

f = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * np.float64(b) + np.asarray(c, dtype=np.float64)).astype(np.float32)


def km_erff(a, nudge=0):
    a = a.astype(np.float32)
    t, s = np.abs(a), (a * a).astype(np.float32)
    r = fma(t, f(-1.72853470e-5), f(3.83197126e-4))
    u = fma(t, f(-3.88396438e-3), f(2.42546219e-2))
    r = (r.astype(np.float64) * s + u).astype(np.float32)
    for c in (-1.06777877e-1, -6.34846687e-1, -1.28717512e-1):
        r = (r.astype(np.float64) * t + f(c)).astype(np.float32)
    r = (r.astype(np.float64) * t - t).astype(np.float32)
    e = np.exp2((r * f(1.4426950408889634)).astype(np.float32).astype(np.float64)).astype(np.float32)
    if nudge:
        e = np.nextafter(e, f(np.inf * nudge))
    big = np.copysign((f(1.0) - e).astype(np.float32), a)
    q = np.full_like(a, f(-5.96761703e-4))
    for c in (4.99119423e-3, -2.67681349e-2, 1.12819925e-1, -3.76125336e-1, 1.28379166e-1):
        q = (q.astype(np.float64) * s + f(c)).astype(np.float32)
    small = (q.astype(np.float64) * a + a).astype(np.float32)
    return np.where(t > f(0.927734375), big, small)


def test_polynomial_erf_is_within_one_and_a_half_ulp():
    x = np.concatenate([np.linspace(-6, 6, 200001), np.random.default_rng(0).normal(size=100000) * 2,
                        np.array([0.0, -0.0, 0.927734375, 0.92773443, 1e-20, -1e-20, 30.0, -30.0])]).astype(np.float32)
    ref = np.array([math.erf(float(v)) for v in x])
    ulp = np.spacing(np.maximum(np.abs(ref), 1e-30).astype(np.float32)).astype(np.float64)
    for nudge in (0, 1, -1):
        got = km_erff(x, nudge).astype(np.float64)
        assert (np.abs(got - ref) / ulp).max() < 1.5
    assert np.isnan(km_erff(np.array([np.nan], dtype=np.float32))[0])
