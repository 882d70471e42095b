"""Synthetic parameter vectors for benchmarks, tests and sampler start-up.

``prior_like_theta`` draws from the reference's starting-value distribution
(CARMA::StartingValue / CARp::StartingAR / CARMA::StartingMA / CAR1::StartingValue,
src/carpack.cpp:38-81,268-311,416-477,515-519) with a numpy Generator instead of the
reference's time-seeded global mt19937.
"""
import numpy as np


def log_quads_from_roots(roots):
    """Inverse of CARp::ARRoots (src/carpack.cpp:137-172): conjugate pairs (negative imaginary part
    first) followed by an optional real root -> log quadratic-factor coefficients."""
    p = len(roots)
    out = []
    for i in range(p // 2):
        r = roots[2 * i]
        out += [np.log(abs(r) ** 2), np.log(-2.0 * r.real)]
    if p % 2:
        out.append(np.log(-roots[-1].real))
    return np.array(out)


def prior_like_theta(rng, p, q, t, y, measerr_dof=50):
    n = y.size
    dt = np.diff(np.sort(t))
    dt = dt[dt > 0]
    max_freq, min_freq = 1.0 / dt.min(), 1.0 / (t.max() - t.min())
    yvar = np.var(y, ddof=1) * (n - 1) / rng.chisquare(n - 1)       # scaled-inv-chi2(n-1, var y)
    mu = rng.normal(np.mean(y), np.sqrt(yvar) / n)
    scale = min(max(measerr_dof / rng.chisquare(measerr_dof), 0.51), 1.99)
    if p == 1:
        log_omega = -np.log(np.median(dt) * rng.uniform(1.0, 50.0))
        log_omega = min(log_omega, max_freq)          # sic: carpack.cpp:56 compares a log with a frequency
        return np.array([np.sqrt(yvar), scale, mu, log_omega])
    nc = (p + 1) // 2
    cent = np.exp(np.log(max_freq / min_freq) * rng.uniform(size=nc) + np.log(min_freq))
    cent = np.sort(cent)[::-1]
    width = np.exp(np.log(max_freq / min_freq) * rng.uniform(size=nc) + np.log(min_freq))
    loga = np.empty(p)
    if p % 2:
        cent[p // 2] = 0.0
        # (a two-point series has max_freq == min_freq, and exp(log(.)) may come back an ulp below: keep the interval ordered)
        lo = np.log(min_freq)
        width[p // 2] = np.exp(rng.uniform(lo, max(lo, np.log(cent[p // 2 - 1]))))
    for i in range(p // 2):
        re_, im_ = -2 * np.pi * width[i], 2 * np.pi * cent[i]
        loga[2 * i] = np.log(re_ * re_ + im_ * im_)
        loga[2 * i + 1] = np.log(-2.0 * re_)
    if p % 2:
        loga[p - 1] = np.log(2 * np.pi * width[p // 2])
    ma = np.abs(rng.standard_normal(q))
    return np.concatenate([[np.sqrt(yvar), scale, mu], loga, ma])


def theta_batch(rng, B, p, q, t, y, theta_center=None, frac_post=0.5):
    """Mixed batch: posterior-like (centre + 0.01 N(0,I)) and prior-like draws (BASELINE config 2)."""
    out = np.empty((B, 4 if p == 1 else 3 + p + q))
    for b in range(B):
        if theta_center is not None and rng.uniform() < frac_post:
            out[b] = theta_center + 0.01 * rng.standard_normal(theta_center.size)
        else:
            out[b] = prior_like_theta(rng, p, q, t, y)
    return out


def irregular_series(n, seed):
    """Irregularly sampled quasi-periodic series with heteroscedastic errors (not a CARMA draw)."""
    rng = np.random.default_rng(seed)
    t = np.cumsum(rng.uniform(1.0, 3.0, n))
    y = 17.0 + 2.3 * np.sin(t / 7.0) + 1.1 * np.cos(t / 31.0) + 0.6 * rng.standard_normal(n)
    yerr = np.full(n, 0.45) * rng.uniform(0.8, 1.2, n)
    return t, y, yerr


def config4_model():
    """BASELINE configs[3]: CARMA(7,6) with three quasi-periodic pairs and one real root.  Returns
    (ar_roots[7], ma_coefs[7], sigma_y)."""
    from .carma_pack import get_ar_roots
    # centroids in descending order: the prior bounds want them non-increasing (carpack.cpp:353-362)
    ar_roots = get_ar_roots(np.array([1.0 / 50.0, 1.0 / 100.0, 1.0 / 300.0, 1.0 / 500.0]),
                            np.array([1.0 / 2.0, 1.0 / 5.0, 1.0 / 25.0]))
    ma_roots = -2.0 * np.pi * np.array([0.3 + 0.4j, 0.3 - 0.4j, 0.08 + 0.15j, 0.08 - 0.15j, 0.9 + 0.0j, 0.02 + 0.0j])
    c = np.poly(ma_roots)                       # highest order first
    ma_coefs = np.real(c / c[-1])[::-1]         # constant term 1, lowest order first
    return ar_roots, ma_coefs, 2.3


def config4_series(n=10000, seed=4):
    """BASELINE configs[3] input: a long irregular series -- time steps 0.1 + |Cauchy| as the reference's
    cpp_tests/generate_test_data.py:17-19 draws them, heteroscedastic errors as :13, and a CARMA(7,6) path drawn by
    this package's own carma_process.  Returns (t, y, yerr, theta_true)."""
    from .carma_pack import carma_process, carma_variance
    rng = np.random.default_rng(seed)
    ar_roots, ma_coefs, sigmay = config4_model()
    yerr = 0.1 * sigmay * np.sqrt(10.0 / rng.chisquare(10.0, n))
    t = np.cumsum(0.1 + np.abs(rng.standard_cauchy(n)))
    t = t - t.min()
    sigsqr = sigmay ** 2 / carma_variance(1.0, ar_roots, ma_coefs)
    y = carma_process(t, sigsqr, ar_roots, ma_coefs, rng=rng) + rng.normal(0.0, yerr)
    ma_roots = np.roots(ma_coefs[::-1])
    # MA log-quadratic parameters in the reference's pairing (conjugate pairs first, real roots paired in order)
    cp_ = sorted([r for r in ma_roots if r.imag < -1e-12], key=lambda r: r.real)
    re_ = sorted([r.real for r in ma_roots if abs(r.imag) <= 1e-12])
    lq = []
    for r in cp_:
        lq += [np.log(abs(r) ** 2), np.log(-2.0 * r.real)]
    for a, b in zip(re_[0::2], re_[1::2]):
        lq += [np.log(a * b), np.log(-(a + b))]
    theta = np.concatenate([[sigmay, 1.0, 0.0], log_quads_from_roots(ar_roots), lq])
    return t, y, yerr, theta
