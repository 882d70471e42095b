"""CPU-only: the sampler core (carma_pt_core.h + carma_rng.h, the source the gfx950 kernel
compiles) on the lane emulator: RNG distributions, the stored-vs-recomputed log-posterior
invariant the reference tests (carma_unit_tests.cpp:783-915), RAM acceptance-rate coercion,
exchange bookkeeping and a small posterior-recovery run (carma_unit_tests.cpp:1319-1376)."""
import os

import numpy as np
import pytest
from scipy import stats

import emu_build as emu
import oracle as orc
from helpers import prior_like_theta


def test_rng_distributions():
    t8, u = emu.rng_draws(seed=12345, chain=7, n=200000)
    assert stats.kstest(u, "uniform").pvalue > 1e-3
    assert stats.kstest(t8, stats.t(8).cdf).pvalue > 1e-3
    assert abs(np.var(t8) - 8.0 / 6.0) < 0.03
    t8b, ub = emu.rng_draws(seed=12345, chain=8, n=1000)
    assert not np.allclose(t8b, t8[:1000]) and abs(np.corrcoef(ub, u[:1000])[0, 1]) < 0.15
    t8c, _ = emu.rng_draws(seed=12345, chain=7, n=1000)
    assert np.array_equal(t8c, t8[:1000])          # counter-based: reproducible


def _start(m, p, q, t, y, rng, T):
    th, lp = [], []
    while len(th) < T:
        x = prior_like_theta(rng, p, q, t, y)
        v = m.logdensity(x)
        if np.isfinite(v):
            th.append(x)
            lp.append(v)
    return np.array(th), np.array(lp)


def _chol0(d, y, n, T):
    var = np.mean(y * y) - np.mean(y) ** 2
    R = np.eye(d) * 0.01
    R[0, 0] = np.sqrt(2 * var * var / n)
    R[2, 2] = np.sqrt(var / n)
    return np.tile(R, (T, 1, 1))


def test_car1_posterior_recovery(golden_dir):
    g = np.load(os.path.join(golden_dir, "car1_n100.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    m = orc.OracleModel(t, y, yerr, 1)
    pr = (m.max_stdev, m.max_freq, m.min_freq)
    rng = np.random.default_rng(5)
    th0, lp0 = _start(m, 1, 0, t, y, rng, 1)
    nb, ns = 4000, 4000
    out = emu.pt_run(t, y, yerr, 1, 0, pr, [1.0], nb, nb + ns, nb, 1, 99, th0, lp0, _chol0(4, y, t.size, 1))
    S = out["samples"]
    # stored log-posterior == LogDensity(sample)   (carma_unit_tests.cpp:917-1114, rel 1e-8)
    idx = np.arange(0, ns, 97)
    ref = m.logdensity_batch(S[idx])
    np.testing.assert_allclose(out["logpost"][idx], ref, rtol=1e-10)
    # truth: sigma_y = 2.3, mu = 0, ln omega = ln 0.01 ; within 4 posterior sd (the reference uses 3)
    truth = np.array([2.3, 1.0, 0.0, np.log(0.01)])
    mean, sd = S.mean(0), S.std(0)
    z = np.abs(mean - truth) / sd
    assert np.all(z[[0, 2, 3]] < 4.0), (mean, sd, z)
    # RAM coerces the acceptance rate towards 0.25 during adaptation
    rate = out["nacc"][0] / (nb + ns)
    assert 0.15 < rate < 0.40, rate


def test_carma21_tempered_invariants():
    rng = np.random.default_rng(8)
    n = 60
    t = np.cumsum(rng.uniform(0.5, 1.5, n))
    y = np.sin(t / 3.0) + 0.3 * rng.standard_normal(n)
    yerr = np.full(n, 0.3)
    p, q, T = 2, 1, 3
    m = orc.OracleModel(t, y, yerr, p, q)
    pr = (m.max_stdev, m.max_freq, m.min_freq)
    th0, lp0 = _start(m, p, q, t, y, rng, T)
    temps = np.exp(np.linspace(0, np.log(100.0), T))
    nb, ns = 1500, 1500
    out = emu.pt_run(t, y, yerr, p, q, pr, temps, nb, nb + ns, nb, 1, 2024, th0, lp0, _chol0(6, y, n, T))
    # final chain states carry their own log-posterior (after swaps)
    np.testing.assert_allclose(out["lp"], m.logdensity_batch(out["theta"]), rtol=1e-10)
    idx = np.arange(0, ns, 53)
    np.testing.assert_allclose(out["logpost"][idx], m.logdensity_batch(out["samples"][idx]), rtol=1e-10)
    assert np.all(np.isfinite(out["logpost"]))
    # exchanges happen, hotter chains accept at comparable (coerced) rates
    assert out["nswap"][1:].sum() > 50
    rates = out["nacc"] / (nb + ns)
    assert np.all((rates > 0.1) & (rates < 0.5)), rates
    # proposal factor stayed upper triangular and positive on the diagonal
    for R in out["chol"]:
        assert np.allclose(R, np.triu(R)) and np.all(np.diag(R) > 0)
    # reproducible
    out2 = emu.pt_run(t, y, yerr, p, q, pr, temps, nb, 200, nb, 1, 2024, th0, lp0, _chol0(6, y, n, T))
    out3 = emu.pt_run(t, y, yerr, p, q, pr, temps, nb, 200, nb, 1, 2024, th0, lp0, _chol0(6, y, n, T))
    assert np.array_equal(out2["theta"], out3["theta"])
