"""CPU-only: runs the kernel-core source (carma_pack_amd/csrc/carma_core.h -- the same code the
gfx950 kernels compile) on a thread-per-lane emulator and checks it against the oracle and the
golden vectors.  This validates the restructured recursion (D = P - V, row-per-lane layout,
lane-distributed LU, mantissa-product log accumulation) without a GPU; it is a test harness,
not a fallback."""
import os

import numpy as np
import pytest

import emu_build as emu
import oracle as orc
from helpers import assert_parity, irregular_series, prior_like_theta


def test_emu_readme_vs_oracle_and_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    m = orc.OracleModel(t, y, yerr, 5, 3)
    pr = (m.max_stdev, m.max_freq, m.min_freq)
    got = emu.logdensity_carma(t, y, yerr, 5, 3, g["theta"], pr)
    assert_parity(got, m.logdensity_batch(g["theta"]), 1e-11, "emu vs oracle")
    ll = emu.logdensity_carma(t, y, yerr, 5, 3, g["theta"], pr, ignore_prior=True) - \
        np.array([m.log_prior(th) for th in g["theta"]])
    assert_parity(ll, g["loglik"], 1e-11, "emu vs golden")


@pytest.mark.parametrize("p,q", [(2, 0), (2, 1), (3, 2), (4, 1), (5, 0), (6, 5), (7, 3), (7, 6)])
def test_emu_orders(p, q):
    t, y, yerr = irregular_series(120, seed=p * 10 + q)
    rng = np.random.default_rng(p * 100 + q)
    th = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(12)])
    m = orc.OracleModel(t, y, yerr, p, q)
    pr = (m.max_stdev, m.max_freq, m.min_freq)
    for ign in (False, True):
        got = emu.logdensity_carma(t, y, yerr, p, q, th, pr, ignore_prior=ign)
        assert_parity(got, m.logdensity_batch(th, ignore_prior=ign), 1e-10, "emu p=%d q=%d" % (p, q))


def test_emu_kfilter_mean_var(golden_dir):
    g = np.load(os.path.join(golden_dir, "cpp_carma_test300.npz"))
    mean, var, ll, rc = emu.kfilter_carma(g["t"], g["y"], g["yerr"], float(g["sigsqr"]), g["omega"], g["ma"])
    assert rc == 0
    np.testing.assert_allclose(var, g["var"], rtol=1e-10)
    np.testing.assert_allclose(mean, g["mean"], rtol=0, atol=1e-10)
    r = g["y"] - mean
    assert abs(ll - np.sum(-0.5 * np.log(var) - 0.5 * r * r / var)) < 1e-10 * abs(ll)


def test_emu_car1(golden_dir):
    g = np.load(os.path.join(golden_dir, "car1_n100.npz"))
    m = orc.OracleModel(g["t"], g["y"], g["yerr"], 1)
    got = emu.logdensity_car1(g["t"], g["y"], g["yerr"], g["theta"], (m.max_stdev, m.max_freq, m.min_freq))
    assert_parity(got, m.logdensity_batch(g["theta"]), 1e-12, "emu car1")


def test_emu_singular_and_bounds(golden_dir):
    g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    m = orc.OracleModel(t, y, yerr, 5, 3)
    pr = (m.max_stdev, m.max_freq, m.min_freq)
    th = g["theta"][0].copy()
    th[5:7] = th[3:5]                      # repeated roots
    assert emu.logdensity_carma(t, y, yerr, 5, 3, th, pr)[0] == -np.inf
    assert not np.isfinite(emu.logdensity_carma(t, y, yerr, 5, 3, th, pr, ignore_prior=True)[0])
    th = g["theta"][0].copy()
    th[1] = 2.5
    assert emu.logdensity_carma(t, y, yerr, 5, 3, th, pr)[0] == -np.inf
