"""CPU check of the algebra behind a BLOCKED covariance recursion (round-5 experiment; numpy prototype
tests/tools/proto/blocked_window.py): a chunk of the co-rotating-frame recursion as ONE symmetric elimination (LDL^T of the
chunk's predictive covariance, gains by back-substitution, one rank-m downdate of S), and the same elimination laid out
one lane per datum with the columns of S riding along as virtual lanes, give the oracle's log-likelihood -- on the README
fixture, the OGLE order grid, the ill-conditioned set, a series with tiny measurement errors, and with chunks cut by
re-base data.  tools/window_proto_report.py prints the error table kept as profiles/r05/blocked_proto_v1.txt."""
import os
import sys

import numpy as np
import pytest

import oracle as orc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "proto"))
from blocked_window import loglik_ldl, loglik_window  # noqa: E402
from carma_pack_amd.synth import theta_batch  # noqa: E402

TOL = 1e-12


def _check(t, y, yerr, theta, p, q, tol=TOL, **kw):
    m = orc.OracleModel(t, y, yerr, p, q, max_stdev=1e300)
    ref = m.logdensity(theta, ignore_prior=True)
    if not np.isfinite(ref):
        return None
    want = ref - m.log_prior(theta)
    out = []
    for f, mc in ((loglik_ldl, 16), (loglik_window, None)):
        st = []
        with np.errstate(all="ignore"):
            got = f(t, y, yerr, theta, p, q, mchunk=mc, stats=st, **kw)
        err = abs(got - want) / max(1.0, abs(want))
        assert err <= tol, (f.__name__, p, q, got, want, err)
        out.append((err, st[0]))
    return out


def test_readme_fixture_and_perturbed_parameters(golden_dir):
    g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    th = np.concatenate([g["theta"][:8], theta_batch(np.random.default_rng(3), 8, 5, 3, t, y, theta_center=g["theta"][0])])
    n = sum(_check(t, y, yerr, x, 5, 3) is not None for x in th)
    assert n >= 12


def test_chunks_cut_by_rebase_data(golden_dir):
    """Small window limits: re-base data every few steps, so chunks of every length 1..m occur."""
    g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    for lim in (0.0, 0.05, 0.5, 5.0):
        out = _check(t, y, yerr, g["theta"][0], 5, 3, tol=1e-11, lim_re=lim, lim_im=1e9)
        assert out is not None
        if lim == 0.0:
            assert out[0][1] == t.size - 1                   # every datum re-based: the stepwise recursion
        if lim == 0.5:
            assert 10 < out[0][1] < t.size - 1


def test_ogle_order_grid(golden_dir):
    g = np.load(os.path.join(golden_dir, "ogle_grid.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    n = 0
    for p in range(2, 8):
        for q in (0, p - 1):
            for x in g["p%dq%d_theta" % (p, q)][:2]:
                n += _check(t, y - y.mean(), yerr, x, p, q, tol=1e-11) is not None
    assert n >= 16


def test_ill_conditioned_and_tiny_errors(golden_dir):
    g = np.load(os.path.join(golden_dir, "illcond_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    n = 0
    for i in range(0, g["theta"].shape[0], 3):
        p, q = int(g["p"][i]), int(g["q"][i])
        # the blocked forms must be as close to the oracle as the stepwise co-rotating recursion is on these
        # (cond up to 1e13: the oracle's own distance from the exact value is 1e-10 ... 1e-4 there)
        n += _check(t, y, yerr, g["theta"][i][:3 + p + q], p, q, tol=1e-6) is not None
    assert n >= 8
    r = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    assert _check(r["t"], r["y"], np.full(r["t"].size, 1e-6), r["theta"][0], 5, 3, tol=1e-9) is not None
