"""CPU check of the algebra behind the TWO-SIDED filter (round 6; numpy prototype tests/tools/proto/two_sided.py): the first half
of the series filtered forward, the second half by the same recursion run backward in the dual coordinates (h and c exchanged,
conjugate roots), and the two states merged at the meeting time -- plain (N = I - Da Db), through the congruence the device uses
(merge_chol) and in the device's lane layout (lane_merge_chol) -- give the oracle's log-likelihood: README fixture, OGLE order
grid, configs[3]'s 10^4-point series, a meeting point inside a season gap, every split position of a short series, tiny
measurement errors; on the ill-conditioned set and on roots 1e-3 ... 1e-6 apart the congruence form stays with the one-pass
recursion's distance from the exact (quad-precision) value, which the plain form does not."""
import os
import sys

import numpy as np
import pytest

import oracle as orc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "proto"))
import two_sided as ts  # noqa: E402
from carma_pack_amd.synth import theta_batch  # noqa: E402
from helpers import loglik_truth  # noqa: E402

FORMS = (("plain", ts.merge), ("congruence", ts.merge_chol), ("lanes", ts.lane_merge_chol))


def _want(t, y, yerr, theta, p, q):
    m = orc.OracleModel(t, y, yerr, p, q, max_stdev=1e300)
    ref = m.logdensity(theta, ignore_prior=True)
    return ref - m.log_prior(theta) if np.isfinite(ref) else None


def _check(t, y, yerr, theta, p, q, tol, forms=FORMS, **kw):
    want = _want(t, y, yerr, theta, p, q)
    if want is None:
        return 0
    for name, f in forms:
        with np.errstate(all="ignore"):
            got = ts.loglik_two_sided(t, y, yerr, theta, p, q, merge_fn=f, **kw)
        err = abs(got - want) / max(1.0, abs(want))
        assert err <= tol, (name, p, q, kw, got, want, err)
    return 1


def test_readme_fixture_and_perturbed_parameters(golden_dir):
    g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    th = np.concatenate([g["theta"][:8], theta_batch(np.random.default_rng(3), 24, 5, 3, t, y, theta_center=g["theta"][0])])
    assert sum(_check(t, y, yerr, x, 5, 3, 1e-12) for x in th) >= 24


def test_meeting_point_in_a_season_gap_and_every_split(golden_dir):
    g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    assert t[90] - t[89] > 150 and t[180] - t[179] > 150              # three seasons of 90
    for m in (90, 180):
        for where in ("left", "mid", "right"):
            assert _check(t, y, yerr, g["theta"][0], 5, 3, 1e-12, m=m, where=where)
    for m in range(1, 40):                                             # forward side shorter than the order, and longer
        assert _check(t[:40], y[:40], yerr[:40], g["theta"][1], 5, 3, 1e-12, m=m)


def test_ogle_order_grid(golden_dir):
    g = np.load(os.path.join(golden_dir, "ogle_grid.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    n = 0
    for p in range(2, 8):
        for q in (0, p - 1):
            for x in g["p%dq%d_theta" % (p, q)][:2]:
                n += _check(t, y - y.mean(), yerr, x, p, q, 1e-11)
    assert n >= 16


def test_config3_long_series(golden_dir):
    g = np.load(os.path.join(golden_dir, "config3_carma76_n10000.npz"))
    assert _check(g["t"], g["y"], g["yerr"], g["theta"][0], 7, 6, 1e-12, forms=FORMS[1:])


def test_tiny_measurement_errors(golden_dir):
    r = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    assert _check(r["t"], r["y"], np.full(r["t"].size, 1e-6), r["theta"][0], 5, 3, 1e-9)


def test_ill_conditioned_never_far_from_the_one_pass_recursion(golden_dir):
    """Distance from the exact value (oracle/carma_truth_q.c): the congruence form within 3x of the one-pass recursion in the same
    coordinates (or at 1e-11); the plain form is NOT (asserted on the closest roots, so that the reason for the congruence stays
    on record)."""
    g = np.load(os.path.join(golden_dir, "illcond_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    cases = [(int(g["p"][i]), int(g["q"][i]), g["theta"][i][:3 + int(g["p"][i]) + int(g["q"][i])]) for i in range(0, g["theta"].shape[0], 2)]
    r = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    th0 = r["theta"][0]
    for eps in (1e-3, 1e-4, 1e-5, 1e-6):
        x = th0.copy()
        x[5:7] = th0[3:5] + eps
        cases.append((5, 3, x))
    worst_plain = 0.0
    for p, q, th in cases:
        truth = loglik_truth(t, y, yerr, th, p, q)[1]
        if not np.isfinite(truth):
            continue
        with np.errstate(all="ignore"):
            one = abs(ts.loglik_one_pass(t, y, yerr, th, p, q) - truth) / abs(truth)
            for name, f in FORMS[1:]:
                got = abs(ts.loglik_two_sided(t, y, yerr, th, p, q, merge_fn=f) - truth) / abs(truth)
                assert got <= max(3.0 * one, 1e-11), (name, p, q, got, one)
            worst_plain = max(worst_plain, abs(ts.loglik_two_sided(t, y, yerr, th, p, q, merge_fn=ts.merge) - truth) / abs(truth) / max(one, 1e-11))
    assert worst_plain > 30.0
