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
    from helpers import loglik_truth
    for ign in (False, True):
        got = emu.logdensity_carma(t, y, yerr, p, q, th, pr, ignore_prior=ign)
        assert_parity(got, m.logdensity_batch(th, ignore_prior=ign), 1e-10, "emu p=%d q=%d" % (p, q),
                      arbiter=lambda i: loglik_truth(t, y, yerr, th[i], p, q)[0])


@pytest.mark.parametrize("p,q", [(2, 1), (3, 0), (4, 3), (5, 3), (6, 2), (7, 6)])
def test_emu_row_loop(p, q, golden_dir):
    """filter_loop_row (one evaluation per 16-lane DPP row, the latency-regime kernel) against the
    oracle, and against the G = 8 loop it must agree with to rounding."""
    if (p, q) == (5, 3):
        g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
        t, y, yerr, th = g["t"], g["y"], g["yerr"], g["theta"]
    else:
        t, y, yerr = irregular_series(90, seed=p * 7 + q)
        rng = np.random.default_rng(p * 31 + q)
        th = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(8)])
    m = orc.OracleModel(t, y, yerr, p, q)
    pr = (m.max_stdev, m.max_freq, m.min_freq)
    from helpers import loglik_truth
    got = emu.logdensity_carma_row(t, y, yerr, p, q, th, pr)
    assert_parity(got, m.logdensity_batch(th), 1e-10, "emu row p=%d q=%d" % (p, q),
                  arbiter=lambda i: loglik_truth(t, y, yerr, th[i], p, q)[0])
    ref = emu.logdensity_carma(t, y, yerr, p, q, th, pr)
    fin = np.isfinite(ref)
    assert np.array_equal(fin, np.isfinite(got))
    assert np.allclose(got[fin], ref[fin], rtol=1e-9, atol=0)


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


def test_emu_predict_vs_oracle_and_golden(golden_dir):
    """carma_predict.h (uniform n+1-point walk with per-group phase selection) vs the literal
    restatement of KalmanFilterp::Predict and the reference-Python golden vectors."""
    g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    pr = np.load(os.path.join(golden_dir, "predict.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    times = np.r_[pr["times"], pr["back"], t[0], t[5]]
    for tag in ("true", "th17"):
        mu, scale = float(pr[tag + "_mu"]), float(pr[tag + "_scale"])
        args = (t, y - mu, np.sqrt(scale) * yerr, float(pr[tag + "_sigsqr"]), pr[tag + "_omega"], pr[tag + "_ma"])
        em, ev = emu.predict_carma(*args, times)
        om, ov = orc.predict_carma(*args, times)
        np.testing.assert_allclose(em, om, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(ev, ov, rtol=1e-9)
        k = pr["times"].size
        np.testing.assert_allclose(em[:k], pr[tag + "_pmean"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(ev[:k + 2], pr[tag + "_dvar"], rtol=1e-6)
    c = np.load(os.path.join(golden_dir, "car1_n100.npz"))
    th = pr["car1_theta"]
    w = np.exp(th[3])
    em, ev = emu.predict_car1(c["t"], c["y"] - th[2], np.sqrt(th[1]) * c["yerr"], 2 * th[0] ** 2 * w, w, pr["car1_times"])
    np.testing.assert_allclose(em, pr["car1_dmean"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(ev, pr["car1_dvar"], rtol=1e-8)


@pytest.mark.parametrize("p,q", [(2, 1), (3, 0), (4, 2), (6, 3), (7, 6)])
def test_emu_predict_orders(p, q):
    t, y, yerr = irregular_series(60, seed=p + q)
    rng = np.random.default_rng(p * 7 + q)
    th = prior_like_theta(rng, p, q, t, y)
    om, ma = orc.ar_roots(th, p), orc.ma_coefs(th, p, q)
    sig = th[0] ** 2 / orc.variance(om, ma)
    times = np.r_[t[0] - 1.0, rng.uniform(t[0], t[-1], 5), t[-1] + 2.0]
    a = emu.predict_carma(t, y - th[2], yerr, sig, om, ma, times)
    b = orc.predict_carma(t, y - th[2], yerr, sig, om, ma, times)
    np.testing.assert_allclose(a[0], b[0], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-8)


@pytest.mark.parametrize("p,q", [(2, 0), (2, 1), (3, 2), (4, 1), (4, 3), (5, 3), (6, 2), (6, 5), (7, 3), (7, 6)])
def test_lane_code_matches_oracle(p, q, golden_dir):
    """carma_lane.h -- one evaluation per lane, the throughput regime's kernel body (plain scalar code, so the host runs
    the very functions the GPU compiles) -- against the oracle on prior-like and golden parameter vectors, with the bounds
    on and off, real root pairs included, and against the lane-group loop it must agree with to rounding."""
    from helpers import loglik_truth
    if (p, q) == (5, 3):
        g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
        t, y, yerr, th = g["t"], g["y"], g["yerr"], g["theta"]
    else:
        t, y, yerr = irregular_series(130, seed=p * 13 + q)
        rng = np.random.default_rng(p * 57 + q)
        th = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(24)])
        # quadratic factors with two REAL roots (b^2 > 4 a): first factor, then every factor
        th[-1, 3:5] = [np.log(0.02), np.log(0.5)]
        for i in range(p // 2):                               # well separated real roots r1 = -0.02 3^i, r2 = -0.5 1.7^i
            r1, r2 = 0.02 * 3.0 ** i, 0.5 * 1.7 ** i
            th[-2, 3 + 2 * i:5 + 2 * i] = [np.log(r1 * r2), np.log(r1 + r2)]
    m = orc.OracleModel(t, y, yerr, p, q)
    pr = (m.max_stdev, m.max_freq, m.min_freq)
    for ign in (False, True):
        got = emu.logdensity_carma_lane(t, y, yerr, p, q, th, pr, ignore_prior=ign)
        want = m.logdensity_batch(th, ignore_prior=ign)
        assert_parity(got, want, 1e-10, "lane p=%d q=%d" % (p, q), arbiter=lambda i: loglik_truth(t, y, yerr, th[i], p, q)[0],
                      max_arb_frac=0.1)
        ref = emu.logdensity_carma(t, y, yerr, p, q, th, pr, ignore_prior=ign)
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(got), fin)
        # (the constructed all-real-roots vectors are ill-conditioned: there the two layouts' roundings are amplified to 1e-8,
        # each closer to the exact value than the oracle -- the arbiter above has seen them)
        rel = np.abs(got[fin] - ref[fin]) / np.abs(ref[fin])
        assert np.all(rel <= 1e-6) and np.sum(rel > 1e-9) <= 2, rel


def test_lane_code_regular_cadence():
    """A series with a constant time step (and gaps): the lane code re-uses the transition factors of a repeated step."""
    rng = np.random.default_rng(3)
    t = np.concatenate([np.arange(60) * 1.5, 200.0 + np.arange(70) * 1.5])
    y = 17.0 + np.sin(t / 9.0) + 0.4 * rng.standard_normal(t.size)
    yerr = np.full(t.size, 0.4)
    for p, q in ((5, 3), (3, 1)):
        th = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(10)])
        m = orc.OracleModel(t, y, yerr, p, q)
        pr = (m.max_stdev, m.max_freq, m.min_freq)
        from helpers import loglik_truth
        assert_parity(emu.logdensity_carma_lane(t, y, yerr, p, q, th, pr), m.logdensity_batch(th), 1e-10, "lane regular cadence",
                      arbiter=lambda i: loglik_truth(t, y, yerr, th[i], p, q)[0], max_arb_frac=0.2)


def test_table_math_accuracy():
    """The table-based exp / complex exponential of carma_math.h (round 4: arguments reduced by ln 2 / 32 and pi / 32, coarse part
    from correctly rounded tables, short polynomials) against QUAD precision on a million arguments spread over the decades
    the prior admits -- the host build of the very functions the kernels compile (tests/tools/proto/table_math_accuracy.cpp)."""
    import ctypes as C
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(here, "tools", "proto", "table_math_accuracy.cpp")
    so = os.path.join(here, "emu", "libtable_math.so")
    csrc = os.path.join(os.path.dirname(here), "carma_pack_amd", "csrc")
    deps = [src, os.path.join(csrc, "carma_math.h"), os.path.join(csrc, "carma_math_tab.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-mfma", "-DTABLE_MATH_NO_MAIN", "-I", csrc, "-o", so, src,
                               "-lquadmath"])
    lib = C.CDLL(so)
    out = (C.c_double * 4)()
    assert lib.table_math_accuracy(1000000, 7, out) == 0
    exp_tab, exp_old, cexp_tab, cexp_old = list(out)
    print("max error (ulp): exp_neg_tab %.2f (exp_neg %.2f), cexp_step_tab %.2f (cexp_step %.2f)" % (exp_tab, exp_old, cexp_tab, cexp_old))
    assert exp_tab < 2.0 and cexp_tab < 3.6
    assert cexp_tab <= cexp_old + 0.25                       # no worse than the polynomial-only form it replaces


def test_math_tables_are_the_generators_output():
    """carma_math_tab.h is generated (tools/gen_math_tables.py, mpmath): regenerate IN MEMORY and compare -- the test never
    writes into csrc/ (a rewritten header would change source_id and rebuild every object)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_math_tables.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_window_asm_header_is_the_generators_output():
    """carma_win_asm.h (the window pipeline's pivot and chunk-start blocks, one inline-asm statement each) is generated by
    tools/gen_win_asm.py: regenerated IN MEMORY and compared (round-5 advice) -- the test never writes into csrc/."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_win_asm.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
