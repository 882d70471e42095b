"""GPU tests of the `_carmcmc` drop-in and the Python API; they read like the reference's own
src/tests/testCarmcmc.py (n = 10 series, tiny samplers, binding overloads and defaults)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
from batched_opt_proto import minimize_batched  # noqa: E402  (numpy prototype of the library's optimiser)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cm():
    import carmcmc
    assert carmcmc._carmcmc._lib.lib.carma_device_count() >= 1
    return carmcmc


@pytest.fixture(scope="module")
def data(cm):
    rng = np.random.RandomState(1)
    npts = 10
    x = 1.0 * np.arange(npts)
    ar_roots = np.array([-0.06283185 - 1.25663706j, -0.06283185 + 1.25663706j, -0.02094395 - 0.25132741j,
                         -0.02094395 + 0.25132741j, -0.03141593 + 0.j])
    sigsqr = 0.00126811439419
    y = cm.carma_process(x, sigsqr, ar_roots, rng=rng)
    dy = np.sqrt(sigsqr) * np.ones(npts)
    xv, yv, dv = cm.vecD(), cm.vecD(), cm.vecD()
    xv.extend(x)
    yv.extend(y)
    dv.extend(dy)
    return dict(x=x, y=y, dy=dy, xv=xv, yv=yv, dv=dv, nSample=100, nBurnin=10, nThin=1, nWalkers=2)


def _first_sample_consistent(sampler, cm):
    s = np.array(sampler.getSamples())
    ll = np.array(sampler.GetLogLikes())
    v = cm.vecD()
    v.extend(s[0])
    assert np.isfinite(sampler.getLogPrior(v))
    assert abs(ll[0] - sampler.getLogDensity(v)) < 1e-7 * max(1.0, abs(ll[0]))    # testCarmcmc.py:46,69,104
    return s, ll


def test_car1(cm, data):
    d = data
    cpp = cm.run_mcmc_car1(d["nSample"], d["nBurnin"], d["xv"], d["yv"], d["dv"], d["nThin"], seed=1)
    ps = cm.Car1Sample(d["x"], d["y"], d["dy"], cpp)
    assert ps.p == 1
    s, ll = _first_sample_consistent(cpp, cm)
    assert s.shape == (100, 4)
    # defaults / overloads (testCar1Defaults)
    cm.run_mcmc_car1(d["nSample"], d["nBurnin"], d["xv"], d["yv"], d["dv"])
    guess = cpp.getSamples()[0]
    cm.run_mcmc_car1(d["nSample"], d["nBurnin"], d["xv"], d["yv"], d["dv"], d["nThin"], guess)


def test_carp(cm, data):
    d = data
    sampler = cm.run_mcmc_carma(d["nSample"], d["nBurnin"], d["xv"], d["yv"], d["dv"], 3, 0, d["nWalkers"], False,
                                d["nThin"], seed=2)
    assert isinstance(sampler, cm.CARp) and not isinstance(sampler, cm.CARMA)
    ps = cm.CarmaSample(np.array(d["xv"]), np.array(d["yv"]), np.array(d["dv"]), sampler)
    assert ps.p == 3
    _first_sample_consistent(sampler, cm)


def test_carp_defaults(cm, data):
    d = data
    cm.run_mcmc_carma(d["nSample"], d["nBurnin"], d["xv"], d["yv"], d["dv"], 3, 1, d["nWalkers"])
    cm.run_mcmc_carma(d["nSample"], d["nBurnin"], d["xv"], d["yv"], d["dv"], 3, 1, d["nWalkers"], False)
    s = cm.run_mcmc_carma(d["nSample"], d["nBurnin"], d["xv"], d["yv"], d["dv"], 3, 1, d["nWalkers"], False, 1)
    guess = s.getSamples()[0]
    s2 = cm.run_mcmc_carma(d["nSample"], d["nBurnin"], d["xv"], d["yv"], d["dv"], 3, 1, d["nWalkers"], False, 1, guess)
    assert len(s2.getSamples()) == d["nSample"]
    with pytest.raises(NotImplementedError):
        cm.run_mcmc_carma(10, 5, d["xv"], d["yv"], d["dv"], 3, 1, 2, True)
    with pytest.raises(RuntimeError):
        cm.run_mcmc_carma(10, 5, d["xv"], d["yv"], d["dv"], 1, 0, 2)


def test_carpq(cm, data):
    d = data
    sampler = cm.run_mcmc_carma(d["nSample"], d["nBurnin"], d["xv"], d["yv"], d["dv"], 3, 2, d["nWalkers"], False,
                                d["nThin"], seed=3)
    assert isinstance(sampler, cm.CARMA)
    ps = cm.CarmaSample(np.array(d["xv"]), np.array(d["yv"]), np.array(d["dv"]), sampler)
    assert ps.p == 3 + 2                # quirk: no q= -> p inferred from trace width (testCarmcmc.py:95)
    _first_sample_consistent(sampler, cm)
    ps2 = cm.CarmaSample(np.array(d["xv"]), np.array(d["yv"]), np.array(d["dv"]), sampler, q=2)
    assert ps2.p == 3 and ps2.get_samples("ma_coefs").shape == (100, 3)
    # loglik = log-density with bounds ignored (prior term kept): never below the bounded value
    assert np.all(ps2.get_samples("loglik")[:, 0] >= ps2.get_samples("logpost")[:, 0] - 1e-9)


def test_kalman_filters(cm, data):
    d = data
    kf = cm.KalmanFilter1(d["xv"], d["yv"], d["dv"], 1.0, 1.0)
    kf.Filter()
    var = np.array(kf.GetVar())
    assert abs(var[0] - (1.0 / 2.0 + d["dy"][0] ** 2)) < 1e-12          # var0 = sigsqr/(2 omega) + yerr0^2
    sampler = cm.run_mcmc_carma(50, 10, d["xv"], d["yv"], d["dv"], 4, 0, 2, False, 1, seed=4)
    ps = cm.CarmaSample(np.array(d["xv"]), np.array(d["yv"]), np.array(d["dv"]), sampler)
    sigsqr = float(ps._samples["sigma"][0][0] ** 2)
    ma = cm.vecD()
    ma.extend(np.append(ps._samples["ma_coefs"][0], np.zeros(3)))
    omega = cm.vecC()
    for i in range(ps.p):
        omega.append(ps._samples["ar_roots"][0][i])
    kfp = cm.KalmanFilterp(d["xv"], d["yv"], d["dv"], sigsqr, omega, ma)
    kfp.Filter()
    v = np.array(kfp.GetVar())
    assert v.shape == (10,) and np.all(v > 0)
    # var0 = process variance + yerr^2 (carma_unit_tests.cpp:443-444)
    assert abs(v[0] - (float(ps._samples["var"][0][0]) + d["dy"][0] ** 2)) < 1e-9 * v[0]
    kf2, mu = ps.makeKalmanFilter("map")
    kf2.Filter()
    assert len(kf2.GetMean()) == 10
    # testKalmanp / testKalman1: extrapolation is less certain than the prediction at a datum
    pred0, predN = kfp.Predict(d["xv"][0]), kfp.Predict(d["xv"][-1] + 1)
    assert predN.second > pred0.second
    p1a, p1b = kf.Predict(d["xv"][0]), kf.Predict(d["xv"][-1] + 1)
    assert p1b.second > p1a.second
    sim = kfp.Simulate(cm.vecD([2.5, 11.0, 4.5]))
    assert len(sim) == 3 and np.all(np.isfinite(sim))
    yhat, yvar = ps.predict(np.array([3.3, 12.0]))
    assert yhat.shape == (2,) and np.all(yvar > 0)
    fit = ps.assess_fit(nplot=32)
    assert fit["mean"].shape == (32,) and fit["std_resid"].shape == (10,)


def test_carma_model_mcmc_and_mle(cm, golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    t, y, e = g["t"], g["y"], g["yerr"]
    model = cm.CarmaModel(t, y, e, p=3, q=1)
    sample = model.run_mcmc(300, nburnin=300, nreplicas=4, seed=5)          # ntemperatures default max(10,p+q)
    assert sample.get_samples("ar_roots").shape == (300, 3)
    assert set(("logpost", "var", "measerr_scale", "mu", "quad_coefs", "ar_roots", "psd_centroid", "psd_width",
                "ar_coefs", "ma_coefs", "sigma", "loglik")) <= set(sample.parameters)
    all_s, all_lp = sample._sampler.getAllSamples()
    assert all_s.shape == (4, 300, 7)
    mle = model.get_mle(2, 0, ntrials=8, seed=6)
    assert np.isfinite(mle.fun) and mle.x.size == 5
    # lock-step batched optimiser vs scipy L-BFGS-B from the same starts: same optimum (or better)
    ref = model.get_mle(2, 0, ntrials=8, seed=6, method="scipy")
    assert mle.fun <= ref.fun + 1e-3 * max(1.0, abs(ref.fun))
    m31 = model.get_mle(3, 1, ntrials=16, seed=9)
    r31 = model.get_mle(3, 1, ntrials=16, seed=9, method="scipy")
    assert m31.fun <= r31.fun + 0.5, (m31.fun, r31.fun)
    # the optimiser inside the library (carma_mle_batched) and its numpy prototype are the same algorithm: from the same
    # starts the same optima, start by start (rounding of the host arithmetic differs, so "to 1e-6", and on the odd start
    # a backtracking decision falls the other way)
    for (pp, qq, nt) in ((1, 0, 8), (2, 0, 12), (3, 1, 16)):
        nat = model.get_mle(pp, qq, ntrials=nt, seed=11, return_all=True)
        proc_, starts_, bnds_ = model._mle_problem(pp, qq, nt, 11)
        pyv = minimize_batched(lambda pts: -np.asarray(proc_.getLogDensityBatch(pts)), starts_, bnds_)
        fa, fb = np.array([r.fun for r in nat]), np.array([r.fun for r in pyv])
        ok = (fa < 1e299) & (fb < 1e299)
        assert ok.sum() >= nt - 2
        same = np.abs(fa[ok] - fb[ok]) <= 1e-6 * np.maximum(1.0, np.abs(fb[ok]))
        assert same.sum() >= ok.sum() - 2, (pp, qq, fa, fb)
        assert abs(fa[ok].min() - fb[ok].min()) <= 0.05
        assert all(r.nit > 0 and r.nfev > r.nit and isinstance(r.message, str) for r in nat)
    best, pqlist, aicc = model.choose_order(2, ntrials=4, seed=7)
    assert pqlist == [(1, 0), (2, 0), (2, 1)] and len(aicc) == 3 and (model.p, model.q) in pqlist
    # njobs: the orders driven by a pool of threads (one context and stream per order) -- same searches, same numbers
    chosen = (model.p, model.q)
    best4, pq4, aicc4 = model.choose_order(2, ntrials=4, seed=7, njobs=3)
    assert pq4 == pqlist and aicc4 == aicc and (model.p, model.q) == chosen and np.array_equal(best4.x, best.x)
    sample.add_mle(mle) if sample.p == 2 else None


def test_simulate_is_a_draw_from_the_dense_gp_conditional(cm, golden_dir):
    """KalmanFilter::Simulate pinned the way carma_unit_tests.cpp:651-781 pins it: the simulated path,
    whitened by the conditional mean / covariance of the dense Gaussian process given the data, is
    standard normal white noise (Anderson-Darling < 3.857, ACF inside the 95 % band)."""
    import os
    import oracle as orc
    from scipy.stats import norm
    g = np.load(os.path.join(golden_dir, "cpp_carma_test300.npz"))
    t, y, e = g["t"][:120], g["y"][:120], g["yerr"][:120]
    roots, ma, sigsqr = g["omega"], g["ma"], float(g["sigsqr"])
    om = cm.vecC()
    for r in roots:
        om.append(complex(r))
    kf = cm.KalmanFilterp(cm.vecD(t.tolist()), cm.vecD(y.tolist()), cm.vecD(e.tolist()), sigsqr, om, cm.vecD(ma.tolist()))
    span = t[-1] - t[0]
    nsim = 80
    tsim = np.linspace(t[0] - 0.05 * span, t[-1] + 0.05 * span, nsim)
    np.random.seed(123)                                                   # Simulate draws from numpy's global stream
    ysim = np.array(kf.Simulate(cm.vecD(tsim.tolist())))
    assert ysim.shape == (nsim,) and np.all(np.isfinite(ysim))
    # dense GP conditional of the simulated epochs given the data
    tc = np.concatenate([tsim, t])
    lags, inv = np.unique(np.abs(tc[:, None] - tc[None, :]).ravel(), return_inverse=True)
    acv = np.array([orc.variance(roots, ma, np.sqrt(sigsqr), float(dt)) for dt in lags])
    cov = acv[inv].reshape(tc.size, tc.size)
    cov[np.arange(nsim, tc.size), np.arange(nsim, tc.size)] += e * e
    Kdd, Ksd, Kss = cov[nsim:, nsim:], cov[:nsim, nsim:], cov[:nsim, :nsim]
    sol = np.linalg.solve(Kdd, np.c_[y, Ksd.T])
    cmean, cvar = Ksd @ sol[:, 0], Kss - Ksd @ sol[:, 1:]
    L = np.linalg.cholesky(0.5 * (cvar + cvar.T))
    z = np.linalg.solve(L, ysim - cmean)
    cdf = norm.cdf(np.sort(z))
    i = np.arange(1, nsim + 1)
    ad = -nsim - np.sum((2.0 * i - 1.0) / nsim * (np.log(cdf) + np.log(1.0 - cdf[::-1])))
    assert ad < 3.857, ad                                                  # 1 % critical value (cpp:756)
    zc = z - z.mean()
    maxlag = 30
    acf = np.array([np.sum(zc[k:] * zc[:nsim - k]) for k in range(1, maxlag + 1)]) / np.sum(zc * zc)
    assert np.sum(np.abs(acf) > 1.96 / np.sqrt(nsim)) <= 5, acf            # binomial(30, 0.05): P(>5) < 1 %


def test_car1_sample_predict_simulate_assess_fit(cm):
    """CarmaModel(p=1).run_mcmc(...).predict / simulate / assess_fit (carma_pack.py:687-837 on a Car1Sample, whose
    makeKalmanFilter builds a KalmanFilter1, :925-948) against the oracle's CAR(1) Predict and Filter."""
    import oracle as orc
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "car1_n100.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    model = cm.CarmaModel(t, y, yerr, p=1)
    sample = model.run_mcmc(200, nburnin=100, seed=5)
    assert isinstance(sample, cm.Car1Sample)
    for bestfit in ("map", "median", "mean"):
        kf, mu = sample.makeKalmanFilter(bestfit)
        tp = np.r_[t[0] - 3.0, 0.5 * (t[10] + t[11]), t[40], t[-1] + 7.5]            # backcast, interpolation, datum, forecast
        pm, pv = sample.predict(tp, bestfit=bestfit)
        wm, wv = orc.predict_car1(t, y - mu, yerr, kf._sigsqr, kf._omega, tp)
        np.testing.assert_allclose(pm - mu, wm, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(pv, wv, rtol=1e-9)
        m1, v1 = sample.predict(float(tp[3]), bestfit=bestfit)                        # scalar in, scalars out
        assert m1 == pm[3] and v1 == pv[3]
    fit = sample.assess_fit(bestfit="map", nplot=32)
    kf, mu = sample.makeKalmanFilter("map")
    om, ov = orc.kfilter_car1(t, y - mu, yerr, kf._sigsqr, kf._omega)
    np.testing.assert_allclose(fit["std_resid"], (y - mu - om) / np.sqrt(ov), rtol=1e-8, atol=1e-10)
    assert fit["mean"].shape == (32,) and np.all(fit["var"] > 0) and abs(fit["resid_acf"][0] - 1.0) < 1e-12
    np.random.seed(3)
    ts = np.r_[t[-1] + 1.0, t[-1] + 2.0, 0.5 * (t[3] + t[4])]
    draws = np.array([sample.simulate(ts, bestfit="map") for _ in range(300)])
    assert draws.shape == (300, 3)
    pm, pv = sample.predict(np.sort(ts)[:1], bestfit="map")                           # first time visited: plain conditional
    z = (draws[:, 0].mean() - pm[0]) / np.sqrt(pv[0] / 300)
    assert abs(z) < 4.5 and 0.7 < draws[:, 0].var() / pv[0] < 1.4


def _philox_normals(seed, path, n):
    """rng_normal(key{seed, path}, i, 0) of carma_rng.h for i < n, restated with numpy (Box-Muller on Philox4x32-10)."""
    from carma_pack_amd import parallel as par
    out = np.empty(n)
    for i in range(n):
        x = par.philox4x32_10(i & 0xFFFFFFFF, (i >> 32) & 0xFFFFFFFF, path, (3 << 24) | 0, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        u1 = ((((x[0] << 32) | x[1]) >> 11) + 0.5) / 9007199254740992.0
        u2 = ((((x[2] << 32) | x[3]) >> 11) + 0.5) / 9007199254740992.0
        out[i] = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return out


def test_device_carma_process_is_the_reference_construction(cm, golden_dir):
    """carma_process_batch (carma_simulate_carma): many paths in one launch.  Exactness: the reference draws
    y_i ~ N(kalman_mean_i, kalman_var_i) from the Kalman recursion without measurement error (carma_pack.py:1226-1257),
    so the ORACLE's filter run over a simulated path with yerr = 0 must give back, as standardised innovations, exactly
    the normal variates the path was built from -- the counter-based Philox draws, restated here with numpy.  Then
    moments over 4096 paths against carma_variance."""
    import os
    import oracle as orc
    g = np.load(os.path.join(golden_dir, "cpp_carma_test300.npz"))
    roots, ma, sigsqr = g["omega"], g["ma"], float(g["sigsqr"])
    t = g["t"][:200]
    seed = 0x1234567
    paths = cm.carma_process_batch(t, sigsqr, roots, ma, npaths=5, seed=seed)
    assert paths.shape == (5, 200) and np.all(np.isfinite(paths))
    again = cm.carma_process_batch(t, sigsqr, roots, ma, npaths=3, seed=seed)
    assert np.array_equal(again, paths[:3])                           # a path does not depend on the batch it is drawn in
    for k in (0, 4):
        mean, var = orc.kfilter_carma(t, paths[k], np.zeros(t.size), sigsqr, roots, ma)
        z = (paths[k] - mean) / np.sqrt(var)
        np.testing.assert_allclose(z, _philox_normals(seed, k, t.size), rtol=0, atol=2e-7)
    big = cm.carma_process_batch(t, sigsqr, roots, ma, npaths=4096, seed=77)
    v0 = cm.carma_variance(sigsqr, roots, ma)
    assert abs(big.var(axis=0).mean() / v0 - 1.0) < 0.03
    for lag_idx in (1, 7, 40):
        dt = t[100 + lag_idx] - t[100]
        acv = np.mean(big[:, 100] * big[:, 100 + lag_idx])
        assert abs(acv - cm.carma_variance(sigsqr, roots, ma, lag=dt)) < 0.06 * v0
    # CAR(1): exact OU recursion
    p1 = cm.car1_process_batch(t, 0.5, 20.0, npaths=2048, seed=5)
    assert p1.shape == (2048, 200) and abs(p1.var(axis=0).mean() / (0.5 * 20.0 / 2.0) - 1.0) < 0.05
    z1 = _philox_normals(5, 3, 4)
    rho = np.exp(-np.diff(t[:4]) / 20.0)
    want = [np.sqrt(5.0) * z1[0]]
    for i in range(1, 4):
        want.append(rho[i - 1] * want[-1] + np.sqrt(5.0 * (1.0 - rho[i - 1] ** 2)) * z1[i])
    np.testing.assert_allclose(p1[3, :4], want, rtol=1e-12)


def test_kalman_filter_objects_keep_their_series_on_the_device(cm, golden_dir):
    """KalmanFilterp / KalmanFilter1 as objects (carma_kf_*): one upload, then Filter() and many Predict() calls -- the
    reference's calling pattern (carma_pack.py:793-803: one Predict per plot point) -- give what the one-shot entry points
    give, and what the oracle gives."""
    import os
    import oracle as orc
    import carma_pack_amd as cpa
    g = np.load(os.path.join(golden_dir, "cpp_carma_test300.npz"))
    t, y, e = g["t"][:150], g["y"][:150], g["yerr"][:150]
    roots, ma, sigsqr = g["omega"], g["ma"], float(g["sigsqr"])
    om = cm.vecC()
    for r in roots[::-1]:                                                  # any order of the roots is accepted
        om.append(complex(r))
    kf = cm.KalmanFilterp(cm.vecD(t.tolist()), cm.vecD(y.tolist()), cm.vecD(e.tolist()), sigsqr, om, cm.vecD(ma.tolist()))
    kf.Filter()
    mean, var = np.array(kf.GetMean()), np.array(kf.GetVar())
    m1, v1 = cpa.kfilter_carma(t, y, e, sigsqr, roots, ma)
    np.testing.assert_allclose(var, v1, rtol=1e-11)                       # (pairs in another order: other summation order)
    np.testing.assert_allclose(mean, m1, rtol=0, atol=1e-10)
    om_, ov_ = orc.kfilter_carma(t, y, e, sigsqr, roots, ma)
    np.testing.assert_allclose(var, ov_, rtol=1e-10)
    tp = np.r_[t[0] - 2.0, 0.5 * (t[20] + t[21]), t[-1] + 3.0]
    for x in tp:                                                            # the reference's one-Predict-per-time pattern
        pr = kf.Predict(float(x))
        wm, wv = orc.predict_carma(t, y, e, sigsqr, roots, ma, [x])
        assert abs(pr.first - wm[0]) <= 1e-9 * max(1.0, abs(wm[0])) and abs(pr.second - wv[0]) <= 1e-9 * wv[0]
    pm, pv = kf.PredictBatch(tp)
    assert pm[1] == kf.Predict(float(tp[1])).first
    # a set of roots that is not closed under conjugation is not a real-valued process
    bad = cm.vecC([complex(-0.1, 0.3), complex(-0.2, 0.0)])
    with pytest.raises(ValueError):
        cm.KalmanFilterp(cm.vecD(t.tolist()), cm.vecD(y.tolist()), cm.vecD(e.tolist()), 1.0, bad, cm.vecD([1.0])).Filter()
    k1 = cm.KalmanFilter1(cm.vecD(t.tolist()), cm.vecD(y.tolist()), cm.vecD(e.tolist()), 0.3, 0.05)
    k1.Filter()
    o1m, o1v = orc.kfilter_car1(t, y, e, 0.3, 0.05)
    np.testing.assert_allclose(np.array(k1.GetVar()), o1v, rtol=1e-11)
    assert abs(k1.Predict(float(tp[2])).first - orc.predict_car1(t, y, e, 0.3, 0.05, [tp[2]])[0][0]) < 1e-10
