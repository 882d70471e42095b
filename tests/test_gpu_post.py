"""GPU tests of SURVEY.md section 8(f) rank 3 -- CarmaSample post-processing on the device (carma_post.hip):
sigma of the driving noise per sample, the power-spectrum grid, its credibility band; and the dictionary of a device-run
sampler (loglik, sigma, PSD band) against the oracle and the reference's own output (tests/golden/psd.npz)."""
import os

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cpa():
    import carma_pack_amd as m
    assert m._lib.lib.carma_device_count() >= 1
    return m


@pytest.fixture(scope="module")
def readme(golden_dir):
    return np.load(os.path.join(golden_dir, "carma53_readme.npz"))


def _derived(cp, th, p, q):
    roots = cp._roots_from_log_quads(th[:, 3:3 + p])
    ar = cp._poly_from_roots(roots).real
    if q:
        c = cp._poly_from_roots(cp._roots_from_log_quads(th[:, 3 + p:3 + p + q]))
        ma = (c / c[:, q:q + 1])[:, ::-1].real
    else:
        ma = np.ones((th.shape[0], 1))
    return roots, ar, ma


def test_sigma_noise_against_reference_vectors_and_restatement(cpa, readme):
    """carma_sigma_noise_batch == CarmaSample._sigma_noise (carma_pack.py:513-546): the reference's own sigma^2 of the 32
    README vectors, and the numpy restatement on prior-like vectors of every order."""
    from carma_pack_amd import _lib, carma_pack as cp
    from helpers import prior_like_theta
    th = readme["theta"]
    roots, ar, ma = _derived(cp, th, 5, 3)
    sig = _lib.sigma_noise_batch(roots, ma, th[:, 0] ** 2)
    np.testing.assert_allclose(sig ** 2, readme["sigsqr"], rtol=1e-11)
    np.testing.assert_allclose(sig, orc.post.sigma_noise(roots, ma, th[:, 0] ** 2), rtol=1e-12)
    rng = np.random.default_rng(5)
    t, y = readme["t"], readme["y"]
    for p in range(1, 8):
        for q in sorted({0, p // 2, p - 1}):
            if p == 1:
                om = np.exp(rng.normal(-3.0, 1.0, 300))
                roots, ma, var = (-om)[:, None] + 0j, np.ones((300, 1)), rng.uniform(0.5, 3.0, 300) ** 2
            else:
                th = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(300)])
                roots, ar, ma = _derived(cp, th, p, q)
                var = th[:, 0] ** 2
            got, want = _lib.sigma_noise_batch(roots, ma, var), orc.post.sigma_noise(roots, ma, var)
            fin = np.isfinite(want)
            assert np.array_equal(np.isfinite(got), fin), (p, q)
            # the sum over the roots cancels when roots cluster: both sides carry cond x eps.  Within 1e-10 of the restatement, or --
            # arbitrated by the same sum in quad precision (oracle/carma_truth_q.c, orc_truth_variance) -- no further from the exact
            # value than the restatement is, or within the forward-error scale of the summation itself, 4 eps x its condition
            # number (which of two double-precision sums lands nearer on ONE entry is luck: the first run of this arbiter met
            # 1.9e-10 against 2.5e-11 on a CARMA(6,0) entry); at most 10 % of the entries may need the arbiter
            from helpers import assert_parity
            idx = np.flatnonzero(fin)

            def exact(k, roots=roots, ma=ma, var=var, idx=idx):
                i = idx[k]
                return float(np.sqrt(var[i] / orc.truth_variance(roots[i], ma[i][:roots.shape[1]])))

            def scale(k, roots=roots, ma=ma, idx=idx):
                i = idx[k]
                return 2.0 * np.finfo(float).eps * orc.truth_variance(roots[i], ma[i][:roots.shape[1]], with_cond=True)[1]
            assert_parity(got[fin], want[fin], 1e-10, "sigma_noise (%d,%d)" % (p, q), arbiter=exact, arb_factor=1.25, max_arb_frac=0.10,
                          noise_scale=scale)


def test_psd_band_matches_reference_output(cpa, readme, golden_dir):
    """carma_psd_band through CarmaSample.plot_power_spectrum / Car1Sample.plot_power_spectrum against the REFERENCE's own
    return values (tests/golden/make_golden_psd.py ran carma_pack.py:548-648 and :950-1035)."""
    from carma_pack_amd import carma_pack as cp
    ref = np.load(os.path.join(golden_dir, "psd.npz"))
    t, y, yerr, th = readme["t"], readme["y"], readme["yerr"], readme["theta"]
    import carmcmc as cm
    xv, yv, ev = cm.vecD(t), cm.vecD(y), cm.vecD(yerr)
    sampler = cm.CARMA(True, "c", xv, yv, ev, 5, 3)

    class Stored(object):                                   # the reference's sampler object holds the samples it drew;
        def getSamples(self):                               # here: the golden parameter vectors, log-densities from the device
            return th.tolist()

        def GetLogLikes(self):
            return [sampler.getLogDensity(cm.vecD(x)) for x in th]

        def SetMLE(self, flag):
            sampler.SetMLE(flag)

        def getLogDensityBatch(self, a):
            return sampler.getLogDensityBatch(a)

    s = cp.CarmaSample(t, y, yerr, Stored(), q=3)
    lo, hi, med, f = s.plot_power_spectrum(percentile=68.0, doShow=False)
    np.testing.assert_allclose(f, ref["freq"], rtol=1e-14)
    for got, key in ((lo, "lo68"), (hi, "hi68"), (med, "med68")):
        np.testing.assert_allclose(got, ref[key], rtol=1e-9)
    lo, hi, med, f = s.plot_power_spectrum(percentile=95.0, nsamples=9, doShow=False)
    for got, key in ((lo, "lo95_n9"), (hi, "hi95_n9"), (med, "med95_n9")):
        np.testing.assert_allclose(got, ref[key], rtol=1e-9)
    # the dictionary of a sample set whose densities came from the device: sigma == reference, loglik == oracle
    np.testing.assert_allclose(np.ravel(s.get_samples("sigma")) ** 2, readme["sigsqr"], rtol=1e-11)
    om = orc.OracleModel(t, y, yerr, 5, 3)
    want = om.logdensity_batch(th, ignore_prior=True)
    np.testing.assert_allclose(np.ravel(s.get_samples("loglik")), want, rtol=1e-10)
    np.testing.assert_allclose(np.ravel(s.get_samples("loglik")) - np.array([om.log_prior(x) for x in th]), readme["loglik"], rtol=1e-10)

    class Car1Stored(object):
        def getSamples(self):
            return ref["car1_theta"].tolist()

        def GetLogLikes(self):
            return np.linspace(-100.0, -90.0, ref["car1_theta"].shape[0]).tolist()

        def getLogPrior(self, theta):
            return -1.0

    s1 = cp.Car1Sample(t, y, yerr, Car1Stored())
    lo, hi, med, f = s1.plot_power_spectrum(percentile=68.0, doShow=False)
    np.testing.assert_allclose(f, ref["car1_freq"], rtol=1e-14)
    for got, key in ((lo, "car1_lo68"), (hi, "car1_hi68"), (med, "car1_med68")):
        np.testing.assert_allclose(got, ref[key], rtol=1e-10)


def test_psd_grid_and_exact_order_statistics(cpa, readme):
    """The grid against the numpy restatement; the band against np.percentile OF THE RETURNED GRID to the last bit but one
    (the selection is exact: ties, rows of equal values, one or two samples, every size around the workgroup width)."""
    from carma_pack_amd import _lib, carma_pack as cp
    from helpers import prior_like_theta
    rng = np.random.default_rng(9)
    t, y = readme["t"], readme["y"]
    freq = np.exp(np.linspace(np.log(1e-3), np.log(0.5), 37))
    for ns in (1, 2, 3, 63, 64, 255, 256, 257, 1000, 5003):
        th = np.array([prior_like_theta(rng, 5, 3, t, y) for _ in range(min(ns, 400))])
        th = th[rng.integers(0, th.shape[0], ns)]                  # repeated samples: ties in every row
        roots, ar, ma = _derived(cp, th, 5, 3)
        sig = orc.post.sigma_noise(roots, ma, th[:, 0] ** 2)
        sig[~np.isfinite(sig)] = 1.0
        pcs = [2.5, 16.0, 50.0, 84.0]
        band, grid = _lib.psd_band(ar, ma, sig, freq, pcs, return_samples=True)
        want_grid = orc.post.psd_samples(ar, ma, sig, freq)
        np.testing.assert_allclose(grid, want_grid, rtol=1e-12)
        want = np.percentile(grid, pcs, axis=1).T
        np.testing.assert_allclose(band, want, rtol=4e-16, atol=0.0)
    # a row of identical values; +inf (alpha(0) = 0 at f = 0); a NaN sample makes its rows NaN (np.percentile)
    ar = np.tile([1.0, 0.3, 0.02], (50, 1))
    band = _lib.psd_band(ar, np.ones((50, 1)), np.full(50, 0.7), [0.01, 0.2], [16.0, 50.0, 84.0])
    assert np.all(band == band[:, :1])
    ar0 = ar.copy()
    ar0[:7, 2] = 0.0
    band, grid = _lib.psd_band(ar0, np.ones((50, 1)), np.full(50, 0.7), [0.0, 0.2], [50.0, 100.0], return_samples=True)
    assert np.isinf(grid[0, :7]).all() and np.isfinite(band[0, 0]) and np.isfinite(band[1]).all()
    with np.errstate(invalid="ignore"):                      # between two infinities numpy's interpolation gives NaN: so do we
        np.testing.assert_array_equal(band, np.percentile(grid, [50.0, 100.0], axis=1).T)
    assert np.isnan(band[0, 1])
    sg = np.full(50, 0.7)
    sg[3] = np.nan
    band = _lib.psd_band(ar, np.ones((50, 1)), sg, [0.01, 0.2], [16.0, 50.0])
    assert np.isnan(band).all()
    with pytest.raises(ValueError):
        _lib.psd_band(ar, np.ones((50, 1)), sg, [0.01], [101.0])


def test_sample_dictionary_of_a_device_run_sampler(cpa, readme):
    """run_mcmc on the device -> CarmaSample: "loglik", "sigma" and the PSD band of the samples the GPU drew equal the
    oracle's LogDensity (prior bounds ignored, carma_pack.py:305-315), the restated _sigma_noise and np.percentile of the
    restated spectrum grid."""
    t, y, yerr = readme["t"], readme["y"], readme["yerr"]
    model = cpa.CarmaModel(t, y, yerr, p=5, q=3)
    s = model.run_mcmc(400, nburnin=300, ntemperatures=6, seed=13)
    trace = np.array(s._sampler.getSamples())
    assert trace.shape == (400, 11)
    om = orc.OracleModel(t, y, yerr, 5, 3)                     # (its default prior bound is RunCarmaSampler's, as run_mcmc's)
    from helpers import assert_parity_states, loglik_truth
    arb = lambda i: loglik_truth(t, y, yerr, trace[i], 5, 3)[0]   # noqa: E731  (in bounds: log-likelihood + log prior either way)
    assert_parity_states(np.ravel(s.get_samples("loglik")), om.logdensity_batch(trace, ignore_prior=True), trace, 5, 3, 1e-10,
                         "CarmaSample loglik", arbiter=arb)
    # log-posterior stored by the sampler == LogDensity of the sample (the reference's criterion, carma_unit_tests.cpp:917-1114)
    assert_parity_states(np.ravel(s.get_samples("logpost")), om.logdensity_batch(trace), trace, 5, 3, 1e-10,
                         "CarmaSample logpost", arbiter=arb)
    roots = orc.post.roots_from_log_quads(trace[:, 3:8])
    np.testing.assert_allclose(s.get_samples("ar_roots"), roots, rtol=1e-9)
    want_sig = orc.post.sigma_noise(s.get_samples("ar_roots"), s.get_samples("ma_coefs"), np.ravel(s.get_samples("var")))
    np.testing.assert_allclose(np.ravel(s.get_samples("sigma")), want_sig, rtol=1e-9)
    lo, hi, med, f = s.plot_power_spectrum(percentile=68.0, doShow=False)
    want = orc.post.psd_band(s.get_samples("ar_coefs"), s.get_samples("ma_coefs"), want_sig, f, [16.0, 50.0, 84.0])
    np.testing.assert_allclose(np.c_[lo, med, hi], want, rtol=1e-8)
    lo2, hi2, med2, _ = s.plot_power_spectrum(percentile=95.0, nsamples=57, doShow=False)
    idx = (np.arange(57) * (400 / 57)).astype(int)
    want = orc.post.psd_band(s.get_samples("ar_coefs")[idx], s.get_samples("ma_coefs")[idx], want_sig[idx], f, [2.5, 50.0, 97.5])
    np.testing.assert_allclose(np.c_[lo2, med2, hi2], want, rtol=1e-8)
