"""CPU-only tests of the host-side Python mirror of carma_pack's API (derived quantities, free
functions, container types); the compute path itself needs a GPU (tests/test_gpu_*.py)."""
import json
import os

import numpy as np
import pytest

import carma_pack_amd as cpa
import carmcmc as cm
import oracle as orc
from carma_pack_amd import carma_pack as cp


@pytest.fixture(autouse=True)
def _post_processing_stand_in(monkeypatch):
    """CarmaSample computes sigma and the PSD band on the device (carma_post.hip).  Without a GPU the two compute hooks of
    the `_carmcmc` mirror are replaced by the oracle's numpy restatement, so that the HOST logic of the classes
    (dictionary, sub-sampling, return contracts) is still exercised here -- and the restatement itself is held to the
    reference's own output by the golden files below.  With a GPU the product's own path runs."""
    if cpa._lib.lib.carma_device_count() == 0:
        monkeypatch.setattr(cp.carmcmcLib, "sigma_noise_batch", orc.post.sigma_noise)
        monkeypatch.setattr(cp.carmcmcLib, "psd_band", orc.post.psd_band)


def test_post_processing_hooks_fail_loudly_without_a_gpu():
    if cpa._lib.lib.carma_device_count() > 0:
        pytest.skip("a GPU is visible")
    from carma_pack_amd import _lib
    with pytest.raises(cpa.CarmaDeviceError):
        _lib.sigma_noise_batch(np.array([[-0.1 + 0.2j, -0.1 - 0.2j]]), np.array([[1.0]]), np.array([1.0]))
    with pytest.raises(cpa.CarmaDeviceError):
        _lib.psd_band(np.array([[1.0, 0.2, 0.05]]), np.array([[1.0]]), np.array([1.0]), [0.1, 0.2], [16.0, 50.0, 84.0])
    with pytest.raises(ValueError):                            # argument checks come before any device work
        _lib.psd_band(np.array([[1.0, 0.2, 0.05]]), np.array([[1.0]]), np.array([1.0]), [0.1], [16.0, 50.0, 84.0, 1.0, 2.0])


def test_numpy_restatement_of_the_roots_is_the_reference_formula(golden_dir):
    """oracle.post.roots_from_log_quads is carma_pack.py:439-468 literally; the product takes the smaller of two REAL roots
    from the product q1 / big root instead (as the kernels do) -- identical for complex pairs, and to rounding otherwise."""
    g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    th = g["theta"]
    np.testing.assert_allclose(orc.post.roots_from_log_quads(th[:, 3:8]), g["omega"], rtol=1e-13)
    np.testing.assert_allclose(cp._roots_from_log_quads(th[:, 3:8]), orc.post.roots_from_log_quads(th[:, 3:8]), rtol=1e-13)
    lq = np.log(np.array([[0.02, 0.5, 0.3], [1e-3, 4.0, 0.1]]))           # two real roots per quadratic factor
    a, b = cp._roots_from_log_quads(lq), orc.post.roots_from_log_quads(lq)
    assert np.all(a.imag == 0) and np.all(b.imag == 0)
    np.testing.assert_allclose(a, b, rtol=1e-10)


def test_alias_package_exports():
    for name in ("vecD", "vecvecD", "vecC", "pairD", "CAR1", "CARp", "CARMA", "run_mcmc_car1", "run_mcmc_carma",
                 "KalmanFilter1", "KalmanFilterp", "CarmaModel", "CarmaSample", "Car1Sample", "get_ar_roots",
                 "power_spectrum", "carma_variance", "carma_process", "car1_process"):
        assert hasattr(cm, name), name
    v = cm.vecD()
    v.extend([1.0, 2.0])
    v.append(3.0)
    assert len(v) == 3 and v[1] == 2.0
    pr = cm.pairD()
    pr.first, pr.second = 1.0, 2.0


def test_variance_kat_and_roots(golden_dir):
    s = json.load(open(os.path.join(golden_dir, "summary.json")))["variance_kat"]
    om = cm.get_ar_roots(np.array([0.01, 0.01, 0.002]), np.array([0.2, 0.02]))
    np.testing.assert_allclose(om.real, s["omega_re"], rtol=1e-15)
    np.testing.assert_allclose(om.imag, s["omega_im"], rtol=1e-15)
    v = cm.carma_variance(2.3 ** 2, om, s["ma"])
    assert abs(v - 223003.230567) / 223003.230567 < 1e-8          # carma_unit_tests.cpp:1313-1316
    for lag, ref in zip(s["lags"], s["lagged"]):
        assert abs(cm.carma_variance(2.3 ** 2, om, s["ma"], lag=lag) - ref) / abs(ref) < 1e-12


def test_derived_quantities_match_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    th = g["theta"]
    roots = cp._roots_from_log_quads(th[:, 3:8])
    np.testing.assert_allclose(roots, g["omega"], rtol=1e-13)
    ma_roots = cp._roots_from_log_quads(th[:, 8:11])
    c = cp._poly_from_roots(ma_roots)
    ma = (c / c[:, 3:4])[:, ::-1].real
    np.testing.assert_allclose(ma, g["ma"][:, :4], rtol=1e-12)
    np.testing.assert_allclose(cp._poly_from_roots(roots)[3], np.poly(roots[3]), rtol=1e-12)


class _FakeSampler(object):
    """Stands in for the object run_mcmc_carma returns; log-densities from the CPU oracle."""

    def __init__(self, t, y, yerr, p, q, samples):
        self.m = orc.OracleModel(t, y, yerr, p, q)
        self.s = samples
        self.mle = False

    def getSamples(self):
        return self.s.tolist()

    def GetLogLikes(self):
        return self.m.logdensity_batch(self.s).tolist()

    def SetMLE(self, flag):
        self.mle = flag

    def getLogDensityBatch(self, th):
        return self.m.logdensity_batch(th, ignore_prior=self.mle)


def test_carma_sample_dictionary(golden_dir):
    g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    fake = _FakeSampler(t, y, yerr, 5, 3, g["theta"][:16])
    s = cp.CarmaSample(t, y, yerr, fake, q=3)
    for key in ("logpost", "var", "measerr_scale", "mu", "quad_coefs", "ar_roots", "psd_centroid", "psd_width",
                "ar_coefs", "ma_coefs", "sigma", "loglik"):
        assert key in s.parameters, key
    assert s.p == 5
    np.testing.assert_allclose(s.get_samples("sigma")[:, 0] ** 2, g["sigsqr"][:16], rtol=1e-11)
    np.testing.assert_allclose(s.get_samples("ar_roots"), g["omega"][:16], rtol=1e-13)
    np.testing.assert_allclose(s.get_samples("psd_width")[0, :2], [0.01, 0.01], rtol=1e-12)
    np.testing.assert_allclose(s.get_samples("loglik")[:, 0] -
                               np.array([fake.m.log_prior(x) for x in g["theta"][:16]]), g["loglik"][:16], rtol=1e-11)
    # quirk kept: without q= the AR order is inferred as p+q (testCarmcmc.py:95)
    s2 = cp.CarmaSample(t, y, yerr, _FakeSampler(t, y, yerr, 5, 3, g["theta"][:4]).__class__(t, y, yerr, 5, 3, g["theta"][:4]))
    assert s2.p == 8
    lo, hi, med, f = s.power_spectrum_band(nsamples=8)
    assert np.all(lo <= med) and np.all(med <= hi) and f.size == 1000
    assert np.isfinite(s.DIC())


def test_samples_from_a_carpack_ascii_file(golden_dir, tmp_path):
    """generate_from_file (reference carma_pack.py:427-437, samplers.py:57-72): the ascii output of the C++ carpack --
    a header line, then parameter vector + log-posterior per row -- fills the same dictionary as a trace does."""
    g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    fake = _FakeSampler(t, y, yerr, 5, 3, g["theta"][:16])
    s = cp.CarmaSample(t, y, yerr, fake, q=3)
    fname = str(tmp_path / "carpack_samples.dat")
    lp = np.array(fake.GetLogLikes())
    np.savetxt(fname, np.c_[g["theta"][:16], lp], header="sigma measerr_scale mu ar... ma... logpost", fmt="%.17g")
    s2 = cp.CarmaSample.__new__(cp.CarmaSample)
    s2.q, s2._samples = 3, {}
    s2.generate_from_file([fname])
    assert s2.p == 5
    for key in ("var", "measerr_scale", "mu", "quad_coefs"):
        np.testing.assert_array_equal(np.squeeze(s2._samples[key]), np.squeeze(s._samples[key]))
    np.testing.assert_array_equal(s2._samples["logpost"], lp)
    # the generic container: one parameter per file, its name on the first line
    one = str(tmp_path / "mu.dat")
    with open(one, "w") as f:
        f.write("mu\n")
        np.savetxt(f, g["theta"][:16, 2])
    m = cp.MCMCSample(filename=one)
    np.testing.assert_allclose(m.get_samples("mu\n"), g["theta"][:16, 2])


def test_carma_process_moments():
    rng = np.random.RandomState(3)
    roots = cm.get_ar_roots(np.array([0.05, 0.02]), np.array([0.1]))
    ma = [1.0, 2.0]
    sigsqr = 1.7 ** 2 / cm.carma_variance(1.0, roots, ma)
    t = np.arange(4000) * 1.0 + rng.uniform(0, 0.5, 4000)
    y = cm.carma_process(t, sigsqr, roots, ma, rng=rng)
    assert abs(y.var() - 1.7 ** 2) < 0.5
    lag = 5
    acf = np.mean(y[lag:] * y[:-lag])
    dtm = np.mean(t[lag:] - t[:-lag])
    assert abs(acf - cm.carma_variance(sigsqr, roots, ma, lag=dtm)) < 0.6
    y1 = cm.car1_process(t, 0.5, 20.0, rng=rng)
    assert abs(y1.var() - 0.5 * 20.0 / 2.0) < 1.5


def test_mle_bounds_and_aicc_bookkeeping():
    rng = np.random.default_rng(0)
    t = np.sort(rng.uniform(0, 100, 50))
    m = cpa.CarmaModel(np.r_[t, t[:3]], rng.standard_normal(53), np.ones(53), p=3, q=1)
    assert m.time.size == 50 and np.all(np.diff(m.time) > 0)       # unique + sorted (:33-35)
    b = m._mle_bounds(3, 1)
    assert len(b) == 3 + 3 + 1 and b[1] == (0.9, 1.1) and b[-1] == (None, None)
    with pytest.raises(ValueError):
        cpa.CarmaModel(t, t, t, p=2, q=2)


def test_plot_power_spectrum_matches_reference(golden_dir):
    """CarmaSample.plot_power_spectrum under the reference's name and return contract (carma_pack.py:548-648):
    (lower, upper, median, frequencies), against the reference's own output (tests/golden/make_golden_psd.py)."""
    g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    ref = np.load(os.path.join(golden_dir, "psd.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    s = cp.CarmaSample(t, y, yerr, _FakeSampler(t, y, yerr, 5, 3, g["theta"]), q=3)
    lo, hi, med, f = s.plot_power_spectrum(percentile=68.0, doShow=False)
    np.testing.assert_allclose(f, ref["freq"], rtol=1e-14)
    np.testing.assert_allclose(lo, ref["lo68"], rtol=1e-9)
    np.testing.assert_allclose(hi, ref["hi68"], rtol=1e-9)
    np.testing.assert_allclose(med, ref["med68"], rtol=1e-9)
    lo, hi, med, f = s.plot_power_spectrum(percentile=95.0, nsamples=9, doShow=False)     # evenly spaced subsample (:572-578)
    np.testing.assert_allclose(lo, ref["lo95_n9"], rtol=1e-9)
    np.testing.assert_allclose(hi, ref["hi95_n9"], rtol=1e-9)
    np.testing.assert_allclose(med, ref["med95_n9"], rtol=1e-9)
    # and the free function sample by sample
    k = 5
    one = cm.power_spectrum(f[::100], s.get_samples("sigma")[k, 0], s.get_samples("ar_coefs")[k], s.get_samples("ma_coefs")[k])
    np.testing.assert_allclose(s._psd_samples(f[::100], np.array([k]))[:, 0], one, rtol=1e-12)


class _FakeCar1(object):
    def __init__(self, samples):
        self.s = np.asarray(samples)

    def getSamples(self):
        return self.s.tolist()

    def GetLogLikes(self):
        return np.linspace(-100.0, -90.0, self.s.shape[0]).tolist()

    def getLogPrior(self, theta):
        return -1.0


def test_car1_sample_psd_and_kalman_filter(golden_dir):
    """Car1Sample: PSD band against the reference (carma_pack.py:950-1035) and makeKalmanFilter building a
    KalmanFilter1 (:925-948) -- round 1 inherited the CARMA(p) version and every predict/simulate/assess_fit raised."""
    g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    ref = np.load(os.path.join(golden_dir, "psd.npz"))
    t, y, yerr, th1 = g["t"], g["y"], g["yerr"], ref["car1_theta"]
    s = cp.Car1Sample(t, y, yerr, _FakeCar1(th1))
    np.testing.assert_allclose(np.ravel(s.get_samples("sigma")), ref["car1_sigma"], rtol=1e-13)
    lo, hi, med, f = s.plot_power_spectrum(percentile=68.0, doShow=False)
    np.testing.assert_allclose(f, ref["car1_freq"], rtol=1e-14)
    np.testing.assert_allclose(lo, ref["car1_lo68"], rtol=1e-10)
    np.testing.assert_allclose(hi, ref["car1_hi68"], rtol=1e-10)
    np.testing.assert_allclose(med, ref["car1_med68"], rtol=1e-10)
    for bestfit in ("map", "median", "mean", 3):
        kf, mu = s.makeKalmanFilter(bestfit)
        assert isinstance(kf, cm.KalmanFilter1)
        assert kf._omega > 0 and kf._sigsqr > 0 and abs(mu - 17.0) < 1.0
    kf, mu = s.makeKalmanFilter("map")                      # logpost is increasing: the last sample is the MAP
    assert mu == th1[-1, 2] and kf._omega == np.exp(th1[-1, 3])
    assert abs(kf._sigsqr - 2.0 * th1[-1, 0] ** 2 * np.exp(th1[-1, 3])) < 1e-12 * kf._sigsqr
    kf, mu = s.makeKalmanFilter("mean")
    assert abs(kf._sigsqr - np.mean(ref["car1_sigma"] ** 2)) < 1e-12 * kf._sigsqr
    # predict() reaches the device entry point with valid arguments: without a GPU it is the device error, not ValueError
    if cpa._lib.lib.carma_device_count() == 0:
        with pytest.raises(cpa._lib.CarmaDeviceError):
            s.predict(t[-1] + 5.0)


def test_root_order_normalisation():
    """carma_normalize_roots (host arithmetic of the KalmanFilterp-type entry points): any order in, ARRoots order out."""
    from carma_pack_amd import _lib
    rng = np.random.default_rng(3)
    roots = np.array([-0.3 + 0.7j, -0.05 + 0j, -0.3 - 0.7j, -1.2 - 0.1j, -2.0 + 0j, -1.2 + 0.1j, -0.9 + 0j])
    for _ in range(5):
        out = _lib.normalize_roots(roots[rng.permutation(7)])
        assert sorted(out, key=lambda z: (z.real, z.imag)) == sorted(roots, key=lambda z: (z.real, z.imag))
        assert np.all(out[0:4:2].imag < 0) and np.array_equal(out[1:4:2], np.conj(out[0:4:2])) and np.all(out[4:].imag == 0)
    # get_ar_roots puts a real root wherever its centroid is zero (carma_pack.py:1038-1059)
    mixed = cm.get_ar_roots(np.array([0.01, 0.02, 0.005]), np.array([0.2, 0.0, 0.03]))
    out = _lib.normalize_roots(mixed)
    assert out.size == 5 and out[4].imag == 0 and out[0].imag < 0 < out[1].imag
    # a conjugate that is off by one ulp is still its conjugate; a lone complex root is not a real-valued process
    r = -0.123456789 + 0.987654321j
    out = _lib.normalize_roots([r, complex(np.nextafter(r.real, 0.0), -r.imag)])
    assert out[1] == np.conj(out[0])
    with pytest.raises(ValueError):
        _lib.normalize_roots([-0.1 + 0.3j, -0.2 + 0j])
    with pytest.raises(ValueError):
        _lib.normalize_roots([-0.1 + 0.3j, -0.1 - 0.31j])
