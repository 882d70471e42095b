"""carma_pack's Python API (src/carmcmc/carma_pack.py in the reference) on top of the MI355X path.

Same public names and call signatures -- ``CarmaModel`` (``run_mcmc``, ``get_mle``, ``choose_order``),
``CarmaSample`` / ``Car1Sample`` (``get_samples``, ``parameters``, ``mle``, derived quantities),
``get_ar_roots``, ``power_spectrum``, ``carma_variance``, ``car1_process``, ``carma_process`` -- but a
fresh implementation: derived quantities are vectorised numpy, every log-density goes through one
batched launch, the sampler runs on the GPU, and ``run_mcmc``/``get_mle`` can use many independent
replicas at once (``nreplicas``).  ``predict``/``simulate``/``assess_fit`` use the batched device
Predict kernel (one launch for all requested times); plotting bodies are not part of the hot path.
"""
import os

import numpy as np
from scipy.optimize import minimize

from . import _carmcmc as carmcmcLib

__all__ = ["CarmaModel", "CarmaSample", "Car1Sample", "MCMCSample", "get_ar_roots", "power_spectrum",
           "carma_variance", "car1_process", "carma_process", "carma_process_batch", "car1_process_batch"]


# ------------------------------------------------------------------------------------------------
# free functions (reference: carma_pack.py:1038-1259)
def get_ar_roots(qpo_width, qpo_centroid):
    """Lorentzian widths/centroids -> roots of the AR polynomial, -2 pi (width + i centroid) with the
    conjugate appended for every centroid > 1e-10 and one extra real root when there is one more width
    than centroids (reference :1038-1059)."""
    qpo_width, qpo_centroid = np.atleast_1d(qpo_width), np.atleast_1d(qpo_centroid)
    roots = []
    for w, c in zip(qpo_width, qpo_centroid):
        roots.append(w + 1j * c)
        if c > 1e-10:
            roots.append(w - 1j * c)
    if qpo_width.size - qpo_centroid.size == 1:
        roots.append(qpo_width[-1] + 0j)
    return -2.0 * np.pi * np.array(roots)


def power_spectrum(freq, sigma, ar_coef, ma_coefs=(1.0,)):
    """sigma^2 |beta(2 pi i f)|^2 / |alpha(2 pi i f)|^2 (reference :1062-1081); ar_coef highest order
    first (np.poly convention), ma_coefs lowest order first."""
    s = 2.0j * np.pi * np.asarray(freq, dtype=float)
    num = np.polyval(np.asarray(ma_coefs)[::-1], s)
    den = np.polyval(np.asarray(ar_coef), s)
    return sigma ** 2 * np.abs(num) ** 2 / np.abs(den) ** 2


def carma_variance(sigsqr, ar_roots, ma_coefs=(1.0,), lag=0.0):
    """Autocovariance of a CARMA(p,q) process at `lag` (reference :1084-1123 == CARp::Variance,
    src/carpack.cpp:377-409)."""
    r = np.asarray(ar_roots, dtype=complex)
    p = r.size
    beta = np.zeros(p)
    beta[:len(ma_coefs)] = ma_coefs
    powers = np.arange(p)
    total = 0.0 + 0.0j
    for k in range(p):
        others = np.delete(r, k)
        denom = -2.0 * r[k].real * np.prod((others - r[k]) * (np.conj(others) + r[k]))
        num = np.sum(beta * r[k] ** powers) * np.sum(beta * (-r[k]) ** powers) * np.exp(r[k] * abs(lag))
        total += num / denom
    return sigsqr * total.real


def car1_process(time, sigsqr, tau, rng=None):
    """Exact Ornstein-Uhlenbeck draw at the given times (reference :1126-1146)."""
    rng = np.random if rng is None else rng
    time = np.asarray(time, dtype=float)
    var = sigsqr * tau / 2.0
    y = np.empty(time.size)
    y[0] = np.sqrt(var) * rng.standard_normal()
    rho = np.exp(-np.diff(time) / tau)
    eps = rng.standard_normal(time.size - 1)
    for i in range(1, time.size):
        y[i] = rho[i - 1] * y[i - 1] + np.sqrt(var * (1.0 - rho[i - 1] ** 2)) * eps[i - 1]
    return y


def _rotated_system(sigsqr, ar_roots, ma_coefs):
    """Diagonalised state space of kfilter.cpp:138-172: returns (b, V) with V Hermitian."""
    r = np.asarray(ar_roots, dtype=complex)
    p = r.size
    E = np.vander(r, p, increasing=True).T           # E[i, j] = r_j ** i
    e = np.zeros(p, dtype=complex)
    e[-1] = 1.0
    J = np.linalg.solve(E, e)
    beta = np.zeros(p)
    beta[:len(ma_coefs)] = ma_coefs
    b = beta @ E
    V = -sigsqr * np.outer(J, np.conj(J)) / (r[:, None] + np.conj(r)[None, :])
    return b, V


def carma_process(time, sigsqr, ar_roots, ma_coefs=(1.0,), rng=None):
    """Draw a CARMA(p,q) path at the (sorted) times by sequential conditional simulation: each value
    is drawn from its one-step predictive distribution (same construction as reference :1148-1259,
    written with the D = P - V recursion used by the device kernel)."""
    rng = np.random if rng is None else rng
    r = np.asarray(ar_roots, dtype=complex)
    time = np.sort(np.asarray(time, dtype=float))
    if r.size == 1:
        return car1_process(time, sigsqr, -1.0 / r.real.item(), rng)
    b, V = _rotated_system(sigsqr, r, ma_coefs)
    c = V @ np.conj(b)
    s0 = float(np.real(b @ c))
    p = r.size
    D = np.zeros((p, p), dtype=complex)
    x = np.zeros(p, dtype=complex)
    y = np.empty(time.size)
    var, mean = s0, 0.0
    y[0] = rng.normal(mean, np.sqrt(var))
    innov = y[0] - mean
    u = c.copy()
    for k in range(1, time.size):
        rho = np.exp(r * (time[k] - time[k - 1]))
        x = rho * (x + u * (innov / var))
        D = np.outer(rho, np.conj(rho)) * (D - np.outer(u, np.conj(u)) / var)
        w = D @ np.conj(b)
        u = w + c
        var = s0 + float(np.real(b @ w))
        mean = float(np.real(b @ x))
        y[k] = rng.normal(mean, np.sqrt(var))
        innov = y[k] - mean
    return y


def carma_process_batch(time, sigsqr, ar_roots, ma_coefs=(1.0,), npaths=1, seed=0, device=None):
    """`npaths` independent draws of carma_process in ONE launch on the GPU (carma_simulate_carma: the same
    value-by-value construction, reference :1148-1259, normal variates from the counter-based generator keyed by
    (seed, path, step)).  Returns [npaths][n] at the sorted times."""
    from . import _lib
    r = np.atleast_1d(np.asarray(ar_roots, dtype=complex))
    if r.size == 1:
        return car1_process_batch(time, sigsqr, -1.0 / r.real.item(), npaths, seed, device)
    return _lib.simulate_carma(time, sigsqr, r, np.asarray(ma_coefs, dtype=float), npaths, seed, device)


def car1_process_batch(time, sigsqr, tau, npaths=1, seed=0, device=None):
    """`npaths` independent Ornstein-Uhlenbeck paths (car1_process, reference :1126-1146) in one launch."""
    from . import _lib
    return _lib.simulate_car1(time, sigsqr, 1.0 / tau, npaths, seed, device)


# ------------------------------------------------------------------------------------------------
class BatchResult(object):
    """scipy.optimize.OptimizeResult look-alike for one start of the lock-step optimiser (carma_mle_batched)."""

    def __init__(self, x, fun, nit, nfev, success, message):
        self.x, self.fun, self.nit, self.nfev, self.success, self.message = x, fun, nit, nfev, success, message

    def __repr__(self):
        return "BatchResult(fun=%r, nit=%d, success=%r)" % (self.fun, self.nit, self.success)


# status codes of carma_mle_batched (include/carma_mi355.h) in words
STATUS_TEXT = ("converged: projected gradient <= gtol", "converged: relative reduction of f <= ftol",
               "maximum number of iterations reached", "line search failed")


class MCMCSample(object):
    """Minimal sample container (the reference's samplers.MCMCSample holds the same `_samples` dict;
    its plotting/diagnostic methods are outside the hot path)."""

    def __init__(self, filename=None, logpost=None, trace=None):
        # reference samplers.py:27-45: a trace wins over a file name
        self._samples = {}
        if trace is not None:
            self.generate_from_trace(trace)
        elif filename is not None:
            self.generate_from_file([filename])
        if logpost is not None:
            self.set_logpost(logpost)

    def get_samples(self, name):
        return self._samples[name].copy()

    def generate_from_file(self, filename):
        """One parameter per ascii file, its name on the first line (reference samplers.py:57-72).  `filename` is a
        list of file names."""
        for fname in filename:
            with open(fname, "r") as f:
                name = f.readline()
            trace = np.genfromtxt(fname, skip_header=1)
            if name not in self._samples:
                self._samples[name] = trace

    def set_logpost(self, logpost):
        self._samples["logpost"] = np.asarray(logpost)

    def newaxis(self):
        for k, v in self._samples.items():
            if v.ndim == 1:
                self._samples[k] = v[:, np.newaxis]

    def posterior_summaries(self, name):
        s = self._samples[name]
        out = dict(median=np.median(s, axis=0), mean=np.mean(s, axis=0), std=np.std(s, axis=0),
                   ci68=np.percentile(s, [16.0, 84.0], axis=0), ci95=np.percentile(s, [2.5, 97.5], axis=0))
        return out


def _roots_from_log_quads(logq):
    """[nsamples, m] log quadratic-factor coefficients -> [nsamples, m] complex roots
    (CARp::ARRoots ordering, src/carpack.cpp:137-172)."""
    logq = np.atleast_2d(logq)
    ns, m = logq.shape
    quad = np.exp(logq)
    roots = np.empty((ns, m), dtype=complex)
    for i in range(m // 2):
        q1, q2 = quad[:, 2 * i], quad[:, 2 * i + 1]
        disc = q2 * q2 - 4.0 * q1
        sq = np.where(disc > 0, np.sqrt(np.abs(disc)) + 0j, 1j * np.sqrt(np.abs(disc)))
        roots[:, 2 * i] = -0.5 * (q2 + sq)
        # two real roots: the smaller one from the product q1 (as the kernels do, carma_core.h quad_root) --
        # -(q2 - sq) / 2 cancels to nothing once q2^2 >> 4 q1
        with np.errstate(divide="ignore", invalid="ignore"):
            roots[:, 2 * i + 1] = np.where((disc > 0) & (q2 - sq != 0), q1 / roots[:, 2 * i], -0.5 * (q2 - sq))
    if m % 2:
        roots[:, -1] = -quad[:, -1]
    return roots


def _poly_from_roots(roots):
    """Vectorised np.poly over the first axis: [ns, m] roots -> [ns, m+1] coefficients, highest first."""
    ns, m = roots.shape
    coefs = np.zeros((ns, m + 1), dtype=complex)
    coefs[:, 0] = 1.0
    for i in range(m):
        coefs[:, 1:i + 2] = coefs[:, 1:i + 2] - roots[:, i:i + 1] * coefs[:, 0:i + 1]
    return coefs


class CarmaSample(MCMCSample):
    """MCMC samples of a CARMA(p,q) model plus derived quantities (reference :263-546)."""

    def __init__(self, time, y, ysig, sampler, q=0, filename=None, MLE=None):
        self.time, self.y, self.ysig, self.q = time, y, ysig, q
        self._sampler = sampler
        logpost = np.array(sampler.GetLogLikes())
        trace = np.array(sampler.getSamples())
        # (as in the reference, :290, `filename` only matters when the sampler holds no trace: generate_from_file below
        # reads the ascii file the C++ carpack wrote)
        super(CarmaSample, self).__init__(filename=filename, logpost=logpost, trace=trace)
        self._ar_roots()
        self._ar_coefs()
        self._ma_coefs(trace)
        self._sigma_noise()
        # "loglik": LogDensity with the prior bounds ignored -- still includes the measurement-error
        # prior, exactly as the reference computes it (:305-315, carpack.hpp:173).  One batched launch.
        if hasattr(sampler, "SetMLE"):
            sampler.SetMLE(True)
        self._samples["loglik"] = np.asarray(sampler.getLogDensityBatch(trace))
        self.parameters = list(self._samples.keys())
        self.newaxis()
        self.mle = {}
        if MLE is not None:
            self.add_mle(MLE)

    def generate_from_file(self, filename):
        """Samples from an ascii file written by the C++ carpack: one header line, then one row per sample holding the
        parameter vector followed by the log-posterior (reference :427-437; `filename` is a list, its first entry is
        read)."""
        trace = np.atleast_2d(np.genfromtxt(filename[0], skip_header=1))
        self.generate_from_trace(trace[:, 0:-1])
        self.set_logpost(trace[:, -1])

    def generate_from_trace(self, trace):
        self.p = trace.shape[1] - 3 - self.q          # sic: p inferred from the trace width (:415)
        self._samples["var"] = trace[:, 0] ** 2
        self._samples["measerr_scale"] = trace[:, 1]
        self._samples["mu"] = trace[:, 2]
        self._samples["quad_coefs"] = np.exp(trace[:, 3:self.p + 3])

    def _ar_roots(self):
        roots = _roots_from_log_quads(np.log(self._samples["quad_coefs"]))
        self._samples["ar_roots"] = roots
        self._samples["psd_centroid"] = np.abs(roots.imag) / (2.0 * np.pi)
        self._samples["psd_width"] = -roots.real / (2.0 * np.pi)

    def _ar_coefs(self):
        self._samples["ar_coefs"] = _poly_from_roots(self._samples["ar_roots"]).real

    def _ma_coefs(self, trace):
        ns = trace.shape[0]
        if self.q == 0:
            self._samples["ma_coefs"] = np.ones((ns, 1))
            return
        roots = _roots_from_log_quads(trace[:, 3 + self.p:3 + self.p + self.q])
        c = _poly_from_roots(roots)
        self._samples["ma_coefs"] = (c / c[:, self.q:self.q + 1])[:, ::-1].real

    def _sigma_noise(self):
        """sigma of the driving noise per sample = sqrt(var / Variance(roots, ma, 1)) (reference :513-546): one launch for all
        samples (carma_sigma_noise_batch)."""
        self._samples["sigma"] = carmcmcLib.sigma_noise_batch(self._samples["ar_roots"], self._samples["ma_coefs"],
                                                              self._samples["var"])

    def add_mle(self, MLE):
        x = np.asarray(MLE.x, dtype=float)
        roots = _roots_from_log_quads(x[None, 3:self.p + 3])[0]
        self.mle = {"loglik": -MLE.fun, "var": x[0] ** 2, "measerr_scale": x[1], "mu": x[2], "ar_roots": roots,
                    "psd_width": -roots.real / (2 * np.pi), "psd_cent": np.abs(roots.imag) / (2 * np.pi),
                    "ar_coefs": np.poly(roots).real}
        if self.q == 0:
            self.mle["ma_coefs"] = 1.0
        else:
            mr = _roots_from_log_quads(x[None, 3 + self.p:])[0]
            c = np.poly(mr)
            self.mle["ma_coefs"] = np.real(c / c[self.q])[::-1]
        unit = carma_variance(1.0, roots, np.atleast_1d(self.mle["ma_coefs"]))
        self.mle["sigma"] = np.sqrt(self.mle["var"] / unit)

    def DIC(self):
        """Deviance information criterion from the stored log-likelihoods."""
        loglik = self._samples["loglik"].ravel()
        dev = -2.0 * loglik
        return float(np.mean(dev) + 0.5 * np.var(dev))

    def _psd_frequencies(self, nfreq=1000):
        """Log-spaced grid between 1 / (time span) and 0.5 / (smallest time step) (reference :583-594)."""
        dt_min = np.diff(self.time).min()
        dt_max = self.time.max() - self.time.min()
        return np.exp(np.linspace(np.log(1.0 / dt_max), np.log(0.5 / dt_min), num=nfreq))

    @staticmethod
    def _subsample(nsamples, nsamples0):
        """The evenly spaced sample indices the reference uses when nsamples < all (:572-578)."""
        if nsamples is None or nsamples >= nsamples0:
            return np.arange(nsamples0)
        return (np.arange(nsamples) * (nsamples0 / nsamples)).astype(int)

    def _psd_inputs(self, index):
        sig = np.ravel(self._samples["sigma"])[index]
        return self._samples["ar_coefs"][index], self._samples["ma_coefs"][index], sig

    def _psd_samples(self, frequencies, index):
        """sigma^2 |delta(2 pi i f)|^2 / |alpha(2 pi i f)|^2 for every (frequency, sample) pair (reference :601-618): the
        [nfreq, nsamples] grid from the device (carma_psd_band without percentiles)."""
        ar, ma, sig = self._psd_inputs(index)
        return carmcmcLib.psd_band(ar, ma, sig, frequencies, [], return_samples=True)[1]

    def _psd_credint(self, percentile, nsamples, frequencies):
        """(lower, median, upper) of the spectrum over the samples at every frequency (reference :596-623): grid and
        percentiles on the device, one call (carma_psd_band)."""
        index = self._subsample(nsamples, self._samples["sigma"].shape[0])
        lower = (100.0 - percentile) / 2.0
        ar, ma, sig = self._psd_inputs(index)
        return carmcmcLib.psd_band(ar, ma, sig, frequencies, [lower, 50.0, 100.0 - lower])

    def plot_power_spectrum(self, percentile=68.0, nsamples=None, plot_log=True, color="b", alpha=0.5, sp=None,
                            doShow=True):
        """Posterior median and `percentile` credibility band of the power spectrum on 1000 log-spaced frequencies
        (reference :548-648).  Returns the reference's tuple (lower PSD, upper PSD, median PSD, frequencies).  The
        numbers are computed for every (frequency, sample) pair at once; drawing happens only when a subplot is
        passed or doShow is true (matplotlib is imported lazily -- plotting is not part of the hot path)."""
        frequencies = self._psd_frequencies()
        ci = self._psd_credint(percentile, nsamples, frequencies)
        if sp is not None or doShow:
            import matplotlib.pyplot as plt
            if sp is None:
                sp = plt.figure().add_subplot(111)
            (sp.loglog if plot_log else sp.plot)(frequencies, ci[:, 1], color=color)
            sp.fill_between(frequencies, ci[:, 2], ci[:, 0], facecolor=color, alpha=alpha)
            sp.set_xlim(frequencies.min(), frequencies.max())
            sp.set_xlabel("Frequency")
            sp.set_ylabel("Power Spectrum")
            if doShow:
                plt.show()
        return ci[:, 0], ci[:, 2], ci[:, 1], frequencies

    def power_spectrum_band(self, percentile=68.0, nsamples=None, freq=None):
        """plot_power_spectrum's numbers on a caller-chosen frequency grid, without any drawing."""
        frequencies = self._psd_frequencies() if freq is None else np.asarray(freq, dtype=float)
        ci = self._psd_credint(percentile, nsamples, frequencies)
        return ci[:, 0], ci[:, 2], ci[:, 1], frequencies

    def makeKalmanFilter(self, bestfit):
        """KalmanFilterp for a point estimate ('map', 'median', 'mean' or a sample index),
        reference :650-685."""
        if bestfit == "map":
            i = int(np.argmax(self._samples["logpost"]))
            pick = lambda a: a[i]  # noqa: E731
        elif bestfit == "median":
            pick = lambda a: np.median(a, axis=0)  # noqa: E731
        elif bestfit == "mean":
            pick = lambda a: np.mean(a, axis=0)  # noqa: E731
        else:
            pick = lambda a: a[int(bestfit)]  # noqa: E731
        sigsqr = float(np.ravel(pick(self._samples["sigma"]))[0]) ** 2
        mu = float(np.ravel(pick(self._samples["mu"]))[0])
        roots = np.atleast_1d(pick(self._samples["ar_roots"]))
        ma = np.atleast_1d(pick(self._samples["ma_coefs"]))
        omega = carmcmcLib.vecC(roots.tolist())
        kf = carmcmcLib.KalmanFilterp(carmcmcLib.vecD(self.time), carmcmcLib.vecD(self.y - mu),
                                      carmcmcLib.vecD(self.ysig), sigsqr, omega, carmcmcLib.vecD(ma.tolist()))
        return kf, mu

    def predict(self, time, bestfit="map"):
        """Expected value and variance of the series at `time` given the data and a point estimate of
        the parameters (reference :755-805).  All times go to the GPU in one batched launch instead of
        one full re-filter per time.  Returns (yhat, yhat_var)."""
        scalar = np.isscalar(time)
        kf, mu = self.makeKalmanFilter(bestfit)
        m, v = kf.PredictBatch(np.atleast_1d(time))
        return (m[0] + mu, v[0]) if scalar else (m + mu, v)

    def simulate(self, time, bestfit="map"):
        """Random draw of the process at `time` conditional on the data (reference :807-837)."""
        kf, mu = self.makeKalmanFilter(bestfit)
        return np.array(kf.Simulate(np.atleast_1d(time))) + mu

    def assess_fit(self, bestfit="map", nplot=256, doShow=False):
        """Numerical part of assess_fit (reference :687-753): the interpolated path on `nplot` times,
        the standardised residuals of the one-step predictions and their autocorrelation function.
        Plotting is outside the hot path."""
        kf, mu = self.makeKalmanFilter(bestfit)
        kf.Filter()
        kmean, kvar = np.array(kf.GetMean()), np.array(kf.GetVar())
        resid = (self.y - mu - kmean) / np.sqrt(kvar)
        tgrid = np.linspace(self.time.min(), self.time.max(), nplot)
        pm, pv = kf.PredictBatch(tgrid)
        r0 = resid - resid.mean()
        acf = np.correlate(r0, r0, mode="full")[r0.size - 1:] / np.sum(r0 * r0)
        return dict(time=tgrid, mean=pm + mu, var=pv, std_resid=resid, resid_acf=acf)


class Car1Sample(CarmaSample):
    """Samples of a CAR(1) model (reference :866-1035): theta = (sigma_y, scale, mu, ln omega)."""

    def __init__(self, time, y, ysig, sampler, filename=None):
        self.time, self.y, self.ysig, self.q, self.p = time, y, ysig, 0, 1
        self._sampler = sampler
        logpost = np.array(sampler.GetLogLikes())
        trace = np.array(sampler.getSamples())
        MCMCSample.__init__(self, logpost=logpost, trace=trace)
        self._samples["loglik"] = self._samples["logpost"] - np.array(
            [sampler.getLogPrior(carmcmcLib.vecD(row)) for row in trace])
        self.parameters = list(self._samples.keys())
        self.newaxis()
        self.mle = {}

    def generate_from_trace(self, trace):
        omega = np.exp(trace[:, 3])
        self._samples["var"] = trace[:, 0] ** 2
        self._samples["measerr_scale"] = trace[:, 1]
        self._samples["mu"] = trace[:, 2]
        self._samples["log_omega"] = trace[:, 3]
        self._samples["ar_roots"] = (-omega)[:, None] + 0j
        self._samples["psd_centroid"] = np.zeros((trace.shape[0], 1))
        self._samples["psd_width"] = omega[:, None] / (2.0 * np.pi)
        self._samples["ar_coefs"] = np.c_[np.ones_like(omega), omega]
        self._samples["ma_coefs"] = np.ones((trace.shape[0], 1))
        self._samples["sigma"] = np.sqrt(2.0 * omega * trace[:, 0] ** 2)

    def makeKalmanFilter(self, bestfit):
        """KalmanFilter1 for a point estimate (reference :925-948): 'map', 'median', anything else = posterior mean
        (of sigma^2, mu and log omega -- as the reference does); an integer picks one sample (as CarmaSample)."""
        sig, mu_s, lw = (np.ravel(self._samples[k]) for k in ("sigma", "mu", "log_omega"))
        if bestfit == "map":
            i = int(np.argmax(self._samples["logpost"]))
            sigsqr, mu, log_omega = sig[i] ** 2, mu_s[i], lw[i]
        elif bestfit == "median":
            sigsqr, mu, log_omega = np.median(sig) ** 2, np.median(mu_s), np.median(lw)
        elif isinstance(bestfit, (int, np.integer)):
            i = int(bestfit)
            sigsqr, mu, log_omega = sig[i] ** 2, mu_s[i], lw[i]
        else:
            sigsqr, mu, log_omega = np.mean(sig ** 2), np.mean(mu_s), np.mean(lw)
        kf = carmcmcLib.KalmanFilter1(carmcmcLib.vecD(self.time), carmcmcLib.vecD(self.y - mu),
                                      carmcmcLib.vecD(self.ysig), float(sigsqr), float(np.exp(log_omega)))
        return kf, float(mu)

    # (the spectrum sigma^2 / (omega^2 + (2 pi f)^2) of the reference (:1004-1013) is the general formula with
    # alpha(s) = s + omega, delta = 1 -- the arrays generate_from_trace stores -- so CarmaSample's device path serves it)


# ------------------------------------------------------------------------------------------------
def _vec(a):
    v = carmcmcLib.vecD()
    v.extend(np.asarray(a, dtype=float).tolist())
    return v


class CarmaModel(object):
    """Statistical inference with a CARMA(p,q) model (reference :12-192)."""

    def __init__(self, time, y, ysig, p=1, q=0):
        time, y, ysig = np.asarray(time, dtype=float), np.asarray(y, dtype=float), np.asarray(ysig, dtype=float)
        if not p > q:
            raise ValueError("Order of AR polynomial, p, must be larger than order of MA polynomial, q.")
        _, idx = np.unique(time, return_index=True)    # sorted, first occurrence of each time
        self.time, self.y, self.ysig = time[idx], y[idx], ysig[idx]
        self._time, self._y, self._ysig = _vec(self.time), _vec(self.y), _vec(self.ysig)
        self.p, self.q = p, q
        self.mcmc_sample = None

    def run_mcmc(self, nsamples, nburnin=None, ntemperatures=None, nthin=1, init=None, nreplicas=1, seed=None, dist=None):
        """Parallel-tempered RAM sampler on the GPU; defaults as the reference (:53-89):
        ntemperatures = max(10, p+q), nburnin = nsamples/2.  `dist`: an initialised torch.distributed module (one
        process per GPU, every rank makes the same call): the `nreplicas` independent ladders are split over the ranks
        and every rank returns the gathered samples of all of them (parallel.sharded_pt_run) -- the same arrays as the
        single-process call with the same seed."""
        if ntemperatures is None:
            ntemperatures = max(10, self.p + self.q)
        if nburnin is None:
            nburnin = nsamples // 2
        init = carmcmcLib.vecD() if init is None else _vec(init)
        if self.p == 1:
            cpp = carmcmcLib.run_mcmc_car1(nsamples, int(nburnin), self._time, self._y, self._ysig, nthin, init,
                                           nreplicas=nreplicas, seed=seed, dist=dist)
            sample = Car1Sample(self.time, self.y, self.ysig, cpp)
        else:
            cpp = carmcmcLib.run_mcmc_carma(nsamples, int(nburnin), self._time, self._y, self._ysig, self.p, self.q,
                                            ntemperatures, False, nthin, init, nreplicas=nreplicas, seed=seed, dist=dist)
            sample = CarmaSample(self.time, self.y, self.ysig, cpp, q=self.q)
        self.mcmc_sample = sample
        return sample

    # -- maximum likelihood ---------------------------------------------------------------------
    def _mle_bounds(self, p, q):
        """L-BFGS-B box of the reference (:219-240)."""
        ysigma = self.y.std()
        dt = np.diff(self.time)
        max_freq, min_freq = 0.9 / dt.min(), 1.0 / (self.time.max() - self.time.min())
        bnds = [(ysigma / 10.0, 10.0 * ysigma), (0.9, 1.1), (None, None)]
        if p == 1:
            bnds.append((np.log(min_freq), np.log(max_freq)))
        else:
            lo = np.log(min(min_freq ** 2, 2.0 * min_freq))
            hi = np.log(max(max_freq ** 2, 2.0 * max_freq))
            bnds += [(lo, hi)] * p + [(None, None)] * q
        return bnds

    def _mle_problem(self, p, q, ntrials, seed):
        """(model object, [ntrials, d] starting points, L-BFGS-B box) of a get_mle call (reference :195-240)."""
        if p == 1:
            proc = carmcmcLib.run_mcmc_car1(1, 25, self._time, self._y, self._ysig, 1, nreplicas=ntrials, seed=seed)
        else:
            proc = carmcmcLib.run_mcmc_carma(1, 25, self._time, self._y, self._ysig, p, q, 10, False, 1,
                                             nreplicas=ntrials, seed=seed)
            proc.SetMLE(True)
        starts = proc.getAllSamples()[0][:, 0, :].copy()
        bnds = self._mle_bounds(p, q)
        rng = np.random.default_rng(seed)
        starts[:, 1] = 1.0                                   # initial guess for the error scale (:217)
        for j, (lo, hi) in enumerate(bnds):
            if lo is not None:
                out = (starts[:, j] < lo) | (starts[:, j] > hi)
                starts[out, j] = rng.uniform(lo, hi, int(out.sum()))
        return proc, starts, bnds

    def get_mle(self, p, q, ntrials=100, njobs=1, seed=None, method="batched", return_all=False):
        """Best of `ntrials` bounded quasi-Newton fits started from short tempered MCMC runs
        (reference :92-129,195-260).  The reference launches ntrials separate 26-iteration samplers
        and calls the C++ log-density once per function evaluation; here ONE sampler call with
        `ntrials` independent replicas provides all starting points, and with method="batched" all
        starts are optimised in lock-step (carma_mle.hip, C ABI carma_mle_batched): one launch evaluates the
        finite-difference stencils of every start.  method="scipy" runs scipy's L-BFGS-B per start
        with a batched gradient.  `njobs` is accepted for compatibility.  Returns an object with
        .x, .fun (= -loglik), .message like scipy's OptimizeResult (return_all=True: the list of all ntrials results,
        in the order of the starts)."""
        proc, starts, bnds = self._mle_problem(p, q, ntrials, seed)
        d = starts.shape[1]

        if method == "batched":
            # the lock-step optimiser inside the library (carma_mle.hip): no interpreter between the launches
            xs, fs, nits, nfevs, sts = proc.minimizeBatch(starts, bnds)
            results = [BatchResult(xs[i].copy(), float(fs[i]), int(nits[i]), int(nfevs[i]), int(sts[i]) < 2, STATUS_TEXT[int(sts[i])])
                       for i in range(xs.shape[0])]
            if return_all:
                return results
            results = [r for r in results if np.isfinite(r.fun) and r.fun < 1e299] or results
            return min(results, key=lambda r: r.fun)
        if method != "scipy":
            raise ValueError("method must be 'batched' or 'scipy'")

        def fun_and_grad(x):
            h = 1e-6 * np.maximum(1.0, np.abs(x))
            pts = np.tile(x, (2 * d + 1, 1))
            pts[1:d + 1] += np.diag(h)
            pts[d + 1:] -= np.diag(h)
            f = -np.asarray(proc.getLogDensityBatch(pts))
            fp, fm = f[1:d + 1], f[d + 1:]
            ok = np.isfinite(fp) & np.isfinite(fm)             # a stencil point outside the bounds is +inf: no inf - inf
            g = np.where(ok, np.where(ok, fp, 0.0) - np.where(ok, fm, 0.0), 0.0) / (2.0 * h)
            return (f[0] if np.isfinite(f[0]) else 1e300), g

        results = [minimize(fun_and_grad, x0, jac=True, method="L-BFGS-B", bounds=bnds) for x0 in starts]
        return results if return_all else min(results, key=lambda r: r.fun)

    def choose_order(self, pmax, qmax=None, pqlist=None, njobs=1, ntrials=100, seed=None, method="batched"):
        """Minimise AICc over a (p,q) grid (reference :131-192); sets self.p, self.q."""
        if pmax < 1:
            raise ValueError("Order of AR polynomial must be at least 1.")
        if qmax is None:
            qmax = pmax - 1
        if pqlist is None:
            pqlist = [(p, q) for p in range(1, pmax + 1) for q in range(min(p, qmax + 1))]
        # njobs (reference :131: processes of a multiprocessing pool, -1 = all cores): here THREADS, each driving its own
        # orders -- every order has its own context and stream, the library calls release the interpreter lock, and the
        # launches of one order (a few thousand evaluations) leave most of the chip to the others
        nthreads = (os.cpu_count() or 1) if njobs is not None and njobs < 0 else max(1, int(njobs or 1))
        if nthreads > 1 and len(pqlist) > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(nthreads, len(pqlist))) as pool:
                MLEs = list(pool.map(lambda pq: self.get_mle(pq[0], pq[1], ntrials=ntrials, seed=seed, method=method), pqlist))
        else:
            MLEs = [self.get_mle(p, q, ntrials=ntrials, njobs=njobs, seed=seed, method=method) for p, q in pqlist]
        AICc, best, best_aicc = [], MLEs[0], 1e300
        n = self.time.size
        for mle, (p, q) in zip(MLEs, pqlist):
            k = 2 + p + q
            a = 2.0 * k + 2.0 * mle.fun + 2.0 * k * (k + 1.0) / (n - k - 1.0)
            AICc.append(a)
            if a < best_aicc:
                best, best_aicc, self.p, self.q = mle, a, p, q
        return best, pqlist, AICc
