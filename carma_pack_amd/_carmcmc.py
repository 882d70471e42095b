"""Drop-in for carma_pack's compiled extension ``carmcmc._carmcmc``.

Same names, positional overloads and return conventions as the Boost.Python module
(/root/reference/src/boost_python_wrapper.cpp:28-101); everything numerical goes through the C ABI of
``libcarma_mi355.so`` (see INTEGRATION.md for the export-by-export map).  Extensions over the
reference are keyword-only (``nreplicas``, ``seed``, ``device``) or new methods
(``getLogDensityBatch``, ``getAllSamples``).
"""
import os

import numpy as np

from . import _lib

__all__ = ["vecD", "vecvecD", "vecC", "pairD", "CAR1", "CARp", "CARMA", "run_mcmc_car1", "run_mcmc_carma",
           "KalmanFilter1", "KalmanFilterp"]


class vecD(list):
    """std::vector<double> (wrapper :32-33): list-like with extend/append/indexing."""


class vecvecD(list):
    """std::vector<std::vector<double>> (wrapper :35-36)."""


class vecC(list):
    """std::vector<std::complex<double>> (wrapper :38-39)."""


class pairD(object):
    """std::pair<double,double> (wrapper :41-43)."""

    def __init__(self, first=0.0, second=0.0):
        self.first, self.second = first, second


def _arr(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64))


def _seed(seed):
    if seed is None:
        # the reference seeds its global mt19937 with time(NULL) (src/random.cpp:20)
        return int.from_bytes(os.urandom(8), "little")
    return int(seed)


class _CarmaBase(object):
    """CARMA_Base (src/include/carpack.hpp:51-249) as seen through the bindings."""

    _p_fixed = None

    def __init__(self, track, label, time, y, yerr, p, q, temperature=1.0, device=None):
        self._track, self._label, self._temperature = bool(track), str(label), float(temperature)
        self.p, self.q = int(p), int(q)
        # default prior bound of the ctor: 10*sqrt(arma::var(y)) (carpack.hpp:71)
        self._ctx = _lib.Context(_arr(time), _arr(y), _arr(yerr), self.p, self.q, device=device)
        self._ignore_prior = False
        self._samples = np.empty((0, self._ctx.d))
        self._logposts = np.empty(0)
        self._all_samples = None
        self._all_logposts = None

    # -- bindings -------------------------------------------------------------------------------
    def getLogPrior(self, theta):
        return float(self._ctx.logprior(_arr(theta)))

    def getLogDensity(self, theta):
        theta = _arr(theta)
        if theta.size != self._ctx.d:
            raise RuntimeError("parameter vector has length %d, expected %d" % (theta.size, self._ctx.d))
        return float(self._ctx.logdensity(theta, ignore_prior=self._ignore_prior))

    def getSamples(self):
        return vecvecD(vecD(row) for row in self._samples.tolist())

    def GetLogLikes(self):
        return vecD(self._logposts.tolist())

    # -- extensions -----------------------------------------------------------------------------
    def getLogDensityBatch(self, thetas):
        """[B][d] -> [B] in one launch (carma_logdensity_batch)."""
        return self._ctx.logdensity(np.asarray(thetas, dtype=float).reshape(-1, self._ctx.d),
                                    ignore_prior=self._ignore_prior)

    def minimizeBatch(self, starts, bounds, **kw):
        """Bounded quasi-Newton searches on -LogDensity from every row of `starts`, in lock-step, host loop in the
        library (carma_mle_batched): (x, fun, nit, nfev, status)."""
        return self._ctx.mle_batched(starts, bounds, ignore_prior=self._ignore_prior, **kw)

    def getAllSamples(self):
        """Samples of every independent replica: ([R][S][d], [R][S])."""
        return self._all_samples, self._all_logposts

    def SetPrior(self, max_stdev):
        self._ctx.set_prior(max_stdev)

    def _run(self, sample_size, burnin, ntemps, thin, init, nreplicas, seed, dist=None):
        init = _arr(init) if init is not None and len(init) else None
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            # replicas split over the ranks of the process group (one process per GPU), coldest chains gathered at the end
            from . import parallel
            dev = "cuda:%d" % self._ctx.device if dist.get_backend() == "nccl" else "cpu"
            s, lp = parallel.sharded_pt_run(lambda: self._ctx, ntemps, nreplicas, int(sample_size), int(burnin), int(thin),
                                            init, _seed(seed), dist, dev)
        else:
            s, lp = self._ctx.pt_run(ntemps, nreplicas, int(sample_size), int(burnin), int(thin), init, _seed(seed))
        self._all_samples, self._all_logposts = s, lp
        self._samples, self._logposts = s[0], lp[0]       # the reference returns one coldest chain
        self.accept_rate, self.swap_rate = self._ctx.pt_stats()


class CAR1(_CarmaBase):
    """CAR1(track, label, time, y, yerr[, temperature]) (wrapper :49-55; no SetMLE binding)."""

    def __init__(self, track, label, time, y, yerr, temperature=1.0, device=None):
        super(CAR1, self).__init__(track, label, time, y, yerr, 1, 0, temperature, device)


class CARp(_CarmaBase):
    """CARp(track, label, time, y, yerr, p[, temperature]) (wrapper :57-64)."""

    def __init__(self, track, label, time, y, yerr, p, temperature=1.0, device=None, _q=0):
        if int(p) < 2:
            raise RuntimeError("CARp needs p >= 2 (use CAR1 for p = 1)")
        super(CARp, self).__init__(track, label, time, y, yerr, p, _q, temperature, device)

    def SetMLE(self, ignore_prior):
        """carpack.hpp:232: skip the CARp prior bounds (the log prior is still added, :173)."""
        self._ignore_prior = bool(ignore_prior)


class CARMA(CARp):
    """CARMA(track, label, time, y, yerr, p, q[, temperature]) (wrapper :66-73)."""

    def __init__(self, track, label, time, y, yerr, p, q, temperature=1.0, device=None):
        if not int(q) < int(p):
            # BOOST_ASSERT_MSG(q < p, ...) carpack.hpp:377
            raise RuntimeError("Order of moving average polynomial must be less than order of "
                               "autoregressive polynomial")
        super(CARMA, self).__init__(track, label, time, y, yerr, p, temperature, device, _q=q)


# CarmaSample post-processing on the device (no counterpart in the reference's binding: there it is Python loops over the
# samples, carma_pack.py:513-546, 596-623); carma_pack.py calls these through this module like every other compute call
sigma_noise_batch = _lib.sigma_noise_batch
psd_band = _lib.psd_band


def _pop_max_stdev(y):
    # RunCar*Sampler: var = E[y^2] - E[y]^2 (population), max_stdev = 10 sqrt(var) (carmcmc.cpp:35-40,85-89)
    y = _arr(y)
    return 10.0 * np.sqrt(np.mean(y * y) - np.mean(y) ** 2)


def run_mcmc_car1(sample_size, burnin, time, y, yerr, thin=1, init=vecD(), **kw):
    """RunCar1Sampler (src/carmcmc.cpp:30-77): one RAM chain at temperature 1 -> CAR1 object."""
    nreplicas, seed, device, dist = kw.pop("nreplicas", 1), kw.pop("seed", None), kw.pop("device", None), kw.pop("dist", None)
    if kw:
        raise TypeError("unexpected arguments %r" % list(kw))
    obj = CAR1(True, "CAR(1)", time, y, yerr, device=device)
    obj.SetPrior(_pop_max_stdev(y))
    obj._run(sample_size, burnin, 1, thin, init, nreplicas, seed, dist)
    return obj


def run_mcmc_carma(sample_size, burnin, time, y, yerr, p, q, nwalkers, do_zcarma=False, thin=1, init=vecD(), **kw):
    """RunCarmaSampler (src/carmcmc.cpp:79-177): `nwalkers` tempered chains (T_i = 100^(i/(nwalkers-1))),
    RAM + exchange steps, returns the coldest chain as a CARp (q == 0) or CARMA object."""
    nreplicas, seed, device, dist = kw.pop("nreplicas", 1), kw.pop("seed", None), kw.pop("device", None), kw.pop("dist", None)
    if kw:
        raise TypeError("unexpected arguments %r" % list(kw))
    if not int(p) > 1:
        raise RuntimeError("run_mcmc_carma: assert(p > 1) (carmcmc.cpp:84)")
    if do_zcarma:
        raise NotImplementedError("ZCAR/ZCARMA parameterisation is out of scope (never reachable from "
                                  "carma_pack's Python API, carma_pack.py:84)")
    if int(q) == 0:
        obj = CARp(True, "CAR(p) Parameters", time, y, yerr, p, device=device)
    else:
        obj = CARMA(True, "CARMA(p,q) Parameters", time, y, yerr, p, q, device=device)
    obj.SetPrior(_pop_max_stdev(y))
    obj._run(sample_size, burnin, int(nwalkers), thin, init, nreplicas, seed, dist)
    return obj


class _KalmanBase(object):
    def __init__(self, time, y, yerr):
        self._t, self._y, self._e = _arr(time), _arr(y), _arr(yerr)
        self._mean = np.zeros(self._t.size)
        self._var = np.zeros(self._t.size)
        self._kf = None          # carma_kf handle: series + model resident in HBM from the first Filter / Predict on

    def _handle(self):
        if self._kf is None:
            self._kf = self._make_handle(self._t, self._y, self._e)
        return self._kf

    def GetMean(self):
        return vecD(self._mean.tolist())

    def GetVar(self):
        return vecD(self._var.tolist())

    def Predict(self, time):
        """(mean, variance) of the process at `time` given the series (kfilter.hpp:122) -> pairD."""
        m, v = self.PredictBatch([time])
        return pairD(float(m[0]), float(v[0]))

    def Simulate(self, time):
        """Draw the (noise-free) process at `time` conditional on the measured series (KalmanFilter::Simulate,
        kfilter.hpp:135-184).  The reference visits the times in ascending order and re-filters the growing series once
        per time (M full filters, each drawn value inserted with zero measurement error).  The same joint conditional
        distribution is drawn here with TWO launches, by Matheron's rule for Gaussian processes:
            f*|y  =  f~*  +  E[f* | y - y~],      f~ = an unconditional path on (data times U requested times),
                                                   y~ = f~(data times) + measurement noise
        -- one batched unconditional simulation (carma_simulate_*) and one batched Predict of the residual series."""
        times = np.sort(_arr(time))
        n = self._t.size
        tt = np.concatenate([self._t, times])
        order = np.argsort(tt, kind="stable")
        inv = np.empty(order.size, dtype=int)
        inv[order] = np.arange(order.size)
        path = self._simulate(tt[order], int(np.random.randint(0, 2 ** 62)))
        f_data, f_new = path[inv[:n]], path[inv[n:]]
        y_tilde = f_data + self._e * np.random.standard_normal(n)
        m, _ = self._predict(self._t, self._y - y_tilde, self._e, times)
        return vecD((f_new + m).tolist())


class KalmanFilter1(_KalmanBase):
    """KalmanFilter1(time, y, yerr[, sigsqr, omega]) (wrapper :83-90; kfilter.hpp:222-245)."""

    def __init__(self, time, y, yerr, sigsqr=None, omega=None):
        super(KalmanFilter1, self).__init__(time, y, yerr)
        self._sigsqr, self._omega = sigsqr, omega

    def Filter(self):
        if self._sigsqr is None or self._omega is None:
            raise RuntimeError("KalmanFilter1: sigsqr and omega are not set")
        self._mean, self._var = self._handle().filter()

    def _make_handle(self, t, y, e):
        if self._sigsqr is None or self._omega is None:
            raise RuntimeError("KalmanFilter1: sigsqr and omega are not set")
        return _lib.KalmanHandle(t, y, e, self._sigsqr, self._omega)

    def PredictBatch(self, times):
        """Extension: all times in one launch (carma_kf_predict)."""
        return self._handle().predict(_arr(times))

    def _predict(self, t, y, e, times):
        return _lib.predict_car1(t, y, e, self._sigsqr, self._omega, times)

    def _simulate(self, times, seed):
        return _lib.simulate_car1(times, self._sigsqr, self._omega, 1, seed)[0]


class KalmanFilterp(_KalmanBase):
    """KalmanFilterp(time, y, yerr[, sigsqr, omega(vecC), ma_coefs(vecD)]) (wrapper :91-101;
    kfilter.hpp:303-334: ma_coefs shorter than p are zero padded)."""

    def __init__(self, time, y, yerr, sigsqr=None, omega=None, ma_coefs=None):
        super(KalmanFilterp, self).__init__(time, y, yerr)
        self._sigsqr = sigsqr
        self._omega = None if omega is None else np.asarray(list(omega), dtype=complex)
        self._ma = None if ma_coefs is None else _arr(ma_coefs)

    def Filter(self):
        if self._sigsqr is None or self._omega is None or self._ma is None:
            raise RuntimeError("KalmanFilterp: sigsqr, omega and ma_coefs are not set")
        self._mean, self._var = self._handle().filter()

    def _make_handle(self, t, y, e):
        if self._sigsqr is None or self._omega is None or self._ma is None:
            raise RuntimeError("KalmanFilterp: sigsqr, omega and ma_coefs are not set")
        return _lib.KalmanHandle(t, y, e, self._sigsqr, self._omega, self._ma)

    def PredictBatch(self, times):
        """Extension: all times in one launch (carma_kf_predict)."""
        return self._handle().predict(_arr(times))

    def _predict(self, t, y, e, times):
        return _lib.predict_carma(t, y, e, self._sigsqr, self._omega, self._ma, times)

    def _simulate(self, times, seed):
        return _lib.simulate_carma(times, self._sigsqr, self._omega, self._ma, 1, seed)[0]
