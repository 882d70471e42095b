"""ctypes front-end of oracle/carma_oracle.c (CPU oracle, test infrastructure only).

The C file restates the reference's algorithm (file:line citations are in the C source);
this module only marshals numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcarma_oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)


def build(force=False):
    """Compile libcarma_oracle.so with gcc (no-op when up to date)."""
    srcs = [os.path.join(_HERE, f) for f in ("carma_oracle.c", "carma_truth_q.c", "Makefile")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcarma_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_model_create.restype = C.c_void_p
        L.orc_model_create.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double]
        L.orc_model_destroy.argtypes = [C.c_void_p]
        L.orc_model_n.argtypes = [C.c_void_p]
        L.orc_model_n.restype = C.c_int
        L.orc_model_get_data.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.orc_model_get_prior.argtypes = [C.c_void_p, _dp]
        L.orc_log_prior.argtypes = [C.c_void_p, _dp]
        L.orc_log_prior.restype = C.c_double
        L.orc_check_prior_bounds_carma.argtypes = [C.c_void_p, _dp, C.c_int]
        L.orc_check_prior_bounds_carma.restype = C.c_int
        L.orc_check_prior_bounds_car1.argtypes = [C.c_void_p, _dp]
        L.orc_check_prior_bounds_car1.restype = C.c_int
        L.orc_logdensity_carma.argtypes = [C.c_void_p, _dp, C.c_int, _dp]
        L.orc_logdensity_carma.restype = C.c_double
        L.orc_logdensity_car1.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_logdensity_car1.restype = C.c_double
        L.orc_logdensity_batch.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, C.c_int, _dp]
        L.orc_kfilter_carma.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int, C.c_double, _dp, _dp, _dp,
                                        _dp, _dp]
        L.orc_kfilter_carma.restype = C.c_int
        L.orc_kfilter_car1.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, _dp, _dp]
        L.orc_ar_roots.argtypes = [_dp, C.c_int, _dp, _dp]
        L.orc_ma_coefs.argtypes = [_dp, C.c_int, C.c_int, _dp]
        L.orc_variance.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, C.c_double]
        L.orc_variance.restype = C.c_double
        L.orc_sort_dedup.argtypes = [C.c_int, _dp, _dp, _dp]
        L.orc_sort_dedup.restype = C.c_int
        L.orc_chol_update_r1.argtypes = [C.c_int, _dp, _dp, C.c_int]
        L.orc_max_threads.restype = C.c_int
        L.orc_predict_carma.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int, C.c_double, _dp, _dp, _dp, C.c_double, _dp, _dp]
        L.orc_predict_carma.restype = C.c_int
        L.orc_predict_car1.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double, _dp, _dp]
        L.orc_truth_logdensity.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, _dp, _dp]
        L.orc_truth_logdensity.restype = C.c_int
        L.orc_truth_filter.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, _dp, _dp, _dp]
        L.orc_truth_filter.restype = C.c_int
        L.orc_truth_variance.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int, _dp]
        L.orc_truth_variance.restype = C.c_int
        L.orc_sampler_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, _dp, _dp, _dp, _dp, _dp]
        L.orc_ram_step.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, C.c_double, C.c_double, C.c_long, C.c_long, _dp, _dp, C.c_double]
        L.orc_ram_step.restype = C.c_int
        L.orc_exchange.argtypes = [C.c_int, _dp, _dp, C.c_double, _dp, _dp, C.c_double, C.c_double]
        L.orc_exchange.restype = C.c_int
        _lib = L
    return _lib


def _a(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def _p(x):
    return x.ctypes.data_as(_dp)


def truth_logdensity(t, y, yerr, theta, p, q):
    """(log-likelihood + log prior, log-likelihood) of the reference's formulas in quad precision (carma_truth_q.c):
    the arbiter of the parity tests; same contract as tests/mp_truth.loglik_truth, ~100x faster."""
    t, y, yerr, theta = _a(t), _a(y), _a(yerr), _a(theta)
    out = np.empty(2)
    rc = lib().orc_truth_logdensity(t.size, _p(t), _p(y), _p(yerr), int(p), int(q), _p(theta), _p(out))
    if rc != 0:
        raise ValueError("orc_truth_logdensity: rc=%d" % rc)
    return float(out[0]), float(out[1])


def truth_variance(roots, ma, with_cond=False):
    """CARp::Variance(roots, ma, sigma = 1) (src/carpack.cpp:377-409) in quad precision, roots / ma taken as the doubles given:
    the arbiter of the device's sigma_noise.  with_cond: also the condition number of the sum over the roots
    (sum |term_k| / |sum term_k|), which bounds what ANY double-precision evaluation of the formula can deliver."""
    roots = np.asarray(roots, dtype=complex).ravel()
    ma = _a(np.asarray(ma, dtype=float).ravel())
    re, im, out = _a(roots.real.copy()), _a(roots.imag.copy()), np.empty(2)
    rc = lib().orc_truth_variance(roots.size, _p(re), _p(im), _p(ma), ma.size, _p(out))
    if rc != 0:
        raise ValueError("orc_truth_variance: rc=%d" % rc)
    return (float(out[0]), float(out[1] / abs(out[0]))) if with_cond else float(out[0])


def truth_filter(t, y, yerr, theta, p, q):
    """(mean[n], var[n]) of the reference's filter for theta in quad precision (oracle/carma_truth_q.c): the data are
    y - theta[2], the errors sqrt(theta[1]) yerr, as CARMA_Base::LogDensity hands them to the filter."""
    t, y, yerr, theta = _a(t), _a(y), _a(yerr), _a(theta)
    mean, var = np.empty(t.size), np.empty(t.size)
    rc = lib().orc_truth_filter(t.size, _p(t), _p(y), _p(yerr), int(p), int(q), _p(theta), _p(mean), _p(var))
    if rc != 0:
        raise RuntimeError("orc_truth_filter failed (%d)" % rc)
    return mean, var


def max_threads():
    return lib().orc_max_threads()


def ar_roots(theta, p):
    theta = _a(theta)
    re, im = np.empty(p), np.empty(p)
    lib().orc_ar_roots(_p(theta), p, _p(re), _p(im))
    return re + 1j * im


def ma_coefs(theta, p, q):
    theta = _a(theta)
    ma = np.empty(p)
    lib().orc_ma_coefs(_p(theta), p, q, _p(ma))
    return ma


def variance(roots, ma, sigma=1.0, dt=0.0):
    roots = np.asarray(roots, dtype=complex)
    p = roots.size
    re, im = _a(roots.real), _a(roots.imag)
    mav = np.zeros(p)
    mav[: len(ma)] = ma
    return lib().orc_variance(p, _p(re), _p(im), _p(mav), float(sigma), float(dt))


def sort_dedup(t, y, yerr):
    t, y, yerr = _a(t).copy(), _a(y).copy(), _a(yerr).copy()
    n = lib().orc_sort_dedup(t.size, _p(t), _p(y), _p(yerr))
    return t[:n], y[:n], yerr[:n]


def kfilter_carma(t, y, yerr, sigsqr, roots, ma):
    """Kalman mean/var for centred y and (already scaled) yerr."""
    t, y, yerr = _a(t), _a(y), _a(yerr)
    roots = np.asarray(roots, dtype=complex)
    p = roots.size
    re, im = _a(roots.real), _a(roots.imag)
    mav = np.zeros(p)
    mav[: len(ma)] = ma
    mean, var = np.empty(t.size), np.empty(t.size)
    rc = lib().orc_kfilter_carma(t.size, _p(t), _p(y), _p(yerr), p, float(sigsqr), _p(re), _p(im),
                                 _p(mav), _p(mean), _p(var))
    if rc != 0:
        raise RuntimeError("singular EigenMat solve")
    return mean, var


def kfilter_car1(t, y, yerr, sigsqr, omega):
    t, y, yerr = _a(t), _a(y), _a(yerr)
    mean, var = np.empty(t.size), np.empty(t.size)
    lib().orc_kfilter_car1(t.size, _p(t), _p(y), _p(yerr), float(sigsqr), float(omega), _p(mean), _p(var))
    return mean, var


def predict_carma(t, y, yerr, sigsqr, roots, ma, times):
    """KalmanFilterp::Predict for each time in `times` (y centred)."""
    t, y, yerr = _a(t), _a(y), _a(yerr)
    roots = np.asarray(roots, dtype=complex)
    p = roots.size
    re, im = _a(roots.real), _a(roots.imag)
    mav = np.zeros(p)
    mav[: len(ma)] = ma
    times = np.atleast_1d(np.asarray(times, dtype=float))
    mean, var = np.empty(times.size), np.empty(times.size)
    m1, v1 = np.empty(1), np.empty(1)
    for i, tp in enumerate(times):
        rc = lib().orc_predict_carma(t.size, _p(t), _p(y), _p(yerr), p, float(sigsqr), _p(re), _p(im), _p(mav),
                                     float(tp), _p(m1), _p(v1))
        if rc != 0:
            raise RuntimeError("singular EigenMat solve")
        mean[i], var[i] = m1[0], v1[0]
    return mean, var


def predict_car1(t, y, yerr, sigsqr, omega, times):
    t, y, yerr = _a(t), _a(y), _a(yerr)
    times = np.atleast_1d(np.asarray(times, dtype=float))
    mean, var = np.empty(times.size), np.empty(times.size)
    m1, v1 = np.empty(1), np.empty(1)
    for i, tp in enumerate(times):
        lib().orc_predict_car1(t.size, _p(t), _p(y), _p(yerr), float(sigsqr), float(omega), float(tp), _p(m1), _p(v1))
        mean[i], var[i] = m1[0], v1[0]
    return mean, var


def chol_update_r1(L, v, downdate):
    L = _a(L).copy()
    v = _a(v).copy()
    lib().orc_chol_update_r1(L.shape[0], _p(L), _p(v), int(bool(downdate)))
    return L, v


class OracleModel:
    """CARMA_Base-like object: data + prior bounds + LogDensity (p==1 -> CAR1)."""

    def __init__(self, t, y, yerr, p, q=0, max_stdev=None):
        t, y, yerr = _a(t), _a(y), _a(yerr)
        if max_stdev is None:
            # RunCarmaSampler's population variance (src/carmcmc.cpp:85-89)
            max_stdev = 10.0 * np.sqrt(np.mean(y * y) - np.mean(y) ** 2)
        self.p, self.q = int(p), int(q)
        self.d = 4 if self.p == 1 else 3 + self.p + self.q
        self._h = C.c_void_p(lib().orc_model_create(_p(t), _p(y), _p(yerr), t.size, self.p, self.q,
                                                    float(max_stdev)))
        self.n = lib().orc_model_n(self._h)
        pr = np.empty(3)
        lib().orc_model_get_prior(self._h, _p(pr))
        self.max_stdev, self.max_freq, self.min_freq = pr

    def __del__(self):
        try:
            lib().orc_model_destroy(self._h)
        except Exception:
            pass

    def data(self):
        t, y, e = np.empty(self.n), np.empty(self.n), np.empty(self.n)
        lib().orc_model_get_data(self._h, _p(t), _p(y), _p(e))
        return t, y, e

    def log_prior(self, theta):
        theta = _a(theta)
        return lib().orc_log_prior(self._h, _p(theta))

    def check_prior_bounds(self, theta, ignore_prior=False):
        theta = _a(theta)
        if self.p == 1:
            return bool(lib().orc_check_prior_bounds_car1(self._h, _p(theta)))
        return bool(lib().orc_check_prior_bounds_carma(self._h, _p(theta), int(ignore_prior)))

    def logdensity(self, theta, ignore_prior=False):
        theta = _a(theta)
        assert theta.size == self.d
        if self.p == 1:
            return lib().orc_logdensity_car1(self._h, _p(theta), None)
        return lib().orc_logdensity_carma(self._h, _p(theta), int(ignore_prior), None)

    def ram_step(self, theta, lp, R, z, u, temperature, niter, maxiter, lnew_rel_shift=0.0):
        """One AdaptiveMetro::DoStep (src/steps.cpp:60-107) with the variates z (t_8 vector) and u (Metropolis uniform) as
        inputs.  Returns (accepted, theta', lp', R', LogDensity(proposal)); the inputs are not modified.
        lnew_rel_shift moves LogDensity(proposal) by that fraction of its magnitude: the step's sensitivity to the
        log-density, for tolerance bands (0 = the reference's step)."""
        th, Rm = _a(theta).copy(), _a(R).copy()
        lpv, lnew = np.array([float(lp)]), np.zeros(1)
        work = np.empty(4 * self.n)
        acc = lib().orc_ram_step(self._h, _p(th), _p(lpv), _p(Rm), _p(_a(z)), float(u), float(temperature), int(niter),
                                 int(maxiter), _p(work), _p(lnew), float(lnew_rel_shift))
        return bool(acc), th, float(lpv[0]), Rm, float(lnew[0])

    @staticmethod
    def exchange(theta_this, lp_this, temp_this, theta_other, lp_other, temp_other, u):
        """One ExchangeStep::DoStep (src/include/steps.hpp:318-362): `this` is the warmer chain.  Returns
        (swapped, theta_this', lp_this', theta_other', lp_other')."""
        a, b = _a(theta_this).copy(), _a(theta_other).copy()
        la, lb = np.array([float(lp_this)]), np.array([float(lp_other)])
        sw = lib().orc_exchange(a.size, _p(a), _p(la), float(temp_this), _p(b), _p(lb), float(temp_other), float(u))
        return bool(sw), a, float(la[0]), b, float(lb[0])

    def sampler_run(self, ntemps, sample_size, burnin, thin, seed, start):
        """Literal restatement of RunCarmaSampler/RunCar1Sampler (serial hot->cold sweep), own RNG.
        start = [ntemps][d].  Returns dict(samples[S][d], logpost[S], accept_rate[T], swap_rate[T])."""
        start = _a(start).reshape(ntemps, self.d)
        samples, lp = np.empty((sample_size, self.d)), np.empty(sample_size)
        acc, swp = np.empty(ntemps), np.empty(ntemps)
        lib().orc_sampler_run(self._h, int(ntemps), int(sample_size), int(burnin), int(thin), C.c_uint64(int(seed)),
                              _p(start), _p(samples), _p(lp), _p(acc), _p(swp))
        return dict(samples=samples, logpost=lp, accept_rate=acc, swap_rate=swp)

    def logdensity_batch(self, thetas, ignore_prior=False, nthreads=1):
        thetas = _a(thetas).reshape(-1, self.d)
        out = np.empty(thetas.shape[0])
        lib().orc_logdensity_batch(self._h, _p(thetas), thetas.shape[0], int(ignore_prior), int(nthreads),
                                   _p(out))
        return out


class NativeComparator:
    """The SAME restatement built with -march=native (no fast-math) on the machine that runs it: the CPU comparator of
    SURVEY.md 8(d), used by bench.py's cpu_baseline leg only.  Never the parity oracle (FMA contraction is on here).
    variant "O3": -O3 -march=native as 8(d) words it; "O2": -O2 -march=native, which is 3x FASTER on this code (gcc's -O3
    vectoriser pessimises its short complex loops)."""

    _libs = {}
    FLAGS = {"O3": "-O3 -march=native", "O2": "-O2 -march=native"}

    @classmethod
    def _lib(cls, variant):
        if variant not in cls._libs:
            name = "libcarma_oracle_native.so" if variant == "O3" else "libcarma_oracle_native_o2.so"
            subprocess.check_call(["make", "-C", _HERE, "-B", name], stdout=subprocess.DEVNULL)
            L = C.CDLL(os.path.join(_HERE, name))
            L.orc_model_create.restype = C.c_void_p
            L.orc_model_create.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double]
            L.orc_model_destroy.argtypes = [C.c_void_p]
            L.orc_logdensity_batch.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, C.c_int, _dp]
            L.orc_max_threads.restype = C.c_int
            cls._libs[variant] = L
        return cls._libs[variant]

    def __init__(self, t, y, yerr, p, q, max_stdev, variant="O3"):
        t, y, yerr = _a(t), _a(y), _a(yerr)
        self.variant = variant
        self.p, self.q = int(p), int(q)
        self.d = 4 if self.p == 1 else 3 + self.p + self.q
        self._h = C.c_void_p(self._lib(variant).orc_model_create(_p(t), _p(y), _p(yerr), t.size, self.p, self.q, float(max_stdev)))

    def __del__(self):
        try:
            self._lib(self.variant).orc_model_destroy(self._h)
        except Exception:
            pass

    def max_threads(self):
        return self._lib(self.variant).orc_max_threads()

    def logdensity_batch(self, thetas, ignore_prior=False, nthreads=1):
        thetas = _a(thetas).reshape(-1, self.d)
        out = np.empty(thetas.shape[0])
        self._lib(self.variant).orc_logdensity_batch(self._h, _p(thetas), thetas.shape[0], int(ignore_prior), int(nthreads), _p(out))
        return out
