"""Builds/loads the CPU lane emulator of the kernel core (tests/emu) -- test harness only."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "emu", "libcarma_emu.so")
_dp = C.POINTER(C.c_double)
_lib = None


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(HERE, "emu", "emu_core.cpp"), os.path.join(HERE, "emu", "grp_emu.h")]
        csrc = os.path.join(ROOT, "carma_pack_amd", "csrc")
        srcs += [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
        if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-mfma",
                                   "-o", SO, srcs[0]])
        _lib = C.CDLL(SO)
    return _lib


def pack_series(t, y, yerr):
    s = np.zeros((t.size, 4))
    s[1:, 0] = np.diff(t)
    s[:, 1] = y
    s[:, 2] = yerr ** 2
    s[:, 3] = t
    return s


def _p(a):
    return a.ctypes.data_as(_dp)


def logdensity_carma(t, y, yerr, p, q, thetas, prior, ignore_prior=False):
    s = pack_series(t, y, yerr)
    th = np.ascontiguousarray(thetas, dtype=float).reshape(-1, 3 + p + q)
    pr = np.array(list(prior) + [50.0])
    out = np.empty(th.shape[0])
    rc = lib().emu_logdensity_carma(p, q, _p(th), th.shape[0], _p(s), t.size, _p(pr), int(ignore_prior), _p(out))
    assert rc == 0
    return out


def logdensity_carma_lane(t, y, yerr, p, q, thetas, prior, ignore_prior=False):
    """Same evaluation through the one-evaluation-per-lane code of the throughput regime (carma_lane.h)."""
    s = pack_series(t, y, yerr)
    th = np.ascontiguousarray(thetas, dtype=float).reshape(-1, 3 + p + q)
    pr = np.array(list(prior) + [50.0])
    out = np.empty(th.shape[0])
    rc = lib().emu_logdensity_carma_lane(p, q, _p(th), th.shape[0], _p(s), t.size, _p(pr), int(ignore_prior), _p(out))
    assert rc == 0
    return out


def logdensity_carma_row(t, y, yerr, p, q, thetas, prior, ignore_prior=False):
    """Same evaluation through the one-evaluation-per-DPP-row loop (G = 16, filter_loop_row)."""
    s = pack_series(t, y, yerr)
    th = np.ascontiguousarray(thetas, dtype=float).reshape(-1, 3 + p + q)
    pr = np.array(list(prior) + [50.0])
    out = np.empty(th.shape[0])
    rc = lib().emu_logdensity_carma_row(p, q, _p(th), th.shape[0], _p(s), t.size, _p(pr), int(ignore_prior), _p(out))
    assert rc == 0
    return out


def logdensity_car1(t, y, yerr, thetas, prior):
    s = pack_series(t, y, yerr)
    th = np.ascontiguousarray(thetas, dtype=float).reshape(-1, 4)
    pr = np.array(list(prior) + [50.0])
    out = np.empty(th.shape[0])
    lib().emu_logdensity_car1(_p(th), th.shape[0], _p(s), t.size, _p(pr), _p(out))
    return out


def kfilter_carma(t, y, yerr, sigsqr, omega, ma):
    s = pack_series(t, y, yerr)
    omega = np.asarray(omega, dtype=complex)
    p = omega.size
    re, im = np.ascontiguousarray(omega.real), np.ascontiguousarray(omega.imag)
    mav = np.zeros(p)
    mav[:len(ma)] = ma
    mean, var, ll = np.empty(t.size), np.empty(t.size), np.empty(1)
    lib().emu_kfilter_carma.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, _dp, C.c_int, _dp, _dp, _dp]
    rc = lib().emu_kfilter_carma(p, _p(re), _p(im), _p(mav), float(sigsqr), _p(s), t.size, _p(mean), _p(var), _p(ll))
    return mean, var, ll[0], rc


def pt_run(t, y, yerr, p, q, prior, temps, maxiter, niter, save_from, thin, seed, theta0, lp0, chol0):
    """Emulated persistent PT kernel for ONE replica (structure of k_pt in carma_pt.hip)."""
    s = pack_series(t, y, yerr)
    d = 4 if p == 1 else 3 + p + q
    T = len(temps)
    temps = np.ascontiguousarray(temps, dtype=float)
    theta = np.ascontiguousarray(theta0, dtype=float).reshape(T, d).copy()
    lp = np.ascontiguousarray(lp0, dtype=float).copy()
    chol = np.ascontiguousarray(chol0, dtype=float).reshape(T, d, d).copy()
    ns = max(0, (niter - save_from) // thin)
    samples, slp = np.zeros((ns, d)), np.zeros(ns)
    nacc, nswap = np.zeros(T, dtype=np.uint32), np.zeros(T, dtype=np.uint32)
    pr = np.array(list(prior) + [50.0])
    up = C.POINTER(C.c_uint)
    L = lib()
    L.emu_pt_run.argtypes = [C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_long, C.c_int, C.c_int,
                             C.c_uint, C.c_uint, _dp, _dp, _dp, _dp, _dp, up, up]
    rc = L.emu_pt_run(p, q, _p(s), t.size, _p(pr), T, _p(temps), maxiter, niter, save_from, thin, seed & 0xffffffff,
                      seed >> 32, _p(theta), _p(lp), _p(chol), _p(samples), _p(slp), nacc.ctypes.data_as(up),
                      nswap.ctypes.data_as(up))
    assert rc == 0
    return dict(theta=theta, lp=lp, chol=chol, samples=samples, logpost=slp, nacc=nacc, nswap=nswap)


def rng_draws(seed, chain, n):
    t8, u = np.empty(n), np.empty(n)
    L = lib()
    L.emu_rng_draws.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_long, _dp, _dp]
    L.emu_rng_draws(seed & 0xffffffff, seed >> 32, chain, n, _p(t8), _p(u))
    return t8, u


def predict_carma(t, y, yerr, sigsqr, omega, ma, times):
    s = pack_series(t, y, yerr)
    omega = np.asarray(omega, dtype=complex)
    p = omega.size
    re, im = np.ascontiguousarray(omega.real), np.ascontiguousarray(omega.imag)
    mav = np.zeros(p)
    mav[:len(ma)] = ma
    times = np.ascontiguousarray(np.atleast_1d(times), dtype=float)
    pm, pv = np.empty(times.size), np.empty(times.size)
    L = lib()
    L.emu_predict_carma.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, _dp, C.c_int, _dp, C.c_int, _dp, _dp]
    rc = L.emu_predict_carma(p, _p(re), _p(im), _p(mav), float(sigsqr), _p(s), t.size, _p(times), times.size, _p(pm), _p(pv))
    assert rc == 0
    return pm, pv


def predict_car1(t, y, yerr, sigsqr, omega, times):
    s = pack_series(t, y, yerr)
    times = np.ascontiguousarray(np.atleast_1d(times), dtype=float)
    pm, pv = np.empty(times.size), np.empty(times.size)
    L = lib()
    L.emu_predict_car1.argtypes = [C.c_double, C.c_double, _dp, C.c_int, _dp, C.c_int, _dp, _dp]
    L.emu_predict_car1(float(sigsqr), float(omega), _p(s), t.size, _p(times), times.size, _p(pm), _p(pv))
    return pm, pv
