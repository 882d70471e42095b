#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own Python.

Run in the build container only (needs /root/reference; nothing here travels to the GPU box
except the .npz/.json outputs):

    python tests/golden/make_golden.py

What is imported from the reference (src/carmcmc/carma_pack.py, with the compiled
``_carmcmc`` extension and ``acor`` stubbed out because they are absent here):

  * KalmanFilterDeprecated (:1264-1375)  -> Kalman mean[n], var[n]   (== kfilter.cpp:138-215)
  * carma_variance (:1084-1123)          -> sigsqr, autocovariance   (== carpack.cpp:377-409)
  * get_ar_roots (:1038-1059), carma_process (:1148-1259), car1_process (:1126-1146)
  * CarmaSample._ar_roots (:439-468), ._ma_coefs (:470-500)  -> theta -> (omega, beta) mirrors
    of carpack.cpp:137-172 / :522-580

The log-likelihood sum itself (carpack.hpp:167-171) has no Python mirror; it is restated here in
numpy exactly as the reference's own test states it (cpp_tests/carma_unit_tests.cpp:829-834) and
cross-checked against the dense Gaussian-process identity (carma_unit_tests.cpp:564-594).
"""
import importlib
import json
import os
import sys
import types

import numpy as np

os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference():
    pkg = types.ModuleType("carmcmc")
    pkg.__path__ = [os.path.join(REF, "src", "carmcmc")]
    sys.modules["carmcmc"] = pkg
    stub = types.ModuleType("carmcmc._carmcmc")
    stub.vecD = list
    stub.vecC = list
    sys.modules["carmcmc._carmcmc"] = stub
    sys.modules["acor"] = types.ModuleType("acor")
    return importlib.import_module("carmcmc.carma_pack")


cp = import_reference()


# ----------------------------------------------------------------------------------------------
def theta_to_model(theta, p, q):
    """theta -> (omega, beta[p], sigsqr) through the reference's Python mirrors."""
    trace = np.atleast_2d(np.asarray(theta, dtype=float))
    s = cp.CarmaSample.__new__(cp.CarmaSample)
    s.p, s.q = p, q
    s._samples = {"var": trace[:, 0] ** 2, "quad_coefs": np.exp(trace[:, 3:p + 3])}
    s._ar_roots()
    s._ma_coefs(trace)
    omega = s._samples["ar_roots"][0].copy()
    ma = np.zeros(p)
    mc = np.atleast_1d(s._samples["ma_coefs"][0])
    ma[: mc.size] = mc
    sigsqr = trace[0, 0] ** 2 / cp.carma_variance(1.0, omega, ma_coefs=ma)
    return omega, ma, float(sigsqr)


def ref_filter(t, y, yerr, theta, p, q):
    """Reference Kalman filter for parameter vector theta -> dict of golden outputs."""
    omega, ma, sigsqr = theta_to_model(theta, p, q)
    scale, mu = theta[1], theta[2]
    yvar = scale * yerr ** 2           # carpack.hpp:150 (sqrt(scale)*yerr)^2 ; ctor takes VARIANCE
    kf = cp.KalmanFilterDeprecated(t, y - mu, yvar, sigsqr, omega, ma_coefs=ma)
    mean, var = kf.filter()
    mean, var = np.array(mean, dtype=float), np.array(var, dtype=float)
    r = y - mu - mean
    loglik = float(np.sum(-0.5 * np.log(var) - 0.5 * r * r / var))   # carpack.hpp:167-171
    E = np.vander(omega, p, increasing=True).T
    return dict(omega=omega, ma=ma, sigsqr=sigsqr, mean=mean, var=var, loglik=loglik,
                cond=float(np.linalg.cond(E)))


def dense_gp_loglik(t, y, yerr, theta, p, q):
    """-1/2 ln det C - 1/2 r^T C^-1 r with C_ij = acf(|t_i-t_j|) + delta_ij scale*yerr_i^2."""
    omega, ma, sigsqr = theta_to_model(theta, p, q)
    n = t.size
    lags = np.abs(t[:, None] - t[None, :])
    Cm = np.zeros((n, n))
    for i in range(n):
        for j in range(i, n):
            Cm[i, j] = Cm[j, i] = cp.carma_variance(sigsqr, omega, ma_coefs=ma, lag=lags[i, j])
    Cm[np.diag_indices(n)] += theta[1] * yerr ** 2
    r = y - theta[2]
    L = np.linalg.cholesky(Cm)
    z = np.linalg.solve(L, r)
    return float(-np.sum(np.log(np.diag(L))) - 0.5 * z @ z)


def log_quads_from_roots(roots):
    """Inverse of ARRoots (carpack.cpp:137-172): conj pairs (negative imag first) + optional real."""
    p = len(roots)
    out = []
    for i in range(p // 2):
        r = roots[2 * i]
        out += [np.log(abs(r) ** 2), np.log(-2.0 * r.real)]
    if p % 2:
        out.append(np.log(-roots[-1].real))
    return out


def prior_like_theta(rng, p, q, t, y):
    """Starting-value distribution of carpack.cpp:268-311,416-477,515-519 (own RNG)."""
    n = y.size
    dt = np.diff(t)
    max_freq, min_freq = 1.0 / dt.min(), 1.0 / (t.max() - t.min())
    nc = (p + 1) // 2
    cent = np.exp(np.log(max_freq / min_freq) * rng.uniform(size=nc) + np.log(min_freq))
    cent = np.sort(cent)[::-1]
    width = np.exp(np.log(max_freq / min_freq) * rng.uniform(size=nc) + np.log(min_freq))
    loga = np.empty(p)
    if p % 2:
        cent[p // 2] = 0.0
        lo = np.log(min_freq)
        hi = np.log(cent[p // 2 - 1]) if p > 1 else np.log(max_freq)
        width[p // 2] = np.exp(rng.uniform(lo, hi))
    for i in range(p // 2):
        re_, im_ = -2 * np.pi * width[i], 2 * np.pi * cent[i]
        loga[2 * i] = np.log(re_ * re_ + im_ * im_)
        loga[2 * i + 1] = np.log(-2.0 * re_)
    if p % 2:
        loga[p - 1] = np.log(2 * np.pi * width[p // 2])
    ma = np.abs(rng.standard_normal(q))
    yvar = np.var(y, ddof=1) * (n - 1) / rng.chisquare(n - 1)
    mu = rng.normal(np.mean(y), np.sqrt(yvar) / n)
    scale = min(max(50.0 / rng.chisquare(50), 0.51), 1.99)
    return np.concatenate([[np.sqrt(yvar), scale, mu], loga, ma])


# ----------------------------------------------------------------------------------------------
def readme_series():
    """README.md:23-46, legacy RNG because carma_process draws from np.random."""
    np.random.seed(0)
    sigmay, p, mu = 2.3, 5, 17.0
    qpo_width = np.array([1.0 / 100.0, 1.0 / 300.0, 1.0 / 200.0])
    qpo_cent = np.array([1.0 / 5.0, 1.0 / 25.0])
    ar_roots = cp.get_ar_roots(qpo_width, qpo_cent)
    ma_coefs = np.zeros(p)
    ma_coefs[0], ma_coefs[1], ma_coefs[2] = 1.0, 4.5, 1.25
    sigsqr = sigmay ** 2 / cp.carma_variance(1.0, ar_roots, ma_coefs=ma_coefs)
    ny = 270
    time = np.empty(ny)
    dt = np.random.uniform(1.0, 3.0, ny)
    time[:90] = np.cumsum(dt[:90])
    time[90:2 * 90] = 180 + time[90 - 1] + np.cumsum(dt[90:2 * 90])
    time[2 * 90:] = 180 + time[2 * 90 - 1] + np.cumsum(dt[2 * 90:])
    y0 = mu + cp.carma_process(time, sigsqr, ar_roots, ma_coefs=ma_coefs)
    ysig = np.ones(ny) * y0.std() / 5.0
    y = y0 + ysig * np.random.standard_normal(ny)
    return time, y, ysig, ar_roots, ma_coefs, float(sigsqr)


def pack_cases(t, y, yerr, thetas, p, q, dense_idx=()):
    res = [ref_filter(t, y, yerr, th, p, q) for th in thetas]
    out = dict(
        theta=np.array(thetas), omega=np.array([r["omega"] for r in res]),
        ma=np.array([r["ma"] for r in res]), sigsqr=np.array([r["sigsqr"] for r in res]),
        mean=np.array([r["mean"] for r in res]), var=np.array([r["var"] for r in res]),
        loglik=np.array([r["loglik"] for r in res]), cond=np.array([r["cond"] for r in res]),
    )
    dense = np.full(len(thetas), np.nan)
    for i in dense_idx:
        dense[i] = dense_gp_loglik(t, y, yerr, thetas[i], p, q)
    out["dense_loglik"] = dense
    return out


def main():
    summary = {}

    # ---- config 2: README CARMA(5,3), n=270 ------------------------------------------------
    t, y, yerr, roots_true, ma_true, sigsqr_true = readme_series()
    p, q = 5, 3
    theta_true = np.array([2.3, 1.0, 17.0] + log_quads_from_roots(roots_true)
                          + [np.log(0.8), np.log(3.6), np.log(50.0)])
    rng = np.random.default_rng(2)
    post = [theta_true] + [theta_true + 0.01 * rng.standard_normal(theta_true.size) for _ in range(15)]
    prior = [prior_like_theta(rng, p, q, t, y) for _ in range(16)]
    thetas = post + prior
    g = pack_cases(t, y, yerr, thetas, p, q, dense_idx=(0, 1, 16, 17))
    # also the exact generating model (q=2 MA given directly): filter with true omega/beta
    kf = cp.KalmanFilterDeprecated(t, y - 17.0, yerr ** 2, sigsqr_true, roots_true, ma_coefs=ma_true)
    m0, v0 = kf.filter()
    np.savez_compressed(os.path.join(HERE, "carma53_readme.npz"), t=t, y=y, yerr=yerr, p=p, q=q,
                        kind=np.array(["post"] * 16 + ["prior"] * 16),
                        true_omega=roots_true, true_ma=ma_true, true_sigsqr=sigsqr_true,
                        true_mean=np.array(m0, dtype=float), true_var=np.array(v0, dtype=float), **g)
    summary["carma53_readme"] = dict(loglik0=g["loglik"][0], dense0=g["dense_loglik"][0],
                                     max_cond=float(g["cond"].max()))

    # ---- config 1: CAR(1), n=100 -----------------------------------------------------------
    rng = np.random.default_rng(1)
    t1 = np.cumsum(rng.uniform(1.0, 3.0, 100))
    np.random.seed(1)
    tau, sy = 100.0, 2.3
    y1 = cp.car1_process(t1, 2.0 * sy ** 2 / tau, tau) + 0.23 * rng.standard_normal(100)
    e1 = np.full(100, 0.23)
    theta1 = np.array([2.3, 1.0, 0.0, np.log(0.01)])
    th1 = [theta1] + [theta1 + np.array([0.1, 0.05, 0.2, 0.3]) * rng.standard_normal(4) for _ in range(7)]
    rows = []
    for th in th1:
        omega = np.exp(th[3])
        # KalmanFilterDeprecated cannot run p=1 (EigenMat[1,:] indexing, carma_pack.py:1296), so the
        # CAR(1) golden vectors come from the closed-form dense GP the reference's own C++ test uses
        # (carma_unit_tests.cpp:305-336): cov = sigma_y^2 exp(-|dt| omega) + diag(scale*yerr^2).
        # With C = L L^T and z = L^-1 r the one-step-ahead predictive moments are
        # var_k = L_kk^2 and mean_k = r_k - L_kk z_k.
        Cm = th[0] ** 2 * np.exp(-np.abs(t1[:, None] - t1[None, :]) * omega) + np.diag(th[1] * e1 ** 2)
        L = np.linalg.cholesky(Cm)
        r = y1 - th[2]
        z = np.linalg.solve(L, r)
        var = np.diag(L) ** 2
        mean = r - np.diag(L) * z
        dense = float(-np.sum(np.log(np.diag(L))) - 0.5 * z @ z)
        rows.append((mean, var, float(np.sum(-0.5 * np.log(var) - 0.5 * (r - mean) ** 2 / var)), dense))
    np.savez_compressed(os.path.join(HERE, "car1_n100.npz"), t=t1, y=y1, yerr=e1, theta=np.array(th1),
                        mean=np.array([r[0] for r in rows]), var=np.array([r[1] for r in rows]),
                        loglik=np.array([r[2] for r in rows]), dense_loglik=np.array([r[3] for r in rows]))
    summary["car1_n100"] = dict(loglik0=rows[0][2], dense0=rows[0][3])

    # ---- config 5: OGLE-LMC-LPV-00007, (p,q) grid ------------------------------------------
    ogle = np.genfromtxt(os.path.join(REF, "examples", "OGLE-LMC-LPV-00007.dat"))
    to, yo, eo = ogle[:, 0] - ogle[:, 0].min(), ogle[:, 1], ogle[:, 2]
    np.savetxt(os.path.join(HERE, "ogle_lmc_lpv_00007.dat"), np.c_[to, yo, eo], fmt="%.8f")
    rng = np.random.default_rng(5)
    grid = {}
    for pp in range(2, 8):
        for qq in range(pp):
            ths = [prior_like_theta(rng, pp, qq, to, yo) for _ in range(3)]
            gg = pack_cases(to, yo, eo, ths, pp, qq)
            for k, v in gg.items():
                grid["p%dq%d_%s" % (pp, qq, k)] = v
    np.savez_compressed(os.path.join(HERE, "ogle_grid.npz"), t=to, y=yo, yerr=eo, **grid)

    # ---- reference C++ test data: KalmanFilterp/Filter fixture (carma_unit_tests.cpp:387-503)
    dat = np.genfromtxt(os.path.join(REF, "cpp_tests", "data", "carma_test.dat"))[:300]
    tc, yc, ec = dat[:, 0], dat[:, 1], dat[:, 2]
    widths, cents = np.array([0.01, 0.01, 0.002]), np.array([0.2, 0.02])
    om = cp.get_ar_roots(widths, cents)
    from scipy.special import comb
    kappa = 0.5
    mac = np.array([comb(4, i) / kappa ** i for i in range(5)])
    sig2 = 2.3 ** 2 / cp.carma_variance(1.0, om, ma_coefs=mac)
    kf = cp.KalmanFilterDeprecated(tc, yc, ec ** 2, sig2, om, ma_coefs=mac)
    mc, vc = kf.filter()
    np.savez_compressed(os.path.join(HERE, "cpp_carma_test300.npz"), t=tc, y=yc, yerr=ec, omega=om, ma=mac,
                        sigsqr=float(sig2), mean=np.array(mc, dtype=float), var=np.array(vc, dtype=float))

    # ---- KAT: ZCAR/variance (carma_unit_tests.cpp:1269-1317) --------------------------------
    om_k = cp.get_ar_roots(np.array([0.01, 0.01, 0.002]), np.array([0.2, 0.02]))
    ma_k = np.array([comb(4, i) / 0.7 ** i for i in range(5)])
    kat = float(cp.carma_variance(2.3 ** 2, om_k, ma_coefs=ma_k))
    lagged = [float(cp.carma_variance(2.3 ** 2, om_k, ma_coefs=ma_k, lag=L)) for L in (0.5, 3.0, 40.0)]
    summary["variance_kat"] = dict(expected_cpp=223003.230567, python=kat, lags=[0.5, 3.0, 40.0], lagged=lagged,
                                   omega_re=om_k.real.tolist(), omega_im=om_k.imag.tolist(), ma=ma_k.tolist())

    with open(os.path.join(HERE, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    print(json.dumps(summary, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
