"""Numerics study (CPU only): how accurate must the once-per-evaluation set-up (theta -> roots, MA
coefficients, sigma^2, rotated observation row h, gain offset c, s0) be for the real-modal recursion to
stay within 1e-10 of the 50-digit value of the reference's algorithm on ill-conditioned, prior-like theta?

  exact-setup : the set-up in 50-digit arithmetic (mpmath), rounded to double; recursion in double
  double-setup: the set-up in plain double (closed-form Vandermonde column, as carma_core.h round 1)

Usage: python tests/tools/proto/setup_precision.py P Q [ntheta]
"""
import os
import sys

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc  # noqa: E402
from helpers import irregular_series, prior_like_theta  # noqa: E402
from mp_truth import loglik_truth  # noqa: E402

mp.mp.dps = 50


def mp_roots(lq, m):
    roots = []
    for i in range(m // 2):
        q1, q2 = mp.exp(lq[2 * i]), mp.exp(lq[2 * i + 1])
        disc = q2 * q2 - 4 * q1
        if disc > 0:
            roots += [mp.mpc(-(q2 + mp.sqrt(disc)) / 2), mp.mpc(-(q2 - mp.sqrt(disc)) / 2)]
        else:
            roots += [mp.mpc(-q2 / 2, -mp.sqrt(-disc) / 2), mp.mpc(-q2 / 2, mp.sqrt(-disc) / 2)]
    if m % 2:
        roots.append(mp.mpc(-mp.exp(lq[m - 1])))
    return roots


def setup_exact(theta, p, q):
    """-> (omega[p] complex, h[p], c[p], s0, cpx[p]) rounded to double from 50-digit arithmetic."""
    th = [mp.mpf(float(v)) for v in theta]
    om = mp_roots(th[3:3 + p], p)
    ma = [mp.mpf(0)] * p
    if q == 0:
        ma[0] = mp.mpf(1)
    else:
        mr = mp_roots(th[3 + p:3 + p + q], q)
        cf = [mp.mpc(1)] + [mp.mpc(0)] * q
        for i, r in enumerate(mr):
            for k in range(i + 1, 0, -1):
                cf[k] = cf[k] - r * cf[k - 1]
        pc = [c.real for c in cf]
        for i in range(q + 1):
            ma[i] = pc[q - i] / pc[q]
    var1 = mp.mpc(0)
    for k in range(p):
        dp = mp.mpc(1)
        for l in range(p):
            if l != k:
                dp *= (om[l] - om[k]) * (mp.conj(om[l]) + om[k])
        den = -2 * om[k].real * dp
        s1 = sum(ma[l] * om[k] ** l for l in range(p))
        s2 = sum(ma[l] * (-om[k]) ** l for l in range(p))
        var1 += s1 * s2 / den
    sigsqr = th[0] ** 2 / var1.real
    J = []
    for k in range(p):
        dp = mp.mpc(1)
        for l in range(p):
            if l != k:
                dp *= (om[k] - om[l])
        J.append(1 / dp)
    b = [sum(ma[i] * om[j] ** i for i in range(p)) for j in range(p)]
    c = []
    for r in range(p):
        acc = mp.mpc(0)
        for j in range(p):
            acc += -sigsqr * J[r] * mp.conj(J[j]) / (om[r] + mp.conj(om[j])) * mp.conj(b[j])
        c.append(acc)
    s0 = sum((b[r] * c[r]).real for r in range(p))
    h, cr, cpx = np.zeros(p), np.zeros(p), np.zeros(p, dtype=bool)
    for r in range(p):
        is_cpx = (om[r].imag != 0) and r < (p & ~1)
        cpx[r] = is_cpx
        if is_cpx:
            e = r & ~1
            h[r] = float(2 * b[r].imag) if (r & 1) else float(2 * b[r].real)
            cr[r] = float(c[e].imag) if (r & 1) else float(c[r].real)
        else:
            h[r] = float(b[r].real)
            cr[r] = float(c[r].real)
    omd = np.array([complex(float(o.real), float(o.imag)) for o in om])
    return omd, h, cr, float(s0), cpx


def recursion(om, h, c, s0, cpx, t, y, yerr, mu, scale):
    """real-modal recursion of filter_loop_real in double (numpy, one evaluation)."""
    p = om.size
    D = np.zeros((p, p))
    z = np.zeros(p)
    k = c.copy()
    w = np.zeros(p)
    ll_logv, chi2 = 0.0, 0.0
    n = t.size
    for i in range(n):
        var = s0 + h @ w + yerr[i] ** 2 * scale
        innov = (y[i] - mu) - h @ z
        ll_logv += np.log(var)
        s = 1.0 / var
        chi2 += innov * innov * s
        if i == n - 1:
            break
        dt = t[i + 1] - t[i]
        rho = np.exp(om * dt)
        Phi = np.zeros((p, p))
        for r in range(p):
            if cpx[r]:
                e = r & ~1
                cc, ss = rho[e].real, rho[e].imag
                if r & 1:
                    Phi[r, r] = cc
                    Phi[r, r - 1] = ss
                else:
                    Phi[r, r] = cc
                    Phi[r, r + 1] = -ss
            else:
                Phi[r, r] = rho[r].real
        z = Phi @ (z + k * (s * innov))
        D = Phi @ (D - np.outer(k, k) * s) @ Phi.T
        w = D @ h
        k = w + c
    return -0.5 * ll_logv - 0.5 * chi2


def main():
    p, q = int(sys.argv[1]), int(sys.argv[2])
    ntheta = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    t, y, yerr = irregular_series(150, seed=100 * p + q)
    rng = np.random.default_rng(7000 + 10 * p + q)
    th = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(ntheta)])
    m = orc.OracleModel(t, y, yerr, p, q, max_stdev=10.0 * np.sqrt(np.var(y, ddof=1)))
    want = m.logdensity_batch(th, nthreads=8)
    fin = np.isfinite(want)
    got = np.full(ntheta, np.nan)
    for i in np.flatnonzero(fin):
        om, h, c, s0, cpx = setup_exact(th[i], p, q)
        ll = recursion(om, h, c, s0, cpx, t, y, yerr, th[i][2], th[i][1])
        got[i] = ll - 0.5 * 50.0 / th[i][1] - 26.0 * np.log(th[i][1])
    rel = np.abs(got[fin] - want[fin]) / np.abs(want[fin])
    idx = np.flatnonzero(fin)[rel > 1e-10]
    print("p=%d q=%d finite %d: exact-setup vs oracle median %.1e 99%% %.1e max %.1e, >1e-10: %d" % (
        p, q, fin.sum(), np.median(rel), np.quantile(rel, 0.99), rel.max(), idx.size), flush=True)
    for i in idx:
        T = float(loglik_truth(t, y, yerr, th[i], p, q)[0])
        eg, eo = abs(got[i] - T) / abs(T), abs(want[i] - T) / abs(T)
        print("   theta %4d: exact-setup err %.1e   oracle err %.1e %s" % (i, eg, eo, "  <-- WORSE" if eg > max(1e-10, 3 * eo) else ""),
              flush=True)


if __name__ == "__main__":
    main()
