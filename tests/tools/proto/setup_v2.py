"""Numerics study, part 2 (CPU only): a cheap set-up that is accurate where it matters.

Closed forms (alpha = AR polynomial with roots w_k, beta = MA polynomial with roots mu_m, beta(0) = 1):
    b_r      = beta(w_r)  = prod_m (mu_m - w_r) / mu_m
    c_r      = (V b^H)_r  = sigma^2 kappa_r,   kappa_r = beta(-w_r) / (alpha'(w_r) alpha(-w_r))       (partial fractions)
    s0       = Re(b V b^H) = theta_0^2                                                                 (stationary variance)
    Var(1)   = sum_k b_k kappa_k,     sigma^2 = theta_0^2 / Var(1)
Only (i) DIFFERENCES of roots and (ii) the cancelling sum Var(1) need more than double precision:
    level "dd":      exp / disc / sqrt / root differences / Var(1) in ~106-bit arithmetic, every product in double
    level "double":  everything in double (what round 1 did, up to operation order)
Usage: python tests/tools/proto/setup_v2.py P Q [ntheta] [level]
"""
import os
import sys

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle as orc  # noqa: E402
from helpers import irregular_series, prior_like_theta  # noqa: E402
from mp_truth import loglik_truth  # noqa: E402
from setup_precision import recursion  # noqa: E402


def quad_dd(lq1, lq2):
    """one quadratic factor in extended precision -> (q1, q2, sq, disc_sign)"""
    q1, q2 = mp.exp(mp.mpf(float(lq1))), mp.exp(mp.mpf(float(lq2)))
    disc = q2 * q2 - 4 * q1
    return q1, q2, mp.sqrt(abs(disc)), (1 if disc > 0 else -1)


def roots_ext(lq, m):
    """roots (as mp complex at the working precision) in the reference's order"""
    out = []
    for i in range(m // 2):
        q1, q2, sq, sgn = quad_dd(lq[2 * i], lq[2 * i + 1])
        if sgn > 0:
            out += [mp.mpc(-(q2 + sq) / 2), mp.mpc(-(q2 - sq) / 2)]
        else:
            out += [mp.mpc(-q2 / 2, -sq / 2), mp.mpc(-q2 / 2, sq / 2)]
    if m % 2:
        out.append(mp.mpc(-mp.exp(mp.mpf(float(lq[m - 1])))))
    return out


def cd(z):
    return complex(float(z.real), float(z.imag))


def setup_v2(theta, p, q, level="dd"):
    prec_ext = 104 if level == "dd" else 53
    with mp.workprec(prec_ext):
        om = roots_ext(theta[3:3 + p], p)
        mu = roots_ext(theta[3 + p:3 + p + q], q) if q else []
        omd = [cd(o) for o in om]
        mud = [cd(m_) for m_ in mu]
        # differences in extended precision, rounded to double
        dAA = [[cd(om[r] - om[l]) for l in range(p)] for r in range(p)]
        dMA = [[cd(mu[m_] - om[r]) for m_ in range(q)] for r in range(p)]
        # Var(1) = sum_k T_k in extended precision, T_k = N_k / D_k with squares u = w^2, v = mu^2
        u = [o * o for o in om]
        v = [m_ * m_ for m_ in mu]
        tot = mp.mpf(0)
        for k in range(p):
            N = mp.mpc(1)
            for m_ in range(q):
                N *= (v[m_] - u[k]) / v[m_]
            D = 2 * om[k] * (-1) ** p
            for l in range(p):
                if l != k:
                    D *= (u[k] - u[l])
            tot += (N / D).real
        var1 = float(tot)
    # everything else in double
    b, kap = [], []
    for r in range(p):
        br = 1.0 + 0j
        bm = 1.0 + 0j
        for m_ in range(q):
            br *= dMA[r][m_] / mud[m_]
            bm *= (mud[m_] + omd[r]) / mud[m_]
        ap = 1.0 + 0j
        for l in range(p):
            if l != r:
                ap *= dAA[r][l]
        am = 1.0 + 0j
        for l in range(p):
            am *= (-omd[r] - omd[l])
        b.append(br)
        kap.append(bm / (ap * am))
    if level == "double2":      # Var(1) from the same factored products, plain double
        var1 = sum((b[r] * kap[r]).real for r in range(p))
    sig2 = theta[0] ** 2 / var1
    c = [sig2 * k_ for k_ in kap]
    s0 = theta[0] ** 2
    h, cr, cpx = np.zeros(p), np.zeros(p), np.zeros(p, dtype=bool)
    for r in range(p):
        is_cpx = (omd[r].imag != 0) and r < (p & ~1)
        cpx[r] = is_cpx
        if is_cpx:
            e = r & ~1
            h[r] = 2 * b[r].imag if (r & 1) else 2 * b[r].real
            cr[r] = c[e].imag if (r & 1) else c[r].real
        else:
            h[r] = b[r].real
            cr[r] = c[r].real
    return np.array(omd), h, cr, s0, cpx


def main():
    p, q = int(sys.argv[1]), int(sys.argv[2])
    ntheta = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    level = sys.argv[4] if len(sys.argv) > 4 else "dd"
    t, y, yerr = irregular_series(150, seed=100 * p + q)
    rng = np.random.default_rng(7000 + 10 * p + q)
    th = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(ntheta)])
    m = orc.OracleModel(t, y, yerr, p, q, max_stdev=10.0 * np.sqrt(np.var(y, ddof=1)))
    want = m.logdensity_batch(th, nthreads=8)
    fin = np.isfinite(want)
    got = np.full(ntheta, np.nan)
    for i in np.flatnonzero(fin):
        om, h, c, s0, cpx = setup_v2(th[i], p, q, level)
        ll = recursion(om, h, c, s0, cpx, t, y, yerr, th[i][2], th[i][1])
        got[i] = ll - 0.5 * 50.0 / th[i][1] - 26.0 * np.log(th[i][1])
    rel = np.abs(got[fin] - want[fin]) / np.abs(want[fin])
    idx = np.flatnonzero(fin)[rel > 1e-10]
    print("p=%d q=%d finite %d: v2(%s) vs oracle median %.1e 99%% %.1e max %.1e, >1e-10: %d" % (
        p, q, fin.sum(), level, np.median(rel), np.quantile(rel, 0.99), rel.max(), idx.size), flush=True)
    nworse = 0
    for i in idx:
        T = float(loglik_truth(t, y, yerr, th[i], p, q)[0])
        eg, eo = abs(got[i] - T) / abs(T), abs(want[i] - T) / abs(T)
        worse = eg > max(1e-10, 3 * eo)
        nworse += worse
        print("   theta %4d: v2 err %.1e   oracle err %.1e %s" % (i, eg, eo, "  <-- WORSE" if worse else ""), flush=True)
    print("p=%d q=%d worse: %d" % (p, q, nworse))


if __name__ == "__main__":
    main()
