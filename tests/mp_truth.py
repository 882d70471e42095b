"""High-precision (mpmath, 50 digits) evaluation of the reference's log-likelihood recursion, used
only to arbitrate ill-conditioned cases: where cond(EigenMat) >~ 1e6 two double-precision
implementations of kfilter.cpp (the reference's LAPACK path, the oracle's LU, the GPU's lane-
distributed LU) legitimately differ by more than 1e-10, and the question becomes which one is
closer to the exact value.  Literal restatement of kfilter.cpp:138-215 + carpack.hpp:167-171."""
import mpmath as mp
import numpy as np

mp.mp.dps = 50


def loglik_truth(t, y, yerr, theta, p, q):
    th = [mp.mpf(float(v)) for v in theta]

    def quad_roots(lq, m):
        roots = []
        for i in range(m // 2):
            q1, q2 = mp.exp(lq[2 * i]), mp.exp(lq[2 * i + 1])
            disc = q2 * q2 - 4 * q1
            if disc > 0:
                roots += [-(q2 + mp.sqrt(disc)) / 2, -(q2 - mp.sqrt(disc)) / 2]
            else:
                roots += [mp.mpc(-q2 / 2, -mp.sqrt(-disc) / 2), mp.mpc(-q2 / 2, mp.sqrt(-disc) / 2)]
        if m % 2:
            roots.append(-mp.exp(lq[m - 1]))
        return [mp.mpc(r) for r in roots]

    om = quad_roots(th[3:3 + p], p)
    ma = [mp.mpf(0)] * p
    if q == 0:
        ma[0] = mp.mpf(1)
    else:
        mr = quad_roots(th[3 + p:3 + p + q], q)
        cf = [mp.mpc(1)] + [mp.mpc(0)] * q
        for i, r in enumerate(mr):
            for k in range(i + 1, 0, -1):
                cf[k] = cf[k] - r * cf[k - 1]
        pc = [c.real for c in cf]
        for i in range(q + 1):
            ma[i] = pc[q - i] / pc[q]
    # Variance(omega, ma, 1)
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
    scale, mu = th[1], th[2]
    E = mp.matrix(p, p)
    for i in range(p):
        for j in range(p):
            E[i, j] = om[j] ** i
    rhs = mp.matrix(p, 1)
    rhs[p - 1] = 1
    J = mp.lu_solve(E, rhs)
    b = [sum(ma[i] * E[i, j] for i in range(p)) for j in range(p)]
    V = [[-sigsqr * J[i] * mp.conj(J[j]) / (om[i] + mp.conj(om[j])) for j in range(p)] for i in range(p)]
    P = [row[:] for row in V]
    x = [mp.mpc(0)] * p
    tt = [mp.mpf(float(v)) for v in t]
    yy = [mp.mpf(float(v)) - mu for v in y]
    ee = [scale * mp.mpf(float(v)) ** 2 for v in yerr]
    var = sum(b[i] * sum(P[i][j] * mp.conj(b[j]) for j in range(p)) for i in range(p)).real + ee[0]
    mean = mp.mpf(0)
    innov = yy[0]
    ll = -mp.log(var) / 2 - innov ** 2 / var / 2
    for k in range(1, len(tt)):
        g = [sum(P[i][j] * mp.conj(b[j]) for j in range(p)) / var for i in range(p)]
        x = [x[i] + g[i] * innov for i in range(p)]
        P = [[P[i][j] - var * g[i] * mp.conj(g[j]) for j in range(p)] for i in range(p)]
        dt = tt[k] - tt[k - 1]
        rho = [mp.exp(om[i] * dt) for i in range(p)]
        x = [rho[i] * x[i] for i in range(p)]
        P = [[rho[i] * mp.conj(rho[j]) * (P[i][j] - V[i][j]) + V[i][j] for j in range(p)] for i in range(p)]
        mean = sum(b[i] * x[i] for i in range(p)).real
        var = sum(b[i] * sum(P[i][j] * mp.conj(b[j]) for j in range(p)) for i in range(p)).real + ee[k]
        innov = yy[k] - mean
        ll += -mp.log(var) / 2 - innov ** 2 / var / 2
    logprior = -mp.mpf(50) / 2 / scale - 26 * mp.log(scale)
    return float(ll + logprior), float(ll)
