"""Prototype (numpy float64, round 6) of the TWO-SIDED filter: the serial chain of kfilter.cpp:189-215 cut in two.

The state z of the real modal recursion (real_modal.py) is a stationary Gauss-Markov process with covariance V, so it
is Markov in reversed time as well, with transition V F^T V^-1.  In the dual coordinates u = V^-1 z that is

    u_k = F_k^T u_{k+1} + noise ,  Cov u = V^-1 ,  y_k = h.z_k = (V h).u_k = c.u_k

-- the SAME recursion with h and c = V h exchanged (c' = V^-1 c = h, s0' = c.V^-1 c = h.c = s0), the rotation sense
reversed (F^T: omega -> conj omega) and the data taken last to first.  V never appears.

Forward over data 0 .. m-1 gives  z_m | y_a ~ N(a, V + Da);  backward over data n-1 .. m gives  u_m | y_b ~ N(beta,
V^-1 + Db), both propagated to ONE meeting time.  With alpha, beta as random variables (functions of the data):
Cov alpha = -Da, Cov beta = -Db, E[alpha beta^T] = Da E[u z^T] Db = Da Db (the halves are independent given z_m), and
alpha / beta are sufficient for z_m, so with N = I - Da Db

    log p(y) = l_a + l_b - 1/2 log det N + beta.N^-1 a + 1/2 beta.N^-1 Da beta + 1/2 (Db a).N^-1 a

-- one p x p factorisation per evaluation, again without V.  (N has its eigenvalues in (0, 1]: -Da < V, -Db < V^-1.)

loglik_two_sided(...) below is the plain recursion on both sides (loglik_std of lazy_frame.py); the window / lane forms of
the device run the same two recursions chunk by chunk."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from real_modal import real_model, phi  # noqa: E402


def half_filter(om, pairs, p, h, c, s0, tt, yc, e, t_meet, transpose):
    """Recursion over the data (tt increasing or decreasing), then the prediction to t_meet.  Returns (l, D, z)."""
    D = np.zeros((p, p))
    z = np.zeros(p)
    ll = 0.0

    def F(dt):
        f = phi(om, pairs, abs(dt), p)
        return f.T if transpose else f

    for k in range(tt.size):
        if k > 0:
            f = F(tt[k] - tt[k - 1])
            D = f @ D @ f.T
            z = f @ z
        w = D @ h
        var = s0 + e[k] + h @ w
        kk = w + c
        innov = yc[k] - h @ z
        ll += -0.5 * np.log(var) - 0.5 * innov * innov / var
        D = D - np.outer(kk, kk) / var
        z = z + kk * (innov / var)
    if tt.size:
        f = F(t_meet - tt[-1])
        D = f @ D @ f.T
        z = f @ z
    return ll, D, z


def merge(Da, a, Db, beta):
    """log of  integral N(z; a, V + Da) N(V^-1 z; beta, V^-1 + Db) |V^-1| / N(z; 0, V) dz  (see the module text)."""
    p = a.size
    N = np.eye(p) - Da @ Db
    sign, logdet = np.linalg.slogdet(N)
    x1 = np.linalg.solve(N, a)
    x2 = np.linalg.solve(N, Da @ beta)
    return -0.5 * logdet + beta @ x1 + 0.5 * beta @ x2 + 0.5 * (Db @ a) @ x1


def loglik_two_sided(t, y, yerr, theta, p, q, m=None, where="left"):
    """m data on the forward side (default n // 2); meeting time t[m-1] ("left"), t[m] ("right") or halfway ("mid")."""
    om, h, Vz, pairs = real_model(theta, p, q)
    n = t.size
    if m is None:
        m = n // 2
    yc = y - theta[2]
    e = theta[1] * yerr ** 2
    c = Vz @ h
    s0 = h @ Vz @ h
    if m <= 0 or m >= n:
        return half_filter(om, pairs, p, h, c, s0, t, yc, e, t[-1], False)[0]
    tm = {"left": t[m - 1], "right": t[m], "mid": 0.5 * (t[m - 1] + t[m])}[where]
    la, Da, a = half_filter(om, pairs, p, h, c, s0, t[:m], yc[:m], e[:m], tm, False)
    lb, Db, beta = half_filter(om, pairs, p, c, h, s0, t[m:][::-1], yc[m:][::-1], e[m:][::-1], tm, True)
    return la + lb + merge(Da, a, Db, beta)


def loglik_one_pass(t, y, yerr, theta, p, q):
    om, h, Vz, pairs = real_model(theta, p, q)
    return half_filter(om, pairs, p, h, Vz @ h, h @ Vz @ h, t, y - theta[2], theta[1] * yerr ** 2, t[-1], False)[0]


if __name__ == "__main__":
    import oracle as orc
    from carma_pack_amd.synth import theta_batch

    g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    p, q = 5, 3
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    th = theta_batch(np.random.default_rng(2), N, p, q, t, y, theta_center=g["theta"][0])
    mdl = orc.OracleModel(t, y, yerr, p, q, max_stdev=10 * y.std())
    ref = mdl.logdensity_batch(th, ignore_prior=True)
    lp = np.array([mdl.log_prior(x) for x in th])
    for where in ("left", "mid", "right"):
        e1, e2 = [], []
        for i in range(N):
            if not np.isfinite(ref[i]):
                continue
            want = ref[i] - lp[i]
            with np.errstate(all="ignore"):
                a = loglik_one_pass(t, y, yerr, th[i], p, q)
                b = loglik_two_sided(t, y, yerr, th[i], p, q, where=where)
            e1.append(abs(a - want) / max(1.0, abs(want)))
            e2.append(abs(b - want) / max(1.0, abs(want)))
        e1, e2 = np.array(e1), np.array(e2)
        for nm, v in (("one pass", e1), ("two-sided " + where, e2)):
            print("%-18s n=%d median %.2e  99%% %.2e  max %.2e  >1e-10: %d" % (nm, v.size, np.median(v), np.quantile(v, 0.99), v.max(), np.sum(v > 1e-10)))
