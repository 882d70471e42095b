"""Prototype (numpy float64) of the co-rotating-frame covariance recursion.

Standard step (filter_loop_row):   D <- Phi_k (D - k k^T / var) Phi_k^T ,  k = D h + c ,  var = s0 + e + h.D.h
With D = A S A^T, A the transition accumulated since the last re-base (block rotations, commuting):
    w~ = S h~ ,  h~ = A^T h ;  var = s0 + e + h~.w~ ;  k~ = w~ + c~ ,  c~ = A^{-1} c ;  S <- S - k~ k~^T / var
and no rotation of the matrix at all inside a window; the mean follows the same way (z = A z~).
A window ends (re-base: S <- A S A^T, z <- A z~, A = I) before |Re omega| dt_acc or |Im omega| dt_acc grow past a
threshold.  Compares both forms with the CPU oracle over the bench parameter batch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle as orc
from carma_pack_amd.synth import theta_batch
from real_modal import real_model, phi


def loglik_std(t, y, yerr, theta, p, q):
    om, h, Vz, pairs = real_model(theta, p, q)
    yc = y - theta[2]; e = theta[1] * yerr ** 2
    c = Vz @ h; s0 = h @ Vz @ h
    D = np.zeros((p, p)); z = np.zeros(p); ll = 0.0
    for k in range(t.size):
        if k > 0:
            F = phi(om, pairs, t[k] - t[k - 1], p)
            D = F @ D @ F.T; z = F @ z
        w = D @ h; var = s0 + e[k] + h @ w; kk = w + c
        innov = yc[k] - h @ z
        ll += -0.5 * np.log(var) - 0.5 * innov * innov / var
        D = D - np.outer(kk, kk) / var; z = z + kk * (innov / var)
    return ll


def loglik_lazy(t, y, yerr, theta, p, q, lim_re=40.0, lim_im=64.0, maxwin=16, stats=None):
    om, h, Vz, pairs = real_model(theta, p, q)
    yc = y - theta[2]; e = theta[1] * yerr ** 2
    c = Vz @ h; s0 = h @ Vz @ h
    amax, bmax = np.max(np.abs(om.real)), np.max(np.abs(om.imag))
    S = np.zeros((p, p)); zt = np.zeros(p); ll = 0.0
    base = t[0]; win = 0; nreb = 0
    ht, ct = h.copy(), c.copy()
    for k in range(t.size):
        if k > 0:
            dta = t[k] - base
            if amax * dta > lim_re or bmax * dta > lim_im or win >= maxwin:
                A = phi(om, pairs, dta, p)                 # re-base including this step
                S = A @ S @ A.T; zt = A @ zt
                base = t[k]; win = 0; nreb += 1
                ht, ct = h, c
            else:
                A = phi(om, pairs, dta, p)
                Ainv = phi(-om, pairs, dta, p)              # block rotation by -angle, scale 1/E
                ht = A.T @ h; ct = Ainv @ c
                win += 1
        w = S @ ht; var = s0 + e[k] + ht @ w; kk = w + ct
        innov = yc[k] - ht @ zt
        ll += -0.5 * np.log(var) - 0.5 * innov * innov / var
        S = S - np.outer(kk, kk) / var; zt = zt + kk * (innov / var)
    if stats is not None:
        stats.append(nreb)
    return ll


if __name__ == "__main__":
    g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    p, q = 5, 3
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    th = theta_batch(np.random.default_rng(2), N, p, q, t, y, theta_center=g["theta"][0])
    m = orc.OracleModel(t, y, yerr, p, q, max_stdev=10 * y.std())
    ref = m.logdensity_batch(th, ignore_prior=True)
    lp = np.array([m.log_prior(x) for x in th])
    es, el, st = [], [], []
    for i in range(N):
        if not np.isfinite(ref[i]):
            continue
        want = ref[i] - lp[i]
        a = loglik_std(t, y, yerr, th[i], p, q); b = loglik_lazy(t, y, yerr, th[i], p, q, stats=st)
        es.append(abs(a - want) / max(1.0, abs(want))); el.append(abs(b - want) / max(1.0, abs(want)))
    es, el = np.array(es), np.array(el)
    print("finite %d ; rebases per eval: mean %.1f max %d" % (es.size, np.mean(st), np.max(st)))
    for nm, v in (("standard", es), ("lazy", el)):
        print("%-9s median %.2e  99%% %.2e  max %.2e  >1e-10: %d" % (nm, np.median(v), np.quantile(v, 0.99), v.max(), np.sum(v > 1e-10)))
    worst = np.argsort(el)[-5:]
    print("worst lazy:", el[worst], "standard there:", es[worst])
