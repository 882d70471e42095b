"""Prototype (CPU, numpy, float64): the CARMA Kalman log-likelihood as an ASSOCIATIVE SCAN over time
(Sarkka & Garcia-Fernandez, "Temporal parallelization of Bayesian smoothers", IEEE TAC 2021, filtering
elements (A, b, C, eta, J)), to see whether the time-parallel form keeps the 1e-10 parity of DESIGN.md §9
on the bench's 1024-theta batch, including its ill-conditioned members.  Not part of the product."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc
from carma_pack_amd.synth import theta_batch


def model(theta, p, q):
    om = np.asarray(orc.ar_roots(theta, p))
    ma = np.asarray(orc.ma_coefs(theta, p, q))
    var1 = orc.variance(om, ma, 1.0)
    sigsqr = theta[0] ** 2 / var1
    J = np.array([1.0 / np.prod([om[r] - om[l] for l in range(p) if l != r]) for r in range(p)])
    b = np.array([np.sum(ma * om[r] ** np.arange(p)) for r in range(p)])
    V = -sigsqr * np.outer(J, J.conj()) / (om[:, None] + om.conj()[None, :])
    return om, b, V


def combine(e1, e2):
    A1, b1, C1, h1, J1 = e1
    A2, b2, C2, h2, J2 = e2
    p = A1.shape[0]
    I = np.eye(p)
    M = np.linalg.solve((I + C1 @ J2).T, A2.T).T          # A2 (I + C1 J2)^-1
    A = M @ A1
    b = M @ (b1 + C1 @ h2) + b2
    C = M @ C1 @ A2.conj().T + C2
    N = np.linalg.solve((I + J2 @ C1).T, A1.conj()).T     # A1^H (I + J2 C1)^-1
    h = N @ (h2 - J2 @ b1) + h1
    Jm = N @ J2 @ A1 + J1
    return A, b, C, h, Jm


def loglik_scan(t, y, yerr, theta, p, q):
    om, bvec, V = model(theta, p, q)
    n = t.size
    yc = y - theta[2]
    e = theta[1] * yerr ** 2
    H = bvec[None, :]
    elems = []
    for k in range(n):
        if k == 0:
            P0 = V
            S = (H @ P0 @ H.conj().T).real.item() + e[0]
            K = (P0 @ H.conj().T) / S
            A = np.zeros((p, p), complex)
            bb = (K * yc[0]).ravel()
            C = P0 - K @ H @ P0
            hh = np.zeros(p, complex)
            Jm = np.zeros((p, p), complex)
        else:
            rho = np.exp(om * (t[k] - t[k - 1]))
            Ak = np.diag(rho)
            Q = V - np.outer(rho, rho.conj()) * V
            S = (H @ Q @ H.conj().T).real.item() + e[k]
            K = (Q @ H.conj().T) / S
            A = Ak - K @ H @ Ak
            bb = (K * yc[k]).ravel()
            C = Q - K @ H @ Q
            hh = (Ak.conj().T @ H.conj().T).ravel() * yc[k] / S
            Jm = Ak.conj().T @ H.conj().T @ H @ Ak / S
        elems.append((A, bb, C, hh, Jm))
    # Hillis-Steele inclusive prefix scan (what the lanes of a wave would do)
    pref = list(elems)
    d = 1
    while d < n:
        new = list(pref)
        for k in range(d, n):
            new[k] = combine(pref[k - d], pref[k])
        pref = new
        d *= 2
    # filtered mean/cov at k = (b, C) of prefix k; predictive for k+1
    ll = -0.5 * np.log((H @ V @ H.conj().T).real.item() + e[0]) - 0.5 * yc[0] ** 2 / ((H @ V @ H.conj().T).real.item() + e[0])
    for k in range(1, n):
        m, P = pref[k - 1][1], pref[k - 1][2]
        rho = np.exp(om * (t[k] - t[k - 1]))
        mp = rho * m
        Pp = np.outer(rho, rho.conj()) * (P - V) + V
        S = (H @ Pp @ H.conj().T).real.item() + e[k]
        inn = yc[k] - (H @ mp).real.item()
        ll += -0.5 * np.log(S) - 0.5 * inn * inn / S
    return ll


if __name__ == "__main__":
    g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    p, q = 5, 3
    th = theta_batch(np.random.default_rng(2), 1024, p, q, t, y, theta_center=g["theta"][0])
    m = orc.OracleModel(t, y, yerr, p, q)
    idx = list(range(24)) + [457]
    want = m.logdensity_batch(th[idx], ignore_prior=True)
    worst = 0.0
    for i, k in enumerate(idx):
        if not np.isfinite(want[i]):
            continue
        ll = loglik_scan(t, y, yerr, th[k], p, q) + m.log_prior(th[k])
        rel = abs(ll - want[i]) / abs(want[i])
        worst = max(worst, rel)
        print("theta %4d  oracle %.12f  scan %.12f  rel %.2e" % (k, want[i], ll, rel), flush=True)
    print("worst rel err %.2e" % worst)
