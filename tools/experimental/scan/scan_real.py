"""Prototype of the planned time-parallel kernel, step by step as the GPU would do it (CPU, numpy float64):
real modal coordinates, one lane per block of consecutive steps, phase 1 = block elements by forward
recursions, phase 2 = Hillis-Steele scan over the lanes with the general combine (no-pivot Gaussian
elimination for (I + C J)), phase 3 = ordinary filter inside each block from the prefix state."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc
from carma_pack_amd.synth import theta_batch


def real_model(theta, p, q):
    om = np.asarray(orc.ar_roots(theta, p))
    ma = np.asarray(orc.ma_coefs(theta, p, q))
    sigsqr = theta[0] ** 2 / orc.variance(om, ma, 1.0)
    J = np.array([1.0 / np.prod([om[r] - om[l] for l in range(p) if l != r]) for r in range(p)])
    b = np.array([np.sum(ma * om[r] ** np.arange(p)) for r in range(p)])
    V = -sigsqr * np.outer(J, J.conj()) / (om[:, None] + om.conj()[None, :])
    # real coordinates: z_{2k} = Re x_{2k}, z_{2k+1} = Im x_{2k} for complex pairs; z_r = x_r for real roots
    T = np.zeros((p, p), complex)
    h = np.zeros(p)
    pairs = []
    r = 0
    while r < p:
        if r + 1 < p and abs(om[r].imag) > 0 and np.isclose(om[r + 1], om[r].conjugate()):
            T[r, r] = 0.5; T[r, r + 1] = 0.5              # Re x_r = (x_r + x_{r+1})/2
            T[r + 1, r] = -0.5j; T[r + 1, r + 1] = 0.5j    # Im x_r = (x_r - x_{r+1})/(2i)
            h[r] = 2 * b[r].real; h[r + 1] = -2 * b[r].imag
            pairs.append((r, True)); r += 2
        else:
            T[r, r] = 1.0; h[r] = b[r].real
            pairs.append((r, False)); r += 1
    Vz = (T @ V @ T.conj().T).real
    return om, h, Vz, pairs


def phi(om, pairs, dt, p):
    F = np.zeros((p, p))
    for r, cpx in pairs:
        rho = np.exp(om[r] * dt)
        if cpx:
            F[r, r] = rho.real; F[r, r + 1] = -rho.imag; F[r + 1, r] = rho.imag; F[r + 1, r + 1] = rho.real
        else:
            F[r, r] = rho.real
    return F


def solve_nopiv(Amat, B):
    """X = A^{-1} B by Gaussian elimination without pivoting (what a lane would do in registers)."""
    A = Amat.copy(); X = B.copy(); n = A.shape[0]
    for k in range(n):
        piv = 1.0 / A[k, k]
        for i in range(k + 1, n):
            l = A[i, k] * piv
            A[i, k:] -= l * A[k, k:]
            X[i] -= l * X[k]
    for k in range(n - 1, -1, -1):
        X[k] = (X[k] - A[k, k + 1:] @ X[k + 1:]) / A[k, k]
    return X


def combine(e1, e2):
    A1, b1, C1, h1, J1 = e1
    A2, b2, C2, h2, J2 = e2
    p = A1.shape[0]
    W = np.eye(p) + C1 @ J2                       # (I + C1 J2)
    # M = A2 W^{-1}  ->  M^T = W^{-T} A2^T
    M = solve_nopiv(W.T, A2.T).T
    A = M @ A1
    b = M @ (b1 + C1 @ h2) + b2
    C = M @ C1 @ A2.T + C2
    C = 0.5 * (C + C.T)
    # N = A1^T (I + J2 C1)^{-1} ;  (I + J2 C1) = W^T
    N = solve_nopiv(W, A1).T
    hh = N @ (h2 - J2 @ b1) + h1
    Jm = N @ J2 @ A1 + J1
    Jm = 0.5 * (Jm + Jm.T)
    return A, b, C, hh, Jm


def loglik_scan(t, y, yerr, theta, p, q, nlanes=64):
    om, h, Vz, pairs = real_model(theta, p, q)
    n = t.size
    yc = y - theta[2]
    e = theta[1] * yerr ** 2
    s = -(-n // nlanes)
    blocks = [(l * s, min(n, (l + 1) * s)) for l in range(nlanes) if l * s < n]
    # phase 1: block elements (A, b, C, eta, J) by forward recursions from "x known exactly"
    elems = []
    for (k0, k1) in blocks:
        A = np.eye(p); b = np.zeros(p); C = np.zeros((p, p)); eta = np.zeros(p); Jm = np.zeros((p, p))
        for k in range(k0, k1):
            if k == 0:
                A = np.zeros((p, p)); C = Vz.copy()          # prior: x_0 ~ N(0, V)
            else:
                F = phi(om, pairs, t[k] - t[k - 1], p)
                A = F @ A; b = F @ b; C = F @ (C - Vz) @ F.T + Vz
            hA = h @ A
            S = h @ C @ h + e[k]
            r = yc[k] - h @ b
            Jm = Jm + np.outer(hA, hA) / S
            eta = eta + hA * (r / S)
            K = C @ h / S
            A = A - np.outer(K, hA)
            b = b + K * r
            C = C - np.outer(K, K) * S
        elems.append((A, b, C, eta, Jm))
    # phase 2: inclusive Hillis-Steele scan over the lanes
    L = len(elems)
    pref = list(elems)
    d = 1
    while d < L:
        new = list(pref)
        for k in range(d, L):
            new[k] = combine(pref[k - d], pref[k])
        pref = new
        d *= 2
    # phase 3: ordinary filter inside each block from the state at its start
    ll = 0.0
    for l, (k0, k1) in enumerate(blocks):
        if l == 0:
            m = np.zeros(p); P = None
        else:
            m, P = pref[l - 1][1].copy(), pref[l - 1][2].copy()
        for k in range(k0, k1):
            if k == 0:
                P = Vz.copy()
            else:
                F = phi(om, pairs, t[k] - t[k - 1], p)
                m = F @ m; P = F @ (P - Vz) @ F.T + Vz
            S = h @ P @ h + e[k]
            r = yc[k] - h @ m
            ll += -0.5 * np.log(S) - 0.5 * r * r / S
            K = P @ h / S
            m = m + K * r
            P = P - np.outer(K, K) * S
    return ll


if __name__ == "__main__":
    g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    p, q = 5, 3
    th = theta_batch(np.random.default_rng(2), 1024, p, q, t, y, theta_center=g["theta"][0])
    m = orc.OracleModel(t, y, yerr, p, q)
    idx = list(range(0, 1024, 16)) + [457]
    want = m.logdensity_batch(th[idx], ignore_prior=True)
    worst, nbad = 0.0, 0
    for i, k in enumerate(idx):
        if not np.isfinite(want[i]):
            continue
        ll = loglik_scan(t, y, yerr, th[k], p, q) + m.log_prior(th[k])
        rel = abs(ll - want[i]) / abs(want[i])
        worst = max(worst, rel)
        if rel > 1e-10:
            nbad += 1
            print("theta %4d  oracle %.12f  scan %.12f  rel %.2e" % (k, want[i], ll, rel), flush=True)
    print("checked %d thetas, worst rel err %.2e, above 1e-10: %d" % (len(idx), worst, nbad))
