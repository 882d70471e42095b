"""Real modal coordinates of the CARMA state space (numpy prototype helpers shared by the prototypes in this
directory): z_{2k} = Re x_{2k}, z_{2k+1} = Im x_{2k} for a complex-conjugate root pair, z_r = x_r for a real root."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import oracle as orc  # noqa: E402


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
