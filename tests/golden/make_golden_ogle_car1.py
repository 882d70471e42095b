"""The (p, q) = (1, 0) member of the OGLE-LMC-LPV-00007 grid (BASELINE configs[4]), which make_golden.py's grid omits
because the reference's Python Kalman filter cannot run p = 1 (carma_pack.py:1296): the CAR(1) golden vectors come from the
closed-form dense Gaussian process the reference's own C++ test uses (carma_unit_tests.cpp:305-336),
cov = sigma_y^2 exp(-|dt| omega) + diag(scale yerr^2); with C = L L^T and z = L^-1 r the one-step predictive moments are
var_k = L_kk^2, mean_k = r_k - L_kk z_k (as for car1_n100.npz).

    python tests/golden/make_golden_ogle_car1.py   ->  tests/golden/ogle_car1.npz
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    og = np.loadtxt(os.path.join(HERE, "ogle_lmc_lpv_00007.dat"))
    t, y, e = og[:, 0], og[:, 1], og[:, 2]
    rng = np.random.default_rng(51)
    dt = np.diff(t)
    thetas, rows = [], []
    for _ in range(4):
        sy = np.std(y) * rng.uniform(0.5, 1.5)
        th = np.array([sy, rng.uniform(0.8, 1.3), np.mean(y) + 0.1 * sy * rng.standard_normal(),
                       -np.log(np.median(dt) * rng.uniform(1.0, 50.0))])
        omega = np.exp(th[3])
        Cm = th[0] ** 2 * np.exp(-np.abs(t[:, None] - t[None, :]) * omega) + np.diag(th[1] * e ** 2)
        L = np.linalg.cholesky(Cm)
        r = y - th[2]
        z = np.linalg.solve(L, r)
        var = np.diag(L) ** 2
        mean = r - np.diag(L) * z
        thetas.append(th)
        rows.append((mean, var, float(np.sum(-0.5 * np.log(var) - 0.5 * (r - mean) ** 2 / var)),
                     float(-np.sum(np.log(np.diag(L))) - 0.5 * z @ z)))
    np.savez_compressed(os.path.join(HERE, "ogle_car1.npz"), theta=np.array(thetas), mean=np.array([r[0] for r in rows]),
                        var=np.array([r[1] for r in rows]), loglik=np.array([r[2] for r in rows]),
                        dense_loglik=np.array([r[3] for r in rows]))
    print("wrote ogle_car1.npz:", [r[2] for r in rows])


if __name__ == "__main__":
    main()
