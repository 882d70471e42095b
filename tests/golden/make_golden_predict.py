#!/usr/bin/env python3
"""Golden vectors for KalmanFilterp::Predict / KalmanFilter1::Predict, from
the REFERENCE's Python (KalmanFilterDeprecated.predict, carma_pack.py:1377-1488) and from the
dense Gaussian-process conditional the reference's C++ tests use (carma_unit_tests.cpp:277-385,
505-649).  Build container only:  python tests/golden/make_golden_predict.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference with _carmcmc / acor stubbed)

cp = mg.cp


def dense_conditional(t, y, yvar, acf, tp):
    """E, Var of the process at tp given noisy data: k^T C^-1 y, acf(0) - k^T C^-1 k."""
    C = acf(np.abs(t[:, None] - t[None, :])) + np.diag(yvar)
    k = acf(np.abs(t - tp))
    sol = np.linalg.solve(C, np.c_[y, k])
    return float(k @ sol[:, 0]), float(acf(np.zeros(1))[0] - k @ sol[:, 1])


def main():
    g = np.load(os.path.join(HERE, "carma53_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    rng = np.random.default_rng(12)
    # interpolation inside the seasons and the gaps, exactly at a datum, forecasts; (backcasts are
    # not supported by the Python mirror: dense GP only)
    times = np.concatenate([rng.uniform(t[0] + 0.01, t[-1], 14), [t[100], t[-1] + 0.5, t[-1] + 30.0, t[-1] + 400.0]])
    back = np.array([t[0] - 0.7, t[0] - 25.0])
    out = {}
    for tag, i in (("true", None), ("th3", 3), ("th17", 17)):
        if i is None:
            om, ma, sig, mu, scale = g["true_omega"], g["true_ma"], float(g["true_sigsqr"]), 17.0, 1.0
        else:
            om, ma, sig, mu, scale = g["omega"][i], g["ma"][i], float(g["sigsqr"][i]), g["theta"][i][2], g["theta"][i][1]
        kf = cp.KalmanFilterDeprecated(t, y - mu, scale * yerr ** 2, sig, om, ma_coefs=ma)
        pm, pv = zip(*[kf.predict(tp) for tp in times])
        acf = np.vectorize(lambda lag: cp.carma_variance(sig, om, ma_coefs=ma, lag=float(lag)))
        dm, dv = zip(*[dense_conditional(t, y - mu, scale * yerr ** 2, acf, tp) for tp in np.r_[times, back]])
        out.update({tag + "_omega": om, tag + "_ma": ma, tag + "_sigsqr": sig, tag + "_mu": mu, tag + "_scale": scale,
                    tag + "_pmean": np.array(pm), tag + "_pvar": np.array(pv), tag + "_dmean": np.array(dm),
                    tag + "_dvar": np.array(dv)})
        print(tag, "max |kalman-dense| mean %.2e var %.2e" % (np.abs(np.array(pm) - np.array(dm)[:len(times)]).max(),
                                                              np.abs(np.array(pv) - np.array(dv)[:len(times)]).max()))
    # CAR(1): closed-form dense GP
    c = np.load(os.path.join(HERE, "car1_n100.npz"))
    t1, y1, e1 = c["t"], c["y"], c["yerr"]
    th = c["theta"][1]
    omega, sy = np.exp(th[3]), th[0]
    times1 = np.concatenate([rng.uniform(t1[0], t1[-1], 8), [t1[10], t1[-1] + 3.0, t1[0] - 2.0]])
    acf1 = lambda lag: sy ** 2 * np.exp(-np.abs(lag) * omega)  # noqa: E731
    dm, dv = zip(*[dense_conditional(t1, y1 - th[2], th[1] * e1 ** 2, acf1, tp) for tp in times1])
    np.savez_compressed(os.path.join(HERE, "predict.npz"), times=times, back=back, car1_times=times1,
                        car1_theta=th, car1_dmean=np.array(dm), car1_dvar=np.array(dv), **out)


if __name__ == "__main__":
    main()
