"""Fuzz of the KalmanFilterp / KalmanFilter1 entry points (Filter's mean / var, Predict) against the oracle: random orders,
series lengths, prior-like models (real-root pairs included), roots handed over in a shuffled order, prediction times before,
inside (also exactly AT data times) and after the series.  python tools/fuzz_predict.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import carma_pack_amd as cpa
import oracle as orc
from helpers import irregular_series, prior_like_theta
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = worst_f = worst_p = 0
for case in range(ncase):
    p = int(rng.integers(1, 8))
    q = int(rng.integers(0, p)) if p > 1 else 0
    n = int(rng.choice([2, 3, 9, 16, 17, 40, 120, 300]))
    t, y, yerr = irregular_series(n, seed=int(rng.integers(1, 10 ** 6)))
    y = y - y.mean()
    tp = np.concatenate([t[0] - rng.uniform(0.1, 30.0, 3), rng.uniform(t[0], t[-1], 6), t[rng.integers(0, n, 2)],
                         t[-1] + rng.uniform(0.01, 200.0, 4)])
    try:
        if p == 1:
            om = float(np.exp(rng.uniform(-4, 1)))
            sig2 = float(rng.uniform(0.1, 3.0))
            mean, var = cpa._lib.kfilter_car1(t, y, yerr, sig2, om)
            wm, wv = orc.kfilter_car1(t, y, yerr, sig2, om)[:2]
            pm, pv = cpa._lib.predict_car1(t, y, yerr, sig2, om, tp)
            qm, qv = orc.predict_car1(t, y, yerr, sig2, om, tp)
        else:
            th = prior_like_theta(rng, p, q, t, y)
            if rng.random() < 0.3:                                   # a quadratic factor with two real roots
                r1 = 10.0 ** rng.uniform(-2.0, -0.5)
                r2 = r1 * rng.uniform(3.0, 20.0)
                th[3], th[4] = np.log(r1 * r2), np.log(r1 + r2)
            roots = np.asarray(orc.ar_roots(th, p))
            ma = np.asarray(orc.ma_coefs(th, p, q))
            sig2 = th[0] ** 2 / orc.variance(roots, ma)
            if not np.isfinite(sig2) or sig2 <= 0:
                continue
            perm = rng.permutation(p) if rng.random() < 0.5 else np.arange(p)      # any order the caller has them in
            mean, var = cpa._lib.kfilter_carma(t, y, yerr, sig2, roots[perm], ma)
            wm, wv = orc.kfilter_carma(t, y, yerr, sig2, roots, ma)[:2]
            pm, pv = cpa._lib.predict_carma(t, y, yerr, sig2, roots[perm], ma, tp)
            qm, qv = orc.predict_carma(t, y, yerr, sig2, roots, ma, tp)
        sc = np.abs(y).max() + 1e-300
        ef = max(np.max(np.abs(mean - wm)) / sc, np.max(np.abs(var - wv) / np.abs(wv)))
        ep = max(np.max(np.abs(pm - qm)) / sc, np.max(np.abs(pv - qv) / np.abs(qv)))
        worst_f, worst_p = max(worst_f, ef), max(worst_p, ep)
        if not (ef < 1e-7 and ep < 1e-7) and p > 1:
            # an ill-conditioned modal basis: the reference's LU solve (and with it the oracle) is itself that far off there.
            # The filter's mean / variance go to the quad-precision arbiter (never further from the exact values than the
            # oracle); the prediction has no arbiter of its own: cond(EigenMat) is printed
            cond = np.linalg.cond(np.vander(roots, p, increasing=True).T)
            th2 = th.copy(); th2[2] = 0.0; th2[1] = 1.0
            tm, tv = orc.truth_filter(t, y, yerr, th2, p, q)
            eg = max(np.max(np.abs(mean - tm)) / sc, np.max(np.abs(var - tv) / np.abs(tv)))
            eo = max(np.max(np.abs(wm - tm)) / sc, np.max(np.abs(wv - tv) / np.abs(tv)))
            print("case %d: p=%d q=%d n=%d deviates from the oracle by %.1e (filter) / %.1e (prediction); cond(EigenMat) %.1e; against the exact "
                  "filter: device %.1e, oracle %.1e" % (case, p, q, n, ef, ep, cond, eg, eo), flush=True)
            assert eg <= max(1e-9, eo), (eg, eo, cond)
            continue
        assert ef < 1e-7 and ep < 1e-7, (ef, ep)
    except AssertionError as ex:
        fails += 1
        print("FAILED case %d: p=%d q=%d n=%d: %s" % (case, p, q, n, str(ex)[:200]), flush=True)
print("%d cases, %d failed; worst filter deviation %.1e, worst prediction deviation %.1e (relative; ill-conditioned models included)" % (
    ncase, fails, worst_f, worst_p))
