"""Wide parity sweep: many random prior-like parameter vectors for every order against the CPU reference
restatement (test infrastructure, tests/ oracle), with the 50-digit arbiter for the entries above 1e-10.
Run on the GPU box:  python tools/parity_sweep.py [ntheta]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import carma_pack_amd as cpa
if os.environ.get("SWEEP_LIB"):
    import carma_pack_amd._lib as _L
    _L.LIB_PATH = os.environ["SWEEP_LIB"]
    _L.lib = _L._load()
import oracle as orc
from helpers import irregular_series, prior_like_theta
from helpers import loglik_truth

ntheta = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
tot = bad = arbitrated = worse3 = worse1 = better3 = within = 0
orders = [int(v) for v in os.environ.get("SWEEP_P", "2,3,4,5,6,7").split(",")]
for p in orders:
    for q in range(0, p):
        t, y, yerr = irregular_series(150, seed=100 * p + q)
        rng = np.random.default_rng(7000 + 10 * p + q)
        th = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(ntheta)])
        ctx = cpa.Context(t, y, yerr, p, q)
        m = orc.OracleModel(t, y, yerr, p, q, max_stdev=ctx.prior()[0])
        want = m.logdensity_batch(th, nthreads=os.cpu_count() or 8)
        got = ctx.logdensity(th)
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), fin), (p, q)
        rel = np.abs(got[fin] - want[fin]) / np.abs(want[fin])
        idx = np.flatnonzero(fin)[rel > 1e-10]
        n3 = n1 = nb = nin = 0
        worst_g = worst_o = 0.0
        for i in idx:                       # every entry above 1e-10 is arbitrated against 50-digit arithmetic
            T = float(loglik_truth(t, y, yerr, th[i], p, q)[0])
            eg, eo = abs(got[i] - T) / abs(T), abs(want[i] - T) / abs(T)
            worst_g, worst_o = max(worst_g, eg), max(worst_o, eo)
            n3 += eg > max(1e-10, 3 * eo)
            n1 += eg > max(1e-10, eo)
            nb += eg < eo / 3
            nin += eg <= 1e-10
            if os.environ.get("SWEEP_VERBOSE"):
                print("   theta %d: gpu err %.1e  reference err %.1e" % (i, eg, eo))
        tot += fin.sum(); bad += idx.size; arbitrated += idx.size
        worse3 += n3; worse1 += n1; better3 += nb; within += nin
        print("p=%d q=%d: finite %4d  median %.1e  99%% %.1e  max %.1e  >1e-10 vs oracle: %2d | vs 50-digit value: GPU worst %.1e "
              "oracle worst %.1e, GPU within 1e-10: %d, worse than the oracle: %d (>3x: %d), >3x better: %d" % (
                  p, q, fin.sum(), np.median(rel), np.quantile(rel, 0.99), rel.max(), idx.size, worst_g, worst_o, nin, n1, n3, nb),
              flush=True)
print("total finite %d, differing from the oracle by more than 1e-10: %d (%.3f%%); of these, against the 50-digit value: GPU within "
      "1e-10 on %d, GPU further away than the oracle on %d (more than 3x further: %d), GPU more than 3x closer: %d" % (
          tot, bad, 100.0 * bad / tot, within, worse1, worse3, better3))
