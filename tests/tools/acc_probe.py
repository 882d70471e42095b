"""Accuracy probe on an ill-conditioned case (p=7, q=4, n=49): GPU error against 50-digit arithmetic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import carma_pack_amd._lib as L0
if len(sys.argv) > 1:
    L0.LIB_PATH = sys.argv[1]
    L0.lib = L0._load()
import oracle as orc
from helpers import irregular_series, prior_like_theta
from helpers import loglik_truth
p, q = 7, 4
rng = np.random.default_rng(900 + p)
for n in (2, 3, 7, 8, 9, 15, 16, 17, 18, 31, 32, 33, 34, 47, 48, 49):
    t, y, yerr = irregular_series(n, seed=n)
    th = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(12)])
ctx = L0.Context(t, y, yerr, p, q)
m = orc.OracleModel(t, y, yerr, p, q, max_stdev=ctx.prior()[0])
want = m.logdensity_batch(th, ignore_prior=True)
for B in (12, 1100, 3000, 20000):
    got = ctx.logdensity(np.tile(th, (B // 12 + 1, 1))[:B], ignore_prior=True)[:12]
    errs = []
    for i in (8, 9, 10):
        T = float(loglik_truth(t, y, yerr, th[i], p, q)[0])
        errs.append("%d: gpu %.2e orc %.2e" % (i, abs(got[i] - T) / abs(T), abs(want[i] - T) / abs(T)))
    print("B=%d  " % B + "  ".join(errs))
