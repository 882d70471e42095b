"""Which stored log-posteriors of the configs[2] run does the oracle not reproduce as finite / non-finite, and why?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import carma_pack_amd as cpa
import oracle as orc
from helpers import loglik_truth
np.set_printoptions(precision=17, linewidth=200)
g = np.load(os.path.join(ROOT, "tests/golden/carma53_readme.npz"))
t, y, yerr = g["t"], g["y"], g["yerr"]
ms = 10.0 * np.sqrt(np.mean(y * y) - np.mean(y) ** 2)
ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=ms)
samples, lp = ctx.pt_run(16, 64, 50000, 25000, 1, seed=2024)
m = orc.OracleModel(t, y, yerr, 5, 3, max_stdev=ms)
for stride in (2503, 997):
    sub = samples[:, ::stride].reshape(-1, 11); got = lp[:, ::stride].reshape(-1)
    want = m.logdensity_batch(sub, nthreads=os.cpu_count() or 8)
    bad = np.flatnonzero(np.isfinite(got) != np.isfinite(want))
    print("stride", stride, ":", sub.shape[0], "samples,", bad.size, "with a different finite pattern")
    for i in bad[:6]:
        th = sub[i]
        print(" entry", i, "gpu stored", got[i], "oracle", want[i], "gpu re-evaluated", ctx.logdensity(th[None, :])[0], "bounds ok (oracle):", m.check_prior_bounds(th))
        print("   theta", th)
        q1, q2 = np.exp(th[8]), np.exp(th[9])
        disc = q2 * q2 - 4 * q1
        sq = np.sqrt(abs(disc))
        print("   MA factor: q1 %.17g q2 %.17g disc>0 %s  q2 - sqrt(disc) = %.3g  (4 q1 / q2^2 = %.3g = 2^%.2f)" % (q1, q2, disc > 0, q2 - sq, 4 * q1 / q2 / q2, np.log2(4 * q1 / q2 / q2)))
        try:
            print("   quad-precision value of the reference's formulas:", loglik_truth(t, y, yerr, th, 5, 3))
        except Exception as ex:
            print("   quad-precision evaluation failed:", ex)
