"""One order's batched maximum-likelihood search (100 starts) timed: iterations, evaluations, seconds."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import carma_pack_amd as cpa
from carma_pack_amd import carma_pack as cm
g = np.load(os.path.join(ROOT, "tests/golden/ogle_grid.npz"))
t, y, yerr = g["t"], g["y"], g["yerr"]
print("n =", len(t))
for (p, q) in ((1, 0), (2, 1), (4, 2), (5, 2), (7, 3), (7, 6)):
    model = cm.CarmaModel(t, y, yerr)
    model.get_mle(p, q, ntrials=100, seed=1)          # warm (context creation, first launches)
    t0 = time.perf_counter()
    mle = model.get_mle(p, q, ntrials=100, seed=1)
    dt = time.perf_counter() - t0
    print("(%d,%d): %.3f s, nit %s nfev %s fun %.4f" % (p, q, dt, getattr(mle, "nit", "?"), getattr(mle, "nfev", "?"), mle.fun), flush=True)
