#!/usr/bin/env python3
"""Where choose_order(7, ntrials=100) on the OGLE series spends its time: per order the wall time inside carma_mle_batched (the
library's lock-step L-BFGS), iterations (the slowest start's and the mean), evaluations."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import carma_pack_amd as cpa
from carma_pack_amd import carma_pack as cp, _lib

d = np.loadtxt(os.path.join(ROOT, "tests", "golden", "ogle_lmc_lpv_00007.dat"))
m = cp.CarmaModel(d[:, 0], d[:, 1], d[:, 2])
m.choose_order(2, ntrials=20)                     # warm-up (module import, first launches)
rows = []
orig = _lib.Context.mle_batched


def timed(self, x0, bounds, **kw):
    t0 = time.perf_counter()
    r = orig(self, x0, bounds, **kw)
    dt = time.perf_counter() - t0
    x, fun, nit, nfev, status = r
    rows.append(dict(p=self.p, q=self.q, d=self.d, starts=len(fun), s=round(dt, 4), nit_max=int(nit.max()), nit_mean=round(float(nit.mean()), 1),
                     nfev_sum=int(nfev.sum()), us_per_iteration=round(1e6 * dt / max(1, int(nit.max())), 1), status=np.bincount(status, minlength=3).tolist()))
    return r


_lib.Context.mle_batched = timed
t0 = time.perf_counter()
best, pq, aicc = m.choose_order(7, ntrials=100)
print(json.dumps(dict(wall_s=round(time.perf_counter() - t0, 3), chosen=[m.p, m.q], inside_mle_batched_s=round(sum(r["s"] for r in rows), 3))))
for r in rows:
    print(json.dumps(r))
