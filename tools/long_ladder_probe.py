import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import carma_pack_amd as cpa
rng = np.random.default_rng(12)
for n in (1024, 3000, 4500, 7000, 10000):
    t = np.cumsum(0.6 + 0.8 * rng.random(n)); y = np.cumsum(rng.normal(0, 0.3, n)); y = y - np.linspace(y[0], y[-1], n) + rng.normal(0, 0.1, n); e = np.full(n, 0.1)
    for p, q in ((5, 3), (3, 1)):
        for sw in (None, 0):
            cpa._lib.tune_reset()
            if sw is not None: cpa._lib.tune_set("PT_ROW_WIN", sw)
            ctx = cpa.Context(t, y, e, p, q)
            ctx.pt_create(10, 1, adapt_iters=10**9, seed=5); ctx.pt_start(None); ctx.pt_iterate(200)
            t0 = time.perf_counter(); ctx.pt_iterate(1000); dt = time.perf_counter() - t0
            print(n, p, q, ctx.pt_row_pipeline(), "%.1f us per iteration" % (1e3 * dt), flush=True)
