import os, sys, time, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import carma_pack_amd as cpa
g = np.load(os.path.join(ROOT, 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=10 * y.std())
for R in (1024, 4096):
    ctx.pt_create(16, R, adapt_iters=10 ** 9, seed=3); ctx.pt_start(None); ctx.pt_iterate(20)
    t0 = time.perf_counter(); ctx.pt_iterate(200); a = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200): ctx.pt_iterate(1)
    b = (time.perf_counter() - t0) / 200
    print("%s: 16 x %d chains (%s): %.1f us per iteration in one call of 200, %.1f us per call of one iteration" % (
        os.path.basename(os.environ.get("CARMA_LIB_PATH", "in-tree")), R, ctx.pt_kernel(), a * 1e6, b * 1e6), flush=True)
