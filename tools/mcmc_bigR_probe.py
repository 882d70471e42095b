"""MCMC iterations/s with many ladders (16 temperatures, CARMA(5,3), n=270).  CARMA_TUNE_PT_ROW_WGS_PER_CU (read once
per process) lets the row sampler kernel take grids of more than one workgroup per CU."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import carma_pack_amd as cpa
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
ms = 10*np.sqrt(np.mean(y*y)-np.mean(y)**2)
print("CARMA_TUNE_PT_ROW_WGS_PER_CU=%s CARMA_PT_KERNEL=%s" % (os.environ.get("CARMA_TUNE_PT_ROW_WGS_PER_CU", "(default)"),
                                                              os.environ.get("CARMA_PT_KERNEL", "(auto)")))
for R in (64, 96, 128, 160, 192, 256, 512):
    ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=ms)
    ctx.pt_create(16, R, adapt_iters=10**9, seed=3)
    ctx.pt_start(None)
    ctx.pt_iterate(100)
    n = 1000 if R <= 256 else 400
    t0 = time.perf_counter(); ctx.pt_iterate(n); dt = time.perf_counter() - t0
    print("R=%4d: %.0f it/s, %.3e chain-evals/s" % (R, n / dt, n * 16 * R / dt), flush=True)
