import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import carma_pack_amd as cpa
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
ms = 10*np.sqrt(np.mean(y*y)-np.mean(y)**2)
ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=ms)
for T, R in ((16, 64), (16, 256), (10, 1), (16, 512)):
    ctx.pt_create(T, R, adapt_iters=100000, seed=1)
    ctx.pt_start(None)
    ctx.pt_iterate(200)
    t0 = time.perf_counter(); ctx.pt_iterate(2000); dt = time.perf_counter()-t0
    acc, swp = ctx.pt_stats()
    print("T=%d R=%d: %.1f it/s, %.3e chain-evals/s, acc %.3f swap %.3f" % (T, R, 2000/dt, 2000*T*R/dt, acc.mean(), swp[:,1:].mean()), flush=True)
