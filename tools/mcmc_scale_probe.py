"""MCMC iterations/s of the row sampler kernel against the number of replicas (chip load)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import carma_pack_amd as cpa
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
ms = 10*np.sqrt(np.mean(y*y)-np.mean(y)**2)
ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=ms)
for T, R, ex in ((16, 1, True), (16, 8, True), (16, 32, True), (16, 64, True), (16, 64, False), (4, 64, True), (4, 256, True)):
    ctx.pt_create(T, R, adapt_iters=100000, seed=1)
    ctx.pt_start(None)
    ctx.pt_iterate(200, do_exchange=ex)
    t0 = time.perf_counter(); ctx.pt_iterate(2000, do_exchange=ex); dt = time.perf_counter()-t0
    print("T=%d R=%d exchange=%s: %.1f it/s (%.1f us/iter)" % (T, R, ex, 2000/dt, dt/2000*1e6), flush=True)
