"""Ladder kernel (k_pt) at large replica counts: producer/consumer variant against the plain one (CARMA_PT_PLAIN=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import carma_pack_amd as cpa
g = np.load(os.path.join(ROOT, 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
for plain in ("0", "1", "auto"):
    os.environ.pop("CARMA_PT_PLAIN", None)
    if plain != "auto":
        os.environ["CARMA_PT_PLAIN"] = plain
    ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=10 * y.std())
    for T, R in ((16, 128), (16, 256), (16, 512), (16, 1024), (10, 512)):
        ctx.pt_create(T, R, adapt_iters=100000, seed=1)
        ctx.pt_start(None)
        ctx.pt_iterate(100)
        t0 = time.perf_counter(); ctx.pt_iterate(500); dt = time.perf_counter() - t0
        print("plain=%s T=%d R=%d: %.1f it/s, %.3e chain-evals/s" % (plain, T, R, 500 / dt, 500 * T * R / dt), flush=True)
