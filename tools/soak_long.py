"""Long run of the bench's sampler shape (16 temperatures x 64 ladders, then x 192): millions of cross-workgroup rendezvous of
the tagged staging without a time-out, a fall-back or a lost chain.  python tools/soak_long.py [iterations]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import carma_pack_amd as cpa
g = np.load(os.path.join(ROOT, "tests/golden/carma53_readme.npz"))
t, y, yerr = g["t"], g["y"], g["yerr"]
niter = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
for R, frac in ((64, 1.0), (192, 0.25), (128, 0.25)):
    ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=10 * y.std())
    ctx.pt_create(16, R, adapt_iters=niter // 10, seed=5 + R)
    ctx.pt_start(None)
    n = int(niter * frac)
    t0 = time.perf_counter()
    done = 0
    while done < n:
        step = min(250000, n - done)
        ctx.pt_iterate(step)
        done += step
        assert ctx.pt_kernel() == "row", "fell back to the ladder kernel after %d iterations" % done
    dt = time.perf_counter() - t0
    th, lp = ctx.pt_get_chains()
    acc, swp = ctx.pt_stats()
    assert np.isfinite(lp).all() and np.isfinite(th).all()
    print("16 x %3d: %d iterations in %.1f s (%.0f it/s), still on k_pt_row, accept %.3f swap %.3f, cold-chain log-posterior %.2f +- %.2f" % (
        R, n, dt, n / dt, acc.mean(), swp[:, 1:].mean(), lp[:, 0].mean(), lp[:, 0].std()), flush=True)
