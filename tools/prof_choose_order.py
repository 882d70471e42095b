"""cProfile of CarmaModel.choose_order on the OGLE series (where does the wall time go: host optimiser or launches)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import carmcmc as cm
d = np.loadtxt(os.path.join(ROOT, "tests", "golden", "ogle_lmc_lpv_00007.dat"))
t, y, e = d[:, 0] - d[:, 0].min(), d[:, 1], d[:, 2]
model = cm.CarmaModel(t, y, e, p=1, q=0)
model.choose_order(3, ntrials=8, seed=1)          # warm-up (contexts, kernels)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
best, pq, aicc = model.choose_order(7, ntrials=100, seed=7)
pr.disable()
print("choose_order(pmax=7, ntrials=100): %.2f s" % (time.perf_counter() - t0))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
