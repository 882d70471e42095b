import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import carmcmc as cm
d = np.loadtxt("tests/golden/ogle_lmc_lpv_00007.dat")
t, y, e = d[:, 0] - d[:, 0].min(), d[:, 1], d[:, 2]
model = cm.CarmaModel(t, y, e, p=1, q=0)
model.choose_order(3, ntrials=8, seed=1)
for nj in (1, 4, 8, 28):
    t0 = time.perf_counter()
    best, pq, aicc = model.choose_order(7, ntrials=100, seed=7, njobs=nj)
    print("njobs=%d: %.2f s  chosen (%d,%d)  min AICc %.3f" % (nj, time.perf_counter() - t0, model.p, model.q, min(aicc)), flush=True)
