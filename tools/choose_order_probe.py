"""choose_order(pmax=7, ntrials=100) on the OGLE light curve (BASELINE configs[4]) with 1 / 4 / 8 / 16 host threads."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import carma_pack_amd as cpa
from carma_pack_amd import carma_pack as cm
g = np.load(os.path.join(ROOT, "tests/golden/ogle_grid.npz"))
t, y, yerr = g["t"], g["y"], g["yerr"]
for nj in (1, 4, 8, 16):
    model = cm.CarmaModel(t, y, yerr)
    t0 = time.perf_counter()
    best, pqlist, aicc = model.choose_order(7, ntrials=100, seed=1, njobs=nj)
    dt = time.perf_counter() - t0
    print("njobs=%2d: %.2f s, chosen (%d,%d), AICc %.3f" % (nj, dt, model.p, model.q, min(aicc)), flush=True)
