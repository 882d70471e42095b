"""Runs ONE 1024-eval launch of the -DCARMA_STAMPS diagnostic build (build_diag/) and lets the kernel
print its per-segment cycle shares."""
import os, sys, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import carma_pack_amd._lib as L0  # product lib (for Context class)
import ctypes as C
diag = os.environ.get("CARMA_DIAG_LIB") or os.path.join(ROOT, "build_diag", "libcarma_mi355_diag.so")
L0.LIB_PATH = diag
L0.lib = L0._load()
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(ROOT, 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
ctx = L0.Context(t, y, yerr, 5, 3, max_stdev=10*np.sqrt(np.mean(y*y)-np.mean(y)**2))
th = theta_batch(np.random.default_rng(2), int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 5, 3, t, y, theta_center=g['theta'][0])
out = ctx.logdensity(th)
print("---- second launch (warm)")
out = ctx.logdensity(th)
print("---- third launch (warm)")
out = ctx.logdensity(th)
print("finite", np.isfinite(out).sum())
