"""Phase timings of the latency-regime pipeline (carma_pipe3l.h): cycles per chunk of every wave, set-up time, from the
diagnostic build  DIAG_FLAGS=-DCARMA_DBG tools/build_diag.sh  (device printf of workgroup 0; third launch = warm)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import carma_pack_amd._lib as L0
L0.LIB_PATH = os.path.join(ROOT, "build_diag", "libcarma_mi355_diag.so")
L0.lib = L0._load()
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(ROOT, 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
th = theta_batch(np.random.default_rng(2), 4, 5, 3, t, y, theta_center=g['theta'][0])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = L0.Context(t[:n], y[:n], yerr[:n], 5, 3, max_stdev=10 * y.std())
ctx.logdensity(th, ignore_prior=True); ctx.logdensity(th, ignore_prior=True); print("---- third"); print(ctx.logdensity(th, ignore_prior=True))
