"""Phase timings of the latency-regime pipeline (carma_pipe3l.h) from the diagnostic build (tools/build_diag.sh,
-DCARMA_STAMPS): core-clock marks of the four waves of workgroup 0 in a FULL launch of 1024 evaluations --
marks: 1 model set up, 2 constants ready, 3 recursion done, 4 result stored.  Third launch = warm."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CARMA_LIB_PATH"] = os.environ.get("DIAG_SO", os.path.join(ROOT, "build_diag", "libcarma_mi355_diag.so"))
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(ROOT, 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
th = theta_batch(np.random.default_rng(2), B, 5, 3, t, y, theta_center=g['theta'][0])
ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=10 * y.std())
for i in range(3):
    print("---- launch %d" % i, flush=True)
    ctx.logdensity(th, ignore_prior=True)
