import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
t, y, e = g["t"], g["y"], g["yerr"]
ctx = cpa.Context(t, y, e, 5, 3, max_stdev=10.0 * y.std())
th = theta_batch(np.random.default_rng(7), 1024, 5, 3, t, y, theta_center=g["theta"][0])
print(ctx.kernel_name(1024))
out = ctx.logdensity(th)
out = ctx.logdensity(th)
print(np.isfinite(out).sum())
