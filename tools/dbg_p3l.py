import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
rng = np.random.default_rng(2)
th = theta_batch(rng, 8, 5, 3, t, y, theta_center=g['theta'][0])
for n in (10, 16, 17, 33, 270):
    ctx = cpa.Context(t[:n], y[:n], yerr[:n], 5, 3, max_stdev=10 * y.std())
    os.environ.pop("CARMA_LOGDENS_KERNEL", None)
    a = ctx.logdensity(th, ignore_prior=True)
    os.environ["CARMA_LOGDENS_KERNEL"] = "p3"
    b = ctx.logdensity(th, ignore_prior=True)
    print(n, "new", a[:4], "\n   old", b[:4], "\n   rel", np.abs(a - b)[:8] / np.abs(b)[:8])
