import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import carma_pack_amd as cpa
import oracle as orc
og = np.loadtxt("tests/golden/ogle_lmc_lpv_00007.dat")
t, y, e = og[:, 0], og[:, 1], og[:, 2]
model = cpa.CarmaModel(t, y, e)
mle = model.get_mle(2, 1, ntrials=24, seed=1, return_all=True)
f = np.array([r.fun for r in mle]); i = int(np.argmin(f))
x = mle[i].x
print("best start", i, "fun", f[i], "x", repr(x))
ctx = cpa.Context(t, y, e, 2, 1, max_stdev=10 * np.std(y))
m = orc.OracleModel(t, y, e, 2, 1, max_stdev=10 * np.std(y))
tr = orc.truth_logdensity(t, y, e, x, 2, 1)
print("oracle ign", m.logdensity(x, ignore_prior=True), "oracle", m.logdensity(x), "truth", tr)
for B in (1, 64, 3200, 70000):
    v = ctx.logdensity(np.tile(x, (B, 1)), ignore_prior=True)
    print(B, ctx.kernel_name(B), repr(v[0]), "all same", bool(np.all(v == v[0])))
print("roots", orc.ar_roots(x, 2))
np.save("gpurun_out/mle21_x.npy", x)
