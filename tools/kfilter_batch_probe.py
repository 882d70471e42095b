"""Filter() of many models in one launch (carma_kfilter_batch_carma, one model per lane) against the one-model entry point
looped and the oracle's C filter on one host core: CARMA(5,3) on the 270-point series of BASELINE configs[1].
Run on the GPU box:  python tools/kfilter_batch_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
import oracle as orc

g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
t, y, yerr = g["t"], g["y"], g["yerr"]
rng = np.random.default_rng(5)
p, q = 5, 3
for B in [int(x) for x in os.environ.get("KFB_PROBE_B", "64,1000,10000,75000").split(",")]:
    th = theta_batch(rng, B, p, q, t, y, theta_center=g["theta"][0])
    roots = np.array([orc.ar_roots(x, p) for x in th])
    ma = np.array([orc.ma_coefs(x, p, q) for x in th])[:, : q + 1]
    sig2 = np.array([x[0] ** 2 / orc.variance(r, m) for x, r, m in zip(th, roots, ma)])
    cpa.kfilter_carma_batch(t, y, yerr, sig2[:64], roots[:64], ma[:64], mu=th[:64, 2])
    t0 = time.perf_counter(); mean, var, sing = cpa.kfilter_carma_batch(t, y, yerr, sig2, roots, ma, mu=th[:, 2]); tb = time.perf_counter() - t0
    k = min(B, 300)
    t0 = time.perf_counter()
    for i in range(k):
        cpa.kfilter_carma(t, y - th[i, 2], yerr, sig2[i], roots[i], ma[i])
    t1 = (time.perf_counter() - t0) / k
    t0 = time.perf_counter()
    for i in range(k):
        orc.kfilter_carma(t, y - th[i, 2], yerr, sig2[i], roots[i], ma[i])
    to = (time.perf_counter() - t0) / k
    print("B = %6d models x %d data: one launch %.4f s (%.2f us per model, host copies included) | one-model entry point %.1f us "
          "per model | oracle C filter, one core %.1f us per model" % (B, t.size, tb, 1e6 * tb / B, 1e6 * t1, 1e6 * to), flush=True)
