import os, sys, numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CARMA_LIB_PATH"] = os.path.join(ROOT, "build_var", "mdebug.so")
import carma_pack_amd as cpa
import oracle as orc
from helpers import loglik_truth, irregular_series, prior_like_theta
p, q = 6, 0
t, y, yerr = irregular_series(150, seed=170 + p)
rng = np.random.default_rng(1700 + 10 * p + q)
cplx = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(32)])
real = cplx.copy()
for f in range(p // 2 if p < 6 else 1):
    r1 = 10.0 ** rng.uniform(-2.0, -0.5, 32) * 3.0 ** f
    r2 = r1 * rng.uniform(3.0, 20.0, 32)
    real[:, 3 + 2 * f] = np.log(r1 * r2)
    real[:, 4 + 2 * f] = np.log(r1 + r2)
ctx = cpa.Context(t, y, yerr, p, q)
th = real[9:10]
print("kernel", ctx.kernel_name(1))
got = ctx.logdensity(th, ignore_prior=True)
truth = loglik_truth(t, y, yerr, th[0], p, q)[0]
print("device", got, "truth", truth, "rel", abs(got[0] - truth) / abs(truth))
