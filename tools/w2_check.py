#!/usr/bin/env python3
"""Round 6: quick parity + timing of the two-sided window kernel against the oracle and the one-sided kernels (README series)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import carma_pack_amd as cpa
import oracle as orc
from carma_pack_amd.synth import theta_batch

g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
t, y, e = g["t"], g["y"], g["yerr"]
ctx = cpa.Context(t, y, e, 5, 3, max_stdev=10.0 * y.std())
m = orc.OracleModel(t, y, e, 5, 3, max_stdev=10.0 * y.std())
th = np.concatenate([g["theta"], theta_batch(np.random.default_rng(2), 224, 5, 3, t, y, theta_center=g["theta"][0])])
ref = m.logdensity_batch(th)
def rel(a, b):
    fin = np.isfinite(b)
    out = np.zeros(b.size)
    out[fin] = np.abs(a[fin] - b[fin]) / np.maximum(1.0, np.abs(b[fin]))
    out[~fin] = np.where((a[~fin] == b[~fin]) | (np.isnan(a[~fin]) & np.isnan(b[~fin])), 0.0, np.inf)
    return out
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import loglik_truth
th0 = g["theta"][0]
ill = []
for eps in (1e-3, 1e-4, 1e-5, 1e-6):
    x = th0.copy(); x[5:7] = th0[3:5] + eps; ill.append(x)
ill = np.array(ill)
tr = np.array([loglik_truth(t, y, e, x, 5, 3)[0] for x in ill])
orc_ill = m.logdensity_batch(ill, ignore_prior=True)
for name, env in (("w2", {}), ("w", {"WIN2_EVALS": 0}), ("p3l", {"WIN2_EVALS": 0, "WIN_ROWS": 0})):
    cpa._lib.tune_reset()
    for k, v in env.items():
        cpa._lib.tune_set(k, v)
    got = ctx.logdensity(th)
    r = rel(got, ref)
    print(name, ctx.kernel_name(th.shape[0]), "max rel %.2e  median %.2e  >1e-10: %d  nonfinite mismatch %d" % (np.max(r[np.isfinite(r)]), np.median(r), np.sum(r > 1e-10), np.sum(~np.isfinite(r))), flush=True)
    gi = ctx.logdensity(ill, ignore_prior=True)
    print("  roots 1e-3 .. 1e-6 apart, distance from the exact value: device", ["%.1e" % v for v in np.abs(gi - tr) / np.abs(tr)],
          "oracle", ["%.1e" % v for v in np.abs(orc_ill - tr) / np.abs(tr)])
    if name == "w2":
        bad = np.argsort(r)[-5:]
        print("  worst:", [(int(i), float(got[i]), float(ref[i])) for i in bad])
    for B in (256, 512, 1024, 1536, 2048):
        thb = theta_batch(np.random.default_rng(2), B, 5, 3, t, y, theta_center=g["theta"][0])
        dev = torch.from_numpy(thb).cuda()
        o = torch.empty(B, dtype=torch.float64, device="cuda")
        for _ in range(50):
            ctx.logdensity_dev(dev.data_ptr(), B, o.data_ptr())
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                ctx.logdensity_dev(dev.data_ptr(), B, o.data_ptr())
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 300)
        print("  ", json.dumps(dict(mode=name, B=B, us=round(best * 1e6, 2), kernel=ctx.kernel_name(B))), flush=True)
