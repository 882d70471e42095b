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
for name, env in (("w2", {}), ("w", {"CARMA_TUNE_WIN2_EVALS": "0"}), ("p3l", {"CARMA_TUNE_WIN2_EVALS": "0", "CARMA_TUNE_WIN_ROWS": "0"})):
    for k in ("CARMA_TUNE_WIN2_EVALS", "CARMA_TUNE_WIN_ROWS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    got = ctx.logdensity(th)
    r = rel(got, ref)
    print(name, ctx.kernel_name(th.shape[0]), "max rel %.2e  median %.2e  >1e-10: %d  nonfinite mismatch %d" % (np.max(r[np.isfinite(r)]), np.median(r), np.sum(r > 1e-10), np.sum(~np.isfinite(r))), flush=True)
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
