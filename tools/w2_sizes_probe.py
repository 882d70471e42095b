#!/usr/bin/env python3
"""Round 6: where does the two-sided window kernel stop paying?  Launch times over batch sizes with the dispatch's own choice and
with the two-sided kernel forced (carma_tune_set WIN2_EVALS), README series, CARMA(5,3) and (7,6)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
t, y, e = g["t"], g["y"], g["yerr"]
for p, q in ((5, 3), (7, 6), (3, 1)):
    ctx = cpa.Context(t, y, e, p, q, max_stdev=10.0 * y.std())
    for B in (256, 512, 768, 1024, 1280, 1536, 2048):
        th = theta_batch(np.random.default_rng(2), B, p, q, t, y, theta_center=g["theta"][0] if (p, q) == (5, 3) else None)
        dev = torch.from_numpy(th).cuda()
        o = torch.empty(B, dtype=torch.float64, device="cuda")
        row = {"p": p, "q": q, "B": B}
        for mode, v in (("dispatch", None), ("two_sided", 1 << 20), ("one_sided", 0)):
            cpa._lib.tune_set("WIN2_EVALS", v)
            for _ in range(30):
                ctx.logdensity_dev(dev.data_ptr(), B, o.data_ptr())
            best = 1e9
            for _ in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    ctx.logdensity_dev(dev.data_ptr(), B, o.data_ptr())
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 200)
            row[mode] = round(best * 1e6, 2)
            row[mode + "_kernel"] = ctx.kernel_name(B)
        cpa._lib.tune_set("WIN2_EVALS", None)
        print(json.dumps(row), flush=True)
