#!/usr/bin/env python3
"""Round 6, pricing of the two-sided filter BEFORE any kernel: the existing latency kernels on the FIRST
HALF of the README series.  A two-sided launch of B evaluations runs 2 B half-length recursions, so
   n = 135 at 2 B evaluations   ~ the same waves and the same work as the split at B (recursion waves on their own)
   n = 135 at   B evaluations   ~ the split at B / 2
Both pipelines (window: CARMA_TUNE_WIN_ROWS large; one-datum: 0), wall clock over back-to-back launches.
usage: split_proxy_probe.py [B ...]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch

g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
t, y, e = g["t"], g["y"], g["yerr"]
Bs = [int(x) for x in sys.argv[1:]] or [256, 512, 1024, 2048]
res = []
for nuse in (270, 135, 68):
    ctx = cpa.Context(t[:nuse], y[:nuse], e[:nuse], 5, 3, max_stdev=10.0 * y.std())
    for mode, env in (("win", 1000000), ("p3l", 0)):
        cpa._lib.tune_set("WIN2_EVALS", 0)
        cpa._lib.tune_set("WIN_ROWS", env)
        for B in Bs:
            th = theta_batch(np.random.default_rng(2), B, 5, 3, t, y, theta_center=g["theta"][0])
            dev = torch.from_numpy(th).cuda()
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
            r = dict(n=nuse, mode=mode, B=B, us=round(best * 1e6, 2), kernel=ctx.kernel_name(B))
            res.append(r)
            print(json.dumps(r), flush=True)
