#!/usr/bin/env python3
"""A/B timing of two builds of the library on ONE box, alternating: headline launch (B evaluations, CARMA(5,3), README series)
and the configs[2]-shape sampler.  usage: ab_time.py LIB_A LIB_B [B ...]   (paths of libcarma_mi355.so variants)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, json, time, numpy as np, torch
sys.path.insert(0, %r)
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
g = np.load(%r)
t, y, e = g["t"], g["y"], g["yerr"]
ctx = cpa.Context(t, y, e, 5, 3, max_stdev=10.0 * y.std())
res = {}
for B in %r:
    th = theta_batch(np.random.default_rng(2), B, 5, 3, t, y, theta_center=g["theta"][0])
    dev = torch.from_numpy(th).cuda(); o = torch.empty(B, dtype=torch.float64, device="cuda")
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): ctx.logdensity_dev(dev.data_ptr(), B, o.data_ptr())
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 200)
    res["B%%d_us" %% B] = best * 1e6
ctx.pt_create(16, 64, adapt_iters=10 ** 9, seed=11); ctx.pt_start(None); ctx.pt_iterate(200)
t0 = time.perf_counter(); ctx.pt_iterate(3000); res["mcmc_it_per_s"] = 3000 / (time.perf_counter() - t0)
c2 = cpa.Context(t, y, e, 5, 3, max_stdev=10.0 * y.std())
c2.pt_create(16, 192, adapt_iters=10 ** 9, seed=11); c2.pt_start(None); c2.pt_iterate(100)
t0 = time.perf_counter(); c2.pt_iterate(1500); res["mcmc_16x192_it_per_s"] = 1500 / (time.perf_counter() - t0)
res["kernel"] = ctx.kernel_name(1024)
print(json.dumps(res))
'''
libs = sys.argv[1:3]
Bs = [int(x) for x in sys.argv[3:]] or [1024]
for rep in range(2):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", CODE % (ROOT, os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"), Bs)],
                           env=dict(os.environ, CARMA_LIB_PATH=os.path.abspath(lib)), capture_output=True, text=True)
        print(os.path.basename(lib), r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-800:], flush=True)
