"""Launch time at a few batch sizes for one build (CARMA_LIB_PATH) -- A/B of builds on one box."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=10*np.sqrt(np.mean(y*y)-np.mean(y)**2))
base = theta_batch(np.random.default_rng(2), 1024, 5, 3, t, y, theta_center=g['theta'][0])
dev = torch.device('cuda'); st = torch.cuda.current_stream().cuda_stream
out_s = []
for B in (1024, 2048, 4096, 8192, 65536):
    th = torch.from_numpy(np.tile(base, (B // 1024 + 1, 1))[:B].copy()).to(dev)
    out = torch.empty(B, dtype=torch.float64, device=dev)
    reps = 300 if B < 20000 else 40
    for _ in range(5): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    out_s.append("%d: %.1f" % (B, dt * 1e6))
print("%-14s" % os.environ.get("CARMA_LIB_PATH", "main").split("/")[-1], " | ".join(out_s), "us", flush=True)
