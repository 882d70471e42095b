"""Per-step time of the latency-regime kernel: B=1024 evaluations at several series lengths."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
rng = np.random.default_rng(2)
dev = torch.device('cuda'); st = torch.cuda.current_stream().cuda_stream
res = []
for rep in (1, 2, 4, 10):
    tt = np.concatenate([t + k * (t[-1] + 5.0) for k in range(rep)]); yy = np.tile(y, rep); ee = np.tile(yerr, rep)
    n = tt.size if rep > 1 else 270
    ms = 10*np.sqrt(np.mean(yy*yy)-np.mean(yy)**2)
    ctx = cpa.Context(tt, yy, ee, 5, 3, max_stdev=ms)
    th = torch.from_numpy(theta_batch(rng, 1024, 5, 3, t, y, theta_center=g['theta'][0])).to(dev)
    out = torch.empty(1024, dtype=torch.float64, device=dev)
    for _ in range(5): ctx.logdensity_dev(th.data_ptr(), 1024, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): ctx.logdensity_dev(th.data_ptr(), 1024, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    res.append((tt.size, dt))
    print("n=%5d  %.1f us/launch" % (tt.size, dt * 1e6), flush=True)
(n0, t0_), (n1, t1_) = res[0], res[-1]
print("per step: %.1f ns ; fixed: %.1f us" % ((t1_ - t0_) / (n1 - n0) * 1e9, (t0_ - (t1_ - t0_) / (n1 - n0) * n0) * 1e6))
