"""Duration of the first launches after the device has been idle / after other work (HIP events around every launch)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=10*np.sqrt(np.mean(y*y)-np.mean(y)**2))
dev = torch.device('cuda'); stream = torch.cuda.current_stream(); st = stream.cuda_stream
th = torch.from_numpy(theta_batch(np.random.default_rng(2), 1024, 5, 3, t, y, theta_center=g['theta'][0])).to(dev)
out = torch.empty(1024, dtype=torch.float64, device=dev)
def burst(n, label):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record(stream)
    for i in range(n):
        ctx.logdensity_dev(th.data_ptr(), 1024, out.data_ptr(), stream=st)
        evs[i + 1].record(stream)
    torch.cuda.synchronize()
    d = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(n)]
    print(label, " ".join("%.1f" % x for x in d[:30]), "| mean of all %.2f" % np.mean(d), flush=True)
burst(30, "cold (first launches of the process):")
time.sleep(0.5)
burst(30, "after 0.5 s idle:")
ctx.pt_create(16, 64, adapt_iters=10**9, seed=1); ctx.pt_start(None); ctx.pt_iterate(2000)
burst(30, "right after 2000 sampler iterations:")
for _ in range(2000): ctx.logdensity_dev(th.data_ptr(), 1024, out.data_ptr(), stream=st)
burst(30, "right after 2000 launches (not synchronised):")
torch.cuda.synchronize()
burst(30, "after a synchronise:")
