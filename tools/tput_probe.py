"""Throughput of the batched log-density kernel vs batch size (device-resident thetas)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
ms = 10*np.sqrt(np.mean(y*y)-np.mean(y)**2)
ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=ms)
rng = np.random.default_rng(2)
base = theta_batch(rng, 1024, 5, 3, t, y, theta_center=g['theta'][0])
dev = torch.device('cuda')
st = torch.cuda.current_stream().cuda_stream
for B in (1024, 2048, 4096, 8192, 16384, 65536, 262144, 1048576):
    th = torch.from_numpy(np.tile(base, (B // 1024, 1))).to(dev)
    out = torch.empty(B, dtype=torch.float64, device=dev)
    reps = max(3, min(200, (1 << 21) // B))
    for _ in range(3): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print("B=%8d  %.3f ms/launch  %.3e evals/s" % (B, dt * 1e3, B / dt), flush=True)
