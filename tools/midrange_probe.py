"""Launch time of the batched log-density kernel around the launch-shape thresholds (CARMA(5,3), n=270).
CARMA_TUNE_P3L_ROWS (read once per process) moves the largest launch that takes the wave pipeline."""
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
print("CARMA_TUNE_P3L_ROWS=%s" % os.environ.get("CARMA_TUNE_P3L_ROWS", "(default)"))
for B in (512, 1024, 1280, 1536, 2048, 2560, 3072, 3584, 4096, 5120, 6144, 8192):
    th = torch.from_numpy(np.tile(base, (B // 1024 + 1, 1))[:B].copy()).to(dev)
    out = torch.empty(B, dtype=torch.float64, device=dev)
    for _ in range(5): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
    print("B=%6d  %.1f us/launch  %.3e evals/s" % (B, dt * 1e6, B / dt), flush=True)
