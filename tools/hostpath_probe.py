"""Host-path overhead of carma_logdensity_batch (H2D + launch + D2H + sync) against the device-pointer entry point."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
t, y, yerr = g["t"], g["y"], g["yerr"]
ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=10 * y.std())
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
for B in (100, 800, 2300):
    th = theta_batch(np.random.default_rng(1), B, 5, 3, t, y, theta_center=g["theta"][0])
    for _ in range(20):
        ctx.logdensity(th)
    t0 = time.perf_counter()
    for _ in range(500):
        ctx.logdensity(th)
    host = (time.perf_counter() - t0) / 500
    d_th = torch.from_numpy(th).to(dev)
    out = torch.empty(B, dtype=torch.float64, device=dev)
    for _ in range(20):
        ctx.logdensity_dev(d_th.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        ctx.logdensity_dev(d_th.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize()
    devt = (time.perf_counter() - t0) / 500
    print("B=%5d  host path %.1f us/call   device path %.1f us/launch" % (B, host * 1e6, devt * 1e6))
