"""Mid-range batch sizes (the producer/consumer G-lane kernel up to 512 waves, the plain kernel beyond)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(ROOT, "tests/golden/carma53_readme.npz"))
t, y, yerr = g["t"], g["y"], g["yerr"]
ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=10 * y.std())
dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
for mw in ("512",):
    for B in (2304, 3072, 4096, 6144, 8192):
        th = torch.from_numpy(theta_batch(np.random.default_rng(2), B, 5, 3, t, y, theta_center=g["theta"][0])).to(dev)
        out = torch.empty(B, dtype=torch.float64, device=dev)
        for _ in range(5): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        print("pc up to %4s waves: B=%5d %.1f us  %.3e evals/s" % (mw, B, dt * 1e6, B / dt), flush=True)
