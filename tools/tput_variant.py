"""tput_probe on an alternative build of the library (tools/build_diag.sh with DIAG_FLAGS): A/B of kernel variants."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import carma_pack_amd._lib as L0
if len(sys.argv) > 1 and sys.argv[1] == "diag":
    L0.LIB_PATH = os.path.join(ROOT, "build_diag", "libcarma_mi355_diag.so")
    L0.lib = L0._load()
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(ROOT, "tests/golden/carma53_readme.npz"))
t, y, yerr = g["t"], g["y"], g["yerr"]
ctx = L0.Context(t, y, yerr, 5, 3, max_stdev=10 * y.std())
dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
for B in (16384, 262144):
    th = torch.from_numpy(theta_batch(np.random.default_rng(2), B, 5, 3, t, y, theta_center=g["theta"][0])).to(dev)
    out = torch.empty(B, dtype=torch.float64, device=dev)
    for _ in range(3): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("B=%7d %.3f ms  %.3e evals/s" % (B, dt * 1e3, B / dt))
