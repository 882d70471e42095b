"""CAR(1) log-density launch time and sampler iterations/s (one lane per evaluation).  CARMA_LIB_PATH selects an A/B build."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import carma_pack_amd as cpa
g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
t, y, yerr = g["t"], g["y"], g["yerr"]
ctx = cpa.Context(t, y, yerr, 1, 0)
rng = np.random.default_rng(1)
dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
for B in (64, 1024, 16384, 65536, 262144):
    th = np.c_[rng.uniform(0.5, 3.0, B), rng.uniform(0.8, 1.5, B), y.mean() + 0.1 * rng.standard_normal(B), rng.uniform(-5.0, -1.0, B)]
    d = torch.from_numpy(th).to(dev); out = torch.empty(B, dtype=torch.float64, device=dev)
    for _ in range(3): ctx.logdensity_dev(d.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ctx.logdensity_dev(d.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print("B=%7d  %7.1f us/launch  %.3e evals/s  finite %d" % (B, 1e6 * dt, B / dt, int(np.isfinite(out.cpu().numpy()).sum())), flush=True)
for R in [int(x) for x in os.environ.get("CAR1_PROBE_R", "64,1024").split(",")]:
    ctx.pt_create(16, R, adapt_iters=10 ** 9, seed=3); ctx.pt_start(None); ctx.pt_iterate(50)
    t0 = time.perf_counter(); ctx.pt_iterate(400); dt = time.perf_counter() - t0
    print("sampler 16 x %4d (%s): %8.1f it/s  %.1f us/iteration" % (R, ctx.pt_kernel(), 400 / dt, 1e6 * dt / 400), flush=True)
