"""MCMC iterations/s of large ensembles (16 temperatures, CARMA(5,3), n = 270) per sampler kernel: the ladder kernel k_pt,
one chain per lane with an iteration as three launches (carma_pt_lane.hip).  LANE_PROBE_R: replica counts."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    kern, R = sys.argv[1], int(sys.argv[2])
    if kern != "auto":
        os.environ["CARMA_PT_KERNEL"] = kern
    import carma_pack_amd as cpa
    g = np.load(os.path.join(ROOT, 'tests/golden/carma53_readme.npz'))
    t, y, yerr = g['t'], g['y'], g['yerr']
    ms = 10 * np.sqrt(np.mean(y * y) - np.mean(y) ** 2)
    P_, Q_ = [int(x) for x in os.environ.get("LANE_PROBE_PQ", "5,3").split(",")]
    ctx = cpa.Context(t, y, yerr, P_, Q_, max_stdev=ms)
    ctx.pt_create(16, R, adapt_iters=10 ** 9, seed=3)
    ctx.pt_start(None)
    ctx.pt_iterate(20)
    n = max(20, min(400, int(3e6 / (16 * R) * 40)))
    t0 = time.perf_counter(); ctx.pt_iterate(n); dt = time.perf_counter() - t0
    acc, swp = ctx.pt_stats()
    print("CARMA(%d,%d) " % (P_, Q_) + "kernel %-7s (%-6s) R=%5d chains=%7d: %8.1f it/s  %.3e chain-evals/s  %.1f us/iteration  accept %.2f swap %.2f" % (
        kern, ctx.pt_kernel(), R, 16 * R, n / dt, n * 16 * R / dt, 1e6 * dt / n, acc.mean(), swp[:, 1:].mean()), flush=True)
    sys.exit(0)
Rs = [int(x) for x in os.environ.get("LANE_PROBE_R", "256,512,768,1024,1536,2048,3072,4096,8192").split(",")]
for R in Rs:
    for kern in os.environ.get("LANE_PROBE_KERNELS", "ladder,lane,auto").split(","):
        subprocess.run([sys.executable, os.path.abspath(__file__), kern, str(R)])
