"""One evaluation per lane (k_logdens_carma_lane) against the lane-group kernels: time and agreement, per batch size.
CARMA_TUNE_LANE_MIN (read once per process) moves the smallest launch that takes the lane kernel: run once with 0
(lane kernel for everything beyond the wave pipeline) and once with a huge value (never)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
if os.environ.get('LANE_PROBE_REGULAR'):      # constant cadence: the transition factors are evaluated once
    t = np.floor(t[0]) + 2.0 * np.arange(len(t))      # (exact differences: every step repeats)
P, Q = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5, 3)
ctx = cpa.Context(t, y, yerr, P, Q, max_stdev=10*np.sqrt(np.mean(y*y)-np.mean(y)**2))
base = theta_batch(np.random.default_rng(2), 4096, P, Q, t, y, theta_center=g['theta'][0] if (P, Q) == (5, 3) else None,
                   frac_post=float(os.environ.get('LANE_PROBE_POST', '0.5')))      # 1.0: posterior-like vectors only
import oracle as orc
m = orc.OracleModel(t, y, yerr, P, Q, max_stdev=ctx.prior()[0])
want = m.logdensity_batch(base[:512], nthreads=8)
dev = torch.device('cuda'); st = torch.cuda.current_stream().cuda_stream
print("  ".join("%s=%s" % (k[11:], os.environ[k]) for k in sorted(os.environ) if k.startswith("CARMA_TUNE_")) or "(defaults)", " CARMA(%d,%d)" % (P, Q))
for B in [int(x) for x in os.environ.get("LANE_PROBE_B", "16384,24576,32768,65536,131072,1048576").split(",")]:
    th = torch.from_numpy(np.tile(base, (B // 4096 + 1, 1))[:B].copy()).to(dev)
    out = torch.empty(B, dtype=torch.float64, device=dev)
    for _ in range(3): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nrep = 200 if B <= 8192 else (20 if B <= 262144 else 5)
    for _ in range(nrep): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / nrep
    k = min(B, 512)
    got = out[:k].cpu().numpy()
    fin = np.isfinite(want[:k])
    ok = np.array_equal(np.isfinite(got), fin)
    rel = np.abs(got[fin] - want[:k][fin]) / np.abs(want[:k][fin])
    print("B=%8d  %-28s %9.1f us/launch  %.3e evals/s | vs oracle: pattern %s, median %.1e, >1e-10: %d of %d, max %.1e" % (
        B, ctx.kernel_name(B), dt * 1e6, B / dt, ok, np.median(rel), int(np.sum(rel > 1e-10)), fin.sum(), rel.max()), flush=True)
