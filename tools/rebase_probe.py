"""Per-datum time of the wave pipeline against the re-base rate of the co-rotating frame: CARMA(7,6) on BASELINE
configs[3]'s 10 000-point series (time steps 0.1 + |Cauchy|) with posterior-like and with prior-like parameter vectors,
and the README CARMA(5,3) case for comparison.  1024 evaluations per launch."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch, prior_like_theta, config4_series

dev = torch.device('cuda'); st = torch.cuda.current_stream().cuda_stream

def run(name, ctx, th, n):
    B = th.shape[0]
    d_th = torch.from_numpy(th.copy()).to(dev)
    out = torch.empty(B, dtype=torch.float64, device=dev)
    for _ in range(3): ctx.logdensity_dev(d_th.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 20 if n > 1000 else 300
    for _ in range(reps): ctx.logdensity_dev(d_th.data_ptr(), B, out.data_ptr(), stream=st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    fin = int(torch.isfinite(out).sum())
    print("%-44s %-26s %9.1f us/launch  %6.1f ns/datum  finite %d/%d" % (name, ctx.kernel_name(B), dt * 1e6, dt * 1e9 / n, fin, B), flush=True)

t, y, e, th0 = config4_series()
ctx = cpa.Context(t, y, e, 7, 6, max_stdev=10 * y.std())
rng = np.random.default_rng(3)
run("config4 CARMA(7,6) n=10000 posterior-like", ctx, th0 + 0.01 * rng.standard_normal((1024, th0.size)), t.size)
run("config4 CARMA(7,6) n=10000 prior-like", ctx, np.array([prior_like_theta(rng, 7, 6, t, y) for _ in range(1024)]), t.size)
run("config4 CARMA(7,6) n=10000 half/half", ctx, theta_batch(rng, 1024, 7, 6, t, y, theta_center=th0), t.size)
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/carma53_readme.npz'))
t, y, e = g['t'], g['y'], g['yerr']
ctx = cpa.Context(t, y, e, 5, 3, max_stdev=10 * y.std())
run("README CARMA(5,3) n=270 posterior-like", ctx, g['theta'][0] + 0.01 * rng.standard_normal((1024, 11)), t.size)
run("README CARMA(5,3) n=270 prior-like", ctx, np.array([prior_like_theta(rng, 5, 3, t, y) for _ in range(1024)]), t.size)
