"""Soak of the sampler path for large ensembles (carma_pt_lane.hip): orders, ladder lengths that do and do not divide 64, ragged
last waves, long runs through several chunks of iterations (state converted out and back every chunk), and after each: still on
that path, every stored log-posterior of a sample of the final chain states is the oracle's LogDensity, the ladders swap, no
NaN in the proposal factors.  Run on the GPU box:  python tools/soak_pt_lane.py [iterations-per-shape]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CARMA_PT_KERNEL"] = "lane"
import carma_pack_amd as cpa
import oracle as orc
from helpers import assert_parity_states, irregular_series, loglik_truth, oracle_noise_scale
niter = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
fails = 0
SHAPES = ((5, 3, 16, 1024, 270), (5, 3, 16, 4099, 150), (7, 6, 8, 2048, 120), (3, 1, 33, 500, 120), (2, 1, 64, 300, 100),
          (4, 2, 10, 1300, 150), (6, 5, 12, 1100, 90), (2, 0, 1, 20000, 90), (5, 0, 7, 2341, 200),
          (1, 0, 16, 64, 270), (1, 0, 8, 1500, 120), (1, 0, 5, 4000, 70))          # CAR(1): the parallel-in-time launch / the lane form
only = os.environ.get("SOAK_ONLY")
for (p, q, T, R, n) in [SHAPES[int(only)]] if only else SHAPES:
    t, y, yerr = irregular_series(n, seed=11 * p + q)
    ctx = cpa.Context(t, y, yerr, p, q)
    ctx.pt_create(T, R, adapt_iters=niter // 2, seed=1000 + T)
    ctx.pt_start(None)
    k0 = ctx.pt_kernel()
    t0 = time.perf_counter()
    ctx.pt_iterate(niter)
    smp, slp = ctx.pt_sample(50, thin=3)
    dt = time.perf_counter() - t0
    th, lp = ctx.pt_get_chains()
    acc, swp = ctx.pt_stats()
    m = orc.OracleModel(t, y, yerr, p, q, max_stdev=ctx.prior()[0])
    flat = th.reshape(-1, 3 + p + q)
    sel = np.random.default_rng(1).choice(flat.shape[0], size=min(600, flat.shape[0]), replace=False)
    ok = True
    try:
        assert np.all(np.isfinite(flat)) and np.all(np.isfinite(smp))
        # (arb_factor 8, as tools/soak_pt_row.py: over thousands of states which side lands nearer the exact value is a coin flip;
        # allowance of arbitrated states 10 %: after 2e4 iterations the tempered CARMA(7,6) chains have drifted so far out in the
        # unconstrained MA parameters that the ORACLE is beyond 1e-10 of the exact value on 55 of 599 states -- up to 2e-2 --
        # while the device is at 1e-13 ... 1e-16 on them: profiles/r04/soak_pt_lane_v3.txt)
        assert_parity_states(lp.reshape(-1)[sel], m.logdensity_batch(flat[sel], nthreads=os.cpu_count() or 8), flat[sel], p, q, 1e-10,
                             "soak", arbiter=lambda i: loglik_truth(t, y, yerr, flat[sel][i], p, q)[0], max_arb_frac=0.10, arb_factor=8.0,
                             max_overflow_frac=0.05, noise_scale=lambda i: oracle_noise_scale(m, t, y, yerr, flat[sel][i], p, q))
        last = smp[:200, -1]                                  # saved samples: stored log-posterior == LogDensity(sample)
        assert_parity_states(slp[:200, -1], m.logdensity_batch(last, nthreads=os.cpu_count() or 8), last, p, q, 1e-10, "saved",
                             arbiter=lambda i: loglik_truth(t, y, yerr, last[i], p, q)[0], max_arb_frac=0.08, arb_factor=8.0, max_overflow_frac=0.05)
    except AssertionError as ex:
        ok = False
        print("   FAILURE:", str(ex)[:300])
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez(os.path.join(ROOT, "gpurun_out", "soak_lane_fail_p%d_q%d_T%d_R%d.npz" % (p, q, T, R)), t=t, y=y, yerr=yerr, theta=flat[sel],
                 got=lp.reshape(-1)[sel], want=m.logdensity_batch(flat[sel], nthreads=os.cpu_count() or 8), max_stdev=ctx.prior()[0])
    still = ctx.pt_kernel()
    good = ok and still == k0 == "lane" and (T == 1 or swp[:, 1:].mean() > 0.01) and acc.mean() > 0.02
    fails += not good
    print("CARMA(%d,%d) T=%2d R=%5d n=%3d (%6d chains): %s -> %s, %d iterations in %.2f s (%.0f it/s, %.2e chain-evals/s), accept %.2f swap %.2f  %s" % (
        p, q, T, R, n, T * R, k0, still, niter + 150, dt, (niter + 150) / dt, (niter + 150) * T * R / dt, acc.mean(),
        swp[:, 1:].mean() if T > 1 else 0.0, "ok" if good else "FAILED"), flush=True)
print("soak:", "all shapes ok" if fails == 0 else "%d shapes FAILED" % fails)
