#!/usr/bin/env python3
"""Runs the five BASELINE.json configs on one MI355X and prints one JSON document
(kept under profiles/ as evidence; bench.py stays the contract benchmark for configs[1])."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import carma_pack_amd as cpa  # noqa: E402
from carma_pack_amd.synth import irregular_series, prior_like_theta  # noqa: E402
from carma_pack_amd import parallel as par  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
out = {}

# ---- config 1: CAR(1), n=100, 1 evaluation ------------------------------------------------------
c = np.load(os.path.join(G, "car1_n100.npz"))
ctx = cpa.Context(c["t"], c["y"], c["yerr"], 1, 0)
ll = ctx.logdensity(c["theta"][0]) - ctx.logprior(c["theta"][0])
out["config1_car1_n100"] = {"loglik_gpu": ll, "loglik_dense_gp": float(c["dense_loglik"][0]),
                            "rel_err": abs(ll - c["dense_loglik"][0]) / abs(ll)}

# ---- config 3: CARMA(5,3), n=270, full PT-MCMC 50k samples + 25k burn-in, 16 temperatures x 64 walkers
g = np.load(os.path.join(G, "carma53_readme.npz"))
t, y, yerr = g["t"], g["y"], g["yerr"]
ms = 10 * np.sqrt(np.mean(y * y) - np.mean(y) ** 2)
ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=ms)
t0 = time.perf_counter()
samples, lp = ctx.pt_run(16, 64, 50000, 25000, 1, seed=2024)
dt = time.perf_counter() - t0
acc, swp = ctx.pt_stats()
pooled = samples[:, ::10].reshape(-1, 11)
truth = g["theta"][0]
out["config3_pt_mcmc"] = {
    "iterations": 75000, "temperatures": 16, "walkers": 64, "seconds": dt, "iters_per_s": 75000 / dt,
    "chain_evals_per_s": 75000 * 1024 / dt, "accept_rate_cold": float(acc[:, 0].mean()),
    "swap_rate": float(swp[:, 1:].mean()), "max_logpost": float(lp.max()),
    "logpost_at_truth": float(ctx.logdensity(truth)),
    "posterior_mean_first5": pooled.mean(0)[:5].tolist(), "posterior_sd_first5": pooled.std(0)[:5].tolist(),
    "truth_first5": truth[:5].tolist(),
}

# ---- config 4: CARMA(7,6), n=10000 (0.1 + |Cauchy| steps, own CARMA draw), 8 temperatures x 128 replicas: the whole
#      ladder on one GPU, then the same ladder as two blocks of 4 / eight blocks of 1 through carma_pt_iterate_sharded
#      (RCCL send/recv to this process's own rank: one GPU here; the blocks run one after the other on it)
from carma_pack_amd import _lib  # noqa: E402
from carma_pack_amd.synth import config4_series  # noqa: E402
t4, y4, e4, th4 = config4_series(10000, seed=4)
ctx4 = cpa.Context(t4, y4, e4, 7, 6)
rng = np.random.default_rng(4)
th = np.array([prior_like_theta(rng, 7, 6, t4, y4) for _ in range(1024)])
ctx4.logdensity(th[:8])
t0 = time.perf_counter()
ld = ctx4.logdensity(th, ignore_prior=True)
dt_eval = time.perf_counter() - t0
ctx4.pt_create(8, 128, adapt_iters=10 ** 6, seed=5)
ctx4.pt_start(None)
ctx4.pt_iterate(5)
t0 = time.perf_counter()
ctx4.pt_iterate(100)
dt_pt = time.perf_counter() - t0
out["config4_carma76_n10000"] = {
    "batch_1024_evals_seconds": dt_eval, "evals_per_s": 1024 / dt_eval, "finite": int(np.isfinite(ld).sum()),
    "logdensity_at_truth": float(ctx4.logdensity(th4)),
    "pt_8temps_128replicas_iters_per_s": 100 / dt_pt, "pt_chain_evals_per_s": 100 * 8 * 128 / dt_pt,
}
comm = _lib.Comm(_lib.Comm.unique_id(), 1, 0, device=0)
temps = par.ladder_temperatures(8)
for blocks in ([4, 4], [1] * 8):
    ctxs, slot0 = [], 0
    for Tl in blocks:
        c = cpa.Context(t4, y4, e4, 7, 6)
        c.pt_create(Tl, 128, 10 ** 6, seed=5, temperatures=temps[slot0:slot0 + Tl])
        c.pt_shard(8, slot0, 0)
        c.pt_start(None)
        ctxs.append(c)
        slot0 += Tl
    _lib.pt_iterate_sharded(ctxs, 4, comm)
    t0 = time.perf_counter()
    _lib.pt_iterate_sharded(ctxs, 40, comm)
    dts = time.perf_counter() - t0
    # the blocks' kernels alone, one after the other (what the sharded run costs without any exchange)
    t0 = time.perf_counter()
    for _ in range(40):
        for c in ctxs:
            c.pt_iterate(1)
    dta = time.perf_counter() - t0
    out["config4_carma76_n10000"]["sharded_%s" % "x".join(map(str, blocks))] = {
        "iters_per_s": 40 / dts, "ms_per_iteration": 1e3 * dts / 40,
        "ms_per_iteration_blocks_alone_with_a_host_sync_each": 1e3 * dta / 40,
        "boundary_swaps": [c.pt_boundary_stats() for c in ctxs]}
comm.close()

# ---- config 5: OGLE-LMC-LPV-00007, all (p,q), 100 prior-like thetas each, ignore_prior ---------------
og = np.loadtxt(os.path.join(G, "ogle_lmc_lpv_00007.dat"))
to, yo, eo = og[:, 0], og[:, 1], og[:, 2]
rng = np.random.default_rng(5)
ctxs, ths = {}, {}
for p in range(1, 8):
    for q in range(p):
        ctxs[(p, q)] = cpa.Context(to, yo, eo, p, q)
        ths[(p, q)] = np.array([prior_like_theta(rng, p, q, to, yo) for _ in range(100)])
        ctxs[(p, q)].logdensity(ths[(p, q)][:2], ignore_prior=True)
t0 = time.perf_counter()
nfin = 0
for k in ctxs:
    nfin += int(np.isfinite(ctxs[k].logdensity(ths[k], ignore_prior=True)).sum())
dt5 = time.perf_counter() - t0
model = cpa.CarmaModel(to, yo, eo)
t0 = time.perf_counter()
best, pqlist, aicc = model.choose_order(7, ntrials=100, seed=1)
dt_co = time.perf_counter() - t0
out["config5_ogle_grid"] = {
    "pairs": len(ctxs), "evals": 100 * len(ctxs), "seconds": dt5, "evals_per_s": 100 * len(ctxs) / dt5, "finite": nfin,
    "choose_order_pmax7_ntrials100_seconds": dt_co, "chosen_pq": [model.p, model.q],
    "aicc": dict(("%d,%d" % pq, a) for pq, a in zip(pqlist, aicc)),
}
print(json.dumps(out, indent=1))
