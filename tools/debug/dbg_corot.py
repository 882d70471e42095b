import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import carma_pack_amd as cpa
import oracle as orc
from carma_pack_amd.synth import log_quads_from_roots as lq
for p, q in ((6, 2), (5, 3), (7, 5)):
    rng = np.random.default_rng(900 + 10 * p + q)
    n = 150
    dt = rng.uniform(0.5, 2.0, n)
    dt[rng.integers(5, n - 5, 8)] = 10.0 ** rng.uniform(1.0, 5.0, 8)
    t = np.cumsum(dt)
    y = 3.0 + np.sin(t / 3.0) + 0.3 * rng.standard_normal(n)
    yerr = np.full(n, 0.3) * rng.uniform(0.7, 1.3, n)
    ctx = cpa.Context(t, y, yerr, p, q)
    m = orc.OracleModel(t, y, yerr, p, q, max_stdev=ctx.prior()[0])
    ths = []
    for _ in range(64):
        width = 10.0 ** rng.uniform(-4.0, 2.0, (p + 1) // 2)
        cent = 10.0 ** rng.uniform(-4.0, 2.0, p // 2)
        cent = np.maximum(cent, 1e-2 * width[:p // 2])
        roots = []
        for i in range(p // 2):
            roots += [complex(-width[i], -cent[i]), complex(-width[i], cent[i])]
        if p % 2:
            roots.append(complex(-width[-1], 0.0))
        ma = rng.normal(0.0, 1.0, q)
        ths.append(np.concatenate([[rng.uniform(0.5, 3.0), rng.uniform(0.6, 1.8), rng.normal(3.0, 0.3)], lq(roots), ma]))
    th = np.array(ths)
    want = m.logdensity_batch(th, ignore_prior=True, nthreads=8)
    for mode, (k, v) in (("default", (None, None)), ("one-datum", ("WIN_ROWS", 0)), ("one-sided window", ("WIN2_EVALS", 0))):
        cpa._lib.tune_reset()
        if k: cpa._lib.tune_set(k, v)
        if mode == "one-sided window": cpa._lib.tune_set("WIN_ROWS", 1 << 20)
        got = ctx.logdensity(th, ignore_prior=True)
        bad = np.flatnonzero(~np.isfinite(got) & np.isfinite(want))
        print(p, q, mode, ctx.kernel_name(64), "nan where the oracle is finite:", len(bad), bad[:12].tolist(), "max rel", np.nanmax(np.abs(got - want) / np.abs(want)))
    cpa._lib.tune_reset()
    if p == 6:
        for k in bad[:3]:
            print("  theta", k, th[k].tolist())
