"""Diagnostic: config-2 sampler run, stored log-posterior against the oracle / quad value; dumps the offending states and
re-evaluates them through every launch shape."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import carma_pack_amd as cpa
import oracle as orc
g = np.load("tests/golden/carma53_readme.npz")
t, y, yerr = g["t"], g["y"], g["yerr"]
ms = float(np.sqrt(np.mean(y * y) - np.mean(y) ** 2)) * float(os.environ.get("MSF", "10"))
ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=ms)
samples, lp = ctx.pt_run(16, 64, 50000, 25000, 1, seed=2024)
m = orc.OracleModel(t, y, yerr, 5, 3, max_stdev=ms)
sub = samples[:, ::2503].reshape(-1, 11); l = lp[:, ::2503].reshape(-1)
want = m.logdensity_batch(sub, nthreads=8)
rel = np.abs(l - want) / np.abs(want)
bad = np.flatnonzero(rel > 1e-10)
print("entries differing from the oracle by > 1e-10:", bad, rel[bad])
nf = np.flatnonzero(~np.isfinite(want))
print("oracle not finite at", nf)
for i in nf[:3]:
    th = sub[i]
    print("theta", repr(th), "stored", l[i], "oracle", want[i], "oracle ign", m.logdensity(th, ignore_prior=True), "logprior", m.log_prior(th))
    print("truth", orc.truth_logdensity(t, y, yerr, th, 5, 3), "roots", orc.ar_roots(th, 5))
for i in bad:
    th = sub[i]
    tr = orc.truth_logdensity(t, y, yerr, th, 5, 3)[0]
    one = ctx.logdensity(th[None, :])[0]
    pc = ctx.logdensity(np.tile(th, (3200, 1)))[0]
    pl = ctx.logdensity(np.tile(th, (70000, 1)))[0]
    print("theta", repr(th))
    print("stored %.15g  oracle %.15g  truth %.15g | p3l %.15g pc %.15g plain %.15g" % (l[i], want[i], tr, one, pc, pl))
    print("errors vs truth: stored %.2e oracle %.2e p3l %.2e pc %.2e plain %.2e" % tuple(abs(v - tr) / abs(tr) for v in (l[i], want[i], one, pc, pl)))
    print("roots", orc.ar_roots(th, 5))
    np.save("gpurun_out/bad_theta_%d.npy" % i, th)
