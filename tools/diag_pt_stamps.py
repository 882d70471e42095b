"""Per-phase cycle stamps of one PT iteration (diagnostic -DCARMA_STAMPS build, tools/build_diag.sh)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import carma_pack_amd._lib as L0
L0.LIB_PATH = os.path.join(ROOT, "build_diag", "libcarma_mi355_diag.so")
L0.lib = L0._load()
if len(sys.argv) > 1 and sys.argv[1] == "config4":          # BASELINE configs[3]: CARMA(7,6), n = 10^4, 8 temperatures x 128 ladders
    from carma_pack_amd.synth import config4_series
    from carma_pack_amd import parallel as par
    t, y, yerr, _ = config4_series(10000, seed=4)
    ctx = L0.Context(t, y, yerr, 7, 6)
    T_, Rs, temps = 8, (128,), par.ladder_temperatures(8)
else:
    g = np.load(os.path.join(ROOT, 'tests/golden/carma53_readme.npz'))
    t, y, yerr = g['t'], g['y'], g['yerr']
    ctx = L0.Context(t, y, yerr, 5, 3, max_stdev=10*np.sqrt(np.mean(y*y)-np.mean(y)**2))
    T_, Rs, temps = 16, (64,), None
    if len(sys.argv) > 2:                                     # diag_pt_stamps.py T R: e.g. 10 1, the README's run_mcmc ladder
        T_, Rs = int(sys.argv[1]), (int(sys.argv[2]),)
for R in Rs:
    print("--- %d temperatures x %d replicas" % (T_, R), flush=True)
    ctx.pt_create(T_, R, adapt_iters=10**9, seed=17, temperatures=temps)
    ctx.pt_shard(T_, 0, 0)
    ctx.pt_start(None)
    ctx.pt_iterate(50)
    ctx.pt_iterate(3)
print("done")
