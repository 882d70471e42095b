"""Per-phase cycle stamps of one PT iteration (diagnostic -DCARMA_STAMPS build, tools/build_diag.sh)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import carma_pack_amd._lib as L0
L0.LIB_PATH = os.path.join(ROOT, "build_diag", "libcarma_mi355_diag.so")
L0.lib = L0._load()
g = np.load(os.path.join(ROOT, 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
ctx = L0.Context(t, y, yerr, 5, 3, max_stdev=10*np.sqrt(np.mean(y*y)-np.mean(y)**2))
for R in (64,):
    print("--- 16 temperatures x %d replicas" % R, flush=True)
    ctx.pt_create(16, R, adapt_iters=10**9, seed=3)
    ctx.pt_shard(16, 0, 0)
    ctx.pt_start(None)
    ctx.pt_iterate(50)
    ctx.pt_iterate(3)
print("done")
