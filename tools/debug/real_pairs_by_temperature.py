import os, sys, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import carma_pack_amd as cpa
g = np.load(os.path.join(ROOT, 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
ms = 10 * np.sqrt(np.mean(y * y) - np.mean(y) ** 2)
ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=ms)
for R, it in ((4096, 300), (4096, 1500)):
    ctx.pt_create(16, R, adapt_iters=10 ** 9, seed=3); ctx.pt_start(None); ctx.pt_iterate(it)
    th, lp = ctx.pt_get_chains()
    th = np.asarray(th).reshape(R, 16, -1)
    real = np.zeros((R, 16), bool)
    for i in range(2):
        a, b = np.exp(th[:, :, 3 + 2 * i]), np.exp(th[:, :, 4 + 2 * i])
        real |= b * b > 4 * a
    print("after %d iterations: fraction of chains with a real pair, per temperature (cold -> hot):" % it)
    print(" ".join("%.2f" % v for v in real.mean(axis=0)), " overall %.2f" % real.mean())
    # waves of 64 consecutive ladders at one temperature: fraction of waves with NO real pair
    w = real.reshape(R // 64, 64, 16).any(axis=1)
    print("temperature-major waves without any real pair, per temperature:", " ".join("%.2f" % v for v in (1 - w.mean(axis=0))))
