"""Out-of-bounds probe of the sampler kernels (companion of tools/fuzz_guard.py): the chain state theta [R][T][d] and the
log-posteriors [R][T] are caller-owned device buffers (carma_pt_bind_state) placed at the very END of 16 MiB allocations of
their own; a sampler kernel reading or writing past them faults.  The same seed without bound buffers must give the same
chains.  One sampler path per process (CARMA_PT_KERNEL is read once):
    python tools/fuzz_guard_sampler.py row|ladder|lane"""
import os, sys
kern = sys.argv[1]
os.environ["CARMA_PT_KERNEL"] = kern
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import carma_pack_amd as cpa
from helpers import irregular_series

dev = torch.device("cuda", 0)
SEG = 16 * 2 ** 20 // 8
ncase = 0
SHAPES = {"row": ((5, 3, 16, 64), (7, 6, 8, 96), (2, 0, 5, 40), (3, 1, 9, 33)),
          "ladder": ((5, 3, 16, 70), (7, 6, 8, 128), (2, 1, 3, 500), (4, 0, 20, 61)),
          "lane": ((5, 3, 16, 600), (7, 6, 8, 1100), (2, 0, 1, 9000), (3, 2, 33, 300), (6, 0, 12, 700))}[kern]
for (p, q, T, R) in SHAPES:
    t, y, yerr = irregular_series(60, seed=7 * p + q)
    d = 3 + p + q
    runs = []
    for bound in (False, True):
        ctx = cpa.Context(t, y, yerr, p, q)
        try:
            ctx.pt_create(T, R, adapt_iters=30, seed=5)
        except ValueError as ex:
            print("CARMA(%d,%d) T=%d R=%d: rejected (%s)" % (p, q, T, R, str(ex)[-50:]))
            runs = None
            break
        if bound:
            pool_t = torch.zeros(SEG, dtype=torch.float64, device=dev)
            pool_l = torch.zeros(SEG, dtype=torch.float64, device=dev)
            tv, lv = pool_t[SEG - R * T * d:], pool_l[SEG - R * T:]
            ctx.pt_bind_state(tv.data_ptr(), lv.data_ptr())
        ctx.pt_start(None)
        ctx.pt_iterate(45)
        smp, slp = ctx.pt_sample(6, thin=2)
        torch.cuda.synchronize()
        th, lp = ctx.pt_get_chains()
        if bound:
            assert np.array_equal(tv.cpu().numpy().reshape(R, T, d), th) and np.array_equal(lv.cpu().numpy().reshape(R, T), lp, equal_nan=True)
        runs.append((th, lp, smp, slp, ctx.pt_kernel()))
        del ctx
    if runs is None:
        continue
    a, b = runs
    assert a[4] == b[4] == kern, (a[4], b[4])
    for u, v in zip(a[:4], b[:4]):
        assert np.array_equal(u, v, equal_nan=True), (p, q, T, R)
    ncase += 1
    torch.cuda.empty_cache()
print("%s: %d shapes, no fault, bound and library-owned state walk the same chains: ok" % (kern, ncase))
