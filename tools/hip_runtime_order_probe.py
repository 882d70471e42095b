"""Does the sampler (cooperative launch of k_pt_row) work whatever the import order of torch and this package?
usage: python tools/hip_runtime_order_probe.py none|torch_first|torch_after|torch_after_used   (round 2: two HIP runtimes
in one process -- this package mapped first, then `import torch` -- made every cooperative launch fail; _lib.py now maps
PyTorch's runtime first whenever PyTorch is installed)"""
import os, sys
sys.path.insert(0, '.')
mode = sys.argv[1]
if mode == "torch_first":
    import torch
import numpy as np
import carma_pack_amd as cpa
if mode == "torch_after":
    import torch
if mode == "torch_after_used":
    import torch
    torch.zeros(4, device="cuda").sum().item()
g = np.load('tests/golden/carma53_readme.npz')
t, y, e = g['t'], g['y'], g['yerr']
for (p, q, T, R) in ((1, 0, 1, 2), (3, 1, 10, 4), (5, 3, 16, 64)):
    ctx = cpa.Context(t, y, e, p, q)
    try:
        ctx.pt_create(T, R, 100, seed=1)
        ctx.pt_start(None)
        ctx.pt_iterate(50)
        print(mode, p, q, T, R, "ok", ctx.pt_iterations_done(), flush=True)
    except Exception as ex:
        print(mode, p, q, T, R, "FAILED", ex, flush=True)
