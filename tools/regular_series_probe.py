"""Throughput-regime launches on a REGULARLY sampled series (constant dt, two gaps) against an irregular one of the same
length: the G-lane kernels skip the exp/sincos evaluation of a step whose dt repeats."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
rng = np.random.default_rng(1)
n = 270
t_irr = np.cumsum(rng.uniform(1.0, 3.0, n))
t_reg = 2.0 * np.arange(n, dtype=float); t_reg[100:] += 37.0; t_reg[200:] += 11.0
dev = torch.device('cuda'); st = torch.cuda.current_stream().cuda_stream
for name, t in (("irregular", t_irr), ("regular", t_reg)):
    y = 17.0 + np.sin(t / 7.0) + 0.3 * rng.standard_normal(n); e = np.full(n, 0.3)
    ctx = cpa.Context(t, y, e, 5, 3, max_stdev=10 * y.std())
    base = theta_batch(np.random.default_rng(2), 1024, 5, 3, t, y)
    out_s = []
    for B in (1024, 4096, 8192, 65536):
        th = torch.from_numpy(np.tile(base, (B // 1024 + 1, 1))[:B].copy()).to(dev)
        out = torch.empty(B, dtype=torch.float64, device=dev)
        reps = 200 if B < 20000 else 30
        for _ in range(3): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        out_s.append("%d: %.1f us" % (B, dt * 1e6))
    print("%-10s" % name, " | ".join(out_s), flush=True)
