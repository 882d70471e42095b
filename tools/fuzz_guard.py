"""Out-of-bounds probe of the batched log-density kernels: the parameter batch and the output array are placed at the very END
of device allocations of their own (16 MiB segments of the caching allocator: what follows is unmapped, or at least not
ours), so that a kernel reading a parameter past theta[B d] or writing past out[B] faults instead of getting away with it --
the staging buffers of the host path end on a page boundary only by accident (once in 9000 fuzz cases: CARMA(5,0), 16 384
evaluations, round 4).  Every order, q = 0 / 1 / p - 1, batch sizes across all launch shapes, a series long enough for the
producer-wave kernels and one too short for them.  A fault ends the process: run it as one (the GPU suite does).
    python tools/fuzz_guard.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import carma_pack_amd as cpa
from helpers import irregular_series, prior_like_theta

dev = torch.device("cuda", 0)
SEG = 16 * 2 ** 20 // 8                                       # doubles in a segment of its own (>= 10 MiB: not shared with other tensors)
stream = torch.cuda.current_stream().cuda_stream
ncase = 0
for p in range(1, 8):
    for q in sorted({0, 1, p - 1} & set(range(p))) if p > 1 else [0]:
        for n in (6, 40):
            t, y, yerr = irregular_series(n, seed=10 * p + q)
            ctx = cpa.Context(t, y, yerr, p, q)
            d = 4 if p == 1 else 3 + p + q
            rng = np.random.default_rng(100 * p + q)
            base = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(29)])
            for B in (1, 3, 4, 1023, 1024, 3072, 3073, 4096, 8192, 8193, 16384, 16385, 24576, 28672, 32768, 49152, 49153, 65536, 70001):
                th = np.tile(base, (B // 29 + 1, 1))[:B]
                want = ctx.logdensity(th)                    # host path (staging buffers)
                pool_t = torch.empty(SEG, dtype=torch.float64, device=dev)
                pool_o = torch.empty(SEG, dtype=torch.float64, device=dev)
                tv = pool_t[SEG - B * d:]
                ov = pool_o[SEG - B:]
                tv.copy_(torch.from_numpy(th.reshape(-1)).to(dev))
                ctx.logdensity_dev(tv.data_ptr(), B, ov.data_ptr(), stream=stream)
                torch.cuda.synchronize()
                got = ov.cpu().numpy()
                assert np.array_equal(got, want, equal_nan=True), (p, q, n, B, ctx.kernel_name(B))
                del pool_t, pool_o, tv, ov
                torch.cuda.empty_cache()                      # (the segments go back: the next case gets fresh ones)
                ncase += 1
print("%d cases, no fault, device-resident and staged batches agree bit for bit: ok" % ncase)
