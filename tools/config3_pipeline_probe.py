#!/usr/bin/env python3
"""Round 6: BASELINE configs[3] -- CARMA(7,6), n = 10^4, 8 temperatures x 128 ladders on one GPU -- under the
one-datum pipeline (PT_ROW_WIN = 0) and the one-sided window pipeline (1), with prior-like chains (right after the start) and with
chains that have run 240 iterations.  carma_tune_set moves the switch between timed calls of the SAME sampler state."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import carma_pack_amd as cpa
from carma_pack_amd.synth import config4_series
from carma_pack_amd import parallel as par
t4, y4, e4, _ = config4_series(10000, seed=4)
res = []
for warm in (0, 240):
    for win in (0, 1):
        ctx = cpa.Context(t4, y4, e4, 7, 6)
        cpa._lib.tune_set("PT_ROW_WIN", win)
        ctx.pt_create(8, 128, 10 ** 9, seed=17, temperatures=par.ladder_temperatures(8))
        ctx.pt_shard(8, 0, 0)
        ctx.pt_start(None)
        cpa._lib.tune_set("PT_ROW_WIN", 0)
        if warm:
            ctx.pt_iterate(warm)
        cpa._lib.tune_set("PT_ROW_WIN", win)
        ctx.pt_iterate(5)
        t0 = time.perf_counter()
        ctx.pt_iterate(60)
        dt = time.perf_counter() - t0
        r = dict(chains="prior-like" if warm == 0 else "after %d iterations" % warm, pipeline="window" if win else "one-datum", it_per_s=round(60 / dt, 1), kernel=ctx.pt_kernel())
        res.append(r)
        print(json.dumps(r), flush=True)
        del ctx
cpa._lib.tune_reset()
