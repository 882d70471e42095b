"""Boundary budget of the sharded ladder (BASELINE configs[3] shape: CARMA(7,6), n = 10 000, 8 temperatures), on ONE GPU:
the ladder as blocks of this process, the boundary chains travelling through carma_pt_iterate_sharded's RCCL send/recv to
the process's own rank.  CARMA_SHARD_STAMPS=1 makes the library print HIP-event times per stage (sampler kernel, pack,
send/recv, swap, block sweep); this script adds the iterations/s per partition and the 8-rank prediction of DESIGN.md
section 7's pipelining argument:  t_iteration(8 ranks, steady state) = t_sampler(1 temperature) + 2 x (pack + send/recv + swap).
    python tools/shard_budget_probe.py [R] [iterations]"""
import os, sys, time
os.environ["CARMA_SHARD_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import carma_pack_amd as cpa
from carma_pack_amd import _lib, parallel as par
from carma_pack_amd.synth import config4_series

R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t, y, e, _ = config4_series(10000, seed=4)
T = 8
temps = par.ladder_temperatures(T)
comm = _lib.Comm(_lib.Comm.unique_id(), 1, 0, device=0)
for blocks in ([8], [4, 4], [2, 2, 2, 2], [1] * 8):
    ctxs, slot0 = [], 0
    for Tl in blocks:
        c = cpa.Context(t, y, e, 7, 6)
        c.pt_create(Tl, R, 10 ** 6, seed=91, temperatures=temps[slot0:slot0 + Tl])
        c.pt_shard(T, slot0, 0)
        c.pt_start(None)
        ctxs.append(c)
        slot0 += Tl
    _lib.pt_iterate_sharded(ctxs, 4, comm if len(blocks) > 1 else None)
    sys.stderr.flush()
    t0 = time.perf_counter()
    _lib.pt_iterate_sharded(ctxs, IT, comm if len(blocks) > 1 else None)
    dt = time.perf_counter() - t0
    print("blocks %-26s R = %d: %.1f it/s, %.2f ms per iteration (kernel of block 0: %s)" % (blocks, R, IT / dt, 1e3 * dt / IT, ctxs[0].pt_kernel()), flush=True)
    del ctxs
comm.close()
