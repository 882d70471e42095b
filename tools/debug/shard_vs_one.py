"""Where does a sharded ladder (blocks in one process, boundaries through RCCL to self) leave the unsharded trajectory?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import carma_pack_amd as cpa
from carma_pack_amd import _lib, parallel as par
P, Q, TG, R, SEED = 3, 1, 5, 6, 4242
rng = np.random.default_rng(11)
n = 80
t = np.cumsum(rng.uniform(1.0, 3.0, n))
y = np.cumsum(rng.standard_normal(n)) * 0.3 + 0.2 * rng.standard_normal(n)
y = y - y.mean(); e = np.full(n, 0.2)
temps = par.ladder_temperatures(TG)
def make(blocks):
    ctxs, slot0 = [], 0
    for Tl in blocks:
        c = cpa.Context(t, y, e, P, Q, max_stdev=10.0 * y.std())
        c.pt_create(Tl, R, 40, seed=SEED, temperatures=temps[slot0:slot0 + Tl])
        c.pt_shard(TG, slot0, 0)
        c.pt_start(None)
        ctxs.append(c); slot0 += Tl
    return ctxs
one = make([TG])[0]
blk = make([3, 2])
comm = _lib.Comm(_lib.Comm.unique_id(), 1, 0, device=0)
def state(cs):
    th = np.concatenate([c.pt_get_chains()[0] for c in cs], axis=1); lp = np.concatenate([c.pt_get_chains()[1] for c in cs], axis=1)
    return th, lp
a, b = state([one]), state(blk)
print("start equal:", np.array_equal(a[0], b[0]), np.array_equal(a[1], b[1]))
for it in range(6):
    one.pt_iterate(1)
    _lib.pt_iterate_sharded(blk, 1, comm)
    a, b = state([one]), state(blk)
    print("after iteration", it, "theta equal:", np.array_equal(a[0], b[0]), "lp equal:", np.array_equal(a[1], b[1]),
          "differing chains (replica, temp):", np.argwhere(a[1] != b[1])[:8].tolist())
# RAM only
one2, blk2 = make([TG])[0], make([3, 2])
one2.pt_iterate(1, do_exchange=False)
for c in blk2: c.pt_iterate(1, do_exchange=False)
a, b = state([one2]), state(blk2)
print("RAM only: equal", np.array_equal(a[0], b[0]), np.array_equal(a[1], b[1]), np.argwhere(a[1] != b[1])[:8].tolist())
