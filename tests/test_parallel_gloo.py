"""world_size-2 gloo tests (CPU) of the N>1 path: batch sharding with its all-gather, and the
ladder-sharded parallel tempering with point-to-point boundary swaps.  The GPU compute backend is
replaced by a CPU stand-in with the same pt_* interface (the product code under test is
carma_pack_amd/parallel.py; the kernels themselves are covered by the -m gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from carma_pack_amd import parallel as par


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_philox_matches_device_rng():
    """The Python Philox used for cross-rank swap decisions == carma_rng.h (via the lane emulator)."""
    import emu_build as emu
    _, u = emu.rng_draws(seed=0x1234ABCD5678, chain=37, n=50)
    mine = [par.philox_uniform(0x1234ABCD5678, 37, i, par.RNG_ACCEPT, 0) for i in range(50)]
    assert np.array_equal(u, np.array(mine))


def test_vectorised_philox_is_the_scalar_one():
    chains = np.array([0, 1, 37, 2 ** 31 + 5, 4000000000])
    for it in (0, 5, 2 ** 33 + 7):
        a = par.philox_uniform_chains(0x1234ABCD5678, chains, it, par.RNG_SWAP, 3)
        b = np.array([par.philox_uniform(0x1234ABCD5678, int(c), it, par.RNG_SWAP, 3) for c in chains])
        assert np.array_equal(a, b)


def test_shard_slice_covers_everything():
    for n in (0, 1, 7, 1024, 1025):
        for w in (1, 2, 3, 8):
            idx = np.concatenate([np.arange(n)[par.shard_slice(n, r, w)] for r in range(w)])
            assert np.array_equal(idx, np.arange(n))


class ToyBackend(object):
    """pt_* interface on the CPU: random-walk Metropolis on a tempered standard normal; the last component of theta is
    an immutable chain tag so that swaps can be audited.  Like the device sampler, every draw is a counter-based
    function of (seed, GLOBAL chain slot, iteration) -- so a ladder or a set of replicas walks the same trajectory
    however it is split over ranks, which is what the tests below pin."""

    def __init__(self, d):
        self.d = d

    @staticmethod
    def target(th):
        return -0.5 * np.sum(th[..., :-1] ** 2, axis=-1)

    def pt_create(self, T, R, adapt_iters, seed=0, temperatures=None):
        self.T, self.R, self.seed, self.it = T, R, int(seed), 0
        self.temps = np.asarray(temperatures, dtype=float) if temperatures is not None else par.ladder_temperatures(T)
        self.T_global, self.slot0, self.replica0 = T, 0, 0
        self.nswap = np.zeros((R, T), dtype=int)

    def pt_shard(self, T_global, slot0, replica0):
        self.T_global, self.slot0, self.replica0 = T_global, slot0, replica0

    def _slots(self):
        return ((self.replica0 + np.arange(self.R))[:, None] * self.T_global + self.slot0 + np.arange(self.T)[None, :]).ravel()

    def _normal(self, it, j):
        g = self._slots()
        u1 = par.philox_uniform_chains(self.seed, g, it, par.RNG_PROPOSAL, 2 * j)
        u2 = par.philox_uniform_chains(self.seed, g, it, par.RNG_PROPOSAL, 2 * j + 1)
        return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).reshape(self.R, self.T)

    def pt_start(self, init=None):
        self.th = np.stack([self._normal(2 ** 40, j) for j in range(self.d)], axis=-1)
        self.th[..., -1] = 1000 * (self.replica0 + np.arange(self.R))[:, None] + self.slot0 + np.arange(self.T)[None, :]   # tag
        self.lp = self.target(self.th)

    def pt_get_chains(self):
        return self.th.copy(), self.lp.copy()

    def pt_set_chains(self, th, lp=None):
        self.th, self.lp = th.copy(), lp.copy()

    def pt_iterate(self, n, do_exchange=True):
        for _ in range(n):
            prop = self.th.copy()
            for j in range(self.d - 1):
                prop[..., j] += 0.5 * self._normal(self.it, j)
            lpn = self.target(prop)
            u = par.philox_uniform_chains(self.seed, self._slots(), self.it, par.RNG_ACCEPT).reshape(self.R, self.T)
            acc = np.log(u) < (lpn - self.lp) / self.temps[None, :]
            self.th[acc], self.lp[acc] = prop[acc], lpn[acc]
            if do_exchange:                                   # the kernels' sweep: hot -> cold over all adjacent pairs
                for i in range(self.T - 1, 0, -1):
                    g = (self.replica0 + np.arange(self.R)) * self.T_global + self.slot0 + i
                    lu = np.log(par.philox_uniform_chains(self.seed, g, self.it, par.RNG_SWAP))
                    sw = lu < (self.lp[:, i - 1] - self.lp[:, i]) * (1.0 / self.temps[i] - 1.0 / self.temps[i - 1])
                    self.th[sw, i], self.th[sw, i - 1] = self.th[sw, i - 1].copy(), self.th[sw, i].copy()
                    self.lp[sw, i], self.lp[sw, i - 1] = self.lp[sw, i - 1].copy(), self.lp[sw, i].copy()
                    self.nswap[sw, i] += 1
            self.it += 1

    def pt_sample(self, nsamples, thin=1):
        s, l = np.empty((self.R, nsamples, self.d)), np.empty((self.R, nsamples))
        for k in range(nsamples):
            self.pt_iterate(thin, do_exchange=True)
            s[:, k], l[:, k] = self.th[:, 0], self.lp[:, 0]
        return s, l


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # ---- batch sharding + all_gather -----------------------------------------------------
        rng = np.random.default_rng(5)
        thetas = rng.standard_normal((37, 4))
        f = lambda x: np.sum(x ** 2, axis=1) + 0.25      # noqa: E731
        out = par.sharded_logdensity(f, thetas, dist)
        assert np.allclose(out, f(thetas))
        # ---- ladder sharding -----------------------------------------------------------------
        Tg, R, d = 5, 6, 4
        sh = par.LadderShard(ToyBackend(d), Tg, R, adapt_iters=0, seed=99, dist=dist)
        assert sh.T_local == (3 if rank == 0 else 2) and sh.slot0 == (0 if rank == 0 else 3)
        sh.start()
        # the same ladder in ONE process (every rank runs its own copy): the sharded ladder must walk its trajectory
        one = ToyBackend(d)
        one.pt_create(Tg, R, 0, seed=99, temperatures=par.ladder_temperatures(Tg))
        one.pt_start()
        for _ in range(60):
            sh.iterate(1)
            one.pt_iterate(1, do_exchange=True)
            th, lp = sh.b.pt_get_chains()
            assert np.allclose(lp, ToyBackend.target(th))              # swapped log-posteriors travel with theta
            sl = slice(sh.slot0, sh.slot0 + sh.T_local)
            assert np.array_equal(th, one.th[:, sl]) and np.array_equal(lp, one.lp[:, sl]), "sharded != unsharded"
            dist.all_gather_object(obj := [None] * world, th[:, :, -1].copy())
            tagsets = np.concatenate(obj, axis=1)                      # [R][Tg]
            for r in range(R):
                assert sorted(tagsets[r].tolist()) == [1000 * r + c for c in range(Tg)], "chain lost or duplicated"
        moved = int(np.sum(th[:, :, -1] != (1000 * np.arange(R)[:, None] + sh.slot0 + np.arange(sh.T_local)[None, :])))
        # ---- replica sharding: independent ladders split by rank, coldest chains gathered -------------------------
        Rt, S = 7, 11
        gs, gl = par.sharded_pt_run(lambda: ToyBackend(d), Tg, Rt, S, burnin=20, thin=2, seed=5, dist=dist)
        ref = ToyBackend(d)
        ref.pt_create(Tg, Rt, 20, seed=5)
        ref.pt_start()
        ref.pt_iterate(20)
        rs, rl = ref.pt_sample(S, 2)
        assert gs.shape == (Rt, S, d) and np.array_equal(gs, rs) and np.array_equal(gl, rl), "gathered != single process"
        q.put((rank, sh.nswap_boundary, sh.nprop_boundary, moved))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get() for _ in range(world))
    # both sides of the single boundary took the same decisions
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]
    assert res[0][1] > 0 and res[0][2] == 6 * 60      # the boundary pair is proposed every iteration, like every other pair
