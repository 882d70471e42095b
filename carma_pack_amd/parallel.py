"""One-process-per-GPU sharding of the path over ``torch.distributed`` (backend "nccl" == RCCL over
xGMI on the GPU box, "gloo" in the CPU test-suite).

The path shards naturally: log-density evaluations and independent replicas need no
data-path collective at all, and the only coupling inside one temperature ladder is the
adjacent-temperature exchange (ExchangeStep, /root/reference/src/include/steps.hpp:318-362).

  * ``shard_slice`` / ``sharded_logdensity`` -- split a batch of parameter vectors by rank, evaluate
    locally, all-gather the results (the only collective; B doubles).
  * ``LadderShard`` -- a ladder of T temperatures split into contiguous blocks, one block per rank
    (BASELINE config 4: 8 temperatures on 8 GPUs).  Every rank advances its block with the
    persistent kernel (local swaps inside the block included); then the chains on either side of a
    rank boundary are exchanged point-to-point: (theta[d], logpost) for all R replicas, R*(d+1)
    doubles each way (<= 20 KB: latency bound, never link bound).  Both sides compute the SAME
    Metropolis decision from a counter-based uniform keyed by the hotter chain's global slot
    (``philox_uniform`` reproduces carma_rng.h bit for bit), so no decision is communicated.
    An iteration is the one-GPU sampler's: RAM steps everywhere, then ONE sweep over all adjacent pairs
    from the hottest to the coldest -- boundary with the hotter rank, the pairs inside the block,
    boundary with the colder rank -- so the sharded ladder walks the unsharded ladder's trajectory.
  * ``sharded_pt_run`` -- independent replicas (whole ladders) split by rank, no data-path collective
    during sampling; the coldest chains are gathered to rank 0 at the end (the reference's driver
    returns the coldest chain, src/carmcmc.cpp:164-176, src/samplers.cpp:118-124).
"""
import numpy as np

RNG_PROPOSAL, RNG_ACCEPT, RNG_SWAP = 0, 1, 2
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
_MASK = 0xFFFFFFFF


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Scalar Philox4x32-10, identical to carma_pack_amd/csrc/carma_rng.h."""
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> 32, p0 & _MASK, p1 >> 32, p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & _MASK, lo1, (hi0 ^ c3 ^ k1) & _MASK, lo0
        k0, k1 = (k0 + _W0) & _MASK, (k1 + _W1) & _MASK
    return c0, c1, c2, c3


def philox_uniform(seed, chain, iteration, purpose, idx=0):
    """rng_uniform(key{seed, chain}, iteration, purpose, idx) of carma_rng.h -> float in (0,1)."""
    x = philox4x32_10(iteration & _MASK, (iteration >> 32) & _MASK, chain & _MASK, ((purpose << 24) | idx) & _MASK,
                      seed & _MASK, (seed >> 32) & _MASK)
    k = ((x[0] << 32) | x[1]) >> 11
    return (k + 0.5) / 9007199254740992.0


def philox_uniform_chains(seed, chains, iteration, purpose, idx=0):
    """`philox_uniform` for an array of chain ids at once (numpy uint64 arithmetic, same bits)."""
    m = np.uint64(_MASK)
    c0 = np.full(len(chains), iteration & _MASK, dtype=np.uint64)
    c1 = np.full(len(chains), (iteration >> 32) & _MASK, dtype=np.uint64)
    c2 = np.asarray(chains, dtype=np.uint64) & m
    c3 = np.full(len(chains), ((purpose << 24) | idx) & _MASK, dtype=np.uint64)
    k0, k1 = seed & _MASK, (seed >> 32) & _MASK
    s32 = np.uint64(32)
    for _ in range(10):
        p0, p1 = np.uint64(_M0) * c0, np.uint64(_M1) * c2          # 32 x 32 -> 64 bits, no overflow
        c0, c1, c2, c3 = ((p1 >> s32) ^ c1 ^ np.uint64(k0)) & m, p1 & m, ((p0 >> s32) ^ c3 ^ np.uint64(k1)) & m, p0 & m
        k0, k1 = (k0 + _W0) & _MASK, (k1 + _W1) & _MASK
    k = ((c0 << s32) | c1) >> np.uint64(11)
    return (k.astype(np.float64) + 0.5) / 9007199254740992.0


def shard_slice(n, rank, world):
    """Contiguous, balanced split of range(n)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return slice(lo, lo + base + (1 if rank < rem else 0))


def sharded_logdensity(evaluate, thetas, dist=None, device="cpu"):
    """Every rank evaluates its slice of `thetas` ([B][d], identical on all ranks) with
    `evaluate(local_thetas) -> [b]` (normally ``Context.logdensity``) and all ranks get the full [B]."""
    import torch
    thetas = np.ascontiguousarray(thetas, dtype=np.float64)
    B = thetas.shape[0]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(evaluate(thetas))
    rank, world = dist.get_rank(), dist.get_world_size()
    sl = shard_slice(B, rank, world)
    local = np.asarray(evaluate(thetas[sl]), dtype=np.float64)
    width = (B + world - 1) // world
    buf = torch.full((width,), float("nan"), dtype=torch.float64, device=device)
    buf[:local.size] = torch.from_numpy(local).to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = np.empty(B)
    for r in range(world):
        s = shard_slice(B, r, world)
        out[s] = parts[r][:s.stop - s.start].cpu().numpy()
    return out


def ladder_temperatures(ntemps, tmax=100.0):
    """T_i = tmax^(i/(ntemps-1)) (src/carmcmc.cpp:92-95)."""
    if ntemps == 1:
        return np.ones(1)
    return np.exp(np.linspace(0.0, np.log(tmax), ntemps))


class LadderShard(object):
    """One rank's contiguous block of a temperature ladder shared by all ranks.

    `backend` is a ``carma_pack_amd.Context`` (or anything with pt_create / pt_shard / pt_start /
    pt_iterate / pt_get_chains / pt_set_chains / d): the tests plug in a CPU stand-in."""

    def __init__(self, backend, ntemps_global, nreplicas, adapt_iters, seed, dist, device="cpu"):
        self.b, self.dist, self.device = backend, dist, device
        self.rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        if ntemps_global < self.world:
            raise ValueError("need at least one temperature per rank")
        sl = shard_slice(ntemps_global, self.rank, self.world)
        self.slot0, self.T_local, self.T_global = sl.start, sl.stop - sl.start, ntemps_global
        self.R, self.seed, self.iteration = nreplicas, int(seed), 0
        self.temps = ladder_temperatures(ntemps_global)
        backend.pt_create(self.T_local, nreplicas, adapt_iters, seed=seed, temperatures=self.temps[sl])
        backend.pt_shard(ntemps_global, self.slot0, 0)
        self.nswap_boundary = 0
        self.nprop_boundary = 0
        # On a GPU the chain state lives in two torch tensors that the sampler is bound to
        # (carma_pt_bind_state): the kernels advance them in place and the boundary exchange below
        # reads, sends and overwrites single temperature rows without the state ever visiting the host.
        self._th = self._lp = None
        if str(device).startswith("cuda") and hasattr(backend, "pt_bind_state"):
            import torch
            self._th = torch.empty((nreplicas, self.T_local, backend.d), dtype=torch.float64, device=device)
            self._lp = torch.empty((nreplicas, self.T_local), dtype=torch.float64, device=device)
            torch.cuda.synchronize(device)
            backend.pt_bind_state(self._th.data_ptr(), self._lp.data_ptr())

    def start(self, init=None):
        self.b.pt_start(init)

    def _swap_with(self, peer, send):
        """Send `send` to `peer` and return what it sent us.  RCCL moves device memory directly; a
        process group that cannot (gloo) gets the R x (d+1) doubles staged through the host."""
        import torch
        dist = self.dist
        staged = send.is_cuda and dist.get_backend() != "nccl"
        buf = send.cpu() if staged else send
        recv = torch.empty_like(buf)
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, buf, peer), dist.P2POp(dist.irecv, recv, peer)]):
            w.wait()
        return recv.to(send.device) if staged else recv

    def _exchange_boundary(self, upper):
        """Exchange with the rank above (`upper`=True: my hottest chain vs its coldest) or below."""
        import torch
        peer = self.rank + 1 if upper else self.rank - 1
        mine = self.T_local - 1 if upper else 0
        hot_slot = self.slot0 + self.T_local if upper else self.slot0          # global slot of the hotter chain
        t_hot, t_cold = float(self.temps[hot_slot]), float(self.temps[hot_slot - 1])
        u = philox_uniform_chains(self.seed, np.arange(self.R) * self.T_global + hot_slot, self.iteration, RNG_SWAP)
        self.nprop_boundary += self.R
        if self._th is not None:                             # device-resident state
            th, lp = self._th, self._lp
            d = th.shape[2]
            other = self._swap_with(peer, torch.cat([th[:, mine, :], lp[:, mine, None]], dim=1).contiguous())
            my_lp, ot_lp = lp[:, mine].clone(), other[:, d]
            hot_lp, cold_lp = (ot_lp, my_lp) if upper else (my_lp, ot_lp)
            # ExchangeStep::DoStep (steps.hpp:331-339) in the kernels' form, log u < (lp_cold - lp_hot) (1/T_hot - 1/T_cold)
            # (a NaN rejects); both ranks evaluate the same expression on the same bits
            acc = torch.from_numpy(np.log(u)).to(th.device) < (cold_lp - hot_lp) * (1.0 / t_hot - 1.0 / t_cold)
            th[:, mine, :] = torch.where(acc[:, None], other[:, :d], th[:, mine, :])
            lp[:, mine] = torch.where(acc, ot_lp, my_lp)
            self.nswap_boundary += int(acc.sum())
            torch.cuda.current_stream(th.device).synchronize()          # the sampler runs on its own stream
            return
        th, lp = self.b.pt_get_chains()                      # [R][Tl][d], [R][Tl]
        d = th.shape[2]
        send = torch.from_numpy(np.concatenate([th[:, mine, :], lp[:, mine, None]], axis=1).copy()).to(self.device)
        other = self._swap_with(peer, send).cpu().numpy()
        my_lp, ot_lp = lp[:, mine], other[:, d]
        hot_lp, cold_lp = (ot_lp, my_lp) if upper else (my_lp, ot_lp)
        with np.errstate(invalid="ignore"):
            acc = np.log(u) < (cold_lp - hot_lp) * (1.0 / t_hot - 1.0 / t_cold)
        if acc.any():
            th[acc, mine, :] = other[acc, :d]
            lp[acc, mine] = other[acc, d]
            self.b.pt_set_chains(th, lp)
        self.nswap_boundary += int(acc.sum())

    def attach_comm(self, comm):
        """Switch to the native exchange: `comm` = carma_pack_amd._lib.Comm over the same ranks.  From now on
        iterate() is ONE call into the C ABI (carma_pt_iterate_sharded): per iteration the sampler kernel, the RCCL
        send/recv of the boundary chains and the swap kernel are enqueued on the sampler's stream -- no host
        synchronisation, no host RNG, no host<->device copy until the last iteration is done.  Same Philox keys and
        decisions as the torch.distributed path below, which stays as the stand-in for process groups that RCCL cannot
        serve (gloo in the CPU test-suite, two ranks sharing one GPU)."""
        if comm.size != self.world or comm.rank != self.rank:
            raise ValueError("communicator does not span the ladder's ranks")
        self._comm = comm

    def _sweep_inside(self):
        """The adjacent pairs inside this block, hottest first (the kernels' exchange sweep, carma_pt_core.h)."""
        if self.T_local < 2:
            return
        if hasattr(self.b, "pt_sweep"):
            self.b.pt_sweep()
            return
        th, lp = self.b.pt_get_chains()
        for i in range(self.T_local - 1, 0, -1):
            slot = self.slot0 + i
            u = philox_uniform_chains(self.seed, np.arange(self.R) * self.T_global + slot, self.iteration, RNG_SWAP)
            dbeta = 1.0 / float(self.temps[slot]) - 1.0 / float(self.temps[slot - 1])
            with np.errstate(invalid="ignore"):
                acc = np.log(u) < (lp[:, i - 1] - lp[:, i]) * dbeta
            if acc.any():
                th[acc, i, :], th[acc, i - 1, :] = th[acc, i - 1, :].copy(), th[acc, i, :].copy()
                lp[acc, i], lp[acc, i - 1] = lp[acc, i - 1].copy(), lp[acc, i].copy()
        self.b.pt_set_chains(th, lp)

    def iterate(self, niter):
        """niter x (RAM steps, then the ladder's sweep hot -> cold: boundary with the hotter rank, the pairs inside the
        block, boundary with the colder rank)."""
        if getattr(self, "_comm", None) is not None:
            from . import _lib
            _lib.pt_iterate_sharded([self.b], niter, self._comm)
            self.iteration += int(niter)
            self.nprop_boundary, self.nswap_boundary = self.b.pt_boundary_stats()
            return
        for _ in range(int(niter)):
            if self.world == 1:
                self.b.pt_iterate(1, do_exchange=True)
            else:
                self.b.pt_iterate(1, do_exchange=False)
                if self.rank + 1 < self.world:
                    self._exchange_boundary(upper=True)
                self._sweep_inside()
                if self.rank > 0:
                    self._exchange_boundary(upper=False)
            self.iteration += 1

    def coldest(self):
        """(theta[R][d], logpost[R]) of temperature 0 -- meaningful on rank 0."""
        if self._th is not None:
            return self._th[:, 0, :].cpu().numpy(), self._lp[:, 0].cpu().numpy()
        th, lp = self.b.pt_get_chains()
        return th[:, 0, :], lp[:, 0]


def sharded_pt_run(make_sampler, ntemps, nreplicas_total, sample_size, burnin, thin=1, init=None, seed=0, dist=None,
                   device="cpu"):
    """Independent replicas of the parallel-tempering sampler split by rank (mode 2).

    `make_sampler()` returns this rank's backend (a ``carma_pack_amd.Context``; the CPU tests plug in a stand-in with the
    same pt_* interface).  Rank r runs replicas [lo, hi) = ``shard_slice(nreplicas_total, r, world)`` with their GLOBAL
    indices (``pt_shard(ntemps, 0, lo)``: the Philox streams are keyed by the global chain slot, so the result does
    not depend on how the replicas are split), no collective while sampling; at the end the coldest-chain samples are
    gathered on every rank (``all_gather`` -- RCCL on device tensors with backend "nccl", gloo in the CPU suite) and
    returned as (samples[R_total][sample_size][d], logposts[R_total][sample_size]) -- the array a single process
    running all replicas returns (the reference returns the coldest chain: src/carmcmc.cpp:164-176,
    src/samplers.cpp:118-124)."""
    import torch
    b = make_sampler()
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if nreplicas_total < world:
        raise ValueError("need at least one replica per rank")
    sl = shard_slice(nreplicas_total, rank, world)
    R = sl.stop - sl.start
    b.pt_create(ntemps, R, burnin, seed=seed)
    b.pt_shard(ntemps, 0, sl.start)
    b.pt_start(init)
    b.pt_iterate(burnin, do_exchange=True)
    samples, logposts = b.pt_sample(sample_size, thin)
    samples, logposts = np.ascontiguousarray(samples), np.ascontiguousarray(logposts)
    if world == 1:
        return samples, logposts
    d = samples.shape[2]
    width = (nreplicas_total + world - 1) // world           # all_gather wants equal shapes: pad the short ranks
    buf = torch.full((width, sample_size, d + 1), float("nan"), dtype=torch.float64, device=device)
    buf[:R, :, :d] = torch.from_numpy(samples).to(device)
    buf[:R, :, d] = torch.from_numpy(logposts).to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out_s = np.empty((nreplicas_total, sample_size, d))
    out_l = np.empty((nreplicas_total, sample_size))
    for r in range(world):
        s_ = shard_slice(nreplicas_total, r, world)
        part = parts[r][: s_.stop - s_.start].cpu().numpy()
        out_s[s_], out_l[s_] = part[:, :, :d], part[:, :, d]
    return out_s, out_l
