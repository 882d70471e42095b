"""GPU test of the N>1 helpers with the real kernels behind them: batch sharding (`sharded_logdensity`) and one temperature ladder sharded over ranks (BASELINE config 4's partitioning,
carma_pack_amd/parallel.py `LadderShard`) with the real sampler behind it.

The GPU box has one device, so both ranks of the world_size-2 group run on cuda:0 and the process
group is gloo (RCCL refuses two ranks on one device): the chain state stays device-resident and bound
to the sampler (carma_pt_bind_state), only the R x (d+1) boundary rows are staged for the transport.
What is pinned: the log-posterior travels with its parameter vector (stored value == a fresh
evaluation by the CPU oracle, as carma_unit_tests.cpp:783-1114 pins the reference sampler), both
sides of the boundary take identical decisions, and the device-resident exchange walks the same
trajectory as the host-mediated one."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P, Q, TG, R, NITER, SEED = 3, 1, 5, 6, 40, 4242


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _series():
    rng = np.random.default_rng(11)
    n = 80
    t = np.cumsum(rng.uniform(1.0, 3.0, n))
    y = np.cumsum(rng.standard_normal(n)) * 0.3 + 0.2 * rng.standard_normal(n)
    return t, y - y.mean(), np.full(n, 0.2)


def _worker(rank, world, port, q, resident):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import carma_pack_amd as cpa
        from carma_pack_amd import parallel as par
        t, y, e = _series()
        ctx = cpa.Context(t, y, e, P, Q, max_stdev=10.0 * y.std())
        # batch sharding with the real kernels: every rank evaluates its slice, all ranks get the full result, and it
        # is the single-process result bit for bit (an evaluation does not depend on its neighbours in the launch)
        from carma_pack_amd.synth import prior_like_theta
        rng = np.random.default_rng(5)
        thetas = np.array([prior_like_theta(rng, P, Q, t, y) for _ in range(37)])
        full = par.sharded_logdensity(lambda x: ctx.logdensity(x, ignore_prior=True), thetas, dist)
        assert np.array_equal(full, ctx.logdensity(thetas, ignore_prior=True), equal_nan=True)
        sh = par.LadderShard(ctx, TG, R, adapt_iters=NITER, seed=SEED, dist=dist, device="cuda:0" if resident else "cpu")
        assert (sh._th is not None) == resident
        sh.start()
        sh.iterate(NITER)
        th, lp = ctx.pt_get_chains()
        if resident:
            assert np.array_equal(th, sh._th.cpu().numpy()) and np.array_equal(lp, sh._lp.cpu().numpy())
        q.put((rank, sh.nswap_boundary, sh.nprop_boundary, sh.slot0, th, lp))
    finally:
        dist.destroy_process_group()


def _run(resident):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, resident)) for r in range(world)]
    for p in procs:
        p.start()
    from helpers import queue_get
    res = sorted((queue_get(q, procs, 300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


def _unsharded(tg=TG):
    """The same ladder on ONE context (the one-GPU sampler kernel with its own sweep): what every sharded run must equal."""
    import carma_pack_amd as cpa
    from carma_pack_amd import parallel as par
    t, y, e = _series()
    ctx = cpa.Context(t, y, e, P, Q, max_stdev=10.0 * y.std())
    ctx.pt_create(tg, R, NITER, seed=SEED, temperatures=par.ladder_temperatures(tg))
    ctx.pt_shard(tg, 0, 0)
    ctx.pt_start(None)
    ctx.pt_iterate(NITER)
    th, lp = ctx.pt_get_chains()
    return th, lp, ctx.pt_stats()


def test_ladder_sharded_over_two_ranks():
    import oracle as orc
    res = _run(resident=True)
    (r0, sw0, pr0, s0, th0, lp0), (r1, sw1, pr1, s1, th1, lp1) = res
    assert (s0, th0.shape[1], s1, th1.shape[1]) == (0, 3, 3, 2)
    assert sw0 == sw1 and pr0 == pr1 == R * NITER                # the boundary pair is proposed every iteration
    assert 0 < sw0 < pr0
    # THE SHARDED LADDER IS THE UNSHARDED ONE: same starting values (keyed by the global chain slot), same RAM steps,
    # same sweep order, same uniforms -> the same chain states bit for bit
    uth, ulp, _ = _unsharded()
    assert np.array_equal(np.concatenate([th0, th1], axis=1), uth) and np.array_equal(np.concatenate([lp0, lp1], axis=1), ulp)
    # stored log-posterior == LogDensity(theta) for every chain on both ranks, swapped ones included
    t, y, e = _series()
    ms = 10.0 * y.std()
    th = np.concatenate([th0, th1], axis=1).reshape(-1, 3 + P + Q)
    lp = np.concatenate([lp0, lp1], axis=1).ravel()
    assert np.all(np.isfinite(lp))
    from helpers import assert_parity
    from helpers import loglik_truth
    m = orc.OracleModel(t, y, e, P, Q, max_stdev=ms)
    assert_parity(lp, m.logdensity_batch(th), 1e-10, "sharded chain states",
                  arbiter=lambda i: loglik_truth(t, y, e, th[i], P, Q)[0])
    # the host-mediated exchange (state copied out and back every boundary) walks the same trajectory
    host = _run(resident=False)
    for a, b in zip(res, host):
        assert a[1] == b[1] and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])


def _native_worker(q, nblocks_T, rccl, nsample=0):
    # (every block's self-check must have agreed)
    """One process owning the whole ladder as consecutive blocks; boundaries go through carma_pt_iterate_sharded."""
    import carma_pack_amd as cpa
    from carma_pack_amd import _lib, parallel as par
    t, y, e = _series()
    temps = par.ladder_temperatures(TG)
    comm = _lib.Comm(_lib.Comm.unique_id(), 1, 0, device=0) if rccl else None
    ctxs, slot0 = [], 0
    for Tl in nblocks_T:
        c = cpa.Context(t, y, e, P, Q, max_stdev=10.0 * y.std())
        c.pt_create(Tl, R, NITER, seed=SEED, temperatures=temps[slot0:slot0 + Tl])
        c.pt_shard(TG, slot0, 0)
        c.pt_start(None)
        ctxs.append(c)
        slot0 += Tl
    _lib.pt_iterate_sharded(ctxs, NITER // 2, comm)
    _lib.pt_iterate_sharded(ctxs, NITER - NITER // 2, comm)        # resumable: two calls == one
    samples = _lib.pt_sample_sharded(ctxs, nsample, 3, comm) if nsample else None
    out = []
    for c in ctxs:
        th, lp = c.pt_get_chains()
        assert c.pt_boundary_check() == (1 if len(nblocks_T) > 1 else 0)
        out.append((th, lp, c.pt_boundary_stats(), c.pt_iterations_done()))
    q.put(out + ([samples] if nsample else []))
    if comm is not None:
        comm.close()


def _run_native(nblocks_T, rccl=True, nsample=0):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_native_worker, args=(q, nblocks_T, rccl, nsample))
    p.start()
    from helpers import queue_get
    out = queue_get(q, [p], 300)
    p.join(120)
    assert p.exitcode == 0
    return out


def test_native_rccl_exchange_walks_the_same_trajectory():
    """carma_pt_iterate_sharded (pack kernel -> ncclSend/ncclRecv -> swap kernel on the sampler's stream, no host work
    per iteration) against the torch.distributed stand-in above: same seed, same blocks (3 + 2 temperatures) -> the same
    chain states bit for bit and the same boundary-swap count.  The box has one GPU, so the two blocks live in one
    process and the boundary rows travel through RCCL send/recv to the process's own rank -- the same code path, and
    the same RCCL kernels, as between two GPUs."""
    ref = _run(resident=True)
    (r0, sw0, pr0, s0, th0, lp0), (r1, sw1, pr1, s1, th1, lp1) = ref
    nat = _run_native([3, 2])
    (nth0, nlp0, (npr0, nsw0), it0), (nth1, nlp1, (npr1, nsw1), it1) = nat
    assert it0 == it1 == NITER
    assert np.array_equal(nth0, th0) and np.array_equal(nlp0, lp0)
    assert np.array_equal(nth1, th1) and np.array_equal(nlp1, lp1)
    assert (npr0, nsw0) == (pr0, sw0) and (npr1, nsw1) == (pr1, sw1)
    # ... and both are the unsharded ladder, whatever the partition: one block per temperature (BASELINE config 4's
    # layout: 5 blocks of 1), 2 + 2 + 1, and without RCCL one block of 5
    uth, ulp, (uacc, uswp) = _unsharded()
    assert np.array_equal(np.concatenate([nth0, nth1], axis=1), uth) and np.array_equal(np.concatenate([nlp0, nlp1], axis=1), ulp)
    for blocks in ([2, 2, 1], [5]):
        out = _run_native(blocks, rccl=len(blocks) > 1)
        assert np.array_equal(np.concatenate([o[0] for o in out], axis=1), uth), blocks
        assert np.array_equal(np.concatenate([o[1] for o in out], axis=1), ulp), blocks
    one = _run_native([1, 1, 1, 1, 1])
    assert np.array_equal(np.concatenate([o[0] for o in one], axis=1), uth)
    assert np.array_equal(np.concatenate([o[1] for o in one], axis=1), ulp)
    import oracle as orc
    t, y, e = _series()
    m = orc.OracleModel(t, y, e, P, Q, max_stdev=10.0 * y.std())
    th = np.concatenate([o[0] for o in one], axis=1).reshape(-1, 3 + P + Q)
    lp = np.concatenate([o[1] for o in one], axis=1).ravel()
    from helpers import assert_parity
    from helpers import loglik_truth
    assert_parity(lp, m.logdensity_batch(th), 1e-10, "one temperature per block",
                  arbiter=lambda i: loglik_truth(t, y, e, th[i], P, Q)[0])
    prop = [o[2][0] for o in one]
    acc = [o[2][1] for o in one]
    assert prop == [R * NITER, 2 * R * NITER, 2 * R * NITER, 2 * R * NITER, R * NITER]     # every boundary, every iteration
    assert all(0 < a < p_ for a, p_ in zip(acc, prop))
    # boundary k is counted once by either side: blocks 0 and 4 see one boundary, the inner blocks two
    assert acc[0] + acc[2] + acc[4] == acc[1] + acc[3]


def test_sharded_ladder_on_the_large_ensemble_path(monkeypatch):
    """The same with the blocks on the sampler path for large ensembles (carma_pt_lane.hip; forced here, CARMA_PT_KERNEL=lane):
    global temperature slots and replica offsets reach its Philox keys, a block's iteration without its own sweep
    (do_exchange = 0) leaves the chain-major state arrays as the sweep / boundary kernels expect them -- blocks 3 + 2 and
    2 + 2 + 1 through RCCL-to-self walk the one-block ladder's trajectory bit for bit, and that one is the ladder kernel's
    to rounding with the same boundary decisions."""
    uth0, ulp0, _ = _unsharded()                            # default kernels
    monkeypatch.setenv("CARMA_PT_KERNEL", "lane")
    one = _run_native([5], rccl=False)
    uth, ulp = one[0][0], one[0][1]
    np.testing.assert_allclose(uth, uth0, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(ulp, ulp0, rtol=1e-8)
    for blocks in ([3, 2], [2, 2, 1]):
        out = _run_native(blocks)
        assert np.array_equal(np.concatenate([o[0] for o in out], axis=1), uth), blocks
        assert np.array_equal(np.concatenate([o[1] for o in out], axis=1), ulp), blocks
        assert all(o[3] == NITER for o in out)


def test_sharded_ladder_saves_the_coldest_chain():
    """carma_pt_sample_sharded: the sharded ladder as a complete sampler -- after every thin-th iteration AND its
    boundary swaps the coldest chain of every replica is saved (Sampler::SaveValues, samplers.cpp:118-124): the last
    sample is the final chain state, every stored log-posterior is the oracle's LogDensity of its sample, and with one
    temperature per block (where the coldest chain IS a boundary chain) the samples are not all the same point."""
    import oracle as orc
    from helpers import assert_parity, loglik_truth
    t, y, e = _series()
    m = orc.OracleModel(t, y, e, P, Q, max_stdev=10.0 * y.std())
    for blocks in ([3, 2], [1, 1, 1, 1, 1]):
        out = _run_native(blocks, nsample=7)
        samples, slp = out.pop()
        assert samples.shape == (R, 7, 3 + P + Q) and slp.shape == (R, 7)
        (th0, lp0, _, it0) = out[0]
        assert it0 == NITER + 21
        assert np.array_equal(samples[:, -1, :], th0[:, 0, :]) and np.array_equal(slp[:, -1], lp0[:, 0])
        flat = samples.reshape(-1, 3 + P + Q)
        assert_parity(slp.reshape(-1), m.logdensity_batch(flat), 1e-10, "sharded samples %s" % blocks,
                      arbiter=lambda i: loglik_truth(t, y, e, flat[i], P, Q)[0])
        assert np.unique(flat[:, 0]).size > R                     # the chains moved between saves


def _two_gpu_worker(rank, port, q):
    """One process per GPU: rank 0 holds temperatures 0..2, rank 1 temperatures 3..4; the boundary goes over RCCL."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)          # bootstrap only: carries the 128-byte RCCL id
    try:
        import carma_pack_amd as cpa
        from carma_pack_amd import _lib, parallel as par
        t, y, e = _series()
        temps = par.ladder_temperatures(TG)
        comm = _lib.Comm.from_torch(dist, device=rank)
        Tl, slot0 = (3, 0) if rank == 0 else (2, 3)
        c = cpa.Context(t, y, e, P, Q, max_stdev=10.0 * y.std(), device=rank)
        c.pt_create(Tl, R, NITER, seed=SEED, temperatures=temps[slot0:slot0 + Tl])
        c.pt_shard(TG, slot0, 0)
        c.pt_start(None)
        _lib.pt_iterate_sharded([c], NITER, comm)
        th, lp = c.pt_get_chains()
        q.put((rank, th, lp, c.pt_boundary_stats()))
        comm.close()
    finally:
        dist.destroy_process_group()


def test_rccl_exchange_between_two_gpus():
    """The same ladder with its two blocks on two GPUs, one process each, boundary rows over RCCL send/recv between the
    devices (xGMI): the same chain states bit for bit as the one-GPU reference.  Needs two GPUs; the one-GPU box of the
    round-end run skips it (the path is then covered through send/recv to the process's own rank, above)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    from helpers import queue_get
    ref = _run(resident=True)
    (r0, sw0, pr0, s0, th0, lp0), (r1, sw1, pr1, s1, th1, lp1) = ref
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_gpu_worker, args=(r, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted((queue_get(q, procs, 300) for _ in range(2)), key=lambda r: r[0])
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    (_, nth0, nlp0, (npr0, nsw0)), (_, nth1, nlp1, (npr1, nsw1)) = res
    assert np.array_equal(nth0, th0) and np.array_equal(nlp0, lp0)
    assert np.array_equal(nth1, th1) and np.array_equal(nlp1, lp1)
    assert (npr0, nsw0) == (pr0, sw0) and (npr1, nsw1) == (pr1, sw1)


def _build_shm_transport():
    """tests/shm_transport/shm_rccl.cpp -> libshm_rccl.so (hipcc; host code only)."""
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shm_transport")
    src, lib = os.path.join(here, "shm_rccl.cpp"), os.path.join(here, "libshm_rccl.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-shared", "-fPIC", "-o", lib, src, "-lrt", "-lpthread"], check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    return lib


def _shm_rank_worker(rank, blocks_by_rank, port, q, lib, nsample):
    """One PROCESS per rank, all on cuda:0, the native sharded path with nranks = len(blocks_by_rank): the boundary rows
    travel through the shared-memory test double of the RCCL entry points (CARMA_RCCL_LIB)."""
    os.environ["CARMA_RCCL_LIB"] = lib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=len(blocks_by_rank))     # carries the 128-byte id, as in production
    try:
        import carma_pack_amd as cpa
        from carma_pack_amd import _lib, parallel as par
        t, y, e = _series()
        tg = sum(sum(b) for b in blocks_by_rank)                     # (5 in most cases; 8 for one temperature per rank x 8)
        temps = par.ladder_temperatures(tg)
        comm = _lib.Comm.from_torch(dist, device=0)
        assert (comm.rank, comm.size) == (rank, len(blocks_by_rank))
        slot0 = sum(sum(b) for b in blocks_by_rank[:rank])
        ctxs = []
        for Tl in blocks_by_rank[rank]:
            c = cpa.Context(t, y, e, P, Q, max_stdev=10.0 * y.std())
            c.pt_create(Tl, R, NITER, seed=SEED, temperatures=temps[slot0:slot0 + Tl])
            c.pt_shard(tg, slot0, 0)
            c.pt_start(None)
            ctxs.append(c)
            slot0 += Tl
        _lib.pt_iterate_sharded(ctxs, NITER // 2, comm)
        _lib.pt_iterate_sharded(ctxs, NITER - NITER // 2, comm)
        samples = _lib.pt_sample_sharded(ctxs, nsample, 3, comm) if nsample else None
        out = []
        for c in ctxs:
            th, lp = c.pt_get_chains()
            assert c.pt_boundary_check() == 1
            out.append((th, lp, c.pt_boundary_stats(), c.pt_iterations_done()))
        q.put((rank, out, samples))
        comm.close()
    finally:
        dist.destroy_process_group()


def _run_shm_ranks(blocks_by_rank, nsample=0):
    import torch.multiprocessing as mp
    from helpers import queue_get
    lib = _build_shm_transport()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shm_rank_worker, args=(r, blocks_by_rank, port, q, lib, nsample)) for r in range(len(blocks_by_rank))]
    for p_ in procs:
        p_.start()
    res = sorted((queue_get(q, procs, 300) for _ in procs), key=lambda r: r[0])
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    return res


@pytest.mark.parametrize("blocks_by_rank", [[[3], [2]], [[2], [2], [1]], [[1, 1], [2, 1]], [[1], [1], [1], [1], [1]], [[1]] * 8])
def test_native_sharded_path_with_more_than_one_rank(blocks_by_rank):
    """carma_pt_iterate_sharded / carma_pt_sample_sharded with nranks = 2, 3 and 5 PROCESSES (one or two blocks each):
    which rank talks to which, in what order, the boundary self-check between processes.  RCCL refuses two ranks on one
    device, so on this one-GPU box the eight RCCL entry points the library binds are a shared-memory test double
    (tests/shm_transport, CARMA_RCCL_LIB) -- everything above them is the production code: `Comm.from_torch` (the id
    travels through torch.distributed), `carma_comm_create` with nranks > 1, the pack / exchange / swap / sweep order.
    The chain states must be the unsharded ladder's, bit for bit, whatever the partition; the rank that owns the coldest
    temperature saves the samples.  [[1]] * 8 is BASELINE configs[3]'s partition: eight ranks, one temperature each."""
    uth, ulp, _ = _unsharded(sum(sum(b) for b in blocks_by_rank))
    res = _run_shm_ranks(blocks_by_rank)
    ths = [o[0] for _, out, _ in res for o in out]
    lps = [o[1] for _, out, _ in res for o in out]
    assert [t_.shape[1] for t_ in ths] == [b for blocks in blocks_by_rank for b in blocks]
    assert np.array_equal(np.concatenate(ths, axis=1), uth) and np.array_equal(np.concatenate(lps, axis=1), ulp)
    assert all(o[3] == NITER for _, out, _ in res for o in out)
    # every boundary proposed every iteration, and counted alike by its two sides
    stats = [o[2] for _, out, _ in res for o in out]
    nb = len(stats)
    assert [s_[0] for s_ in stats] == [R * NITER * ((i > 0) + (i < nb - 1)) for i in range(nb)]
    if nb == 2:
        assert stats[0] == stats[1] and 0 < stats[0][1] < stats[0][0]


def test_native_sharded_sampler_with_two_ranks():
    """... and as a complete sampler: 7 saves, thin 3, on two processes -- rank 0 (which owns the coldest temperature) returns
    the samples, and they are those of the one-process run of the same partition."""
    one = _run_native([3, 2], nsample=7)
    ref_samples, ref_slp = one.pop()
    res = _run_shm_ranks([[3], [2]], nsample=7)
    (_, out0, smp0), (_, out1, smp1) = res
    assert smp0 is not None and np.array_equal(smp0[0], ref_samples) and np.array_equal(smp0[1], ref_slp)
    assert np.array_equal(out0[0][0], one[0][0]) and np.array_equal(out1[0][0], one[1][0])


def _replica_worker(rank, world, port, q):
    """CarmaModel.run_mcmc(dist=): independent ladders split over the ranks, coldest chains gathered on every rank."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import carmcmc as cm
        t, y, e = _series()
        model = cm.CarmaModel(t, y + 3.0, e, p=P, q=Q)
        s = model.run_mcmc(60, nburnin=40, ntemperatures=4, nreplicas=5, seed=99, dist=dist)
        q.put((rank, np.asarray(s._sampler._all_samples), np.asarray(s._sampler._all_logposts)))
    finally:
        dist.destroy_process_group()


def test_run_mcmc_with_replicas_split_over_ranks():
    """SURVEY 8(e) mode 2 through the Python API: run_mcmc(nreplicas=5, dist=) on two ranks (3 + 2 ladders; gloo, both on
    the one GPU) returns, on EVERY rank, the samples and log-posteriors of all five ladders -- the very arrays the
    single-process call with the same seed returns (starting values and sampler streams are keyed by the global chain
    slot, so the split does not enter)."""
    import torch.multiprocessing as mp
    from helpers import queue_get
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replica_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted((queue_get(q, procs, 300) for _ in range(2)), key=lambda r: r[0])
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    import carmcmc as cm
    t, y, e = _series()
    one = cm.CarmaModel(t, y + 3.0, e, p=P, q=Q).run_mcmc(60, nburnin=40, ntemperatures=4, nreplicas=5, seed=99)
    ref_s, ref_l = np.asarray(one._sampler._all_samples), np.asarray(one._sampler._all_logposts)
    assert ref_s.shape == (5, 60, 3 + P + Q)
    for _, s, l in res:
        assert np.array_equal(s, ref_s) and np.array_equal(l, ref_l)
