// carma_shard.hip -- ONE temperature ladder sharded across GPUs (BASELINE config 4).
//
// Every rank (one process per GPU) owns a contiguous block of the ladder's temperatures for all R replicas.  The only
// coupling between ranks is the adjacent-temperature exchange across a block boundary (ExchangeStep::DoStep,
// /root/reference/src/include/steps.hpp:318-362; wiring src/carmcmc.cpp:147-157).
//
// Round 3: THE SHARDED LADDER WALKS THE UNSHARDED LADDER'S TRAJECTORY, bit for bit.  An iteration of the one-GPU kernel
// is "every chain takes its RAM step, then ONE sweep over the adjacent pairs from the hottest to the coldest"; here
//     every block:  sampler kernel without its sweep (RAM steps only)
//     then, from the hottest block to the coldest:
//         boundary with the hotter neighbour   k_shard_pack -> ncclSend / ncclRecv (grouped) -> k_shard_swap
//         the pairs inside the block           k_shard_sweep (hot -> cold, physical swaps)
//         boundary with the colder neighbour   (the same exchange, seen from the other side)
// i.e. the same pairs in the same order with the same Philox uniforms (keyed by seed, global slot of the hotter chain,
// iteration): both sides of a boundary compute the same decision from the same bits, nothing but chain states crosses
// the link, and the chain states after any number of iterations equal those of the unsharded run
// (tests/test_gpu_ladder_shard.py).  (Round 2 proposed a boundary pair only every other iteration -- even boundaries on
// even iterations -- which was a valid but DIFFERENT kernel: a sharded ladder mixed more slowly across its boundaries.)
// All of it is enqueued on the sampler's stream: no host synchronisation, no host RNG, no host<->device copy per
// iteration.  The hot -> cold order makes the boundary exchanges of one iteration a chain through the ranks, but a rank
// only ever waits for its two neighbours: it starts its next RAM steps as soon as its own colder boundary is done, so
// in the steady state an iteration costs a rank its kernel + two exchanges, whatever the number of ranks.
// A process may own several consecutive blocks (`nlocal`): a boundary between two of its own blocks goes through
// ncclSend/ncclRecv to itself, which is what lets a one-GPU box exercise the RCCL path.
// SELF-CHECK: every boundary side folds (iteration, replica, decision, the two log-posteriors it compared) into a
// checksum; at the end of a call the two sides of every boundary exchange their sums and a mismatch -- a transfer that
// delivered something else than was sent, ranks that disagree about the ladder -- fails the call loudly.
//
// RCCL is bound at run time (dlopen of librccl.so.1: the copy PyTorch already loaded when there is one), so the
// library itself has no link-time dependency on it; without RCCL carma_comm_* fail loudly.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <utility>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/carma_mi355.h"
#include "carma_host.h"

#define CARMA_DEV __device__ __forceinline__
#include "carma_rng.h"

namespace carma {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
};

static Rccl* rccl()
{
    static Rccl api;
    static std::once_flag once;                              // (choose_order drives library calls from a thread pool)
    std::call_once(once, [] {
        // CARMA_RCCL_LIB: bind THIS library instead (tests/shm_transport: a shared-memory test double of the eight entry
        // points, with which a one-GPU box runs the nranks > 1 paths below -- RCCL refuses two ranks on one device)
        const char* override_path = getenv("CARMA_RCCL_LIB");
        const char* names[] = {"librccl.so.1", "librccl.so"};
        if (override_path && *override_path) {
            api.lib = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char* nm : names) {
                api.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
                if (api.lib) break;
            }
        }
        if (api.lib) {
#define CARMA_RCCL_SYM(field, sym) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, sym))
            CARMA_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
            CARMA_RCCL_SYM(CommInitRank, "ncclCommInitRank");
            CARMA_RCCL_SYM(CommDestroy, "ncclCommDestroy");
            CARMA_RCCL_SYM(GetErrorString, "ncclGetErrorString");
            CARMA_RCCL_SYM(GroupStart, "ncclGroupStart");
            CARMA_RCCL_SYM(GroupEnd, "ncclGroupEnd");
            CARMA_RCCL_SYM(Send, "ncclSend");
            CARMA_RCCL_SYM(Recv, "ncclRecv");
#undef CARMA_RCCL_SYM
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.GetErrorString || !api.GroupStart ||
                !api.GroupEnd || !api.Send || !api.Recv) {
                dlclose(api.lib);
                api.lib = nullptr;
            }
        }
    });
    return api.lib ? &api : nullptr;
}

struct Comm {
    ncclComm_t nccl = nullptr;
    int nranks = 1, rank = 0, device = 0;
};

static int rccl_fail(ncclResult_t r, const char* what)
{
    Rccl* api = rccl();
    set_error("%s: %s", what, api ? api->GetErrorString(r) : "RCCL not loaded");
    return CARMA_EHIP;
}

// (theta[d], logpost) of temperature `mine` of every replica -> buf[R][d+1], followed by that temperature itself (so
// the other side needs no knowledge of this block's ladder)
__global__ void k_shard_pack(const double* __restrict__ theta, const double* __restrict__ lp, int R, int T, int d, int mine,
                             double temperature, double* __restrict__ buf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > R * (d + 1)) return;
    if (i == R * (d + 1)) {
        buf[i] = temperature;
        return;
    }
    const int r = i / (d + 1), j = i - r * (d + 1);
    buf[i] = j < d ? theta[((size_t)r * T + mine) * d + j] : lp[(size_t)r * T + mine];
}

// order-independent fold of one boundary decision: both sides of a boundary must arrive at the same sum
__device__ __forceinline__ unsigned long long shard_mix(unsigned long long iter, unsigned r, bool acc, double hot, double cold)
{
    unsigned long long h = iter * 0x9E3779B97F4A7C15ull + (unsigned long long)r * 0xC2B2AE3D27D4EB4Full + (acc ? 0x165667B19E3779F9ull : 0ull);
    h ^= (unsigned long long)__double_as_longlong(hot) * 0xFF51AFD7ED558CCDull;
    h ^= (unsigned long long)__double_as_longlong(cold) * 0xC4CEB9FE1A85EC53ull;
    h ^= h >> 29;
    return h * 0xBF58476D1CE4E5B9ull;
}

// ExchangeStep::DoStep (steps.hpp:318-362) for the pair (hot_slot, hot_slot - 1), one thread per replica.  `upper`: this
// block holds the COLDER chain of the pair (its hottest temperature, `mine` = T - 1), the peer the hotter one.
// alpha = (lp_cold - lp_hot) (1/T_hot - 1/T_cold)  (steps.hpp:331-332);  accept when log u < alpha (NaN rejects, :336-338)
// -- the expression, the uniform and the comparison of the sampler kernels' own sweep (exchange_decide, carma_pt_core.h).
__global__ void k_shard_swap(double* __restrict__ theta, double* __restrict__ lp, int R, int T, int d, int mine,
                             const double* __restrict__ recv, int upper, double t_mine, unsigned seed0, unsigned seed1,
                             unsigned long long iter, unsigned T_global, unsigned replica0, unsigned hot_slot,
                             unsigned* __restrict__ nswap_bnd, unsigned* __restrict__ nswap, unsigned long long* __restrict__ checksum)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const double my_lp = lp[(size_t)r * T + mine], ot_lp = recv[(size_t)r * (d + 1) + d];
    const double hot = upper ? ot_lp : my_lp, cold = upper ? my_lp : ot_lp;
    const double t_peer = recv[(size_t)R * (d + 1)];
    const double t_hot = upper ? t_peer : t_mine, t_cold = upper ? t_mine : t_peer;
    const double a = (cold - hot) * (1.0 / t_hot - 1.0 / t_cold);
    RngKey key{seed0, seed1, (replica0 + (unsigned)r) * T_global + hot_slot};
    const double logu = log(rng_uniform(key, iter, RNG_SWAP, 0));
    const bool acc = logu < a;
    if (acc) {
        for (int j = 0; j < d; j++) theta[((size_t)r * T + mine) * d + j] = recv[(size_t)r * (d + 1) + j];
        lp[(size_t)r * T + mine] = ot_lp;
        atomicAdd(nswap_bnd, 1u);
        if (!upper) nswap[(size_t)r * T] += 1u;            // entry i of the statistics = swaps between temperature i and i - 1
    }
    atomicAdd(checksum, shard_mix(iter, (unsigned)r, acc, hot, cold));
}

// The pairs INSIDE a block, hottest first (the part of the sweep the sampler kernel was told to leave out): one thread
// per replica walks hot -> cold and swaps the two chains' parameter vectors and stored log-posteriors in place
// (exchange_sweep of carma_pt_core.h with the kernels' decision rule: log u < (lp_{i-1} - lp_i) (1/T_i - 1/T_{i-1})).
__global__ void k_shard_sweep(double* __restrict__ theta, double* __restrict__ lp, int R, int T, int d,
                              const double* __restrict__ temps, unsigned seed0, unsigned seed1, unsigned long long iter,
                              unsigned T_global, unsigned replica0, unsigned slot0, unsigned* __restrict__ nswap)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    double* th = theta + (size_t)r * T * d;
    double* l = lp + (size_t)r * T;
    const unsigned chain_base = (replica0 + (unsigned)r) * T_global + slot0;
    for (int i = T - 1; i > 0; i--) {
        const double hot = l[i], cold = l[i - 1];
        const double dbeta = 1.0 / temps[i] - 1.0 / temps[i - 1];
        const double a = (cold - hot) * dbeta;
        RngKey key{seed0, seed1, chain_base + (unsigned)i};
        const double logu = log(rng_uniform(key, iter, RNG_SWAP, 0));
        if (logu < a) {
            for (int j = 0; j < d; j++) {
                const double tmp = th[(size_t)i * d + j];
                th[(size_t)i * d + j] = th[(size_t)(i - 1) * d + j];
                th[(size_t)(i - 1) * d + j] = tmp;
            }
            l[i] = cold;
            l[i - 1] = hot;
            nswap[(size_t)r * T + i] += 1u;
        }
    }
}

// Sampler::SaveValues (src/samplers.cpp:118-124) for the sharded ladder: the coldest chain of every replica, AFTER the
// iteration's boundary swaps -> samples[r][sidx][d], logposts[r][sidx]
__global__ void k_shard_save(const double* __restrict__ theta, const double* __restrict__ lp, int R, int T, int d, long sidx,
                             long cap, double* __restrict__ samples, double* __restrict__ slp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * (d + 1)) return;
    const int r = i / (d + 1), j = i - r * (d + 1);
    if (j < d)
        samples[((size_t)r * cap + sidx) * d + j] = theta[((size_t)r * T) * d + j];
    else
        slp[(size_t)r * cap + sidx] = lp[(size_t)r * T];
}

}  // namespace carma

using namespace carma;

extern "C" {

int carma_comm_unique_id(void* out128)
{
    if (!out128) return CARMA_EINVAL;
    Rccl* api = rccl();
    if (!api) {
        set_error("carma_comm_unique_id: librccl.so.1 could not be loaded (%s)", dlerror() ? dlerror() : "no such library");
        return CARMA_ENODEV;
    }
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
    std::memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return CARMA_OK;
}

carma_comm* carma_comm_create(const void* id128, int nranks, int rank, int device)
{
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) {
        set_error("carma_comm_create: bad argument");
        return nullptr;
    }
    Rccl* api = rccl();
    if (!api) {
        set_error("carma_comm_create: librccl.so.1 could not be loaded");
        return nullptr;
    }
    if (select_device(device) != CARMA_OK) return nullptr;
    ncclUniqueId id;
    std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    Comm* cm = new Comm();
    cm->nranks = nranks;
    cm->rank = rank;
    cm->device = device;
    ncclResult_t r = api->CommInitRank(&cm->nccl, nranks, id, rank);
    if (r != ncclSuccess) {
        rccl_fail(r, "ncclCommInitRank");
        delete cm;
        return nullptr;
    }
    return reinterpret_cast<carma_comm*>(cm);
}

void carma_comm_destroy(carma_comm* h)
{
    if (!h) return;
    Comm* cm = reinterpret_cast<Comm*>(h);
    Rccl* api = rccl();
    if (api && cm->nccl) {
        (void)hipSetDevice(cm->device);
        (void)api->CommDestroy(cm->nccl);
    }
    delete cm;
}

int carma_comm_rank(const carma_comm* h) { return h ? reinterpret_cast<const Comm*>(h)->rank : CARMA_EINVAL; }
int carma_comm_size(const carma_comm* h) { return h ? reinterpret_cast<const Comm*>(h)->nranks : CARMA_EINVAL; }

}  // extern "C"

// niter iterations of the sharded ladder; save_thin > 0: after every save_thin-th iteration (and its boundary swaps)
// the coldest chain of every replica is appended to d_samples / d_slp -- only on the process that owns temperature 0.
static int iterate_sharded(carma_ctx* const* shards, int nlocal, long niter, carma_comm* comm, int save_thin, long sample_cap,
                           double* d_samples, double* d_slp)
{
    if (!shards || nlocal < 1 || niter < 0) {
        set_error("carma_pt_iterate_sharded: bad argument");
        return CARMA_EINVAL;
    }
    Comm* cm = reinterpret_cast<Comm*>(comm);
    const int nranks = cm ? cm->nranks : 1, rank = cm ? cm->rank : 0;
    const int nblocks = nranks * nlocal;
    std::vector<Ctx*> cs(nlocal);
    for (int i = 0; i < nlocal; i++) {
        cs[i] = reinterpret_cast<Ctx*>(shards[i]);
        if (!cs[i] || !cs[i]->pt || !cs[i]->pt->started) {
            set_error("carma_pt_iterate_sharded: shard %d has no started sampler", i);
            return CARMA_EINVAL;
        }
    }
    PtState* s0 = cs[0]->pt;
    for (int i = 0; i < nlocal; i++) {
        PtState* s = cs[i]->pt;
        const bool chained = i == 0 || s->slot0 == cs[i - 1]->pt->slot0 + (unsigned)cs[i - 1]->pt->T;
        if (s->R != s0->R || cs[i]->d != cs[0]->d || s->T_global != s0->T_global || s->replica0 != s0->replica0 ||
            s->seed != s0->seed || s->iter != s0->iter || cs[i]->device != cs[0]->device || !chained) {
            set_error("carma_pt_iterate_sharded: shard %d does not continue the ladder of shard 0 (same replicas, seed, "
                      "iteration and device, contiguous temperature slots)", i);
            return CARMA_EINVAL;
        }
    }
    if ((rank == 0 && s0->slot0 != 0) ||
        (rank == nranks - 1 && cs[nlocal - 1]->pt->slot0 + (unsigned)cs[nlocal - 1]->pt->T != s0->T_global)) {
        set_error("carma_pt_iterate_sharded: the blocks of the ranks must tile the ladder in rank order (carma_pt_shard)");
        return CARMA_EINVAL;
    }
    if (nblocks > 1 && !cm) {
        set_error("carma_pt_iterate_sharded: more than one block needs a communicator (carma_comm_create)");
        return CARMA_EINVAL;
    }
    Rccl* api = rccl();
    if (nblocks > 1 && !api) {
        set_error("carma_pt_iterate_sharded: RCCL is not available");
        return CARMA_ENODEV;
    }
    hipError_t e = hipSetDevice(cs[0]->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    const int R = s0->R, d = cs[0]->d;
    const size_t nbuf = (size_t)R * (d + 1) + 1;     // boundary chains of all replicas + the boundary temperature
    for (int i = 0; i < nlocal; i++) {
        PtState* s = cs[i]->pt;
        if (!s->d_send) {
            e = dev_malloc(&s->d_send, sizeof(double) * nbuf);
            if (e == hipSuccess) e = dev_malloc(&s->d_recv, sizeof(double) * nbuf);
            if (e == hipSuccess) e = dev_malloc(&s->d_bnd_swaps, sizeof(unsigned));
            if (e == hipSuccess) e = hipMemset(s->d_bnd_swaps, 0, sizeof(unsigned));
            if (e == hipSuccess) e = dev_malloc(&s->d_checksum, 4 * sizeof(unsigned long long));
            if (e == hipSuccess) e = hipMemset(s->d_checksum, 0, 4 * sizeof(unsigned long long));
            if (e != hipSuccess) return hip_fail(e, "carma_pt_iterate_sharded: boundary buffers");
        }
    }
    // every block of this process on ONE stream (block 0's): kernels, RCCL calls and swap kernels are ordered by it
    hipStream_t st = cs[0]->stream;
    // The self-check covers THIS call: the call is collective, so both sides of every boundary clear their folds at the
    // same point of the ladder's history (sums kept over a context's lifetime made a re-created neighbour a mismatch).
    for (int i = 0; i < nlocal; i++) {
        e = hipMemsetAsync(cs[i]->pt->d_checksum, 0, 4 * sizeof(unsigned long long), st);
        if (e != hipSuccess) return hip_fail(e, "carma_pt_iterate_sharded: checksum reset");
    }
    for (int i = 1; i < nlocal; i++) {
        e = hipStreamSynchronize(cs[i]->stream);      // work enqueued earlier on a block's own stream (start-up)
        if (e != hipSuccess) return hip_fail(e, "carma_pt_iterate_sharded");
    }
    const unsigned tpb = 64;
    // CARMA_SHARD_STAMPS=1 (measurements only; read once): HIP events between the stages of the first 64 iterations of a call,
    // summed per stage and printed when the call ends -- the boundary budget behind DESIGN.md section 7's pipelining argument
    static const bool stamps_on = [] {
        const char* v = getenv("CARMA_SHARD_STAMPS");
        return v && v[0] == '1';
    }();
    enum { ST_START, ST_RAM, ST_PACK, ST_XFER, ST_SWAP, ST_SWEEP, ST_N };
    std::vector<std::pair<int, hipEvent_t>> stamps;
    auto stamp = [&](int tag, long it) {
        if (!stamps_on || it >= 64) return;
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return;
        (void)hipEventRecord(ev, st);
        stamps.emplace_back(tag, ev);
    };
    long cur_it = 0;
    struct Side {
        int local;      // index of the local block
        int mine;       // its boundary temperature (local index)
        int upper;      // 1: the peer block is the hotter one
        int peer;       // rank that owns the block on the other side
    };
    // one boundary, as seen from this process: one side (the peer is another rank) or both (lower block first)
    auto exchange = [&](const Side* sides, int nsides, unsigned long long iter) -> int {
        for (int a = 0; a < nsides; a++) {
            PtState* s = cs[sides[a].local]->pt;
            hipLaunchKernelGGL(k_shard_pack, dim3((unsigned)((nbuf + tpb - 1) / tpb)), dim3(tpb), 0, st, s->d_theta, s->d_lp, R,
                               s->T, d, sides[a].mine, s->temps[sides[a].mine], s->d_send);
        }
        hipError_t el = hipGetLastError();
        if (el != hipSuccess) return hip_fail(el, "k_shard_pack");
        stamp(ST_PACK, cur_it);
        // Sends and receives between one pair of ranks are matched in issue order; the only pair with two transfers in
        // flight is this rank with itself, where the lower block's data has to land in the upper block's buffer and vice
        // versa -- so every side issues its send, then the receive INTO THE OTHER SIDE'S BUFFER when the peer is this rank.
        ncclResult_t nr = api->GroupStart();
        if (nr != ncclSuccess) return rccl_fail(nr, "ncclGroupStart");
        for (int a = 0; a < nsides; a++) {
            PtState* s = cs[sides[a].local]->pt;
            double* recv_into = nsides == 2 ? cs[sides[1 - a].local]->pt->d_recv : s->d_recv;
            nr = api->Send(s->d_send, nbuf, ncclDouble, sides[a].peer, cm->nccl, st);
            if (nr == ncclSuccess) nr = api->Recv(recv_into, nbuf, ncclDouble, sides[a].peer, cm->nccl, st);
            if (nr != ncclSuccess) {
                (void)api->GroupEnd();
                return rccl_fail(nr, "ncclSend/ncclRecv");
            }
        }
        nr = api->GroupEnd();
        if (nr != ncclSuccess) return rccl_fail(nr, "ncclGroupEnd");
        stamp(ST_XFER, cur_it);
        for (int a = 0; a < nsides; a++) {
            PtState* s = cs[sides[a].local]->pt;
            const unsigned hot_slot = sides[a].upper ? s->slot0 + (unsigned)s->T : s->slot0;      // global slot of the hotter chain
            hipLaunchKernelGGL(k_shard_swap, dim3((unsigned)((R + tpb - 1) / tpb)), dim3(tpb), 0, st, s->d_theta, s->d_lp, R, s->T, d,
                               sides[a].mine, s->d_recv, sides[a].upper, s->temps[sides[a].mine], (unsigned)(s->seed & 0xffffffffu),
                               (unsigned)(s->seed >> 32), iter, s->T_global, s->replica0, hot_slot, s->d_bnd_swaps, s->d_nswap,
                               s->d_checksum + (sides[a].upper ? 1 : 0));
            s->bnd_proposed += (unsigned long long)R;
        }
        el = hipGetLastError();
        if (el != hipSuccess) return hip_fail(el, "k_shard_swap");
        stamp(ST_SWAP, cur_it);
        return CARMA_OK;
    };
    int rc_loop = CARMA_OK;
    for (long it = 0; it < niter && rc_loop == CARMA_OK; it++) {
        const unsigned long long iter = s0->iter;     // index of the iteration about to run (== every block's)
        cur_it = it;
        stamp(ST_START, it);
        // one block: the sampler kernel with its own sweep; several: RAM steps only, the sweep follows piece by piece
        for (int i = 0; i < nlocal && rc_loop == CARMA_OK; i++) rc_loop = pt_enqueue(cs[i], 1, nblocks == 1 ? 1 : 0, 0, nullptr, st);
        if (rc_loop != CARMA_OK) break;
        stamp(ST_RAM, it);
        if (nblocks > 1) {
            // the sweep of the whole ladder, hottest pair first, as far as this process holds a side of it
            for (int i = nlocal - 1; i >= 0 && rc_loop == CARMA_OK; i--) {
                const int gb = rank * nlocal + i;
                PtState* s = cs[i]->pt;
                if (i == nlocal - 1 && gb + 1 < nblocks) {             // the hotter neighbour is another rank's block
                    const Side sd{i, s->T - 1, 1, rank + 1};
                    rc_loop = exchange(&sd, 1, iter);
                    if (rc_loop != CARMA_OK) break;
                }
                if (s->T > 1) {
                    hipLaunchKernelGGL(k_shard_sweep, dim3((unsigned)((R + tpb - 1) / tpb)), dim3(tpb), 0, st, s->d_theta, s->d_lp, R, s->T,
                                       d, s->d_temps, (unsigned)(s->seed & 0xffffffffu), (unsigned)(s->seed >> 32), iter, s->T_global,
                                       s->replica0, s->slot0, s->d_nswap);
                    e = hipGetLastError();
                    if (e != hipSuccess) {
                        rc_loop = hip_fail(e, "k_shard_sweep");
                        break;
                    }
                    stamp(ST_SWEEP, it);
                }
                if (gb > 0) {
                    if (i > 0) {                                        // the colder neighbour is this process's block i - 1
                        const Side both[2] = {{i - 1, cs[i - 1]->pt->T - 1, 1, rank}, {i, 0, 0, rank}};
                        rc_loop = exchange(both, 2, iter);
                    } else {
                        const Side sd{i, 0, 0, rank - 1};
                        rc_loop = exchange(&sd, 1, iter);
                    }
                }
            }
            if (rc_loop != CARMA_OK) break;
        }
        if (save_thin > 0 && d_samples && s0->slot0 == 0 && ((it + 1) % save_thin) == 0) {
            const long sidx = (it + 1) / save_thin - 1;
            if (sidx < sample_cap)
                hipLaunchKernelGGL(k_shard_save, dim3((unsigned)((R * (d + 1) + tpb - 1) / tpb)), dim3(tpb), 0, st, s0->d_theta, s0->d_lp,
                                   R, s0->T, d, sidx, sample_cap, d_samples, d_slp);
        }
    }
    // SELF-CHECK: the two sides of every boundary must have folded the same decisions on the same log-posteriors.  One
    // more (8-byte) exchange per boundary and CALL, in the sweep's order.
    std::vector<unsigned long long> mine_sum, peer_sum;
    if (rc_loop == CARMA_OK && nblocks > 1) {
        ncclResult_t nr = ncclSuccess;
        for (int i = nlocal - 1; i >= 0 && nr == ncclSuccess; i--) {
            const int gb = rank * nlocal + i;
            PtState* s = cs[i]->pt;
            // d_checksum: [0] lower side, [1] upper side of this block, [2] / [3] what the peers of those sides report
            if (i == nlocal - 1 && gb + 1 < nblocks) {
                nr = api->GroupStart();
                if (nr == ncclSuccess) nr = api->Send(s->d_checksum + 1, 1, ncclUint64, rank + 1, cm->nccl, st);
                if (nr == ncclSuccess) nr = api->Recv(s->d_checksum + 3, 1, ncclUint64, rank + 1, cm->nccl, st);
                if (nr == ncclSuccess) nr = api->GroupEnd();
            }
            if (gb > 0 && nr == ncclSuccess) {
                nr = api->GroupStart();
                if (i > 0) {
                    PtState* sl = cs[i - 1]->pt;
                    if (nr == ncclSuccess) nr = api->Send(sl->d_checksum + 1, 1, ncclUint64, rank, cm->nccl, st);
                    if (nr == ncclSuccess) nr = api->Recv(s->d_checksum + 2, 1, ncclUint64, rank, cm->nccl, st);
                    if (nr == ncclSuccess) nr = api->Send(s->d_checksum, 1, ncclUint64, rank, cm->nccl, st);
                    if (nr == ncclSuccess) nr = api->Recv(sl->d_checksum + 3, 1, ncclUint64, rank, cm->nccl, st);
                } else {
                    if (nr == ncclSuccess) nr = api->Send(s->d_checksum, 1, ncclUint64, rank - 1, cm->nccl, st);
                    if (nr == ncclSuccess) nr = api->Recv(s->d_checksum + 2, 1, ncclUint64, rank - 1, cm->nccl, st);
                }
                if (nr == ncclSuccess) nr = api->GroupEnd();
            }
        }
        if (nr != ncclSuccess) rc_loop = rccl_fail(nr, "boundary self-check exchange");
    }
    e = hipStreamSynchronize(st);
    if (e != hipSuccess && rc_loop == CARMA_OK) rc_loop = hip_fail(e, "carma_pt_iterate_sharded");
    if (!stamps.empty()) {
        double sum[ST_N] = {0, 0, 0, 0, 0, 0};
        long cnt[ST_N] = {0, 0, 0, 0, 0, 0};
        for (size_t k = 1; k < stamps.size(); k++) {
            float ms = 0.f;
            if (stamps[k].first != ST_START && hipEventElapsedTime(&ms, stamps[k - 1].second, stamps[k].second) == hipSuccess) {
                sum[stamps[k].first] += 1e3 * ms;
                cnt[stamps[k].first]++;
            }
        }
        const long its = cnt[ST_RAM] ? cnt[ST_RAM] : 1;
        fprintf(stderr, "carma_shard stamps: rank %d, %d local block(s) of %d, R = %d, %ld iterations | per iteration (us): sampler kernel(s) %.1f, "
                        "pack %.1f (%ld), send/recv %.1f (%ld), swap %.1f (%ld), block sweep %.1f (%ld) | per stage (us): pack %.2f, send/recv %.2f, "
                        "swap %.2f, sweep %.2f\n", rank, nlocal, nblocks, R, its, sum[ST_RAM] / its, sum[ST_PACK] / its, cnt[ST_PACK] / its,
                sum[ST_XFER] / its, cnt[ST_XFER] / its, sum[ST_SWAP] / its, cnt[ST_SWAP] / its, sum[ST_SWEEP] / its, cnt[ST_SWEEP] / its,
                cnt[ST_PACK] ? sum[ST_PACK] / cnt[ST_PACK] : 0.0, cnt[ST_XFER] ? sum[ST_XFER] / cnt[ST_XFER] : 0.0,
                cnt[ST_SWAP] ? sum[ST_SWAP] / cnt[ST_SWAP] : 0.0, cnt[ST_SWEEP] ? sum[ST_SWEEP] / cnt[ST_SWEEP] : 0.0);
        for (auto& pr_ : stamps) (void)hipEventDestroy(pr_.second);
    }
    bool any_abort = false;
    for (int i = 0; i < nlocal; i++) {
        bool aborted = false;
        (void)pt_check_abort(cs[i], &aborted);
        if (aborted) {
            any_abort = true;
            if (cs[i]->pt->d_abort) (void)hipMemset(cs[i]->pt->d_abort, 0, sizeof(unsigned));
        }
    }
    if (any_abort && rc_loop == CARMA_OK) {
        set_error("carma_pt_iterate_sharded: a cross-workgroup swap barrier of the sampler kernel timed out");
        rc_loop = CARMA_EHIP;
    }
    if (rc_loop == CARMA_OK && nblocks > 1) {
        for (int i = 0; i < nlocal && rc_loop == CARMA_OK; i++) {
            const int gb = rank * nlocal + i;
            unsigned long long h[4] = {0, 0, 0, 0};
            e = hipMemcpy(h, cs[i]->pt->d_checksum, sizeof h, hipMemcpyDeviceToHost);
            if (e != hipSuccess) {
                rc_loop = hip_fail(e, "boundary self-check");
                break;
            }
            const bool lower_ok = gb == 0 || h[0] == h[2], upper_ok = gb + 1 >= nblocks || h[1] == h[3];
            cs[i]->pt->bnd_check = (lower_ok && upper_ok) ? 1 : -1;
            if (!lower_ok || !upper_ok) {
                set_error("carma_pt_iterate_sharded: rank %d block %d and its %s neighbour took different swap decisions (checksums %016llx vs "
                          "%016llx): the boundary exchange is broken", rank, gb, lower_ok ? "hotter" : "colder",
                          lower_ok ? h[1] : h[0], lower_ok ? h[3] : h[2]);
                rc_loop = CARMA_EHIP;
            }
        }
    }
    if (rc_loop != CARMA_OK) {
        // A failure somewhere inside the loop leaves the blocks at different points of an iteration: the chain states
        // are not a sample of anything any more.  Refuse to continue from them (carma_pt_start / carma_pt_set_chains
        // re-arm the sampler); peers of a failed rank are left waiting in their receive -- tear the communicator down.
        for (int i = 0; i < nlocal; i++) cs[i]->pt->started = false;
    }
    return rc_loop;
}

extern "C" {

int carma_pt_iterate_sharded(carma_ctx* const* shards, int nlocal, long niter, carma_comm* comm)
{
    return iterate_sharded(shards, nlocal, niter, comm, 0, 0, nullptr, nullptr);
}

int carma_pt_sample_sharded(carma_ctx* const* shards, int nlocal, int nsamples, int thin, carma_comm* comm, double* samples,
                            double* logposts)
{
    if (!shards || nlocal < 1 || nsamples < 1 || thin < 1 || !shards[0] || !reinterpret_cast<Ctx*>(shards[0])->pt) {
        set_error("carma_pt_sample_sharded: bad argument");
        return CARMA_EINVAL;
    }
    Ctx* c0 = reinterpret_cast<Ctx*>(shards[0]);
    const bool owner = c0->pt->slot0 == 0;            // this process holds temperature 0: it collects the samples
    if (owner && (!samples || !logposts)) {
        set_error("carma_pt_sample_sharded: the process that owns temperature 0 must pass sample buffers");
        return CARMA_EINVAL;
    }
    double *d_s = nullptr, *d_l = nullptr;
    const size_t R = (size_t)c0->pt->R, d = (size_t)c0->d;
    if (owner) {
        hipError_t e = hipSetDevice(c0->device);
        if (e == hipSuccess) e = dev_malloc(&d_s, sizeof(double) * R * nsamples * d);
        if (e == hipSuccess) e = dev_malloc(&d_l, sizeof(double) * R * nsamples);
        if (e != hipSuccess) {
            if (d_s) (void)dev_free(d_s);
            return hip_fail(e, "carma_pt_sample_sharded: sample buffers");
        }
    }
    int rc = iterate_sharded(shards, nlocal, (long)nsamples * thin, comm, thin, nsamples, d_s, d_l);
    if (owner) {
        hipError_t e = hipSuccess;
        if (rc == CARMA_OK) e = hipMemcpy(samples, d_s, sizeof(double) * R * nsamples * d, hipMemcpyDeviceToHost);
        if (rc == CARMA_OK && e == hipSuccess) e = hipMemcpy(logposts, d_l, sizeof(double) * R * nsamples, hipMemcpyDeviceToHost);
        (void)dev_free(d_s);
        (void)dev_free(d_l);
        if (e != hipSuccess) rc = hip_fail(e, "carma_pt_sample_sharded: D2H");
    }
    return rc;
}

int carma_pt_boundary_check(carma_ctx* h)
{
    if (!h || !reinterpret_cast<Ctx*>(h)->pt) return CARMA_EINVAL;
    return reinterpret_cast<Ctx*>(h)->pt->bnd_check;
}

// ---- debug pair for the deterministic sampler tests (tests/test_gpu_sampler_steps.py) -------------------------------
// The sampler's variates come from a counter-based generator keyed by (seed, global chain slot, iteration, purpose), so
// the numbers chain (replica, temperature) WILL use at iteration `iter` can be produced on demand -- by the device, with
// the very functions the kernels call (the host's libm rounds log / sincos differently in the last place).
__global__ void k_pt_debug_draws(unsigned seed0, unsigned seed1, unsigned chain, unsigned long long iter, int d, double* out)
{
    const RngKey key{seed0, seed1, chain};
    const int j = threadIdx.x;
    if (j < d) out[j] = rng_student_t8(key, iter, (uint32_t)j);          // unit proposal (steps.cpp:65-69)
    if (j == d) out[d] = rng_uniform(key, iter, RNG_ACCEPT, 0);          // Metropolis uniform (steps.cpp:48)
    if (j == d + 1) out[d + 1] = rng_uniform(key, iter, RNG_SWAP, 0);    // swap uniform of the pair (this, next colder) (steps.hpp:333)
}

int carma_pt_debug_draws(carma_ctx* h, int replica, int temperature, unsigned long long iter, double* z, double* u_accept,
                         double* u_swap)
{
    if (!h || !reinterpret_cast<Ctx*>(h)->pt || !z) return CARMA_EINVAL;
    Ctx* c = reinterpret_cast<Ctx*>(h);
    PtState* s = c->pt;
    if (replica < 0 || replica >= s->R || temperature < 0 || temperature >= s->T) {
        set_error("carma_pt_debug_draws: chain (%d, %d) outside %d x %d", replica, temperature, s->R, s->T);
        return CARMA_EINVAL;
    }
    const int d = c->d;
    hipError_t e = hipSetDevice(c->device);
    double* dbuf = nullptr;
    if (e == hipSuccess) e = dev_malloc(&dbuf, sizeof(double) * (d + 2));
    if (e != hipSuccess) return hip_fail(e, "carma_pt_debug_draws");
    const unsigned chain = (s->replica0 + (unsigned)replica) * s->T_global + s->slot0 + (unsigned)temperature;
    hipLaunchKernelGGL(k_pt_debug_draws, dim3(1), dim3(64), 0, c->stream, (unsigned)(s->seed & 0xffffffffu), (unsigned)(s->seed >> 32),
                       chain, iter, d, dbuf);
    std::vector<double> hb(d + 2);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(hb.data(), dbuf, sizeof(double) * (d + 2), hipMemcpyDeviceToHost);
    (void)dev_free(dbuf);
    if (e != hipSuccess) return hip_fail(e, "carma_pt_debug_draws");
    std::memcpy(z, hb.data(), sizeof(double) * d);
    if (u_accept) *u_accept = hb[d];
    if (u_swap) *u_swap = hb[d + 1];
    return CARMA_OK;
}

// the Cholesky factors of the proposal scale matrices, [R][T][d*d] (upper triangular, row-major; AdaptiveMetro::chol_factor_)
int carma_pt_get_factor(carma_ctx* h, double* chol)
{
    if (!h || !reinterpret_cast<Ctx*>(h)->pt || !chol) return CARMA_EINVAL;
    Ctx* c = reinterpret_cast<Ctx*>(h);
    PtState* s = c->pt;
    hipError_t e = pt_sync_factor(c, c->stream);            // (lane sampler: the factors live in its working state between calls)
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(chol, s->d_chol, sizeof(double) * (size_t)s->T * s->R * c->d * c->d, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "carma_pt_get_factor");
    return CARMA_OK;
}

int carma_pt_set_factor(carma_ctx* h, const double* chol)
{
    if (!h || !reinterpret_cast<Ctx*>(h)->pt || !chol) return CARMA_EINVAL;
    Ctx* c = reinterpret_cast<Ctx*>(h);
    PtState* s = c->pt;
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(s->d_chol, chol, sizeof(double) * (size_t)s->T * s->R * c->d * c->d, hipMemcpyHostToDevice);
    if (e != hipSuccess) return hip_fail(e, "carma_pt_set_factor");
    pt_factor_written(c);
    return CARMA_OK;
}

int carma_pt_sweep(carma_ctx* h)
{
    if (!h || !reinterpret_cast<Ctx*>(h)->pt || !reinterpret_cast<Ctx*>(h)->pt->started || reinterpret_cast<Ctx*>(h)->pt->iter == 0) {
        set_error("carma_pt_sweep: no iteration to sweep after");
        return CARMA_EINVAL;
    }
    Ctx* c = reinterpret_cast<Ctx*>(h);
    PtState* s = c->pt;
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    if (s->T > 1) {
        hipLaunchKernelGGL(k_shard_sweep, dim3((unsigned)((s->R + 63) / 64)), dim3(64), 0, c->stream, s->d_theta, s->d_lp, s->R, s->T, c->d,
                           s->d_temps, (unsigned)(s->seed & 0xffffffffu), (unsigned)(s->seed >> 32), s->iter - 1, s->T_global,
                           s->replica0, s->slot0, s->d_nswap);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return hip_fail(e, "carma_pt_sweep");
    }
    return CARMA_OK;
}

int carma_pt_boundary_stats(carma_ctx* h, unsigned long long* proposed, unsigned long long* accepted)
{
    if (!h || !reinterpret_cast<Ctx*>(h)->pt) return CARMA_EINVAL;
    PtState* s = reinterpret_cast<Ctx*>(h)->pt;
    unsigned acc = 0;
    if (s->d_bnd_swaps) {
        hipError_t e = hipMemcpy(&acc, s->d_bnd_swaps, sizeof(unsigned), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return hip_fail(e, "carma_pt_boundary_stats");
    }
    if (proposed) *proposed = s->bnd_proposed;
    if (accepted) *accepted = acc;
    return CARMA_OK;
}

}  // extern "C"
