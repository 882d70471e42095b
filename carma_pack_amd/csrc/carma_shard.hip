// carma_shard.hip -- ONE temperature ladder sharded across GPUs (BASELINE config 4).
//
// Every rank (one process per GPU) owns a contiguous block of the ladder's temperatures for all R replicas and
// advances it with the sampler kernel of carma_pt.hip (RAM steps + the swaps inside the block).  The only coupling
// between ranks is the adjacent-temperature exchange across a block boundary (ExchangeStep::DoStep,
// /root/reference/src/include/steps.hpp:318-362; wiring src/carmcmc.cpp:147-157): per iteration and active boundary
//     k_shard_pack   gathers (theta[d], logpost) of the boundary chain of all R replicas   -> send buffer, R (d+1) doubles
//     ncclSend / ncclRecv (RCCL, grouped) with the rank on the other side of the boundary  <= ~20 KB: latency bound
//     k_shard_swap   draws the swap uniform and applies the exchange
// all enqueued on the sampler's stream: no host synchronisation, no host RNG, no host<->device copy per iteration.
// The uniform is Philox keyed by (seed, global slot of the hotter chain, iteration), so both sides compute the same
// decision from the same bits and nothing but the chain states crosses the link.  Boundaries alternate even / odd per
// iteration (deterministic even-odd tempering), so a rank that holds a single temperature never has both of its
// boundaries active at once.  A process may own several consecutive blocks (`nlocal`): a boundary between two of its
// own blocks goes through ncclSend/ncclRecv to itself, which is what lets a one-GPU box exercise the RCCL path.
//
// RCCL is bound at run time (dlopen of librccl.so.1: the copy PyTorch already loaded when there is one), so the
// library itself has no link-time dependency on it; without RCCL carma_comm_* fail loudly.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cmath>
#include <cstring>
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
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* nm : names) {
            api.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
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
    }
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

// ExchangeStep::DoStep (steps.hpp:318-362) for the pair (hot_slot, hot_slot - 1), one thread per replica.  `upper`: this
// block holds the COLDER chain of the pair (its hottest temperature, `mine` = T - 1), the peer the hotter one.
// alpha = (lp_cold - lp_hot) (1/T_hot - 1/T_cold)  (steps.hpp:331-332);  accept when log u < alpha (NaN rejects, :336-338).
__global__ void k_shard_swap(double* __restrict__ theta, double* __restrict__ lp, int R, int T, int d, int mine,
                             const double* __restrict__ recv, int upper, double t_mine, unsigned seed0, unsigned seed1,
                             unsigned long long iter, unsigned T_global, unsigned replica0, unsigned hot_slot,
                             unsigned* __restrict__ nswap)
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
    if (logu < a) {
        for (int j = 0; j < d; j++) theta[((size_t)r * T + mine) * d + j] = recv[(size_t)r * (d + 1) + j];
        lp[(size_t)r * T + mine] = ot_lp;
        atomicAdd(nswap, 1u);
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
            e = hipMalloc(&s->d_send, sizeof(double) * nbuf);
            if (e == hipSuccess) e = hipMalloc(&s->d_recv, sizeof(double) * nbuf);
            if (e == hipSuccess) e = hipMalloc(&s->d_bnd_swaps, sizeof(unsigned));
            if (e == hipSuccess) e = hipMemset(s->d_bnd_swaps, 0, sizeof(unsigned));
            if (e != hipSuccess) return hip_fail(e, "carma_pt_iterate_sharded: boundary buffers");
        }
    }
    // every block of this process on ONE stream (block 0's): kernels, RCCL calls and swap kernels are ordered by it
    hipStream_t st = cs[0]->stream;
    for (int i = 1; i < nlocal; i++) {
        e = hipStreamSynchronize(cs[i]->stream);      // work enqueued earlier on a block's own stream (start-up)
        if (e != hipSuccess) return hip_fail(e, "carma_pt_iterate_sharded");
    }
    const unsigned tpb = 64;
    for (long it = 0; it < niter; it++) {
        const unsigned long long iter = s0->iter;     // index of the iteration about to run (== every block's)
        for (int i = 0; i < nlocal; i++) {
            int rc = pt_enqueue(cs[i], 1, 1, 0, nullptr, st);
            if (rc != CARMA_OK) return rc;
        }
        const bool save_now = save_thin > 0 && d_samples && s0->slot0 == 0 && ((it + 1) % save_thin) == 0;
        auto save = [&]() {
            if (!save_now) return;
            const long sidx = (it + 1) / save_thin - 1;
            if (sidx < sample_cap)
                hipLaunchKernelGGL(k_shard_save, dim3((unsigned)((R * (d + 1) + tpb - 1) / tpb)), dim3(tpb), 0, st, s0->d_theta, s0->d_lp,
                                   R, s0->T, d, sidx, sample_cap, d_samples, d_slp);
        };
        if (nblocks == 1) {
            save();
            continue;
        }
        const int parity = (int)(iter & 1ull);
        // boundary k sits between block k and block k + 1; active when k has the iteration's parity.  A block has at
        // most one active boundary per iteration, so one send and one receive buffer per block are enough.
        struct Side {
            int local;      // index of the local block
            int mine;       // its boundary temperature (local index)
            int upper;      // 1: the peer block is the hotter one
            int peer;       // rank that owns the block on the other side
            int k;          // boundary
        };
        std::vector<Side> sides;
        for (int i = 0; i < nlocal; i++) {
            const int gb = rank * nlocal + i;
            if (gb + 1 < nblocks && (gb & 1) == parity) sides.push_back({i, cs[i]->pt->T - 1, 1, (gb + 1) / nlocal, gb});
            if (gb > 0 && ((gb - 1) & 1) == parity) sides.push_back({i, 0, 0, (gb - 1) / nlocal, gb - 1});
        }
        if (sides.empty()) {
            save();
            continue;
        }
        for (const Side& sd : sides) {
            PtState* s = cs[sd.local]->pt;
            hipLaunchKernelGGL(k_shard_pack, dim3((unsigned)((nbuf + tpb - 1) / tpb)), dim3(tpb), 0, st, s->d_theta, s->d_lp, R,
                               s->T, d, sd.mine, s->temps[sd.mine], s->d_send);
        }
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "k_shard_pack");
        // Sends and receives between one pair of ranks are matched in issue order; the only pair with two transfers in
        // flight is this rank with itself (a boundary between two of its own blocks), where the lower block's data has
        // to land in the upper block's buffer and vice versa -- so every side issues its send, then the receive INTO THE
        // OTHER SIDE'S BUFFER when the peer is this rank (sides of one boundary are adjacent in `sides`: lower first).
        ncclResult_t nr = api->GroupStart();
        if (nr != ncclSuccess) return rccl_fail(nr, "ncclGroupStart");
        for (size_t a = 0; a < sides.size(); a++) {
            const Side& sd = sides[a];
            PtState* s = cs[sd.local]->pt;
            double* recv_into = s->d_recv;
            if (sd.peer == rank) {
                const size_t other = (sd.upper ? a + 1 : a - 1);      // the other side of the same boundary
                recv_into = cs[sides[other].local]->pt->d_recv;
            }
            nr = api->Send(s->d_send, nbuf, ncclDouble, sd.peer, cm->nccl, st);
            if (nr == ncclSuccess) nr = api->Recv(recv_into, nbuf, ncclDouble, sd.peer, cm->nccl, st);
            if (nr != ncclSuccess) {
                (void)api->GroupEnd();
                return rccl_fail(nr, "ncclSend/ncclRecv");
            }
        }
        nr = api->GroupEnd();
        if (nr != ncclSuccess) return rccl_fail(nr, "ncclGroupEnd");
        for (const Side& sd : sides) {
            PtState* s = cs[sd.local]->pt;
            const unsigned hot_slot = sd.upper ? s->slot0 + (unsigned)s->T : s->slot0;      // global slot of the hotter chain
            hipLaunchKernelGGL(k_shard_swap, dim3((unsigned)((R + tpb - 1) / tpb)), dim3(tpb), 0, st, s->d_theta, s->d_lp, R, s->T, d,
                               sd.mine, s->d_recv, sd.upper, s->temps[sd.mine], (unsigned)(s->seed & 0xffffffffu),
                               (unsigned)(s->seed >> 32), iter, s->T_global, s->replica0, hot_slot, s->d_bnd_swaps);
            s->bnd_proposed += (unsigned long long)R;
        }
        save();
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "k_shard_swap");
    }
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hip_fail(e, "carma_pt_iterate_sharded");
    for (int i = 0; i < nlocal; i++) {
        bool aborted = false;
        int rc = pt_check_abort(cs[i], &aborted);
        if (rc != CARMA_OK) return rc;
        if (aborted) {
            set_error("carma_pt_iterate_sharded: a cross-workgroup swap barrier of the sampler kernel timed out");
            return CARMA_EHIP;
        }
    }
    return CARMA_OK;
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
        if (e == hipSuccess) e = hipMalloc(&d_s, sizeof(double) * R * nsamples * d);
        if (e == hipSuccess) e = hipMalloc(&d_l, sizeof(double) * R * nsamples);
        if (e != hipSuccess) {
            if (d_s) (void)hipFree(d_s);
            return hip_fail(e, "carma_pt_sample_sharded: sample buffers");
        }
    }
    int rc = iterate_sharded(shards, nlocal, (long)nsamples * thin, comm, thin, nsamples, d_s, d_l);
    if (owner) {
        hipError_t e = hipSuccess;
        if (rc == CARMA_OK) e = hipMemcpy(samples, d_s, sizeof(double) * R * nsamples * d, hipMemcpyDeviceToHost);
        if (rc == CARMA_OK && e == hipSuccess) e = hipMemcpy(logposts, d_l, sizeof(double) * R * nsamples, hipMemcpyDeviceToHost);
        (void)hipFree(d_s);
        (void)hipFree(d_l);
        if (e != hipSuccess) rc = hip_fail(e, "carma_pt_sample_sharded: D2H");
    }
    return rc;
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
