// carma_host.h -- host-side state behind the C ABI (carma_capi.hip, carma_pt_host.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "carma_launch.h"
#include "carma_types.h"

// Device allocations of the library go through these two.  CARMA_DEBUG_GUARD=1 (read once; a test switch, off by default)
// gives every allocation a virtual-memory mapping of its own whose END is the end of the buffer, with unmapped address space
// behind it: a kernel that reads or writes past a buffer faults instead of getting away with it (round 4: a read of two
// doubles past the parameter batch had lived in the lane-group kernels for two rounds, caught only when a batch happened to
// end on a page boundary).  tests/test_gpu_parity.py runs a cross-section of the entry points in that mode.
hipError_t carma_dev_malloc(void** p, size_t n);
hipError_t carma_dev_free(void* p);
template <class T>
static inline hipError_t dev_malloc(T** p, size_t n)          // (typed front end: dev_malloc(&d_x, bytes))
{
    return carma_dev_malloc(reinterpret_cast<void**>(p), n);
}
static inline hipError_t dev_free(void* p) { return carma_dev_free(p); }

namespace carma {

// Parallel-tempering sampler state of one context (carma_pt_host.hip, carma_shard.hip)
struct PtState {
    int T = 0, R = 0;
    unsigned T_global = 0, slot0 = 0, replica0 = 0;
    int maxiter = 0;
    uint64_t seed = 0;
    unsigned long long iter = 0;
    std::vector<double> temps;
    double *d_temps = nullptr, *d_theta = nullptr, *d_lp = nullptr, *d_chol = nullptr;
    // lane sampler (use_lane): the factors live in the chain-minor scratch between calls
    bool lane_factor_loaded = false;   // the scratch holds the current factors (else: take them from d_chol at the next launch)
    bool chol_stale = false;           // d_chol is behind the scratch (pt_sync_factor brings it up to date)
    bool ext_state = false;
    unsigned *d_nacc = nullptr, *d_nswap = nullptr;
    double *d_samples = nullptr, *d_slp = nullptr;
    long cap = 0;
    bool started = false;
    unsigned long long stat_iters = 0;
    // row-variant kernel (k_pt_row): ladders spread over wpl workgroups, swap through global staging
    bool use_row = false;
    int wpl = 0;
    unsigned long long* d_stage = nullptr;      // tagged staging words of the swap step (PtRowSync::stage)
    unsigned long long epoch = 0;               // launches so far (PtRowSync::epoch)
    unsigned* d_abort = nullptr;
    double* d_backup = nullptr;         // chain state before the chunk in flight (theta, logpost, chol): abort recovery
    // large ensembles (carma_pt_lane.hip): one chain per lane, an iteration as propose kernel + batched log-density + finish kernel
    bool use_lane = false;
    double* d_lane_scratch = nullptr;   // chain-minor working state (current value, R^T z, packed factor, proposals, ...)
    // ladder sharded across ranks (carma_shard.hip): boundary staging and statistics
    double *d_send = nullptr, *d_recv = nullptr;       // [R][d+1] each
    unsigned* d_bnd_swaps = nullptr;                   // [1] accepted boundary swaps (this block's side)
    unsigned long long bnd_proposed = 0;
    unsigned long long* d_checksum = nullptr;          // [4] folds of the boundary decisions: lower / upper side, the peers' reports
    int bnd_check = 0;                                 // result of the last self-check: 1 agreed, -1 differed, 0 none yet
};

struct Ctx {
    int device = 0;
    int p = 0, q = 0, d = 0, n = 0;
    std::vector<double> t, y, yerr;   // after sort/dedup
    Prior pr{};
    bool repeated_dt = false;         // >= 25 % of the time steps equal their predecessor (regular cadence)
    bool window_ok = false;           // the series suits the windowed wave pipeline (carma_capi.hip, carma_ctx_create)
    int window2 = 0;                  // the TWO-SIDED window pipeline: 2 = suits it, 1 = with a CU per workgroup only (carma_types.h)
    int series_flags() const
    {
        return (repeated_dt ? SERIES_REPEATED_DT : 0) | (window_ok ? SERIES_WINDOW_OK : 0) | (window2 == 2 ? SERIES_WINDOW2_OK : 0) |
               (window2 >= 1 ? SERIES_WINDOW2_SMALL : 0);
    }
    double* d_series = nullptr;       // records {dt, y, yerr^2, t}[n + 16 pads], then yerr^2[n + 16], y[n + 16] (carma_types.h)
    double* d_theta = nullptr;        // staging for the host-pointer entry points
    double* d_out = nullptr;
    double* h_stage = nullptr;        // pinned host staging: [cap * d] parameter vectors followed by [cap] results
    int cap = 0;
    hipStream_t stream = nullptr;
    PtState* pt = nullptr;
    int ensure_staging(int B);
};

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
int select_device(int device);
void pt_state_free(Ctx* c);
// Enqueue `niter` iterations of the sampler kernel on `st` (no synchronisation).  thin > 0: save the coldest chain
// every `thin` iterations starting at sample index *save_offset (advanced).
int pt_enqueue(Ctx* c, long niter, int do_exchange, int thin, long* save_offset, hipStream_t st);
// after the stream has been synchronised: did a cross-workgroup rendezvous of the row kernel time out?
int pt_check_abort(Ctx* c, bool* aborted);
// The lane sampler keeps the proposal factors in its chain-minor working state between calls.  pt_sync_factor: bring the chain-major
// array d_chol up to date before it is read (enqueued on st; no-op for the other kernels); pt_factor_written: d_chol was written, the
// next launch reloads.
hipError_t pt_sync_factor(Ctx* c, hipStream_t st);
void pt_factor_written(Ctx* c);

}  // namespace carma
