// carma_launch.h -- host-callable launchers implemented in carma_kernels.hip / carma_pt.hip
#pragma once
#include <hip/hip_runtime.h>

#include "carma_types.h"

namespace carma {

// compute units of the current device (cached per device)
int device_cus();

// Launch-shape switches that measurements and the parity tests move: read from the environment ONCE, when the
// first launch asks (CARMA_TUNE_WIN_ROWS, CARMA_TUNE_WIN2_EVALS, CARMA_TUNE_PT_ROW_WIN), held in atomics, and moved afterwards through
// carma_tune_set only -- no getenv() in the launch path (it raced with setenv() in multi-threaded callers).  TUNE_UNSET: the default.
enum { TUNE_WIN_ROWS = 0, TUNE_WIN2_EVALS = 1, TUNE_PT_ROW_WIN = 2, TUNE_COUNT = 3 };
constexpr long TUNE_UNSET = -0x7fffffffffffffffL - 1;
long tune_get(int which);                    // TUNE_UNSET or the override
int tune_set(const char* name, long value);  // "WIN_ROWS" / "WIN2_EVALS" / "PT_ROW_WIN" (or the full CARMA_TUNE_ name); 0 or -1

// repeated_dt: a good part of the series' time steps equal their predecessor (regular cadence): the throughput kernels
// then run the variant that re-uses the transition factors of such steps (carma_core.h, RhoInline DTC)
hipError_t launch_logdens_carma(int p, const double* theta, int B, int d, int q, const double4* series, int n,
                                const Prior& pr, int ignore_prior, double* out, hipStream_t st, int series_flags = 0);
// name of the kernel launch_logdens_* picks for B evaluations of a series of n points (as rocprofv3 prints it, up to
// the namespace and the argument list)
int logdens_kernel_name(int p, long B, int n, char* buf, int len, int series_flags = 0);
hipError_t launch_logdens_car1(const double* theta, int B, const double4* series, int n, const Prior& pr, double* out,
                               hipStream_t st);
hipError_t launch_kfilter_carma(int p, const double* om_re_im, const double* ma, double sigsqr, const double4* series,
                                int n, double* mean, double* var, int* singular, hipStream_t st);
hipError_t launch_kfilter_car1(double sigsqr, double omega, const double4* series, int n, double* mean, double* var,
                               hipStream_t st);

// Filter() of B models in one launch (one model per lane): par = [B][3 p + 2] (roots re/im in normalised order, p MA
// coefficients, sigsqr, mu), mv = scratch of 2 n (B + 64) doubles; mean / var = [B][n] (device)
hipError_t launch_kfilter_batch(int p, const double* par, int B, const double4* series, int n, double* mv, int* singular,
                                double* mean, double* var, hipStream_t st);
hipError_t launch_predict_carma(int p, const double* om_re_im, const double* ma, double sigsqr, const double4* series,
                                int n, const double* tpred, int M, double* pmean, double* pvar, int* singular,
                                hipStream_t st);
hipError_t launch_predict_car1(double sigsqr, double omega, const double4* series, int n, const double* tpred, int M,
                               double* pmean, double* pvar, hipStream_t st);

// npaths independent paths of the process at n sorted times (carma_simulate.h); out = [npaths][n]
hipError_t launch_simulate_carma(int p, const double* om_re_im, const double* ma, double sigsqr, const double* times, int n,
                                 int npaths, unsigned seed0, unsigned seed1, unsigned path0, double* out, int* singular,
                                 hipStream_t st);
hipError_t launch_simulate_car1(double sigsqr, double omega, const double* times, int n, int npaths, unsigned seed0,
                                unsigned seed1, unsigned path0, double* out, hipStream_t st);

// one chunk of the persistent PT sampler kernel (carma_pt.hip)
hipError_t launch_pt(int p, const PtLaunch& L, const double4* series, const Prior& pr, const double* temps,
                     double* theta, double* logpost, double* chol, unsigned* naccept, unsigned* nswap, double* samples,
                     double* sample_lp, hipStream_t st);
size_t pt_lds_bytes(int P, int d, int T, int* nthreads_out, int* pc_out);
// row variant (one chain per 16-lane DPP row, ladders spread over several workgroups): number of
// workgroups that can be resident at once for this (p, d, T), 0 if the variant does not apply
long pt_row_capacity(int p, int d, int T, int n);
int pt_row_last_pipeline();     // 0 / 1 / 2 as carma_pt_row_pipeline (include/carma_mi355.h); -1 before the first launch
hipError_t launch_pt_row(int p, const PtLaunch& L, const PtRowSync& S, const double4* series, const Prior& pr,
                         const double* temps, double* theta, double* logpost, double* chol, unsigned* naccept,
                         unsigned* nswap, double* samples, double* sample_lp, hipStream_t st);

// sampler for large ensembles (carma_pt_lane.hip): one chain per lane, an iteration = propose kernel + the batched
// log-density launch above + finish kernel; a ladder's T <= 64 chains are T consecutive lanes.  Enqueues L.niter
// iterations on st.  scratch: pt_lane_scratch_doubles(d, T * R) doubles of device memory (chain-minor working state).
size_t pt_lane_scratch_doubles(int d, long nchain);
hipError_t launch_pt_lane(int p, const PtLaunch& L, double* scratch, const double4* series, const Prior& pr,
                          const double* temps, double* theta, double* logpost, double* chol, unsigned* naccept,
                          unsigned* nswap, double* samples, double* sample_lp, int series_flags, bool load_factor, hipStream_t st);
// the lane sampler keeps the proposal factors in its chain-minor working state between calls; this writes them to the chain-major
// array the other entry points read (carma_pt_get_factor, the shard packers): enqueued on st
hipError_t pt_lane_store_factor(int d, int T, int R, double* scratch, double* chol, hipStream_t st);

}  // namespace carma
