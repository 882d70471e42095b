/*
 * carma_mi355.h -- C ABI of libcarma_mi355.so, the MI355X (gfx950) implementation of
 * carma_pack's Kalman-filter log-likelihood path.
 *
 * This is the drop-in boundary: each entry point replaces one binding of the reference's
 * Boost.Python module `carmcmc._carmcmc` (src/boost_python_wrapper.cpp:28-101) or the C++
 * method behind it.  Plain pointers and sizes only; vectors are copied, nothing is retained
 * except what a context owns.  All arithmetic is IEEE FP64.
 *
 * Conventions
 *   - return value 0 = success, negative = error (CARMA_E*); carma_last_error() has the text.
 *   - numerical failure is reported IN BAND, exactly as the reference does: a log-density of
 *     -INFINITY for a prior-bound violation or a repeated AR root (the reference's singular Vandermonde solve)
 *     (src/include/carpack.hpp:134-138,154-164), NaN where the reference's arithmetic gives NaN.
 *   - theta layout (src/carpack.cpp:205-207,432): theta[0]=sigma_y, theta[1]=measurement-error
 *     scale, theta[2]=mu, theta[3..3+p) log quadratic-factor coefficients of the AR polynomial,
 *     theta[3+p..3+p+q) same for the MA polynomial.  CAR(1) (p==1): theta[3]=ln(omega), d=4.
 *   - "_dev" entry points take DEVICE pointers (hipMalloc'ed or a torch tensor's data_ptr) and a
 *     hipStream_t passed as void*; they only enqueue work.  The others take host pointers and
 *     return when the result is in host memory.
 *   - There is no CPU fallback: without a usable gfx950 device every compute call fails with
 *     CARMA_ENODEV.
 *   - ILL-CONDITIONED parameter vectors (deliberate, documented deviation).  The reference forms the constants of the
 *     recursion through an LU solve of the Vandermonde system of the AR roots (KalmanFilterp::Reset, src/kfilter.cpp:157-158)
 *     and p-term sums that cancel when roots cluster -- the prior admits roots 1e-4 apart (src/carpack.cpp:330), where
 *     cond(EigenMat) reaches 1e13 and the reference's own arithmetic is 1e-10 ... 1e-3 away from the exact value of its
 *     formulas.  This library evaluates the same quantities by closed-form products: on such
 *     vectors it returns the closed-form value, NOT the LU's -- within 1e-10 of the reference wherever the reference is
 *     itself within 1e-10 of the exact value, and never further from the exact value than the reference elsewhere
 *     (the tests arbitrate such entries against a quad-precision evaluation and print a census per class of chain:
 *     tests/helpers.py parity_census).
 */
#ifndef CARMA_MI355_H
#define CARMA_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CARMA_OK 0
#define CARMA_EINVAL (-22)
#define CARMA_ENODEV (-19)
#define CARMA_ENOMEM (-12)
#define CARMA_EHIP (-5)

#define CARMA_PMAX 7

typedef struct carma_ctx carma_ctx;
typedef struct carma_comm carma_comm;
typedef struct carma_kf carma_kf;

/* Library / device queries. */
const char* carma_version(void);
const char* carma_last_error(void);
int carma_device_count(void);                      /* 0 when no HIP device is visible */

/*
 * Model context == one CAR1 / CARp / CARMA parameter object of the reference
 * (ctors src/include/carpack.hpp:56-72,254-262,289-298,374-379): copies the series, sorts it by
 * time and drops duplicate times (KalmanFilter::init, src/include/kfilter.hpp:43-76), sets the
 * prior bounds (SetPrior, carpack.hpp:201-207: max_freq = 1/min dt, min_freq = 1/(tmax-tmin))
 * and uploads the series to HBM.  p==1 selects CAR(1) (q must be 0); 2<=p<=7 needs q<p.
 * device = HIP device ordinal.  Returns NULL on error.
 */
carma_ctx* carma_ctx_create(const double* time, const double* y, const double* yerr, int n,
                            int p, int q, double max_stdev, int device);
void carma_ctx_destroy(carma_ctx* h);

int carma_ctx_n(const carma_ctx* h);               /* length after sort/dedup */
int carma_ctx_dim(const carma_ctx* h);             /* d = 3+p+q, or 4 for CAR(1) */
int carma_ctx_get_data(const carma_ctx* h, double* time, double* y, double* yerr);
int carma_ctx_get_prior(const carma_ctx* h, double* out3 /* max_stdev, max_freq, min_freq */);
int carma_ctx_set_prior(carma_ctx* h, double max_stdev);          /* SetPrior, carpack.hpp:201 */

/*
 * Batched CARMA_Base::LogDensity (src/include/carpack.hpp:131-176) == getLogDensity
 * (boost_python_wrapper.cpp:51,59,68) for B parameter vectors, theta = [B][d] row-major.
 * ignore_prior != 0 is SetMLE(true) (carpack.hpp:232; skips the CARp bounds only, the log
 * prior is still added, carpack.hpp:173).  out = [B].
 */
int carma_logdensity_batch(carma_ctx* h, const double* theta, int B, int ignore_prior, double* out);
int carma_logdensity_batch_dev(carma_ctx* h, const double* d_theta, int B, int ignore_prior,
                               double* d_out, void* stream);

/*
 * CarmaModel.get_mle (src/carmcmc/carma_pack.py:195-260: `ntrials` separate scipy L-BFGS-B searches, one FFI crossing
 * per function evaluation) as ONE call: B bounded quasi-Newton searches on f(x) = -LogDensity(x) advanced in lock-step
 * -- per iteration one batched launch evaluates the central-difference stencils of every active start, one more eight
 * backtracking step lengths of every start; projected L-BFGS update (memory `mem`), stopping rules of L-BFGS-B
 * (projected gradient <= gtol; relative decrease <= ftol, three iterations in a row after one restart of the memory).
 * x0 = [B][d] starts; lo / hi = [d] box (NULL or non-finite entries = unbounded); maxiter per start; fd_step = relative
 * finite-difference step; ignore_prior as carma_logdensity_batch (SetMLE(true), carma_pack.py:242).
 * Outputs: x = [B][d], fun = [B] (-LogDensity at x), and optionally nit / nfev / status = [B]
 * (0 converged on the gradient, 1 converged on f, 2 maxiter reached, 3 line search failed).
 */
int carma_mle_batched(carma_ctx* h, const double* x0, int B, const double* lo, const double* hi, int maxiter, int mem,
                      double ftol, double gtol, double fd_step, int ignore_prior, double* x, double* fun, int* nit,
                      int* nfev, int* status);

/* Which kernel a launch of B evaluations on this context takes (the launch shape depends on B: DESIGN.md section 3);
 * the name as rocprofv3 lists it, without namespace and argument list.  For measurement scripts. */
int carma_logdensity_kernel_name(const carma_ctx* h, int B, char* buf, int len);

/* Launch-shape switches for measurements and the parity tests (no counterpart in the reference): "WIN_ROWS", "WIN2_EVALS",
 * "PT_ROW_WIN" -- the CARMA_TUNE_* environment variables of the same names give their initial values, read once when the first
 * launch asks; afterwards only this call moves them (process-wide, thread-safe).  CARMA_EINVAL for an unknown name. */
int carma_tune_set(const char* name, long value);

/* getLogPrior (carpack.hpp:118-126, wrapper :50,58,67); host arithmetic, one vector. */
double carma_logprior(const carma_ctx* h, const double* theta);

/*
 * KalmanFilterp(time,y,yerr,sigsqr,omega,ma).Filter() + GetMean()/GetVar()
 * (src/include/kfilter.hpp:303-334,126-132; wrapper :91-101).  y is already centred, yerr is
 * used as given; omega_re_im = [p][2]; ma has p entries (zero padded by the caller or here if
 * nma < p, kfilter.hpp:318-320).  The AR roots may come in any order but must be closed under conjugation (real
 * roots, conjugate pairs): anything else is not a real-valued process and is CARMA_EINVAL.  The series is
 * sorted/deduplicated first; *n_out receives the resulting length and mean/var must have room for n values.
 * Returns 1 (and fills nothing useful) for a repeated AR root (singular eigenvector matrix) -- the reference
 * throws std::runtime_error there.
 */
int carma_kfilter_carma(const double* time, const double* y, const double* yerr, int n, int p,
                        double sigsqr, const double* omega_re_im, const double* ma, int nma,
                        double* mean, double* var, int* n_out, int device);
/*
 * Filter() of nmodels CARMA(p, nma-1) models on ONE series in one launch (one model per lane): what
 * a caller looping KalmanFilterp(...).Filter() over posterior samples does (carma_pack.py:374-386 builds
 * one filter per call).  sigsqr [nmodels], omega_re_im [nmodels][p][2], ma [nmodels][nma] (ma[.][0] = 1),
 * mu [nmodels] or NULL (0): subtracted from y and added back to mean, as carma_pack.py:399-400 / :442 do
 * around the filter.  mean, var: [nmodels][n_out] row-major (allocate [nmodels][n]).  singular
 * [nmodels] or NULL: 1 where a model has coincident roots (its rows are then not meaningful: the reference's solve throws).
 */
int carma_kfilter_batch_carma(const double* time, const double* y, const double* yerr, int n, int p,
                              int nmodels, const double* sigsqr, const double* omega_re_im,
                              const double* ma, int nma, const double* mu, double* mean, double* var,
                              int* singular, int* n_out, int device);
/* KalmanFilter1(time,y,yerr,sigsqr,omega).Filter() (kfilter.hpp:222-245; kfilter.cpp:19-48). */
int carma_kfilter_car1(const double* time, const double* y, const double* yerr, int n,
                       double sigsqr, double omega, double* mean, double* var, int* n_out,
                       int device);

/*
 * KalmanFilterp::Predict / KalmanFilter1::Predict (src/kfilter.cpp:218-337, 72-135, 51-69; wrapper
 * :87,98) for M times in ONE launch: conditional mean and variance of the (noise-free) process at
 * tpred[i] given the whole measured series -- interpolation, forecast (tpred > max time) and
 * backcast (tpred < min time).  The reference re-filters the series once per requested time; here
 * every time is one lane group.  Inputs as for carma_kfilter_*; pmean/pvar = [M].  Returns 1 on a
 * singular eigenvector system.
 */
int carma_predict_carma(const double* time, const double* y, const double* yerr, int n, int p,
                        double sigsqr, const double* omega_re_im, const double* ma, int nma,
                        const double* tpred, int M, double* pmean, double* pvar, int device);
int carma_predict_car1(const double* time, const double* y, const double* yerr, int n, double sigsqr,
                       double omega, const double* tpred, int M, double* pmean, double* pvar, int device);

/* The order in which the KalmanFilterp-type entry points want the AR roots -- conjugate pairs adjacent (negative
 * imaginary part first), then the real roots: what CARp::ARRoots emits (src/carpack.cpp:137-172) -- from roots in any
 * order.  They do this themselves; exported for callers that keep per-root quantities aligned.  Host arithmetic.
 * CARMA_EINVAL when the set is not closed under conjugation (to 1e-12).  out = p (re, im) pairs. */
int carma_normalize_roots(int p, const double* omega_re_im, double* out);

/*
 * The same as OBJECTS, as the reference has them (KalmanFilter1 / KalmanFilterp, kfilter.hpp:211-334; wrapper :83-101):
 * carma_kf_create_* copies, sorts and deduplicates the series and uploads it with the model ONCE; Filter and any number
 * of Predict calls then only launch and copy results (the free functions above build such an object per call).
 * carma_kf_n = length after sort/dedup (size of mean/var).  Return codes as above.
 */
carma_kf* carma_kf_create_carma(const double* time, const double* y, const double* yerr, int n, int p, double sigsqr,
                                const double* omega_re_im, const double* ma, int nma, int device);
carma_kf* carma_kf_create_car1(const double* time, const double* y, const double* yerr, int n, double sigsqr,
                               double omega, int device);
void carma_kf_destroy(carma_kf* kf);
int carma_kf_n(const carma_kf* kf);
int carma_kf_filter(carma_kf* kf, double* mean, double* var);
int carma_kf_predict(carma_kf* kf, const double* tpred, int M, double* pmean, double* pvar);

/*
 * carma_process / car1_process (src/carmcmc/carma_pack.py:1148-1259, 1126-1146) for npaths independent paths in ONE
 * launch: exact draws of the process at the n (sorted here) times, value by value from the one-step predictive
 * distribution of the Kalman recursion without measurement error.  Normal variates from the counter-based generator
 * keyed by (seed, path, step): reproducible, and a path does not depend on the batch it is drawn in.
 * out = [npaths][n] (host).  CAR(1): omega = 1 / tau.  Returns 1 on a repeated AR root.
 */
int carma_simulate_carma(const double* time, int n, int p, double sigsqr, const double* omega_re_im,
                         const double* ma, int nma, int npaths, uint64_t seed, double* out, int device);
int carma_simulate_car1(const double* time, int n, double sigsqr, double omega, int npaths, uint64_t seed,
                        double* out, int device);

/*
 * CarmaSample post-processing (src/carmcmc/carma_pack.py), one launch per quantity instead of a
 * Python loop over the MCMC samples.
 *
 * carma_sigma_noise_batch == CarmaSample._sigma_noise (carma_pack.py:513-546; the Python twin of CARp::Variance,
 *   src/carpack.cpp:377-409, with sigma = 1): sigma[s] = sqrt(var[s] / Variance(roots_s, ma_s, 1)) for ns samples.
 *   ar_roots_re_im = [ns][p][2], ma_coefs = [ns][nma] (lowest order first, nma <= p), var = [ns]; all host pointers.
 *   A sample whose variance sum is not positive gives NaN, as numpy's sqrt does.
 *
 * carma_psd_band == the numbers behind CarmaSample.plot_power_spectrum (carma_pack.py:548-648) and
 *   Car1Sample.plot_power_spectrum (:950-1035): psd[f][s] = sigma_s^2 |delta_s(2 pi i f)|^2 / |alpha_s(2 pi i f)|^2 on the
 *   nf x ns grid, then np.percentile(psd, percentiles, axis=samples) -- the exact order statistics, numpy's default linear
 *   interpolation.  ar_coefs = [ns][nar] highest order first (np.poly order, nar = p + 1), ma_coefs = [ns][nma] lowest
 *   order first, sigma = [ns], freq = [nf]; band = [nf][nperc] (nperc <= 4; 0: grid only); psd_samples = NULL or
 *   [nf][ns] for the grid itself.  A NaN in a row makes that row's percentiles NaN (np.percentile).
 */
int carma_sigma_noise_batch(int p, int nma, const double* ar_roots_re_im, const double* ma_coefs, const double* var, int ns,
                            double* sigma, int device);
int carma_psd_band(int nar, int nma, const double* ar_coefs, const double* ma_coefs, const double* sigma, int ns,
                   const double* freq, int nf, const double* percentiles, int nperc, double* band, double* psd_samples,
                   int device);

/*
 * Parallel-tempered Robust-Adaptive-Metropolis sampler == RunCarmaSampler / RunCar1Sampler
 * (src/carmcmc.cpp:30-177; bindings run_mcmc_car1 / run_mcmc_carma, boost_python_wrapper.cpp:76-77)
 * with every chain advanced on the GPU by one persistent kernel (carma_pt.hip).
 *
 *   ntemps     number of tempered chains per ladder = the reference's `nwalkers`; default ladder
 *              T_i = 100^(i/(ntemps-1)) (carmcmc.cpp:92-95), or `temperatures[ntemps]` ascending.
 *   nreplicas  number of INDEPENDENT ladders run side by side (the reference runs exactly one);
 *              each yields its own stream of coldest-chain samples.
 *   adapt_iters  the RAM proposal adapts while iteration < adapt_iters (= burn-in, carmcmc.cpp:149).
 *   seed       counter-based RNG key; runs are reproducible (the reference seeds with time(NULL)).
 *
 * carma_pt_run does the whole Sampler::Run (src/samplers.cpp:57-115): starting values (user
 * `init` honoured only when ninit == d and its posterior is finite, else drawn from the
 * reference's starting-value distribution until finite), `burnin` iterations, then sample_size
 * saves of the coldest chain every `thin` iterations.  samples = [nreplicas][sample_size][d],
 * logposts = [nreplicas][sample_size] (host pointers).
 * The finer-grained calls expose the same machinery for chunked runs, benchmarks and for a
 * temperature ladder sharded across GPUs (carma_pt_shard gives the local block its global
 * temperature / replica indices so RNG streams and swap decisions line up across ranks;
 * carma_pt_bind_state lets the caller own the theta/logpost device buffers so that boundary
 * chains can be exchanged with RCCL send/recv between iterations).
 */
int carma_pt_run(carma_ctx* h, int ntemps, int nreplicas, int sample_size, int burnin, int thin,
                 const double* init, int ninit, uint64_t seed, double* samples, double* logposts);

/* ntemps: the ladder must fit ONE workgroup of the fall-back kernel (k_pt: 8 chains per wave at p >= 5, at most 1024
 * threads and 160 KiB of LDS) -- up to 88 temperatures at CARMA(5,3), 52 at CARMA(7,6); longer ladders are CARMA_EINVAL
 * with the sizes in carma_last_error() (the reference's own runs use about ten), or are split into blocks over
 * ranks (carma_pt_shard). */
int carma_pt_create(carma_ctx* h, int ntemps, int nreplicas, const double* temperatures,
                    int adapt_iters, uint64_t seed);
int carma_pt_shard(carma_ctx* h, int ntemps_global, int slot0, int replica0);
int carma_pt_bind_state(carma_ctx* h, double* d_theta /* [R][T][d] */, double* d_logpost /* [R][T] */);
int carma_pt_start(carma_ctx* h, const double* init, int ninit);
int carma_pt_set_chains(carma_ctx* h, const double* theta /* [R][T][d] */, const double* logpost /* [R][T] or NULL */);
int carma_pt_get_chains(carma_ctx* h, double* theta, double* logpost);
int carma_pt_iterate(carma_ctx* h, long niter, int do_exchange);
int carma_pt_sample(carma_ctx* h, int nsamples, int thin, double* samples, double* logposts);
/* acceptance rate per chain [R][T]; swap_rate[r][i] = accepted swaps between temperatures i and i-1 */
int carma_pt_stats(carma_ctx* h, double* accept_rate, double* swap_rate, int reset);
long carma_pt_iterations_done(const carma_ctx* h);
/* which sampler kernel the context is on: 1 = k_pt_row (one chain per DPP row, ladders spread over workgroups), 0 = k_pt
 * (one workgroup per ladder), 2 = one chain per lane with an iteration as propose kernel + batched log-density launch +
 * finish kernel (carma_pt_lane.hip: ensembles of tens of thousands of chains, ladders of at most 64 temperatures).  A context created on k_pt_row drops to
 * k_pt -- silently -- when a cooperative launch is refused or a cross-workgroup exchange times out; tests assert that it
 * did not.  The kernels draw from the same Philox keys and take the same accept / swap decisions, but round the RAM
 * update and the log-density differently (launch shapes of the same evaluation, 1e-8 apart at most on well-conditioned
 * states): after a fall-back the chains are STATISTICALLY equivalent to, not bit-identical with, an uninterrupted run. */
int carma_pt_kernel_in_use(const carma_ctx* h);
/* which recursion the LAST k_pt_row launch of this process ran on: 0 = one-datum wave pipeline, 1 = windowed pipeline (one-sided),
 * 2 = two-sided windowed pipeline; -1 before the first such launch.  For tests and measurements: the choice (launch_pt_row_p in
 * carma_pt.hip) depends on the whole ladder set's grid, the series (SERIES_WINDOW2_OK) and its length. */
int carma_pt_row_pipeline(void);

/*
 * ONE temperature ladder sharded across the GPUs of a node (one process per GPU): the reference has no counterpart --
 * its ExchangeStep (src/include/steps.hpp:318-362, wired hot -> cold in src/carmcmc.cpp:147-157) swaps two chains of
 * the same process; here the two chains of a pair may live on different GPUs and travel over RCCL / xGMI.
 *
 * carma_comm_*: an RCCL communicator owned by this library (bound at run time to librccl.so.1; no link-time
 * dependency).  Rank 0 calls carma_comm_unique_id and hands the 128 bytes to the other ranks by whatever bootstrap the
 * host program has (torch.distributed broadcast, MPI, a file); every rank then calls carma_comm_create (collective).
 *
 * carma_pt_iterate_sharded: `shards` = the nlocal (normally 1) contexts of this process, each created with
 * carma_pt_create + carma_pt_shard for a contiguous block of temperature slots, consecutive blocks in ascending order;
 * the blocks of all ranks tile the ladder in rank order with the same nlocal everywhere.  Per iteration: the sampler
 * kernel advances every block (RAM steps only), then the ladder's sweep runs hot -> cold over ALL adjacent pairs --
 * inside a block by a sweep kernel, across a block boundary by pack kernel -> ncclSend/ncclRecv of R (d+1) + 1 doubles ->
 * swap kernel -- all on the sampler's stream with no host synchronisation; both sides of a boundary draw the same Philox
 * uniform (seed, hotter chain's global slot, iteration) and take the same decision.  Same pairs, same order, same
 * uniforms as the one-GPU sampler: the chain states equal those of the unsharded run bit for bit.
 * Returns after the last iteration has completed, and after the two sides of every boundary have compared a checksum of
 * their decisions (a mismatch fails the call).  After a failed call the blocks refuse to continue (carma_pt_start /
 * carma_pt_set_chains re-arm them).  comm may be NULL when the process owns the whole ladder as one block.
 */
int carma_comm_unique_id(void* out128);
carma_comm* carma_comm_create(const void* id128, int nranks, int rank, int device);
void carma_comm_destroy(carma_comm* comm);
int carma_comm_rank(const carma_comm* comm);
int carma_comm_size(const carma_comm* comm);
int carma_pt_iterate_sharded(carma_ctx* const* shards, int nlocal, long niter, carma_comm* comm);
/* nsamples x thin more iterations; after every thin-th (and its boundary swaps) the coldest chain of every replica is
 * saved (Sampler::SaveValues, src/samplers.cpp:118-124).  Collective like carma_pt_iterate_sharded; the process that
 * owns temperature 0 receives samples = [R][nsamples][d], logposts = [R][nsamples] (host), the others pass NULL. */
int carma_pt_sample_sharded(carma_ctx* const* shards, int nlocal, int nsamples, int thin, carma_comm* comm,
                            double* samples, double* logposts);
/* boundary swaps this block took part in: proposed = R per boundary and iteration; accepted */
int carma_pt_boundary_stats(carma_ctx* h, unsigned long long* proposed, unsigned long long* accepted);
/* result of the last call's boundary self-check for this block: 1 = both sides of its boundaries folded the same
 * decisions, -1 = they differed (the call failed), 0 = no sharded call yet */
int carma_pt_boundary_check(carma_ctx* h);
/* The sweep over the adjacent pairs INSIDE this block for the iteration just run with do_exchange = 0 (hot -> cold,
 * ExchangeStep, src/include/steps.hpp:318-362): the building block of a sharded iteration for host programs that move
 * the boundary chains themselves (carma_pack_amd/parallel.py over torch.distributed). */
int carma_pt_sweep(carma_ctx* h);

/* ---- debug pair: ties the device sampler step to the reference's arithmetic (tests only; no reference counterpart,
 * the reference's generator is a time-seeded global, src/random.cpp:20) --------------------------------------------
 * carma_pt_debug_draws: the variates chain (replica, temperature) of this block uses at iteration `iter` (counted as
 * AdaptiveMetro::niter_, src/steps.cpp:101), computed on the device by the sampler kernels' own functions: z[d] = the
 * unit proposal (proposal_.Draw, src/steps.cpp:65-69), *u_accept = the Metropolis uniform (src/steps.cpp:48),
 * *u_swap = the uniform of the exchange between this chain and the next colder one (src/include/steps.hpp:333; keyed
 * by the warmer chain's global slot).
 * carma_pt_get_factor / carma_pt_set_factor: chol_factor_ of every chain (src/steps.cpp:32), [R][T][d*d] row-major
 * upper triangular. */
int carma_pt_debug_draws(carma_ctx* h, int replica, int temperature, unsigned long long iter, double* z, double* u_accept,
                         double* u_swap);
int carma_pt_get_factor(carma_ctx* h, double* chol);
int carma_pt_set_factor(carma_ctx* h, const double* chol);

#ifdef __cplusplus
}
#endif
#endif /* CARMA_MI355_H */
