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
 *     -INFINITY for a prior-bound violation or a singular Vandermonde solve
 *     (src/include/carpack.hpp:134-138,154-164), NaN where the reference's arithmetic gives NaN.
 *   - theta layout (src/carpack.cpp:205-207,432): theta[0]=sigma_y, theta[1]=measurement-error
 *     scale, theta[2]=mu, theta[3..3+p) log quadratic-factor coefficients of the AR polynomial,
 *     theta[3+p..3+p+q) same for the MA polynomial.  CAR(1) (p==1): theta[3]=ln(omega), d=4.
 *   - "_dev" entry points take DEVICE pointers (hipMalloc'ed or a torch tensor's data_ptr) and a
 *     hipStream_t passed as void*; they only enqueue work.  The others take host pointers and
 *     return when the result is in host memory.
 *   - There is no CPU fallback: without a usable gfx950 device every compute call fails with
 *     CARMA_ENODEV.
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

/* getLogPrior (carpack.hpp:118-126, wrapper :50,58,67); host arithmetic, one vector. */
double carma_logprior(const carma_ctx* h, const double* theta);

/*
 * KalmanFilterp(time,y,yerr,sigsqr,omega,ma).Filter() + GetMean()/GetVar()
 * (src/include/kfilter.hpp:303-334,126-132; wrapper :91-101).  y is already centred, yerr is
 * used as given; omega_re_im = [p][2]; ma has p entries (zero padded by the caller or here if
 * nma < p, kfilter.hpp:318-320).  The series is sorted/deduplicated first; *n_out receives the
 * resulting length and mean/var must have room for n values.
 * Returns 1 (and fills nothing useful) if the Vandermonde solve is singular -- the reference
 * throws std::runtime_error there.
 */
int carma_kfilter_carma(const double* time, const double* y, const double* yerr, int n, int p,
                        double sigsqr, const double* omega_re_im, const double* ma, int nma,
                        double* mean, double* var, int* n_out, int device);
/* KalmanFilter1(time,y,yerr,sigsqr,omega).Filter() (kfilter.hpp:222-245; kfilter.cpp:19-48). */
int carma_kfilter_car1(const double* time, const double* y, const double* yerr, int n,
                       double sigsqr, double omega, double* mean, double* var, int* n_out,
                       int device);

#ifdef __cplusplus
}
#endif
#endif /* CARMA_MI355_H */
