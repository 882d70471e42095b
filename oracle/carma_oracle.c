/*
 * carma_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C restatement of the reference's Kalman-filter log-likelihood hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product path (carma_pack_amd/) never links, imports or calls it.
 *
 * Every function cites the reference file:line (relative to /root/reference) it
 * follows.  Operation order follows the reference literally (full complex p x p
 * covariance, no Hermitian / conjugate-pair shortcuts), complex arithmetic is
 * C99 <complex.h> (gcc lowers * and / to __muldc3/__divdc3 exactly as libstdc++'s
 * std::complex<double> does), built with -ffp-contract=off.
 *
 * Pinning: checked against golden vectors generated in the build container from
 * the reference's own pure-NumPy filter (src/carmcmc/carma_pack.py:1264-1375
 * KalmanFilterDeprecated, :1084-1123 carma_variance, :439-500 root/MA mirrors),
 * the reference's numeric KAT 223003.230567 (cpp_tests/carma_unit_tests.cpp:1313),
 * the var(0) identity (:215,:443) and the dense-GP identity (:564-594).
 * The reference's C++ cannot be built here (Armadillo/Boost/LAPACK absent), so
 * there is no oracle/_ref; third-party call sites (arma::solve -> LAPACK zgesv,
 * kfilter.cpp:158) are restated as LU with partial pivoting (izamax |re|+|im|
 * pivot rule, reciprocal scaling as in zgetf2).
 * Sampler RNG streams: parity unpinned (reference seeds with time(NULL),
 * src/random.cpp:20) -- the sampler restatement below uses its own RNG and is
 * compared distributionally.
 */
#define _GNU_SOURCE
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PMAX 16
typedef double complex cplx;

/* ------------------------------------------------------------------ */
/* libstdc++ std::pow(complex<double>, int) == __complex_pow_unsigned  */
/* used by CARp::Variance (src/carpack.cpp:399-400).                   */
static cplx cpow_uint(cplx x, unsigned n)
{
    cplx y = (n % 2) ? x : 1.0;
    while (n >>= 1) {
        x *= x;
        if (n % 2) y *= x;
    }
    return y;
}

/* src/carpack.cpp:137-172  CARp::ARRoots (also the root half of
 * CARMA::ExtractMA, src/carpack.cpp:522-553, with offset = 3+p).      */
static void quad_roots(const double *logq, int m, cplx *roots)
{
    for (int i = 0; i < m / 2; i++) {
        double quad_term1 = exp(logq[2 * i]);
        double quad_term2 = exp(logq[2 * i + 1]);
        double discriminant = quad_term2 * quad_term2 - 4.0 * quad_term1;
        if (discriminant > 0) {
            double root1 = -0.5 * (quad_term2 + sqrt(discriminant));
            double root2 = -0.5 * (quad_term2 - sqrt(discriminant));
            roots[2 * i] = root1;
            roots[2 * i + 1] = root2;
        } else {
            double real_part = -0.5 * quad_term2;
            double imag_part = -0.5 * sqrt(-discriminant);
            roots[2 * i] = real_part + imag_part * I;
            roots[2 * i + 1] = real_part - imag_part * I;
        }
    }
    if (m % 2 == 1) {
        roots[m - 1] = -exp(logq[m - 1]);
    }
}

void orc_ar_roots(const double *theta, int p, double *re, double *im)
{
    cplx r[ORC_PMAX];
    quad_roots(theta + 3, p, r);
    for (int i = 0; i < p; i++) { re[i] = creal(r[i]); im[i] = cimag(r[i]); }
}

/* src/carpack.cpp:742-756 polycoefs: coefficients of prod (x - r_i),
 * coefs[0] = 1; the span assignment is evaluated from a temporary, i.e.
 * a simultaneous update.                                               */
static void polycoefs(const cplx *roots, int m, double *out /* m+1 */)
{
    cplx coefs[ORC_PMAX + 1], tmp[ORC_PMAX + 1];
    for (int i = 0; i <= m; i++) coefs[i] = 0.0;
    coefs[0] = 1.0;
    for (int i = 0; i < m; i++) {
        for (int k = 1; k <= i + 1; k++) tmp[k] = coefs[k] - roots[i] * coefs[k - 1];
        for (int k = 1; k <= i + 1; k++) coefs[k] = tmp[k];
    }
    for (int i = 0; i <= m; i++) out[i] = creal(coefs[i]);
}

/* src/carpack.cpp:522-580 CARMA::ExtractMA; q == 0 is CARp's fixed
 * ma_coefs_ = (1,0,...,0) (src/include/carpack.hpp:294-295).           */
void orc_ma_coefs(const double *theta, int p, int q, double *ma /* p */)
{
    for (int i = 0; i < p; i++) ma[i] = 0.0;
    if (q == 0) { ma[0] = 1.0; return; }
    cplx r[ORC_PMAX];
    double pc[ORC_PMAX + 1];
    quad_roots(theta + 3 + p, q, r);
    polycoefs(r, q, pc);
    double cq = pc[q];
    for (int i = 0; i <= q; i++) pc[i] = pc[i] / cq;
    for (int i = 0; i < q + 1; i++) ma[i] = pc[q - i];
}

/* src/carpack.cpp:377-409 CARp::Variance (autocovariance at lag dt).   */
double orc_variance(int p, const double *re, const double *im, const double *ma,
                    double sigma, double dt)
{
    cplx roots[ORC_PMAX];
    for (int i = 0; i < p; i++) roots[i] = re[i] + im[i] * I;
    cplx car_var = 0.0;
    for (int k = 0; k < p; k++) {
        cplx denom_product = 1.0;
        for (int l = 0; l < p; l++) {
            if (l != k) denom_product *= (roots[l] - roots[k]) * (conj(roots[l]) + roots[k]);
        }
        cplx denom = -2.0 * creal(roots[k]) * denom_product;
        cplx ma_sum1 = 0.0, ma_sum2 = 0.0;
        for (int l = 0; l < p; l++) {
            ma_sum1 += ma[l] * cpow_uint(roots[k], (unsigned)l);
            ma_sum2 += ma[l] * cpow_uint(-roots[k], (unsigned)l);
        }
        cplx numer = ma_sum1 * ma_sum2 * cexp(roots[k] * dt);
        car_var += numer / denom;
    }
    return sigma * sigma * creal(car_var);
}

/* src/carpack.cpp:709-732 unique_roots */
static int unique_roots(const cplx *roots, int p, double tolerance)
{
    double min_frac_diff = 100.0 * tolerance;
    for (int i = 0; i < p - 1; i++)
        for (int j = i + 1; j < p; j++) {
            double frac_diff = cabs((roots[i] - roots[j]) / (roots[i] + roots[j]));
            if (frac_diff < min_frac_diff) min_frac_diff = frac_diff;
        }
    return min_frac_diff > tolerance;
}

/* Data + prior context shared by the model-level entry points.
 * Mirrors CARMA_Base members (src/include/carpack.hpp:236-248) after
 * KalmanFilter::init (src/include/kfilter.hpp:43-76) and SetPrior (:201-207). */
typedef struct {
    int n, p, q;
    double *t, *y, *yerr;
    double max_stdev, max_freq, min_freq;
    int measerr_dof;
} orc_model;

static int cmp_idx(const void *a, const void *b, void *arg)
{
    const double *t = (const double *)arg;
    size_t ia = *(const size_t *)a, ib = *(const size_t *)b;
    if (t[ia] < t[ib]) return -1;
    if (t[ia] > t[ib]) return 1;
    return (ia > ib) - (ia < ib); /* stable */
}

/* src/include/kfilter.hpp:43-76 init(): sort by time if any dt<0, then drop
 * entries whose dt to the previous kept-or-dropped sample is zero.
 * Returns the new length.                                                   */
int orc_sort_dedup(int n, double *t, double *y, double *yerr)
{
    int need_sort = 0, has_dup = 0;
    for (int i = 1; i < n; i++) if (t[i] - t[i - 1] < 0) need_sort = 1;
    if (need_sort) {
        size_t *idx = malloc(sizeof(size_t) * n);
        double *tt = malloc(sizeof(double) * n), *yy = malloc(sizeof(double) * n),
               *ee = malloc(sizeof(double) * n);
        for (int i = 0; i < n; i++) idx[i] = i;
        qsort_r(idx, n, sizeof(size_t), cmp_idx, t);
        for (int i = 0; i < n; i++) { tt[i] = t[idx[i]]; yy[i] = y[idx[i]]; ee[i] = yerr[idx[i]]; }
        memcpy(t, tt, sizeof(double) * n); memcpy(y, yy, sizeof(double) * n);
        memcpy(yerr, ee, sizeof(double) * n);
        free(idx); free(tt); free(yy); free(ee);
    }
    for (int i = 1; i < n; i++) if (t[i] - t[i - 1] == 0) has_dup = 1;
    if (has_dup) {
        /* unique_values = {0} U {1 + find(dt != 0)} computed on the sorted dt */
        int m = 1;
        double prev = t[0];
        for (int i = 1; i < n; i++) {
            double d = t[i] - prev;
            prev = t[i];
            if (d != 0) { t[m] = t[i]; y[m] = y[i]; yerr[m] = yerr[i]; m++; }
        }
        n = m;
    }
    return n;
}

orc_model *orc_model_create(const double *t, const double *y, const double *yerr, int n,
                            int p, int q, double max_stdev)
{
    orc_model *m = calloc(1, sizeof(orc_model));
    m->t = malloc(sizeof(double) * n); m->y = malloc(sizeof(double) * n);
    m->yerr = malloc(sizeof(double) * n);
    memcpy(m->t, t, sizeof(double) * n); memcpy(m->y, y, sizeof(double) * n);
    memcpy(m->yerr, yerr, sizeof(double) * n);
    m->n = orc_sort_dedup(n, m->t, m->y, m->yerr);
    m->p = p; m->q = q;
    m->measerr_dof = 50;                       /* src/include/carpack.hpp:63 */
    m->max_stdev = max_stdev;                  /* src/include/carpack.hpp:201-207 SetPrior */
    double dtmin = INFINITY, tmin = INFINITY, tmax = -INFINITY;
    for (int i = 1; i < m->n; i++) { double d = m->t[i] - m->t[i - 1]; if (d < dtmin) dtmin = d; }
    for (int i = 0; i < m->n; i++) { if (m->t[i] < tmin) tmin = m->t[i]; if (m->t[i] > tmax) tmax = m->t[i]; }
    m->max_freq = 1.0 / dtmin;
    m->min_freq = 1.0 / (tmax - tmin);
    return m;
}

void orc_model_destroy(orc_model *m)
{
    if (!m) return;
    free(m->t); free(m->y); free(m->yerr); free(m);
}
int orc_model_n(const orc_model *m) { return m->n; }
void orc_model_get_data(const orc_model *m, double *t, double *y, double *yerr)
{
    memcpy(t, m->t, sizeof(double) * m->n); memcpy(y, m->y, sizeof(double) * m->n);
    memcpy(yerr, m->yerr, sizeof(double) * m->n);
}
void orc_model_get_prior(const orc_model *m, double *out3)
{ out3[0] = m->max_stdev; out3[1] = m->max_freq; out3[2] = m->min_freq; }

/* src/include/carpack.hpp:118-126 LogPrior */
double orc_log_prior(const orc_model *m, const double *theta)
{
    double measerr_scale = theta[1];
    return -0.5 * m->measerr_dof / measerr_scale -
           (1.0 + m->measerr_dof / 2.0) * log(measerr_scale);
}

/* src/carpack.cpp:314-374 CARp::CheckPriorBounds (ignore_prior short-circuit :316) */
int orc_check_prior_bounds_carma(const orc_model *m, const double *theta, int ignore_prior)
{
    if (ignore_prior) return 1;
    int p = m->p;
    double ysigma = theta[0], measerr_scale = theta[1];
    cplx roots[ORC_PMAX];
    quad_roots(theta + 3, p, roots);
    double cent[ORC_PMAX], width[ORC_PMAX];
    int nv1 = 0, nv2 = 0, nv3 = 0;
    for (int i = 0; i < p; i++) {
        cent[i] = fabs(cimag(roots[i])) / 2.0 / M_PI;
        width[i] = -creal(roots[i]) / 2.0 / M_PI;
        if (cent[i] < m->max_freq) nv1++;
        if (width[i] < m->max_freq) nv2++;
        if (width[i] > m->min_freq) nv3++;
    }
    int prior_satisfied = unique_roots(roots, p, 1e-4);
    if (nv1 != p || nv2 != p || nv3 != p || (ysigma > m->max_stdev) || (ysigma < 0) ||
        (measerr_scale < 0.5) || (measerr_scale > 2.0))
        prior_satisfied = 0;
    for (int i = 1; i < p; i++) {                   /* order_lorentzians_ (:353-362) */
        double d = cent[i] - cent[i - 1];
        if (d > 1e-8) prior_satisfied = 0;
    }
    return prior_satisfied;
}

/* src/carpack.cpp:116-130 CAR1::CheckPriorBounds (no ignore_prior branch) */
int orc_check_prior_bounds_car1(const orc_model *m, const double *theta)
{
    double ysigma = theta[0], measerr_scale = theta[1], omega = exp(theta[3]);
    if ((omega > m->max_freq) || (omega < m->min_freq) || (ysigma > m->max_stdev) || (ysigma < 0) ||
        (measerr_scale < 0.5) || (measerr_scale > 2.0))
        return 0;
    return 1;
}

/* Restatement of arma::solve(EigenMat, Rvector) (src/kfilter.cpp:158) ==
 * LAPACK zgesv on a p x p system with one right-hand side: zgetf2-style LU
 * with partial pivoting (pivot = first max of |re|+|im|, izamax), column
 * scaling by the reciprocal of the pivot, then L and U triangular solves.
 * A is row-major a[i*p+j].  Returns 0, or k+1 if U(k,k) is exactly zero.     */
static int zgesv_small(int p, cplx *a, cplx *b)
{
    for (int k = 0; k < p; k++) {
        int piv = k;
        double best = fabs(creal(a[k * p + k])) + fabs(cimag(a[k * p + k]));
        for (int i = k + 1; i < p; i++) {
            double v = fabs(creal(a[i * p + k])) + fabs(cimag(a[i * p + k]));
            if (v > best) { best = v; piv = i; }
        }
        if (best == 0.0) return k + 1;
        if (piv != k) {
            for (int j = 0; j < p; j++) { cplx tmp = a[k * p + j]; a[k * p + j] = a[piv * p + j]; a[piv * p + j] = tmp; }
            cplx tb = b[k]; b[k] = b[piv]; b[piv] = tb;
        }
        cplx rinv = 1.0 / a[k * p + k];
        for (int i = k + 1; i < p; i++) a[i * p + k] *= rinv;
        for (int i = k + 1; i < p; i++) {
            cplx l = a[i * p + k];
            for (int j = k + 1; j < p; j++) a[i * p + j] -= l * a[k * p + j];
        }
    }
    /* forward substitution with unit-lower L (ztrsm L,L,N,U) */
    for (int k = 0; k < p; k++)
        for (int i = k + 1; i < p; i++) b[i] -= b[k] * a[i * p + k];
    /* back substitution with U (ztrsm L,U,N,N) */
    for (int k = p - 1; k >= 0; k--) {
        b[k] = b[k] / a[k * p + k];
        for (int i = 0; i < k; i++) b[i] -= b[k] * a[i * p + k];
    }
    return 0;
}

/* src/kfilter.cpp:138-215 KalmanFilterp::Reset + Update, driven by
 * src/include/kfilter.hpp:126-132 Filter().  y is already centred, yerr already
 * scaled (the caller does carpack.hpp:150-153).  Returns 0 or -1 (singular
 * solve == the std::runtime_error path of carpack.hpp:154-164).
 * If cond_out != NULL also reports nothing (kept for ABI symmetry).            */
int orc_kfilter_carma(int n, const double *t, const double *y, const double *yerr, int p,
                      double sigsqr, const double *om_re, const double *om_im, const double *ma,
                      double *mean, double *var)
{
    cplx omega[ORC_PMAX], E[ORC_PMAX * ORC_PMAX], J[ORC_PMAX], b[ORC_PMAX];
    cplx V[ORC_PMAX * ORC_PMAX], P[ORC_PMAX * ORC_PMAX], x[ORC_PMAX], g[ORC_PMAX], rho[ORC_PMAX];
    for (int i = 0; i < p; i++) omega[i] = om_re[i] + om_im[i] * I;

    /* Reset (:138-186) */
    cplx Ework[ORC_PMAX * ORC_PMAX];
    for (int j = 0; j < p; j++) {
        cplx pw = 1.0;
        for (int i = 0; i < p; i++) {       /* row i = omega^i (:144-149) */
            E[i * p + j] = pw;
            pw *= omega[j];
        }
    }
    for (int i = 0; i < p; i++) J[i] = 0.0;
    J[p - 1] = 1.0;
    memcpy(Ework, E, sizeof(cplx) * p * p);
    if (zgesv_small(p, Ework, J) != 0) return -1;
    for (int j = 0; j < p; j++) {            /* rotated_ma_coefs_ = ma * E (:162) */
        cplx s = 0.0;
        for (int i = 0; i < p; i++) s += ma[i] * E[i * p + j];
        b[j] = s;
    }
    for (int i = 0; i < p; i++)              /* :165-172, symmatu conj-reflects */
        for (int j = i; j < p; j++) {
            V[i * p + j] = -sigsqr * J[i] * conj(J[j]) / (omega[i] + conj(omega[j]));
            V[j * p + i] = conj(V[i * p + j]);
        }
    memcpy(P, V, sizeof(cplx) * p * p);
    for (int i = 0; i < p; i++) x[i] = 0.0;
    mean[0] = 0.0;
    {
        cplx acc = 0.0;                       /* (b*V)*b^H (:181) */
        for (int j = 0; j < p; j++) {
            cplx s = 0.0;
            for (int i = 0; i < p; i++) s += b[i] * V[i * p + j];
            acc += s * conj(b[j]);
        }
        var[0] = creal(acc) + yerr[0] * yerr[0];
    }
    double innovation = y[0];

    /* Update (:189-215) */
    for (int k = 1; k < n; k++) {
        double vprev = var[k - 1];
        for (int i = 0; i < p; i++) {         /* gain (:191) */
            cplx s = 0.0;
            for (int j = 0; j < p; j++) s += P[i * p + j] * conj(b[j]);
            g[i] = s / vprev;
        }
        for (int i = 0; i < p; i++) x[i] += g[i] * innovation;                  /* :194 */
        for (int i = 0; i < p; i++)                                             /* :197 */
            for (int j = 0; j < p; j++) P[i * p + j] -= vprev * (g[i] * conj(g[j]));
        double dt = t[k] - t[k - 1];
        for (int i = 0; i < p; i++) rho[i] = cexp(omega[i] * dt);               /* :200 */
        for (int i = 0; i < p; i++) x[i] = rho[i] * x[i];                       /* :201 */
        for (int i = 0; i < p; i++)                                             /* :204 */
            for (int j = 0; j < p; j++)
                P[i * p + j] = (rho[i] * conj(rho[j])) * (P[i * p + j] - V[i * p + j]) + V[i * p + j];
        cplx m = 0.0;                                                           /* :207 */
        for (int i = 0; i < p; i++) m += b[i] * x[i];
        mean[k] = creal(m);
        cplx acc = 0.0;                                                         /* :209 */
        for (int j = 0; j < p; j++) {
            cplx s = 0.0;
            for (int i = 0; i < p; i++) s += b[i] * P[i * p + j];
            acc += s * conj(b[j]);
        }
        var[k] = creal(acc) + yerr[k] * yerr[k];                                /* :210 */
        innovation = y[k] - mean[k];                                            /* :213 */
    }
    return 0;
}

/* src/kfilter.cpp:19-48 KalmanFilter1::Reset/Update */
void orc_kfilter_car1(int n, const double *t, const double *y, const double *yerr, double sigsqr,
                      double omega, double *mean, double *var)
{
    mean[0] = 0.0;
    var[0] = sigsqr / (2.0 * omega) + yerr[0] * yerr[0];
    for (int k = 1; k < n; k++) {
        double rho = exp(-1.0 * omega * (t[k] - t[k - 1]));
        double previous_var = var[k - 1] - yerr[k - 1] * yerr[k - 1];
        double var_ratio = previous_var / var[k - 1];
        mean[k] = rho * mean[k - 1] + rho * var_ratio * (y[k - 1] - mean[k - 1]);
        var[k] = sigsqr / (2.0 * omega) * (1.0 - rho * rho) + rho * rho * previous_var * (1.0 - var_ratio);
        var[k] += yerr[k] * yerr[k];
    }
}

/* src/include/carpack.hpp:131-176 CARMA_Base::LogDensity for CARp / CARMA.
 * work = 4n doubles of scratch (or NULL to malloc).                           */
double orc_logdensity_carma(const orc_model *m, const double *theta, int ignore_prior, double *work)
{
    int n = m->n, p = m->p, q = m->q;
    if (!orc_check_prior_bounds_carma(m, theta, ignore_prior)) return -INFINITY;   /* :134-138 */
    double re[ORC_PMAX], im[ORC_PMAX], ma[ORC_PMAX];
    orc_ar_roots(theta, p, re, im);                                                /* :140 */
    orc_ma_coefs(theta, p, q, ma);                                                 /* :141 */
    double sigsqr = theta[0] * theta[0] / orc_variance(p, re, im, ma, 1.0, 0.0);   /* :142, hpp:316-319,391-395 */
    double measerr_scale = theta[1], mu = theta[2];
    int own = 0;
    if (!work) { work = malloc(sizeof(double) * 4 * n); own = 1; }
    double *ycent = work, *perr = work + n, *mean = work + 2 * n, *var = work + 3 * n;
    double s = sqrt(measerr_scale);
    for (int i = 0; i < n; i++) { perr[i] = s * m->yerr[i]; ycent[i] = m->y[i] - mu; }    /* :150-153 */
    double logpost;
    if (orc_kfilter_carma(n, m->t, ycent, perr, p, sigsqr, re, im, ma, mean, var) != 0) {
        logpost = -INFINITY;                                                       /* :156-164 */
    } else {
        logpost = 0.0;
        for (int i = 0; i < n; i++) {                                              /* :167-171 */
            double yc = m->y[i] - mean[i] - mu;
            logpost += -0.5 * log(var[i]) - 0.5 * yc * yc / var[i];
        }
        logpost += orc_log_prior(m, theta);                                        /* :173 */
    }
    if (own) free(work);
    return logpost;
}

/* Same for CAR1 (ExtractAR hpp:265, ExtractSigsqr hpp:273-275). theta has 4 entries. */
double orc_logdensity_car1(const orc_model *m, const double *theta, double *work)
{
    int n = m->n;
    if (!orc_check_prior_bounds_car1(m, theta)) return -INFINITY;
    double omega = exp(theta[3]);
    double sigsqr = 2.0 * theta[0] * theta[0] * exp(theta[3]);
    double measerr_scale = theta[1], mu = theta[2];
    int own = 0;
    if (!work) { work = malloc(sizeof(double) * 4 * n); own = 1; }
    double *ycent = work, *perr = work + n, *mean = work + 2 * n, *var = work + 3 * n;
    double s = sqrt(measerr_scale);
    for (int i = 0; i < n; i++) { perr[i] = s * m->yerr[i]; ycent[i] = m->y[i] - mu; }
    orc_kfilter_car1(n, m->t, ycent, perr, sigsqr, omega, mean, var);
    double logpost = 0.0;
    for (int i = 0; i < n; i++) {
        double yc = m->y[i] - mean[i] - mu;
        logpost += -0.5 * log(var[i]) - 0.5 * yc * yc / var[i];
    }
    logpost += orc_log_prior(m, theta);
    if (own) free(work);
    return logpost;
}

/* Batched driver used by tests and by bench.py's cpu_baseline leg ("port").
 * theta is [B][d] row-major, d = 3+p+q (p>=2) or 4 (p==1).  nthreads<=1 is a
 * scalar single-thread run; otherwise OpenMP over evals.                      */
void orc_logdensity_batch(const orc_model *m, const double *theta, int B, int ignore_prior,
                          int nthreads, double *out)
{
    int d = (m->p == 1) ? 4 : 3 + m->p + m->q;
#ifdef _OPENMP
    if (nthreads > 1) {
#pragma omp parallel num_threads(nthreads)
        {
            double *work = malloc(sizeof(double) * 4 * m->n);
#pragma omp for schedule(static)
            for (int b = 0; b < B; b++)
                out[b] = (m->p == 1) ? orc_logdensity_car1(m, theta + (size_t)b * d, work)
                                     : orc_logdensity_carma(m, theta + (size_t)b * d, ignore_prior, work);
            free(work);
        }
        return;
    }
#endif
    (void)nthreads;
    double *work = malloc(sizeof(double) * 4 * m->n);
    for (int b = 0; b < B; b++)
        out[b] = (m->p == 1) ? orc_logdensity_car1(m, theta + (size_t)b * d, work)
                             : orc_logdensity_carma(m, theta + (size_t)b * d, ignore_prior, work);
    free(work);
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* src/steps.cpp:111-131 CholUpdateR1 on an upper-triangular factor stored
 * row-major L[k*d+j]; v is overwritten.                                       */
void orc_chol_update_r1(int d, double *L, double *v, int downdate)
{
    double sign = downdate ? -1.0 : 1.0;
    for (int k = 0; k < d; k++) {
        double r = sqrt(L[k * d + k] * L[k * d + k] + sign * v[k] * v[k]);
        double c = r / L[k * d + k];
        double s = v[k] / L[k * d + k];
        L[k * d + k] = r;
        if (k < d - 1) {
            for (int j = k + 1; j < d; j++) L[k * d + j] = (L[k * d + j] + sign * s * v[j]) / c;
            for (int j = k + 1; j < d; j++) v[j] = c * v[j] - s * L[k * d + j];
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * ONE AdaptiveMetro::DoStep and ONE ExchangeStep::DoStep with the random variates as INPUTS, so that a
 * sampler with its own generator (the device's counter-based one) can be checked decision by decision
 * and digit by digit against the reference's arithmetic.
 *
 * orc_ram_step: src/steps.cpp:60-107 (DoStep) with Accept :36-56 and CholUpdateR1 :111-131.
 *   theta[d], *lp (stored log-posterior, Parameter::GetLogDensity), R[d*d] (upper-triangular factor,
 *   row-major, Sigma = R^T R as arma::chol returns, :32) are updated in place.  z[d] = the unit proposal
 *   (proposal_.Draw, :65-69), u = the Metropolis uniform (:48; not consumed when alpha is not finite,
 *   :41-46), temperature = parameter_.GetTemperature(), niter = niter_ BEFORE the step, maxiter = maxiter_.
 *   Returns 1 when the proposal was accepted; *lnew = LogDensity(new_value).  lnew_rel_shift: 0 (see below).        */
int orc_ram_step(const orc_model *m, double *theta, double *lp, double *R, const double *z, double u,
                 double temperature, long niter, long maxiter, double *work, double *lnew, double lnew_rel_shift)
{
    const int d = (m->p == 1) ? 4 : 3 + m->p + m->q;
    double scaled[32], newv[32];
    int accepted = 0;
    /* scaled_proposal = chol_factor_.t() * unit_proposal ; new_value = old_value + scaled_proposal (:72-73) */
    for (int j = 0; j < d; j++) {
        double acc = 0.0;
        for (int k = 0; k <= j; k++) acc += R[k * d + j] * z[k];
        scaled[j] = acc;
        newv[j] = theta[j] + acc;
    }
    /* Accept (:36-56) */
    double l1 = (m->p == 1) ? orc_logdensity_car1(m, newv, work) : orc_logdensity_carma(m, newv, 0, work);
    /* (test instrument: the step's sensitivity to the log-density -- how far the factor may move when LogDensity(new_value)
     * moves by the parity bar; 0 for the reference's step) */
    if (lnew_rel_shift != 0.0 && isfinite(l1)) l1 += lnew_rel_shift * fabs(l1);
    double alpha = (l1 - *lp) / temperature;
    if (!isfinite(alpha)) {
        alpha = 0.0;                                   /* :41-46: rejected, and alpha_ = 0 stays FINITE */
    } else {
        alpha = fmin(exp(alpha), 1.0);
        if (u < alpha) accepted = 1;
    }
    if (accepted) {                                    /* parameter_.Save(new_value) (:77), carpack.hpp:90-108 */
        memcpy(theta, newv, sizeof(double) * d);
        *lp = l1;
    }
    if (niter < maxiter && isfinite(alpha)) {          /* :82 -- always finite here: a rejected -inf proposal downdates */
        double step = fmin(1.0, (double)d / pow((double)niter, 2.0 / 3.0));   /* :87, niter_ = 0 -> 1 */
        double nrm = 0.0;
        for (int k = 0; k < d; k++) nrm += z[k] * z[k];
        nrm = sqrt(nrm);                               /* arma::norm(unit_proposal, 2) (:89) */
        double fac = sqrt(step * fabs(alpha - 0.25)) / nrm;                   /* :92 */
        for (int k = 0; k < d; k++) scaled[k] *= fac;
        orc_chol_update_r1(d, R, scaled, alpha < 0.25);                       /* :95-98 */
    }
    if (lnew) *lnew = l1;
    return accepted;
}

/* orc_exchange: src/include/steps.hpp:318-362.  "this" = the warmer chain i, "other" = chain i-1.
 * Swaps theta and the stored log-posteriors in place when accepted; returns 1 then.                   */
int orc_exchange(int d, double *theta_this, double *lp_this, double temp_this, double *theta_other,
                 double *lp_other, double temp_other, double u)
{
    double this_logpost = *lp_this, other_logpost = *lp_other;
    double alpha = 1.0 / temp_this * (other_logpost - this_logpost) + 1.0 / temp_other * (this_logpost - other_logpost);
    alpha = fmin(exp(alpha), 1.0);
    if (!isfinite(alpha)) alpha = 0.0;
    if (u < alpha) {
        for (int k = 0; k < d; k++) { double tmp = theta_this[k]; theta_this[k] = theta_other[k]; theta_other[k] = tmp; }
        *lp_this = other_logpost;                      /* SetLogDensity overrides (:343-350) */
        *lp_other = this_logpost;
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Prediction (interpolation / forecast / backcast) -- SURVEY.md §8(f) rank 1.
 * src/kfilter.cpp:218-286 KalmanFilterp::Predict with :290-337 InitializeCoefs / UpdateCoefs.
 * y centred, yerr as given.  Returns 0, or -1 on a singular Vandermonde solve.                    */
int orc_predict_carma(int n, const double *t, const double *y, const double *yerr, int p, double sigsqr,
                      const double *om_re, const double *om_im, const double *ma, double time,
                      double *pmean, double *pvar)
{
    cplx omega[ORC_PMAX], E[ORC_PMAX * ORC_PMAX], Ework[ORC_PMAX * ORC_PMAX], J[ORC_PMAX], b[ORC_PMAX];
    cplx V[ORC_PMAX * ORC_PMAX], P[ORC_PMAX * ORC_PMAX], x[ORC_PMAX], g[ORC_PMAX], rho[ORC_PMAX];
    cplx sconst[ORC_PMAX], sslope[ORC_PMAX];
    double *var = malloc(sizeof(double) * n), *mean = malloc(sizeof(double) * n);
    int rc = 0;
    for (int i = 0; i < p; i++) omega[i] = om_re[i] + om_im[i] * I;
    int ipredict = 0;                                        /* :221-229 */
    while (time > t[ipredict]) { ipredict++; if (ipredict == n) break; }
    /* Reset (:138-186) */
    for (int j = 0; j < p; j++) { cplx pw = 1.0; for (int i = 0; i < p; i++) { E[i * p + j] = pw; pw *= omega[j]; } }
    for (int i = 0; i < p; i++) J[i] = 0.0;
    J[p - 1] = 1.0;
    memcpy(Ework, E, sizeof(cplx) * p * p);
    if (zgesv_small(p, Ework, J) != 0) { rc = -1; goto done; }
    for (int j = 0; j < p; j++) { cplx s = 0.0; for (int i = 0; i < p; i++) s += ma[i] * E[i * p + j]; b[j] = s; }
    for (int i = 0; i < p; i++)
        for (int j = i; j < p; j++) {
            V[i * p + j] = -sigsqr * J[i] * conj(J[j]) / (omega[i] + conj(omega[j]));
            V[j * p + i] = conj(V[i * p + j]);
        }
    memcpy(P, V, sizeof(cplx) * p * p);
    for (int i = 0; i < p; i++) x[i] = 0.0;
#define BPB(OUT) do { cplx acc_ = 0.0; for (int j_ = 0; j_ < p; j_++) { cplx s_ = 0.0; \
        for (int i_ = 0; i_ < p; i_++) s_ += b[i_] * P[i_ * p + j_]; acc_ += s_ * conj(b[j_]); } (OUT) = creal(acc_); } while (0)
#define GAIN(DEN) do { for (int i_ = 0; i_ < p; i_++) { cplx s_ = 0.0; \
        for (int j_ = 0; j_ < p; j_++) s_ += P[i_ * p + j_] * conj(b[j_]); g[i_] = s_ / (DEN); } } while (0)
#define DOWNDATE(DEN) do { for (int i_ = 0; i_ < p; i_++) for (int j_ = 0; j_ < p; j_++) \
        P[i_ * p + j_] -= (DEN) * (g[i_] * conj(g[j_])); } while (0)
#define TIMEUPD(DT) do { for (int i_ = 0; i_ < p; i_++) rho[i_] = cexp(omega[i_] * (DT)); \
        for (int i_ = 0; i_ < p; i_++) for (int j_ = 0; j_ < p; j_++) \
            P[i_ * p + j_] = (rho[i_] * conj(rho[j_])) * (P[i_ * p + j_] - V[i_ * p + j_]) + V[i_ * p + j_]; } while (0)
    mean[0] = 0.0;
    BPB(var[0]);
    var[0] += yerr[0] * yerr[0];
    double innovation = y[0];
    for (int k = 1; k < ipredict; k++) {                      /* Update x (ipredict-1)  (:231-234) */
        GAIN(var[k - 1]);
        for (int i = 0; i < p; i++) x[i] += g[i] * innovation;
        DOWNDATE(var[k - 1]);
        TIMEUPD(t[k] - t[k - 1]);
        for (int i = 0; i < p; i++) x[i] = rho[i] * x[i];
        cplx m = 0.0;
        for (int i = 0; i < p; i++) m += b[i] * x[i];
        mean[k] = creal(m);
        BPB(var[k]);
        var[k] += yerr[k] * yerr[k];
        innovation = y[k] - mean[k];
    }
    double ypredict_mean, ypredict_var, yprecision;
    if (ipredict == 0) {                                      /* backcast (:238-241) */
        ypredict_mean = 0.0;
        BPB(ypredict_var);                                    /* P == V here */
    } else {                                                  /* :242-255 */
        GAIN(var[ipredict - 1]);
        for (int i = 0; i < p; i++) x[i] += g[i] * innovation;
        DOWNDATE(var[ipredict - 1]);
        double dt = fabs(time - t[ipredict - 1]);
        TIMEUPD(dt);
        for (int i = 0; i < p; i++) x[i] = rho[i] * x[i];
        cplx m = 0.0;
        for (int i = 0; i < p; i++) m += b[i] * x[i];
        ypredict_mean = creal(m);
        BPB(ypredict_var);
    }
    if (ipredict == n) { *pmean = ypredict_mean; *pvar = ypredict_var; goto done; }   /* forecast (:257-261) */
    yprecision = 1.0 / ypredict_var;
    ypredict_mean *= yprecision;
    {   /* InitializeCoefs(time, ipredict, ypredict_mean / yprecision, ypredict_var)  (:290-314) */
        double ymean = ypredict_mean / yprecision, yvar = ypredict_var;
        GAIN(yvar);
        for (int i = 0; i < p; i++) { sconst[i] = x[i] - g[i] * ymean; sslope[i] = g[i]; }
        DOWNDATE(yvar);
        double dt = fabs(t[ipredict] - time);
        TIMEUPD(dt);
        for (int i = 0; i < p; i++) { sconst[i] = rho[i] * sconst[i]; sslope[i] = rho[i] * sslope[i]; }
    }
    double yconst, yslope;
    {
        cplx a = 0.0, c = 0.0;
        for (int i = 0; i < p; i++) { a += b[i] * sconst[i]; c += b[i] * sslope[i]; }
        yconst = creal(a); yslope = creal(c);
        BPB(var[ipredict]);
        var[ipredict] += yerr[ipredict] * yerr[ipredict];
    }
    yprecision += yslope * yslope / var[ipredict];                              /* :272-273 */
    ypredict_mean += yslope * (y[ipredict] - yconst) / var[ipredict];
    for (int k = ipredict + 1; k < n; k++) {                                     /* UpdateCoefs (:318-337) */
        GAIN(var[k - 1]);
        for (int i = 0; i < p; i++) { sconst[i] += g[i] * (y[k - 1] - yconst); sslope[i] -= g[i] * yslope; }
        DOWNDATE(var[k - 1]);
        TIMEUPD(t[k] - t[k - 1]);
        for (int i = 0; i < p; i++) { sconst[i] = rho[i] * sconst[i]; sslope[i] = rho[i] * sslope[i]; }
        cplx a = 0.0, c = 0.0;
        for (int i = 0; i < p; i++) { a += b[i] * sconst[i]; c += b[i] * sslope[i]; }
        yconst = creal(a); yslope = creal(c);
        BPB(var[k]);
        var[k] += yerr[k] * yerr[k];
        yprecision += yslope * yslope / var[k];
        ypredict_mean += yslope * (y[k] - yconst) / var[k];
    }
    ypredict_var = 1.0 / yprecision;
    ypredict_mean *= ypredict_var;
    *pmean = ypredict_mean;
    *pvar = ypredict_var;
done:
    free(var); free(mean);
    return rc;
#undef BPB
#undef GAIN
#undef DOWNDATE
#undef TIMEUPD
}

/* src/kfilter.cpp:72-135 KalmanFilter1::Predict with :51-69 InitializeCoefs / UpdateCoefs. */
void orc_predict_car1(int n, const double *t, const double *y, const double *yerr, double sigsqr, double omega,
                      double time, double *pmean, double *pvar)
{
    double *mean = malloc(sizeof(double) * n), *var = malloc(sizeof(double) * n);
    int ipredict = 0;
    while (time > t[ipredict]) { ipredict++; if (ipredict == n) break; }
    mean[0] = 0.0;
    var[0] = sigsqr / (2.0 * omega) + yerr[0] * yerr[0];
    for (int k = 1; k < ipredict; k++) {
        double rho = exp(-1.0 * omega * (t[k] - t[k - 1]));
        double previous_var = var[k - 1] - yerr[k - 1] * yerr[k - 1];
        double var_ratio = previous_var / var[k - 1];
        mean[k] = rho * mean[k - 1] + rho * var_ratio * (y[k - 1] - mean[k - 1]);
        var[k] = sigsqr / (2.0 * omega) * (1.0 - rho * rho) + rho * rho * previous_var * (1.0 - var_ratio);
        var[k] += yerr[k] * yerr[k];
    }
    double ypredict_mean, ypredict_var;
    if (ipredict == 0) {
        ypredict_mean = 0.0;
        ypredict_var = sigsqr / (2.0 * omega);
    } else {
        double dt = time - t[ipredict - 1];
        double rho = exp(-dt * omega);
        double previous_var = var[ipredict - 1] - yerr[ipredict - 1] * yerr[ipredict - 1];
        double var_ratio = previous_var / var[ipredict - 1];
        ypredict_mean = rho * mean[ipredict - 1] + rho * var_ratio * (y[ipredict - 1] - mean[ipredict - 1]);
        ypredict_var = sigsqr / (2.0 * omega) * (1.0 - rho * rho) + rho * rho * previous_var * (1.0 - var_ratio);
    }
    if (ipredict == n) { *pmean = ypredict_mean; *pvar = ypredict_var; free(mean); free(var); return; }
    double yprecision = 1.0 / ypredict_var;
    ypredict_mean *= yprecision;
    /* InitializeCoefs(time, ipredict, 0, 0) (:51-56) */
    double yconst = 0.0;
    double yslope = exp(-fabs(t[ipredict] - time) * omega);
    var[ipredict] = sigsqr / (2.0 * omega) * (1.0 - yslope * yslope) + yerr[ipredict] * yerr[ipredict];
    yprecision += yslope * yslope / var[ipredict];
    ypredict_mean += yslope * (y[ipredict] - yconst) / var[ipredict];
    for (int k = ipredict + 1; k < n; k++) {                 /* UpdateCoefs (:59-69) */
        double rho = exp(-1.0 * (t[k] - t[k - 1]) * omega);
        double previous_var = var[k - 1] - yerr[k - 1] * yerr[k - 1];
        double var_ratio = previous_var / var[k - 1];
        yslope *= rho * (1.0 - var_ratio);
        yconst = yconst * rho * (1.0 - var_ratio) + rho * var_ratio * y[k - 1];
        var[k] = sigsqr / (2.0 * omega) * (1.0 - rho * rho) + rho * rho * previous_var * (1.0 - var_ratio) +
                 yerr[k] * yerr[k];
        yprecision += yslope * yslope / var[k];
        ypredict_mean += yslope * (y[k] - yconst) / var[k];
    }
    ypredict_var = 1.0 / yprecision;
    ypredict_mean *= ypredict_var;
    *pmean = ypredict_mean;
    *pvar = ypredict_var;
    free(mean); free(var);
}

/* ------------------------------------------------------------------------------------------------
 * Literal restatement of the reference's sampler for DISTRIBUTIONAL checks of the GPU sampler.
 * RunCarmaSampler (src/carmcmc.cpp:79-177) + Sampler::Run/Iterate (src/samplers.cpp:37-115):
 * per iteration, for i = T-1 .. 1: AdaptiveMetro(chain i) (src/steps.cpp:60-107), ExchangeStep(i, i-1)
 * (src/include/steps.hpp:318-362); then AdaptiveMetro(chain 0).  Serial hot -> cold sweep exactly as
 * the reference orders its step stack.  RNG: xoshiro256** (the reference's mt19937 is time-seeded, so
 * its streams are unpinned anyway); Student-t(8) = N(0,1)/sqrt(chi2_8/8).
 * start = [T][d] finite starting values (the caller draws them); p==1 runs the single CAR(1) chain of
 * RunCar1Sampler (T must be 1).                                                                    */
typedef struct { uint64_t s[4]; } orc_rng;
static uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t rng_next(orc_rng *r)
{
    uint64_t *s = r->s, result = rotl64(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl64(s[3], 45);
    return result;
}
static double rng_u01(orc_rng *r) { return ((double)(rng_next(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
static double rng_normal(orc_rng *r)
{
    double u1 = rng_u01(r), u2 = rng_u01(r);
    return sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
}
static double rng_t8(orc_rng *r)
{
    double z = rng_normal(r), chi2 = 0.0;
    for (int i = 0; i < 8; i++) { double w = rng_normal(r); chi2 += w * w; }
    return z / sqrt(chi2 / 8.0);
}

int orc_sampler_run(const orc_model *m, int T, int sample_size, int burnin, int thin, uint64_t seed,
                    const double *start, double *samples, double *logposts, double *accept_rate, double *swap_rate)
{
    const int d = (m->p == 1) ? 4 : 3 + m->p + m->q;
    orc_rng rng;
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 4; i++) { z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; rng.s[i] = z; z += 0x9E3779B97F4A7C15ull; }
    double *work = malloc(sizeof(double) * 4 * m->n);
    double *theta = malloc(sizeof(double) * T * d), *lp = malloc(sizeof(double) * T), *temps = malloc(sizeof(double) * T);
    double *chol = calloc((size_t)T * d * d, sizeof(double));
    double *unit = malloc(sizeof(double) * d), *scaled = malloc(sizeof(double) * d), *newv = malloc(sizeof(double) * d);
    long *nacc = calloc(T, sizeof(long)), *nswap = calloc(T, sizeof(long));
    /* temperature ladder and initial proposal covariance (carmcmc.cpp:92-95, 132-136) */
    double sum = 0, sq = 0;
    for (int i = 0; i < m->n; i++) { sum += m->y[i]; sq += m->y[i] * m->y[i]; }
    double mean = sum / m->n, var = sq / m->n - mean * mean;
    for (int i = 0; i < T; i++) {
        temps[i] = (T == 1) ? 1.0 : exp(log(100.0) * (double)i / (double)(T - 1));
        for (int k = 0; k < d; k++) chol[(size_t)i * d * d + k * d + k] = 0.01;
        chol[(size_t)i * d * d + 0] = sqrt(2.0 * var * var / m->n);
        chol[(size_t)i * d * d + 2 * d + 2] = sqrt(var / m->n);
        memcpy(theta + i * d, start + i * d, sizeof(double) * d);
        lp[i] = (m->p == 1) ? orc_logdensity_car1(m, theta + i * d, work) : orc_logdensity_carma(m, theta + i * d, 0, work);
    }
    const long total = (long)burnin + (long)thin * sample_size;
    for (long it = 0; it < total; it++) {
        for (int i = T - 1; i >= 0; i--) {
            /* AdaptiveMetro::DoStep (steps.cpp:60-107); niter_ == it for every chain */
            double *R = chol + (size_t)i * d * d, *old = theta + i * d;
            for (int k = 0; k < d; k++) unit[k] = rng_t8(&rng);
            for (int j = 0; j < d; j++) {
                double acc = 0.0;
                for (int k = 0; k <= j; k++) acc += R[k * d + j] * unit[k];
                scaled[j] = acc;
                newv[j] = old[j] + acc;
            }
            double lnew = (m->p == 1) ? orc_logdensity_car1(m, newv, work) : orc_logdensity_carma(m, newv, 0, work);
            double alpha = (lnew - lp[i]) / temps[i];
            if (!isfinite(alpha)) {
                alpha = 0.0;
            } else {
                double u = rng_u01(&rng);
                alpha = fmin(exp(alpha), 1.0);
                if (u < alpha) { memcpy(old, newv, sizeof(double) * d); lp[i] = lnew; nacc[i]++; }
            }
            if (it < burnin) {
                double step = fmin(1.0, (double)d / pow((double)it, 2.0 / 3.0));
                double nrm = 0.0;
                for (int k = 0; k < d; k++) nrm += unit[k] * unit[k];
                double fac = sqrt(step * fabs(alpha - 0.25)) / sqrt(nrm);
                for (int k = 0; k < d; k++) scaled[k] *= fac;
                orc_chol_update_r1(d, R, scaled, alpha < 0.25);
            }
            if (i > 0) {
                /* ExchangeStep::DoStep for (i, i-1) (steps.hpp:318-362) */
                double a = 1.0 / temps[i] * (lp[i - 1] - lp[i]) + 1.0 / temps[i - 1] * (lp[i] - lp[i - 1]);
                double u = rng_u01(&rng);
                a = fmin(exp(a), 1.0);
                if (!isfinite(a)) a = 0.0;
                if (u < a) {
                    for (int k = 0; k < d; k++) { double tmp = theta[i * d + k]; theta[i * d + k] = theta[(i - 1) * d + k]; theta[(i - 1) * d + k] = tmp; }
                    double tl = lp[i]; lp[i] = lp[i - 1]; lp[i - 1] = tl;
                    nswap[i]++;
                }
            }
        }
        if (it >= burnin && ((it - burnin + 1) % thin) == 0) {         /* SaveValues (samplers.cpp:118-124) */
            long s = (it - burnin + 1) / thin - 1;
            memcpy(samples + s * d, theta, sizeof(double) * d);
            logposts[s] = lp[0];
        }
    }
    for (int i = 0; i < T; i++) { if (accept_rate) accept_rate[i] = (double)nacc[i] / total; if (swap_rate) swap_rate[i] = (double)nswap[i] / total; }
    free(work); free(theta); free(lp); free(temps); free(chol); free(unit); free(scaled); free(newv); free(nacc); free(nswap);
    return 0;
}
