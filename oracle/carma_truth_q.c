/* carma_truth_q.c -- the reference's log-density formulas in QUAD precision (gcc __float128, 113-bit mantissa).
 *
 * TEST INFRASTRUCTURE ONLY (like everything under oracle/): the arbiter of the parity tests.  Where AR roots cluster
 * (the prior admits roots 1e-4 apart, src/carpack.cpp:330) the reference's double-precision arithmetic -- LU solve of
 * the Vandermonde system, p-term sums that cancel -- is itself 1e-10 ... 1e-3 away from the exact value of its own
 * formulas, so "who is right" needs more digits.  This file is a literal restatement, operation by operation, of
 *     CARp::ARRoots            src/carpack.cpp:137-172
 *     CARMA::ExtractMA         src/carpack.cpp:522-580 (polycoefs :742-756)
 *     CARp::Variance           src/carpack.cpp:377-409
 *     KalmanFilterp::Reset     src/kfilter.cpp:138-186   (Gaussian elimination with partial pivoting for arma::solve)
 *     KalmanFilterp::Update    src/kfilter.cpp:189-215
 *     CARMA_Base::LogDensity   src/include/carpack.hpp:131-176 (sum :167-171, prior :118-126; no bounds check)
 * with every double replaced by __float128.  Conditioning up to 1e13 leaves > 20 correct digits.  It is pinned
 * against tests/mp_truth.py (mpmath, 50 digits) in tests/test_oracle_golden.py; it exists because mpmath needs a
 * minute for one 10^4-point evaluation and this needs half a second.
 */
#include <quadmath.h>
#include <stdlib.h>

typedef __float128 Q;
typedef struct {
    Q re, im;
} QC;

#define PMAXQ 8

static QC qc(Q re, Q im)
{
    QC z = {re, im};
    return z;
}
static QC qadd(QC a, QC b) { return qc(a.re + b.re, a.im + b.im); }
static QC qsub(QC a, QC b) { return qc(a.re - b.re, a.im - b.im); }
static QC qmul(QC a, QC b) { return qc(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static QC qconj(QC a) { return qc(a.re, -a.im); }
static QC qscale(QC a, Q s) { return qc(a.re * s, a.im * s); }
static QC qdiv(QC a, QC b)
{
    const Q den = b.re * b.re + b.im * b.im;
    return qc((a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den);
}
static Q qabs2(QC a) { return a.re * a.re + a.im * a.im; }
static QC qexp(QC a)
{
    const Q e = expq(a.re);
    return qc(e * cosq(a.im), e * sinq(a.im));
}

/* roots of a polynomial given by m log quadratic-factor coefficients (carpack.cpp:137-172, :522-552) */
static void quad_roots_q(const double* lq, int m, QC* roots)
{
    for (int i = 0; i < m / 2; i++) {
        const Q q1 = expq((Q)lq[2 * i]), q2 = expq((Q)lq[2 * i + 1]);
        const Q disc = q2 * q2 - 4 * q1;
        if (disc > 0) {
            const Q s = sqrtq(disc);
            roots[2 * i] = qc(-(q2 + s) / 2, 0);
            roots[2 * i + 1] = qc(-(q2 - s) / 2, 0);
        } else {
            const Q s = sqrtq(-disc);
            roots[2 * i] = qc(-q2 / 2, -s / 2);
            roots[2 * i + 1] = qc(-q2 / 2, s / 2);
        }
    }
    if (m % 2) roots[m - 1] = qc(-expq((Q)lq[m - 1]), 0);
}

static QC qpowi(QC a, int k)
{
    QC r = qc(1, 0);
    for (int i = 0; i < k; i++) r = qmul(r, a);
    return r;
}

/* returns 0 on success; out[0] = log-likelihood + log prior, out[1] = log-likelihood */
static int truth_run(int n, const double* t, const double* y, const double* yerr, int p, int q, const double* theta,
                     double* out, double* mean_out, double* var_out)
{
    if (p < 2 || p >= PMAXQ || q < 0 || q >= p || n < 1) return -1;
    QC om[PMAXQ];
    Q ma[PMAXQ];
    quad_roots_q(theta + 3, p, om);
    for (int i = 0; i < p; i++) ma[i] = 0;
    if (q == 0) {
        ma[0] = 1;
    } else {
        QC mr[PMAXQ], cf[PMAXQ + 1];
        quad_roots_q(theta + 3 + p, q, mr);
        cf[0] = qc(1, 0);
        for (int i = 1; i <= q; i++) cf[i] = qc(0, 0);
        for (int i = 0; i < q; i++)
            for (int k = i + 1; k >= 1; k--) cf[k] = qsub(cf[k], qmul(mr[i], cf[k - 1]));
        for (int i = 0; i <= q; i++) ma[i] = cf[q - i].re / cf[q].re;
    }
    /* Variance(omega, ma, 1)  (carpack.cpp:377-409) */
    QC var1 = qc(0, 0);
    for (int k = 0; k < p; k++) {
        QC dp = qc(1, 0);
        for (int l = 0; l < p; l++)
            if (l != k) dp = qmul(dp, qmul(qsub(om[l], om[k]), qadd(qconj(om[l]), om[k])));
        const QC den = qscale(dp, -2 * om[k].re);
        QC s1 = qc(0, 0), s2 = qc(0, 0);
        for (int l = 0; l < p; l++) {
            s1 = qadd(s1, qscale(qpowi(om[k], l), ma[l]));
            s2 = qadd(s2, qscale(qpowi(qscale(om[k], -1), l), ma[l]));
        }
        var1 = qadd(var1, qdiv(qmul(s1, s2), den));
    }
    const Q sigsqr = (Q)theta[0] * (Q)theta[0] / var1.re;
    const Q scale = (Q)theta[1], mu = (Q)theta[2];
    /* Reset (kfilter.cpp:138-186): E J = e_{p-1} by Gaussian elimination with partial pivoting */
    QC E[PMAXQ][PMAXQ], J[PMAXQ], A[PMAXQ][PMAXQ + 1];
    for (int i = 0; i < p; i++)
        for (int j = 0; j < p; j++) E[i][j] = qpowi(om[j], i);
    for (int i = 0; i < p; i++) {
        for (int j = 0; j < p; j++) A[i][j] = E[i][j];
        A[i][p] = qc(i == p - 1 ? 1 : 0, 0);
    }
    for (int k = 0; k < p; k++) {
        int piv = k;
        for (int i = k + 1; i < p; i++)
            if (qabs2(A[i][k]) > qabs2(A[piv][k])) piv = i;
        if (qabs2(A[piv][k]) == 0) return 1;
        if (piv != k)
            for (int j = 0; j <= p; j++) {
                QC tmp = A[k][j];
                A[k][j] = A[piv][j];
                A[piv][j] = tmp;
            }
        for (int i = k + 1; i < p; i++) {
            const QC f = qdiv(A[i][k], A[k][k]);
            for (int j = k; j <= p; j++) A[i][j] = qsub(A[i][j], qmul(f, A[k][j]));
        }
    }
    for (int i = p - 1; i >= 0; i--) {
        QC s = A[i][p];
        for (int j = i + 1; j < p; j++) s = qsub(s, qmul(A[i][j], J[j]));
        J[i] = qdiv(s, A[i][i]);
    }
    QC b[PMAXQ], V[PMAXQ][PMAXQ], P[PMAXQ][PMAXQ], x[PMAXQ];
    for (int j = 0; j < p; j++) {
        b[j] = qc(0, 0);
        for (int i = 0; i < p; i++) b[j] = qadd(b[j], qscale(E[i][j], ma[i]));
        x[j] = qc(0, 0);
    }
    for (int i = 0; i < p; i++)
        for (int j = 0; j < p; j++) {
            V[i][j] = qdiv(qscale(qmul(J[i], qconj(J[j])), -sigsqr), qadd(om[i], qconj(om[j])));
            P[i][j] = V[i][j];
        }
    Q var = 0;
    for (int i = 0; i < p; i++) {
        QC s = qc(0, 0);
        for (int j = 0; j < p; j++) s = qadd(s, qmul(P[i][j], qconj(b[j])));
        var += qmul(b[i], s).re;
    }
    var += scale * (Q)yerr[0] * (Q)yerr[0];
    Q mean = 0, innov = (Q)y[0] - mu;
    if (mean_out) mean_out[0] = 0.0;
    if (var_out) var_out[0] = (double)var;
    Q ll = -logq(var) / 2 - innov * innov / var / 2;
    for (int k = 1; k < n; k++) {
        /* Update (kfilter.cpp:189-215) */
        QC g[PMAXQ], rho[PMAXQ];
        for (int i = 0; i < p; i++) {
            QC s = qc(0, 0);
            for (int j = 0; j < p; j++) s = qadd(s, qmul(P[i][j], qconj(b[j])));
            g[i] = qscale(s, 1 / var);
        }
        for (int i = 0; i < p; i++) x[i] = qadd(x[i], qscale(g[i], innov));
        for (int i = 0; i < p; i++)
            for (int j = 0; j < p; j++) P[i][j] = qsub(P[i][j], qscale(qmul(g[i], qconj(g[j])), var));
        const Q dt = (Q)t[k] - (Q)t[k - 1];
        for (int i = 0; i < p; i++) rho[i] = qexp(qscale(om[i], dt));
        for (int i = 0; i < p; i++) x[i] = qmul(rho[i], x[i]);
        for (int i = 0; i < p; i++)
            for (int j = 0; j < p; j++)
                P[i][j] = qadd(qmul(qmul(rho[i], qconj(rho[j])), qsub(P[i][j], V[i][j])), V[i][j]);
        mean = 0;
        var = 0;
        for (int i = 0; i < p; i++) {
            mean += qmul(b[i], x[i]).re;
            QC s = qc(0, 0);
            for (int j = 0; j < p; j++) s = qadd(s, qmul(P[i][j], qconj(b[j])));
            var += qmul(b[i], s).re;
        }
        var += scale * (Q)yerr[k] * (Q)yerr[k];
        if (mean_out) mean_out[k] = (double)mean;
        if (var_out) var_out[k] = (double)var;
        innov = (Q)y[k] - mu - mean;
        ll += -logq(var) / 2 - innov * innov / var / 2;
    }
    const Q logprior = -(Q)50 / 2 / scale - 26 * logq(scale); /* carpack.hpp:118-126, measerr_dof = 50 */
    out[0] = (double)(ll + logprior);
    out[1] = (double)ll;
    return 0;
}

int orc_truth_logdensity(int n, const double* t, const double* y, const double* yerr, int p, int q, const double* theta,
                         double* out)
{
    return truth_run(n, t, y, yerr, p, q, theta, out, 0, 0);
}

/* the same filter, returning mean[n] / var[n] (KalmanFilter::Filter's public vectors, kfilter.hpp:31-32) of the data
 * y - theta[2] with errors sqrt(theta[1]) yerr -- the arbiter of the mean / variance comparisons */
int orc_truth_filter(int n, const double* t, const double* y, const double* yerr, int p, int q, const double* theta,
                     double* mean, double* var)
{
    double out[2];
    return truth_run(n, t, y, yerr, p, q, theta, out, mean, var);
}

/* CARp::Variance(omega, ma, sigma = 1, lag 0) (src/carpack.cpp:377-409; CarmaSample._sigma_noise, carma_pack.py:513-546, forms the
 * same sum) in quad precision, from roots and MA coefficients GIVEN AS DOUBLES: the arbiter of carma_sigma_noise_batch.
 * nma coefficients (nma <= p).  Returns 0; out[0] = the variance for unit driving noise (sigma_noise = sqrt(var / out[0])).       */
int orc_truth_variance(int p, const double* om_re, const double* om_im, const double* ma, int nma, double* out)
{
    /* out[0] = the variance; out[1] = sum over the roots of |term_k| (the summation's condition number is out[1] / |out[0]|) */
    if (p < 1 || p >= PMAXQ || nma < 1 || nma > p) return -1;
    Q sabs = 0;
    QC om[PMAXQ];
    for (int k = 0; k < p; k++) om[k] = qc((Q)om_re[k], (Q)om_im[k]);
    QC var1 = qc(0, 0);
    for (int k = 0; k < p; k++) {
        QC dp = qc(1, 0);
        for (int l = 0; l < p; l++)
            if (l != k) dp = qmul(dp, qmul(qsub(om[l], om[k]), qadd(qconj(om[l]), om[k])));
        const QC den = qscale(dp, -2 * om[k].re);
        QC s1 = qc(0, 0), s2 = qc(0, 0);
        for (int l = 0; l < nma; l++) {
            s1 = qadd(s1, qscale(qpowi(om[k], l), (Q)ma[l]));
            s2 = qadd(s2, qscale(qpowi(qscale(om[k], -1), l), (Q)ma[l]));
        }
        const QC term = qdiv(qmul(s1, s2), den);
        var1 = qadd(var1, term);
        sabs += sqrtq(qabs2(term));
    }
    out[0] = (double)var1.re;
    out[1] = (double)sabs;
    return 0;
}

