// carma_scan.h -- the Kalman log-likelihood as an ASSOCIATIVE SCAN OVER TIME (latency regime).
//
// With <= 1024 evaluations in flight the sequential filter (carma_core.h / carma_pipe3l.h) is bound by
// n - 1 dependent steps of one wave.  Here ONE WAVE serves one evaluation and every LANE owns a block of
// s = ceil(n / 64) consecutive data: the filter over a block is an element (A, b, C, eta, J) of the
// associative operator of Sarkka & Garcia-Fernandez (IEEE TAC 2021, "Temporal parallelization of
// Bayesian smoothers", eqs. for the filtering elements):
//     p(x_end | x_start, y_block) = N(A x_start + b, C),   p(y_block | x_start) ~ N_info(x_start; eta, J)
//   phase 1  each lane builds its block element by forward recursions from "x_start known exactly"
//   phase 2  inclusive Hillis-Steele scan over the 64 lanes with the combine below (6 levels, LDS exchange)
//   phase 3  each lane runs the ordinary filter over its block from the prefix state of the lane before
//            it and accumulates its share of  -1/2 sum(log var_k + innov_k^2 / var_k)
// Same model, same likelihood as kfilter.cpp:138-215 + carpack.hpp:167-171 in the REAL modal coordinates of
// filter_loop_real (state z, transition Phi block diagonal, y = h.z, stationary covariance Vz); the
// covariances are carried as D = C - Vz wherever the sequential code does.  ~4x the arithmetic of the
// sequential filter, ~1/10 of its dependent depth, all 64 lanes busy.  Plain C++ (fma), shared with the
// CPU test harness (tests/emu).
//
// STATUS: experimental, opt-in (CARMA_LOGDENS_KERNEL=scan).  It matches the CPU reference path to 1e-10 on ordinary
// parameter vectors, but its rounding errors are amplified by the signal-to-noise ratio of the model (the
// combine solves with W = I + C J, J ~ 1/yerr^2): 1e-9 on theta #330 of the bench batch, where the sequential
// filter is at 4e-13 -- it does not meet the parity bar, and at 40 us per 1024-evaluation launch it is not
// faster than the three-wave pipeline either.
#pragma once
#include "carma_core.h"

namespace carma {

template <int P>
struct ScanDim {
    static constexpr int NS = P * (P + 1) / 2;                 // packed symmetric
    static constexpr int NE = P * P + 2 * P + 2 * NS;          // doubles per element
};
template <int P>
CARMA_DEV constexpr int sym_idx(int i, int j)                   // i <= j
{
    return i * P - i * (i - 1) / 2 + (j - i);
}
template <int P>
CARMA_DEV double sym_get(const double (&S)[ScanDim<P>::NS], int i, int j)
{
    return i <= j ? S[sym_idx<P>(i, j)] : S[sym_idx<P>(j, i)];
}

// Model of one evaluation in real modal coordinates, replicated in every lane.
template <int P>
struct ScanModel {
    double wre[P], wim[P];                   // roots (a complex pair occupies two consecutive slots)
    bool cpx[P];                             // coordinate belongs to a complex pair
    double h[P];                             // observation row
    double Vz[ScanDim<P>::NS];               // stationary covariance of z
    double c[P];                             // Vz h^T
    double s0;                               // h Vz h^T
    double mu, scale;
};

template <int P>
struct ScanElem {
    double A[P][P];
    double b[P];
    double C[ScanDim<P>::NS];
    double eta[P];
    double J[ScanDim<P>::NS];
};

template <int P>
CARMA_DEV void scan_identity(ScanElem<P>& e)
{
#pragma unroll
    for (int i = 0; i < P; i++) {
#pragma unroll
        for (int j = 0; j < P; j++) e.A[i][j] = (i == j) ? 1.0 : 0.0;
        e.b[i] = 0.0;
        e.eta[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < ScanDim<P>::NS; i++) {
        e.C[i] = 0.0;
        e.J[i] = 0.0;
    }
}

// transition of one step: (c_r, s_r) per coordinate as in filter_loop_real (odd member of a pair holds the
// conjugate): (Phi v)_r = c_r v_r - s_r v_{r^1}
template <int P>
struct ScanPhi {
    double c[P], s[P];
};
template <int P>
CARMA_DEV void scan_phi(const ScanModel<P>& m, double dt, ScanPhi<P>& f)
{
#pragma unroll
    for (int r = 0; r < P; r++) {
        if (m.cpx[r] && (r & 1)) {                 // odd member: conjugate of the even one
            f.c[r] = f.c[r - 1];
            f.s[r] = -f.s[r - 1];
        } else {
            cexp_step(m.wre[r], m.wim[r], dt, &f.c[r], &f.s[r]);
        }
    }
}
// S <- Phi S Phi^T for a packed symmetric S (via the full matrix)
template <int P>
CARMA_DEV void phi_sym(const ScanPhi<P>& f, double (&S)[ScanDim<P>::NS])
{
    double F[P][P], G[P][P];
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j < P; j++) F[i][j] = sym_get<P>(S, i, j);
    // columns: G_ij = F_ij c_j - F_i,j^1 s_j
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j < P; j++) {
            const int jp = (j ^ 1) < P ? (j ^ 1) : j;
            G[i][j] = fma(F[i][j], f.c[j], -(F[i][jp] * f.s[j]));
        }
    // rows (upper triangle only): S_ij = c_i G_ij - s_i G_{i^1,j}
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = i; j < P; j++) {
            const int ip = (i ^ 1) < P ? (i ^ 1) : i;
            S[sym_idx<P>(i, j)] = fma(f.c[i], G[i][j], -(f.s[i] * G[ip][j]));
        }
}

// One measurement update shared by phase 1 and phase 3 (D = C - Vz):  var = h D h + s0 + e,
// gain K = (D h + c) / var;  returns var and leaves K, w = D h in the caller's arrays.
template <int P>
CARMA_DEV double scan_gain(const ScanModel<P>& m, const double (&D)[ScanDim<P>::NS], double e, double (&K)[P])
{
    double var = m.s0 + e * m.scale;
    double w[P];
#pragma unroll
    for (int i = 0; i < P; i++) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < P; j++) a = fma(sym_get<P>(D, i, j), m.h[j], a);
        w[i] = a;
    }
#pragma unroll
    for (int i = 0; i < P; i++) var = fma(m.h[i], w[i], var);
    const double sv = recip(var);
#pragma unroll
    for (int i = 0; i < P; i++) K[i] = (w[i] + m.c[i]) * sv;
    return var;
}

// Same for a covariance carried as C itself (phase 1: a block starts from C = 0, where the D form would
// compute var = s0 - h Vz h + e by cancellation):  var = h C h + e,  K = C h / var.
template <int P>
CARMA_DEV double scan_gain_c(const ScanModel<P>& m, const double (&C)[ScanDim<P>::NS], double e, double (&K)[P])
{
    double var = e * m.scale;
    double w[P];
#pragma unroll
    for (int i = 0; i < P; i++) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < P; j++) a = fma(sym_get<P>(C, i, j), m.h[j], a);
        w[i] = a;
    }
#pragma unroll
    for (int i = 0; i < P; i++) var = fma(m.h[i], w[i], var);
    const double sv = recip(var);
#pragma unroll
    for (int i = 0; i < P; i++) K[i] = w[i] * sv;
    return var;
}

// phase 1: the element of the block of data [k0, k1) (k0 < k1).  `first`: the block starts the series
// (x_0 ~ N(0, Vz) instead of a known x_start).  phis[] receives the transition factors of the block's
// steps for reuse in phase 3.
template <int P, int SMAX>
CARMA_DEV void scan_block_element(const ScanModel<P>& m, const double4* __restrict__ series, int k0, int k1, bool first,
                                  ScanElem<P>& el, ScanPhi<P> (&phis)[SMAX])
{
    double D[ScanDim<P>::NS];                        // the block's C (x_start known exactly: C = 0; series start: Vz)
    scan_identity<P>(el);
#pragma unroll
    for (int i = 0; i < ScanDim<P>::NS; i++) D[i] = first ? m.Vz[i] : 0.0;
    if (first) {
#pragma unroll
        for (int i = 0; i < P; i++) el.A[i][i] = 0.0;
    }
#pragma unroll
    for (int t = 0; t < SMAX; t++) {
        const int k = k0 + t;
        if (k < k1) {
            const double4 rec = series[k];
            if (!(first && t == 0)) {
                scan_phi<P>(m, rec.x, phis[t]);
                const ScanPhi<P>& f = phis[t];
                // A <- Phi A (rows), b <- Phi b, D <- Phi D Phi^T
#pragma unroll
                for (int j = 0; j < P; j++) {
                    double col[P];
#pragma unroll
                    for (int i = 0; i < P; i++) col[i] = el.A[i][j];
#pragma unroll
                    for (int i = 0; i < P; i++) {
                        const int ip = (i ^ 1) < P ? (i ^ 1) : i;
                        el.A[i][j] = fma(f.c[i], col[i], -(f.s[i] * col[ip]));
                    }
                }
                {
                    double bb[P];
#pragma unroll
                    for (int i = 0; i < P; i++) bb[i] = el.b[i];
#pragma unroll
                    for (int i = 0; i < P; i++) {
                        const int ip = (i ^ 1) < P ? (i ^ 1) : i;
                        el.b[i] = fma(f.c[i], bb[i], -(f.s[i] * bb[ip]));
                    }
                }
                // C <- Phi (C - Vz) Phi^T + Vz   (kfilter.cpp:204)
#pragma unroll
                for (int i = 0; i < ScanDim<P>::NS; i++) D[i] -= m.Vz[i];
                phi_sym<P>(f, D);
#pragma unroll
                for (int i = 0; i < ScanDim<P>::NS; i++) D[i] += m.Vz[i];
            }
            double hA[P], K[P];
#pragma unroll
            for (int j = 0; j < P; j++) {
                double a = 0.0;
#pragma unroll
                for (int i = 0; i < P; i++) a = fma(m.h[i], el.A[i][j], a);
                hA[j] = a;
            }
            const double var = scan_gain_c<P>(m, D, rec.z, K);
            const double sv = recip(var);
            double r = rec.y - m.mu;
#pragma unroll
            for (int i = 0; i < P; i++) r = fma(-m.h[i], el.b[i], r);
            const double rs = r * sv;
#pragma unroll
            for (int i = 0; i < P; i++) {
                el.eta[i] = fma(hA[i], rs, el.eta[i]);
                const double hs = hA[i] * sv;
#pragma unroll
                for (int j = i; j < P; j++) el.J[sym_idx<P>(i, j)] = fma(hs, hA[j], el.J[sym_idx<P>(i, j)]);
            }
#pragma unroll
            for (int i = 0; i < P; i++) {
#pragma unroll
                for (int j = 0; j < P; j++) el.A[i][j] = fma(-K[i], hA[j], el.A[i][j]);
                el.b[i] = fma(K[i], r, el.b[i]);
                const double kv = K[i] * var;
#pragma unroll
                for (int j = i; j < P; j++) D[sym_idx<P>(i, j)] = fma(-kv, K[j], D[sym_idx<P>(i, j)]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < ScanDim<P>::NS; i++) el.C[i] = D[i];
}

// out = e1 (x) e2  (e1 earlier in time).  Gaussian elimination without pivoting on W = I + C1 J2, whose
// spectrum is that of I + C1^(1/2) J2 C1^(1/2) >= I.
template <int P>
CARMA_DEV void scan_combine(const ScanElem<P>& e1, const ScanElem<P>& e2, ScanElem<P>& o)
{
    double W[P][P];
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j < P; j++) {
            double a = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < P; k++) a = fma(sym_get<P>(e1.C, i, k), sym_get<P>(e2.J, k, j), a);
            W[i][j] = a;
        }
    // LU in place (unit lower), reciprocal pivots
    double ipiv[P];
#pragma unroll
    for (int k = 0; k < P; k++) {
        ipiv[k] = recip(W[k][k]);
#pragma unroll
        for (int i = k + 1; i < P; i++) {
            const double l = W[i][k] * ipiv[k];
            W[i][k] = l;
#pragma unroll
            for (int j = k + 1; j < P; j++) W[i][j] = fma(-l, W[k][j], W[i][j]);
        }
    }
    // M = A2 W^{-1}: row by row, x U = a (forward over columns), then y L = x (backward)
    double M[P][P];
#pragma unroll
    for (int i = 0; i < P; i++) {
        double x[P];
#pragma unroll
        for (int j = 0; j < P; j++) {
            double a = e2.A[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) a = fma(-x[k], W[k][j], a);
            x[j] = a * ipiv[j];
        }
#pragma unroll
        for (int j = P - 1; j >= 0; j--) {
            double a = x[j];
#pragma unroll
            for (int k = j + 1; k < P; k++) a = fma(-x[k], W[k][j], a);
            x[j] = a;
        }
#pragma unroll
        for (int j = 0; j < P; j++) M[i][j] = x[j];
    }
    // A = M A1 ; b = M (b1 + C1 eta2) + b2 ; C = M C1 A2^T + C2
    double v[P];
#pragma unroll
    for (int i = 0; i < P; i++) {
        double a = e1.b[i];
#pragma unroll
        for (int k = 0; k < P; k++) a = fma(sym_get<P>(e1.C, i, k), e2.eta[k], a);
        v[i] = a;
    }
    double MC[P][P];
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j < P; j++) {
            double a = 0.0, c = 0.0;
#pragma unroll
            for (int k = 0; k < P; k++) {
                a = fma(M[i][k], e1.A[k][j], a);
                c = fma(M[i][k], sym_get<P>(e1.C, k, j), c);
            }
            o.A[i][j] = a;
            MC[i][j] = c;
        }
#pragma unroll
    for (int i = 0; i < P; i++) {
        double a = e2.b[i];
#pragma unroll
        for (int k = 0; k < P; k++) a = fma(M[i][k], v[k], a);
        o.b[i] = a;
#pragma unroll
        for (int j = i; j < P; j++) {
            double c = sym_get<P>(e2.C, i, j);
#pragma unroll
            for (int k = 0; k < P; k++) c = fma(MC[i][k], e2.A[j][k], c);
            o.C[sym_idx<P>(i, j)] = c;
        }
    }
    // N = A1^T (I + J2 C1)^{-1} = (W^{-1} A1)^T ;  X = W^{-1} A1 column by column (L y = a, then U x = y)
    double X[P][P];
#pragma unroll
    for (int j = 0; j < P; j++) {
        double y[P];
#pragma unroll
        for (int i = 0; i < P; i++) {
            double a = e1.A[i][j];
#pragma unroll
            for (int k = 0; k < i; k++) a = fma(-W[i][k], y[k], a);
            y[i] = a;
        }
#pragma unroll
        for (int i = P - 1; i >= 0; i--) {
            double a = y[i];
#pragma unroll
            for (int k = i + 1; k < P; k++) a = fma(-W[i][k], y[k], a);
            y[i] = a * ipiv[i];
        }
#pragma unroll
        for (int i = 0; i < P; i++) X[i][j] = y[i];
    }
    // (C1 and J2 are symmetric, so I + J2 C1 = W^T and N = A1^T W^{-T} = X^T)
    //   eta = N (eta2 - J2 b1) + eta1 ,  J = N J2 A1 + J1
    double u[P];
#pragma unroll
    for (int i = 0; i < P; i++) {
        double a = e2.eta[i];
#pragma unroll
        for (int k = 0; k < P; k++) a = fma(-sym_get<P>(e2.J, i, k), e1.b[k], a);
        u[i] = a;
    }
    double JA[P][P];                                 // J2 A1
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j < P; j++) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < P; k++) a = fma(sym_get<P>(e2.J, i, k), e1.A[k][j], a);
            JA[i][j] = a;
        }
#pragma unroll
    for (int i = 0; i < P; i++) {
        double a = e1.eta[i];
#pragma unroll
        for (int k = 0; k < P; k++) a = fma(X[k][i], u[k], a);
        o.eta[i] = a;
#pragma unroll
        for (int j = i; j < P; j++) {
            double c = e1.J[sym_idx<P>(i, j)];
#pragma unroll
            for (int k = 0; k < P; k++) c = fma(X[k][i], JA[k][j], c);
            o.J[sym_idx<P>(i, j)] = c;
        }
    }
}

// phase 3: the ordinary filter over the block [k0, k1) from the state (m, C) at its start (prefix of the
// lane before); accumulates the block's share of the log-likelihood.
template <int P, int SMAX>
CARMA_DEV void scan_block_loglik(const ScanModel<P>& m, const double4* __restrict__ series, int k0, int k1, bool first,
                                 const double (&m0)[P], const double (&C0)[ScanDim<P>::NS],
                                 const ScanPhi<P> (&phis)[SMAX], LogLikAcc& acc)
{
    double z[P], D[ScanDim<P>::NS];
#pragma unroll
    for (int i = 0; i < P; i++) z[i] = first ? 0.0 : m0[i];
#pragma unroll
    for (int i = 0; i < ScanDim<P>::NS; i++) D[i] = first ? m.Vz[i] : C0[i];
#pragma unroll
    for (int t = 0; t < SMAX; t++) {
        const int k = k0 + t;
        if (k < k1) {
            const double4 rec = series[k];
            if (!(first && t == 0)) {
                const ScanPhi<P>& f = phis[t];
                double zz[P];
#pragma unroll
                for (int i = 0; i < P; i++) zz[i] = z[i];
#pragma unroll
                for (int i = 0; i < P; i++) {
                    const int ip = (i ^ 1) < P ? (i ^ 1) : i;
                    z[i] = fma(f.c[i], zz[i], -(f.s[i] * zz[ip]));
                }
#pragma unroll
                for (int i = 0; i < ScanDim<P>::NS; i++) D[i] -= m.Vz[i];
                phi_sym<P>(f, D);
#pragma unroll
                for (int i = 0; i < ScanDim<P>::NS; i++) D[i] += m.Vz[i];
            }
            double K[P];
            const double var = scan_gain_c<P>(m, D, rec.z, K);
            double r = rec.y - m.mu;
#pragma unroll
            for (int i = 0; i < P; i++) r = fma(-m.h[i], z[i], r);
            acc.add_var(var);
            acc.chi2 += r * (recip(var) * r);
#pragma unroll
            for (int i = 0; i < P; i++) {
                z[i] = fma(K[i], r, z[i]);
                const double kv = K[i] * var;
#pragma unroll
                for (int j = i; j < P; j++) D[sym_idx<P>(i, j)] = fma(-kv, K[j], D[sym_idx<P>(i, j)]);
            }
        }
    }
}

}  // namespace carma
