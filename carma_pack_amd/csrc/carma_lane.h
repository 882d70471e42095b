// carma_lane.h -- the CARMA(p,q) Kalman log-density with ONE EVALUATION PER LANE (round 3): the throughput regime.
//
// The lane-group kernels of carma_core.h give an evaluation G = 8 lanes (p = 5..7) of which p work, and every step pays
// for talking between them: two 3-stage DPP butterflies (var, mean), an LDS all-gather of the gain, DPP moves of the
// pair partner's row in the rotation -- 127.6 VALU instructions per wave-step for EIGHT evaluations (16 per
// evaluation-step, profiles/r03/pmc_v1_tput.json).  With tens of thousands of evaluations in flight none of that is
// needed: here a lane holds the whole p x p matrix D of its evaluation (symmetric: p (p + 1) / 2 registers), nothing
// crosses lanes, all 64 lanes work, and a wave-step of ~330 instructions serves SIXTY-FOUR evaluations (~5 per
// evaluation-step).  The series record of a step is still wave-uniform (scalar loads), and so is the time step: a step
// that repeats its predecessor's dt re-uses the transition factors (the regular-cadence variant of the other kernels
// is a wave-uniform branch here, always on).
//
// Same model, same recursion as filter_loop_real (REAL modal coordinates; reference: KalmanFilterp::Reset / Update,
// src/kfilter.cpp:138-215; CARMA_Base::LogDensity, src/include/carpack.hpp:131-176; set-up closed forms: struct Model
// in carma_core.h) -- only the distribution of the work over lanes differs:
//     coordinates     z_{2k} = Re x_{2k}, z_{2k+1} = Im x_{2k} for a complex pair, z_r = x_r for a real root
//     h_{2k} = 2 Re b_{2k}, h_{2k+1} = -2 Im b_{2k} (real root: b_r);   c_{2k} = Re (V b^H)_{2k}, c_{2k+1} = Im (V b^H)_{2k}
//     var  = s0 + h.w + e,  mean = h.z,  k = w + c,  w = D h
//     z   <- Phi (z + k innov / var),   D <- Phi (D - k k^T / var) Phi^T
//     Phi : rotation-scaling [[c,-s],[s,c]] per complex pair (c + i s = exp(omega_{2k} dt)), a scalar per real root;
//           stored per coordinate as (c_r, s_r) with s_{2k+1} = -s_{2k}, so that
//           (d Phi^T)_{ij} = d_ij c_j - d_{i,j^1} s_j   and   (Phi m)_{ij} = c_i m_ij - s_i m_{i^1,j}
//           hold for every coordinate (s = 0 for a real root, whose partner index is never dereferenced).
// Plain C++ over doubles: compiled for the host by the test harness as well (tests/emu).
#pragma once
#include "carma_core.h"

namespace carma {

template <int P>
struct LaneModel {
    double h[P], c[P];         // observation row and gain offset in real coordinates
    double wre[P], wim[P];     // AR roots
    double s0, scale, mu;
    bool valid, sing;
    bool realpair[(P + 1) / 2];   // quadratic factor i has two REAL roots (its members then rotate separately)
};

// theta -> everything the recursion needs (ARRoots, ExtractMA, ExtractSigsqr, CheckPriorBounds; closed forms of struct Model)
template <int P>
CARMA_DEV void lane_model_from_theta(const double* theta, int q, const Prior& pr, int ignore_prior, LaneModel<P>& m)
{
    constexpr int NMA = P > 1 ? P - 1 : 1;
    Cx w[P], mu[NMA];
#pragma unroll
    for (int j = 0; j < P; j++) w[j] = poly_root(theta + 3, P, j);
#pragma unroll
    for (int k = 0; k < NMA; k++) mu[k] = k < q ? poly_root(theta + 3 + P, q, k) : Cx{-1.0, 0.0};
    // prod_k mu_k is real (conjugate pairs and real roots)
    Cx pmu = {1.0, 0.0};
#pragma unroll
    for (int k = 0; k < NMA; k++)
        if (k < q) pmu = cmul(pmu, mu[k]);
    const double rmu = 1.0 / pmu.re;
    Cx b[P], kap[P];
    bool sing = false;
    double var1 = 0.0;
#pragma unroll
    for (int r = 0; r < P; r++) {
        // b_r = beta(omega_r) = prod_k (mu_k - omega_r) / mu_k,  beta(-omega_r) = prod_k (mu_k + omega_r) / mu_k
        Cx pb = {1.0, 0.0}, pm = {1.0, 0.0};
#pragma unroll
        for (int k = 0; k < NMA; k++) {
            if (k < q) {
                pb = cmul(pb, csub(mu[k], w[r]));
                pm = cmul(pm, cadd(mu[k], w[r]));
            }
        }
        b[r] = cscale(pb, rmu);
        // kappa_r = beta(-omega_r) / (alpha'(omega_r) alpha(-omega_r))
        Cx ap = {1.0, 0.0}, am = {1.0, 0.0};
#pragma unroll
        for (int l = 0; l < P; l++) {
            const Cx dl = csub(w[r], w[l]);
            const Cx sl = {-(w[r].re + w[l].re), -(w[r].im + w[l].im)};
            if (l != r) ap = cmul(ap, dl);
            am = cmul(am, sl);
        }
        if (ap.re == 0.0 && ap.im == 0.0) sing = true;      // repeated AR root (arma::solve throws, carpack.hpp:154-164)
        kap[r] = cdiv(cscale(pm, rmu), cmul(ap, am));
        var1 += b[r].re * kap[r].re - b[r].im * kap[r].im;
    }
    // sigma^2 = theta0^2 / Variance(omega, beta, 1)   (carpack.cpp:377-409, carpack.hpp:316-319, 391-395)
    const double sigsqr = theta[0] * theta[0] / var1;
    m.s0 = theta[0] * theta[0];
    m.scale = theta[1];
    m.mu = theta[2];
    m.sing = sing;
#pragma unroll
    for (int r = 0; r < P; r++) {
        m.wre[r] = w[r].re;
        m.wim[r] = w[r].im;
        const bool cpx = (w[r].im != 0.0) && (r < (P & ~1));
        const int ev = r & ~1;                               // the even member of r's pair
        if (cpx) {
            // (the odd member is the conjugate of the even one: h = 2 Im b_odd = -2 Im b_even, c = Im (V b^H)_even)
            m.h[r] = (r & 1) ? 2.0 * b[r].im : 2.0 * b[r].re;
            m.c[r] = sigsqr * ((r & 1) ? kap[ev].im : kap[r].re);
        } else {
            m.h[r] = b[r].re;
            m.c[r] = sigsqr * kap[r].re;
        }
    }
#pragma unroll
    for (int i = 0; i < (P + 1) / 2; i++) m.realpair[i] = (2 * i + 1 < P) && (w[2 * i].im == 0.0);
    // --- prior bounds (carpack.cpp:314-374, unique_roots :709-732)
    m.valid = true;
    if (!ignore_prior) {
        bool viol = false;
        double cent_prev = 0.0;
#pragma unroll
        for (int r = 0; r < P; r++) {
            const double cent = fabs(w[r].im) / 2.0 / (TWO_PI / 2.0);
            const double width = -w[r].re / 2.0 / (TWO_PI / 2.0);
            if (!(cent < pr.max_freq) || !(width < pr.max_freq) || !(width > pr.min_freq)) viol = true;
            if (r >= 1 && (cent - cent_prev) > 1e-8) viol = true;
            cent_prev = cent;
#pragma unroll
            for (int j = r + 1; j < P; j++) {
                // |(w - w_j) / (w + w_j)| <= 1e-4 (carpack.cpp:709-732), compared as squared moduli
                const Cx dn = csub(w[r], w[j]), sm = cadd(w[r], w[j]);
                const double n2 = dn.re * dn.re + dn.im * dn.im, d2 = sm.re * sm.re + sm.im * sm.im;
                if (n2 <= 1e-8 * d2) viol = true;
            }
        }
        const double ysigma = theta[0], ms = theta[1];
        if (viol || (ysigma > pr.max_stdev) || (ysigma < 0) || (ms < 0.5) || (ms > 2.0)) m.valid = false;
    }
}

// index of (i, j), i <= j, in the packed upper triangle
template <int P>
CARMA_DEV constexpr int tri(int i, int j)
{
    return i <= j ? i * P - i * (i - 1) / 2 + (j - i) : j * P - j * (j - 1) / 2 + (i - j);
}

// wave-uniform "does any lane need it" (the host build runs one evaluation at a time)
CARMA_DEV bool lane_any(bool b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(b) != 0;
#else
    return b;
#endif
}

// Transition factors of one step, per coordinate: one exp/sincos per PAIR (the members are conjugates), one exp per real root.
// Round 4: the table-based forms of carma_math.h (tab: the 160 table entries, in LDS on the device) -- 45 instead of 65
// instructions per complex exponential, which were 46 % of this kernel's instruction stream.
template <int P>
CARMA_DEV void lane_factors(const double (&wre)[P], const double (&wim)[P], const bool (&realpair)[(P + 1) / 2], bool anyreal,
                            double dt, double (&cr)[P], double (&sr)[P], const double* tab)
{
#if defined(CARMA_AB_NOTAB)                                  // A/B builds only: the polynomial-only forms of round 3
#define cexp_step_tab(a, b, dt, c, s, tab) cexp_step(a, b, dt, c, s)
#define exp_neg_tab(x, tab) exp_neg(x)
#endif
#pragma unroll
    for (int i = 0; i < P / 2; i++) {
        double c, s;
        cexp_step_tab(wre[2 * i], wim[2 * i], dt, &c, &s, tab);
        cr[2 * i] = c;
        sr[2 * i] = s;
        cr[2 * i + 1] = c;
        sr[2 * i + 1] = -s;
    }
    if (anyreal) {
        // a quadratic factor with two real roots: the second member has its own modulus (and no phase: s = 0)
#pragma unroll
        for (int i = 0; i < P / 2; i++) {
            const double e1 = exp_neg_tab(wre[2 * i + 1] * dt, tab);
            if (realpair[i]) cr[2 * i + 1] = e1;
        }
    }
    if (P & 1) cr[P - 1] = exp_neg_tab(wre[P - 1] * dt, tab);
#if defined(CARMA_AB_NOTAB)
#undef cexp_step_tab
#undef exp_neg_tab
#endif
}

// Where lane_filter takes a NEW time step's factors from.  LaneFactorsInline: the lane computes them itself.
template <int P>
struct LaneFactorsInline {
    static constexpr int UNROLL = 1;
    const LaneModel<P>& m;
    bool anyreal;
    const double* tab;
    CARMA_DEV void step(int) const {}
    CARMA_DEV void get(int, double dt, double (&cr)[P], double (&sr)[P]) const
    {
        lane_factors<P>(m.wre, m.wim, m.realpair, anyreal, dt, cr, sr, tab);
#if defined(__HIP_DEVICE_COMPILE__)
        // the factors enter the recursion as opaque values, as the producers' ring hands them to its consumer: what the
        // compiler knows about them in line (c_{2k+1} = c_{2k}, s_{2k+1} = -s_{2k}) would contract the products of the step
        // differently, and the two kernels are held to the same bits (test_launch_shapes_agree)
#pragma unroll
        for (int r = 0; r < P; r++) {
            asm("" : "+v"(cr[r]));
            asm("" : "+v"(sr[r]));
        }
#endif
    }
};

#if defined(__HIPCC__)
// LaneFactorsRing: PRODUCER WAVES compute them (lane_produce: lane l of a producer serves lane l of its consumer) into a
// two-buffer LDS ring, CH steps per buffer, one workgroup barrier per CH steps; with NP producers per consumer, producer k
// takes the steps s = k (mod NP) of a buffer.  The factors do not depend on the state of the recursion, so the split takes
// the ~220 exp/sincos instructions of a step (p = 5) out of the ~410 of a lone wave's instruction stream -- which IS the run
// time while there is at most one wave per SIMD.  NP = 1: the producer is the longer of the two (224 vs 197 instructions);
// NP = 3: the consumer is all that is left.  A step that repeats its predecessor's time step is skipped by both sides
// (the test is wave-uniform).
template <int P, int NP>
struct LaneRingGeom {
    static constexpr int CH = NP == 1 ? 4 : 2 * NP;
    static constexpr int NC = 4 / (1 + NP);                   // consumers per workgroup of four waves
    static constexpr int NV = P + P / 2;                      // c per coordinate, s per pair
    static constexpr size_t DOUBLES = (size_t)2 * CH * NV * 64;                    // per consumer
    static constexpr size_t BYTES = NC * DOUBLES * sizeof(double);                 // p = 5: 56 KiB (NP = 1), 42 KiB (NP = 3)
};
template <int P, int NP, bool UNROLLED = false>
struct LaneFactorsRing {
    using Geo = LaneRingGeom<P, NP>;
    // UNROLLED (round 5): lane_filter takes a whole buffer of CH steps as ONE basic block (series without repeated time steps) --
    // the barrier in front, the slots at compile-time offsets, the reads of the later steps' factors move up across the earlier
    // steps: 110 -> 95 us at p = 5 while a CU holds one workgroup (profiles/r05/ab_lpc_unroll_v1.txt; p = 3: 64 -> 55, p = 7:
    // 165 -> 147); with two workgroups per CU the step-by-step loop is 1-4 % ahead and stays.  Same bits.
    // (half a buffer at a time -- three steps, or two -- is slower at p = 6, 7, where the whole buffer spills into AGPRs:
    // 131 / 155 us against 124 / 150, profiles/r05/ab_lpc_unroll_hi_v1.txt)
    static constexpr int UNROLL = UNROLLED ? Geo::CH : 1;
    const double* ring;                                       // [2][CH][NV][64], + lane
    // first step kk of a buffer (kk - 1 a multiple of CH)
    CARMA_DEV const double* chunk(int kk) const
    {
        __syncthreads();                                      // barrier c: chunk c is in the ring
        return ring + (size_t)((((kk - 1) / Geo::CH) & 1) * Geo::CH) * Geo::NV * 64;
    }
    CARMA_DEV static void get_at(const double* cb, int s, double (&cr)[P], double (&sr)[P])
    {
        const double* b = cb + (size_t)s * Geo::NV * 64;
#pragma unroll
        for (int r = 0; r < P; r++) cr[r] = b[r * 64];
#pragma unroll
        for (int i = 0; i < P / 2; i++) {
            const double sv = b[(P + i) * 64];
            sr[2 * i] = sv;
            sr[2 * i + 1] = -sv;
        }
    }
    CARMA_DEV void step(int kk) const
    {
        if ((kk - 1) % Geo::CH == 0) __syncthreads();          // barrier c: chunk c is in the ring
    }
    CARMA_DEV void get(int kk, double, double (&cr)[P], double (&sr)[P]) const
    {
        const int c = (kk - 1) / Geo::CH, s = (kk - 1) % Geo::CH;
        const double* b = ring + (size_t)((c & 1) * Geo::CH + s) * Geo::NV * 64;
#pragma unroll
        for (int r = 0; r < P; r++) cr[r] = b[r * 64];
#pragma unroll
        for (int i = 0; i < P / 2; i++) {
            const double sv = b[(P + i) * 64];
            sr[2 * i] = sv;
            sr[2 * i + 1] = -sv;
        }
    }
};
template <int P, int NP, bool REPDT = true>
__device__ __forceinline__ void lane_produce(int k, const double* theta, double* ring /* + lane */, const double4* __restrict__ series,
                                             int n, const double* tab)
{
    using Geo = LaneRingGeom<P, NP>;
    double wre[P], wim[P];
    bool realpair[(P + 1) / 2];
#pragma unroll
    for (int j = 0; j < P; j++) {
        const Cx w = poly_root(theta + 3, P, j);
        wre[j] = w.re;
        wim[j] = w.im;
    }
    bool anyreal = false;
#pragma unroll
    for (int i = 0; i < (P + 1) / 2; i++) {
        realpair[i] = (2 * i + 1 < P) && (wim[2 * i] == 0.0);
        anyreal = anyreal || realpair[i];
    }
    anyreal = __builtin_amdgcn_ballot_w64(anyreal) != 0;
    const int nc = (n - 1 + Geo::CH - 1) / Geo::CH;
    for (int c = 0; c < nc; c++) {
#if defined(CARMA_AB_NOPROD)                                  // timing-only A/B build: the consumer alone
        if (c >= 2) {
            __syncthreads();
            continue;
        }
#endif
#pragma unroll 1
        for (int s = k; s < Geo::CH; s += NP) {
            const int kk = 1 + c * Geo::CH + s;
            if (kk >= n) break;
            const double dt = series[kk].x;
            // as the consumer decides: new factors unless the step repeats its predecessor's time step
            if (!REPDT || kk == 1 || dt != series[kk - 1].x) {
                double cr[P], sr[P];
                lane_factors<P>(wre, wim, realpair, anyreal, dt, cr, sr, tab);
                double* b = ring + (size_t)((c & 1) * Geo::CH + s) * Geo::NV * 64;
#pragma unroll
                for (int r = 0; r < P; r++) b[r * 64] = cr[r];
#pragma unroll
                for (int i = 0; i < P / 2; i++) b[(P + i) * 64] = sr[2 * i];
            }
        }
        __syncthreads();                                      // barrier c
    }
}
#endif

// Reset + n - 1 Updates -> log-likelihood sum (no prior)
// WRITE_MV: also store mean_k = h.z + mu and var_k of every datum (KalmanFilter::GetMean / GetVar,
// src/include/kfilter.hpp:116-117) to mv[k * mv_stride] and mv[(n + k) * mv_stride]
// REPDT: the series has time steps that repeat their predecessor (regular cadence): such a step re-uses the factors (a
// wave-uniform test).  Without (the launcher's choice for irregular series) a step is ONE basic block -- the factors' LDS
// reads or exp / sincos chains interleave with the recursion: 122 -> 113 us with producer waves, 2 % for the lone wave
// (profiles/r04/ab_lane_noskip_v1.txt).
template <int P, class Src, bool WRITE_MV = false, bool REPDT = true>
CARMA_DEV double lane_filter(const LaneModel<P>& m, const double4* __restrict__ series, int n, const Src& src, double* mv = nullptr,
                             long mv_stride = 0)
{
    constexpr int NT = P * (P + 1) / 2;
    constexpr int PE = P & ~1;                               // coordinates that belong to pairs
    double D[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) D[i] = 0.0;
    double z[P], k[P], w[P];
#pragma unroll
    for (int r = 0; r < P; r++) {
        z[r] = 0.0;
        w[r] = 0.0;
        k[r] = m.c[r];
    }
    double cr[P], sr[P];                                     // transition factors per coordinate
#pragma unroll
    for (int r = 0; r < P; r++) {
        cr[r] = 1.0;
        sr[r] = 0.0;
    }
    LogLikAcc acc;
    acc.init();
    double4 rprev = series[0];
    double dt_prev = -1.0;                                   // (a time step is never negative)
    int kk0 = 1;
    if constexpr (Src::UNROLL > 1 && !REPDT && !WRITE_MV) {
        // Whole ring buffers, U steps as one basic block.  (The statements of the loop below, once more: routing BOTH loops through
        // one body cost the in-line kernel 5 % at 65 536 evaluations -- profiles/r05/ab_lpc_unroll_v2.txt -- so that loop stays as it is.)
        constexpr int U = Src::UNROLL;
        for (; kk0 + U <= n; kk0 += U) {                      // (kk0 - 1 is a multiple of U)
            const double* cb = src.chunk(kk0);
#pragma unroll
            for (int us = 0; us < U; us++) {
                const int kk = kk0 + us;
                const double4 rec = series[kk];
                Src::get_at(cb, us, cr, sr);
            // --- var_{kk-1} = s0 + h D h^T + e, mean_{kk-1} = h.z   (kfilter.cpp:180-184, 207-213)
            double pv = 0.0, pm = 0.0;
#pragma unroll
            for (int r = 0; r < P; r++) {
                pv = fma(m.h[r], w[r], pv);
                pm = fma(m.h[r], z[r], pm);
            }
            const double var = m.s0 + pv + rprev.z * m.scale;
            const double innov = (rprev.y - m.mu) - pm;
            if constexpr (WRITE_MV) {
                mv[(long)(kk - 1) * mv_stride] = pm + m.mu;
                mv[(long)(n + kk - 1) * mv_stride] = var;
            }
            acc.add_var(var);
            const double s = recip(var);
            const double si = s * innov;
            acc.chi2 += innov * si;
            // --- state (kfilter.cpp:191-194, 200-201)
            double zu[P];
#pragma unroll
            for (int r = 0; r < P; r++) zu[r] = fma(k[r], si, z[r]);
#pragma unroll
            for (int r = 0; r < P; r++) z[r] = (r < PE) ? cr[r] * zu[r] - sr[r] * zu[r ^ 1] : cr[r] * zu[r];
            // --- covariance (kfilter.cpp:197, 204): d = D - k k^T / var (upper triangle), mm = d Phi^T, D = Phi mm
            double d[NT];
#pragma unroll
            for (int i = 0; i < P; i++) {
                const double t = k[i] * s;
#pragma unroll
                for (int j = i; j < P; j++) d[tri<P>(i, j)] = fma(-t, k[j], D[tri<P>(i, j)]);
            }
            // mm_ij for j >= i, and below the diagonal the one entry a pair's even row needs from its partner: (i + 1, i)
            double mm[P][P];
#pragma unroll
            for (int i = 0; i < P; i++) {
#pragma unroll
                for (int j = 0; j < P; j++) {
                    const bool need = (j >= i) || (j == i - 1 && (i & 1) && i < PE);
                    if (need) {
                        if (j < PE)
                            mm[i][j] = d[tri<P>(i, j)] * cr[j] - d[tri<P>(i, j ^ 1)] * sr[j];
                        else
                            mm[i][j] = d[tri<P>(i, j)] * cr[j];
                    } else {
                        mm[i][j] = 0.0;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < P; i++) {
#pragma unroll
                for (int j = i; j < P; j++) {
                    if (i < PE)
                        D[tri<P>(i, j)] = cr[i] * mm[i][j] - sr[i] * mm[i ^ 1][j];
                    else
                        D[tri<P>(i, j)] = cr[i] * mm[i][j];
                }
            }
            // --- w = D h, gain of the next step k = w + c   (kfilter.cpp:191 of the next Update)
#pragma unroll
            for (int i = 0; i < P; i++) {
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < P; j++) a = fma(D[tri<P>(i, j)], m.h[j], a);
                w[i] = a;
                k[i] = a + m.c[i];
            }
                rprev = rec;
            }
        }
    }
    for (int kk = kk0; kk < n; kk++) {
        const double4 rec = series[kk];
        // --- transition factors of this step; a repeated time step (wave-uniform: the series is shared) re-uses them
        src.step(kk);
        if constexpr (REPDT) {
            if (rec.x != dt_prev) {
                dt_prev = rec.x;
                src.get(kk, rec.x, cr, sr);
            }
        } else {
            src.get(kk, rec.x, cr, sr);
        }
        // --- var_{kk-1} = s0 + h D h^T + e, mean_{kk-1} = h.z   (kfilter.cpp:180-184, 207-213)
        double pv = 0.0, pm = 0.0;
#pragma unroll
        for (int r = 0; r < P; r++) {
            pv = fma(m.h[r], w[r], pv);
            pm = fma(m.h[r], z[r], pm);
        }
        const double var = m.s0 + pv + rprev.z * m.scale;
        const double innov = (rprev.y - m.mu) - pm;
        if constexpr (WRITE_MV) {
            mv[(long)(kk - 1) * mv_stride] = pm + m.mu;
            mv[(long)(n + kk - 1) * mv_stride] = var;
        }
        acc.add_var(var);
        const double s = recip(var);
        const double si = s * innov;
        acc.chi2 += innov * si;
        // --- state (kfilter.cpp:191-194, 200-201)
        double zu[P];
#pragma unroll
        for (int r = 0; r < P; r++) zu[r] = fma(k[r], si, z[r]);
#pragma unroll
        for (int r = 0; r < P; r++) z[r] = (r < PE) ? cr[r] * zu[r] - sr[r] * zu[r ^ 1] : cr[r] * zu[r];
        // --- covariance (kfilter.cpp:197, 204): d = D - k k^T / var (upper triangle), mm = d Phi^T, D = Phi mm
        double d[NT];
#pragma unroll
        for (int i = 0; i < P; i++) {
            const double t = k[i] * s;
#pragma unroll
            for (int j = i; j < P; j++) d[tri<P>(i, j)] = fma(-t, k[j], D[tri<P>(i, j)]);
        }
        // mm_ij for j >= i, and below the diagonal the one entry a pair's even row needs from its partner: (i + 1, i)
        double mm[P][P];
#pragma unroll
        for (int i = 0; i < P; i++) {
#pragma unroll
            for (int j = 0; j < P; j++) {
                const bool need = (j >= i) || (j == i - 1 && (i & 1) && i < PE);
                if (need) {
                    if (j < PE)
                        mm[i][j] = d[tri<P>(i, j)] * cr[j] - d[tri<P>(i, j ^ 1)] * sr[j];
                    else
                        mm[i][j] = d[tri<P>(i, j)] * cr[j];
                } else {
                    mm[i][j] = 0.0;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < P; i++) {
#pragma unroll
            for (int j = i; j < P; j++) {
                if (i < PE)
                    D[tri<P>(i, j)] = cr[i] * mm[i][j] - sr[i] * mm[i ^ 1][j];
                else
                    D[tri<P>(i, j)] = cr[i] * mm[i][j];
            }
        }
        // --- w = D h, gain of the next step k = w + c   (kfilter.cpp:191 of the next Update)
#pragma unroll
        for (int i = 0; i < P; i++) {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < P; j++) a = fma(D[tri<P>(i, j)], m.h[j], a);
            w[i] = a;
            k[i] = a + m.c[i];
        }
        rprev = rec;
    }
    {   // last point: var_{n-1}, mean_{n-1}
        double pv = 0.0, pm = 0.0;
#pragma unroll
        for (int r = 0; r < P; r++) {
            pv = fma(m.h[r], w[r], pv);
            pm = fma(m.h[r], z[r], pm);
        }
        const double var = m.s0 + pv + rprev.z * m.scale;
        const double innov = (rprev.y - m.mu) - pm;
        if constexpr (WRITE_MV) {
            mv[(long)(n - 1) * mv_stride] = pm + m.mu;
            mv[(long)(2 * n - 1) * mv_stride] = var;
        }
        acc.add_var(var);
        acc.chi2 += innov * (recip(var) * innov);
    }
    return acc.total();
}

// KalmanFilterp(time, y, yerr, sigsqr, omega, ma_coefs) as one lane: the model from its roots (adjacent conjugate pairs first,
// then the real roots: carma_normalize_roots) and MA coefficients (lowest order first, zero padded to P) instead of theta --
// b_r = beta(omega_r) and beta(-omega_r) by Horner, kappa_r = beta(-omega_r) / (alpha'(omega_r) alpha(-omega_r)) (struct Model),
// s0 = sigsqr * sum_r Re(b_r kappa_r), c_r = sigsqr kappa_r, all in the real modal coordinates of lane_model_from_theta.
template <int P>
CARMA_DEV void lane_model_from_roots(const double* om_re_im, const double* ma, double sigsqr, double mu, LaneModel<P>& m)
{
    Cx w[P], b[P], kap[P];
#pragma unroll
    for (int j = 0; j < P; j++) w[j] = Cx{om_re_im[2 * j], om_re_im[2 * j + 1]};
    bool sing = false;
    double var1 = 0.0;
#pragma unroll
    for (int r = 0; r < P; r++) {
        Cx br = {0.0, 0.0}, bm = {0.0, 0.0};
        const Cx nw = {-w[r].re, -w[r].im};
#pragma unroll
        for (int i = P - 1; i >= 0; i--) {
            br = cadd(cmul(br, w[r]), Cx{ma[i], 0.0});
            bm = cadd(cmul(bm, nw), Cx{ma[i], 0.0});
        }
        b[r] = br;
        Cx ap = {1.0, 0.0}, am = {1.0, 0.0};
#pragma unroll
        for (int l = 0; l < P; l++) {
            const Cx dl = csub(w[r], w[l]);
            const Cx sl = {-(w[r].re + w[l].re), -(w[r].im + w[l].im)};
            if (l != r) ap = cmul(ap, dl);
            am = cmul(am, sl);
        }
        if (ap.re == 0.0 && ap.im == 0.0) sing = true;
        kap[r] = cdiv(bm, cmul(ap, am));
        var1 += b[r].re * kap[r].re - b[r].im * kap[r].im;
    }
    m.s0 = sigsqr * var1;
    m.scale = 1.0;
    m.mu = mu;
    m.sing = sing;
    m.valid = true;
#pragma unroll
    for (int r = 0; r < P; r++) {
        m.wre[r] = w[r].re;
        m.wim[r] = w[r].im;
        const bool cpx = (w[r].im != 0.0) && (r < (P & ~1));
        const int ev = r & ~1;
        if (cpx) {
            m.h[r] = (r & 1) ? 2.0 * b[r].im : 2.0 * b[r].re;
            m.c[r] = sigsqr * ((r & 1) ? kap[ev].im : kap[r].re);
        } else {
            m.h[r] = b[r].re;
            m.c[r] = sigsqr * kap[r].re;
        }
    }
#pragma unroll
    for (int i = 0; i < (P + 1) / 2; i++) m.realpair[i] = (2 * i + 1 < P) && (w[2 * i].im == 0.0);
}

// Filter() of one model per lane: mean[n], var[n] into mv (see lane_filter); returns the repeated-root flag
template <int P>
CARMA_DEV bool kfilter_lane(const double* om_re_im, const double* ma, double sigsqr, double mu, const double4* __restrict__ series,
                            int n, const double* tab, double* mv, long mv_stride)
{
    LaneModel<P> m;
    lane_model_from_roots<P>(om_re_im, ma, sigsqr, mu, m);
    bool anyreal = false;
#pragma unroll
    for (int i = 0; i < P / 2; i++) anyreal = anyreal || m.realpair[i];
    const LaneFactorsInline<P> src{m, lane_any(anyreal), tab};
    (void)lane_filter<P, LaneFactorsInline<P>, true>(m, series, n, src, mv, mv_stride);
    return m.sing;
}

// CARMA_Base::LogDensity (carpack.hpp:131-176): -inf outside the prior bounds or on a repeated root, else
// log-likelihood + log prior
template <int P, bool REPDT = true>
CARMA_DEV double logdensity_lane(const double* theta, int q, const double4* __restrict__ series, int n, const Prior& pr,
                                 int ignore_prior, const double* tab)
{
    LaneModel<P> m;
    lane_model_from_theta<P>(theta, q, pr, ignore_prior, m);
    bool anyreal = false;
#pragma unroll
    for (int i = 0; i < P / 2; i++) anyreal = anyreal || m.realpair[i];
    const LaneFactorsInline<P> src{m, lane_any(anyreal), tab};
    double ll = lane_filter<P, LaneFactorsInline<P>, false, REPDT>(m, series, n, src);
    ll += log_prior(m.scale, pr.measerr_dof);
    if (m.sing || !m.valid) ll = -1.0 / 0.0;
    return ll;
}

#if defined(__HIPCC__)
// the same with the factors from the producer wave's ring (consumer side of k_logdens_carma_lpc)
template <int P, int NP, bool REPDT = true, bool UNROLLED = false>
__device__ __forceinline__ double logdensity_lane_ring(const double* theta, int q, const double4* __restrict__ series, int n,
                                                       const Prior& pr, int ignore_prior, const double* ring /* + lane */)
{
    LaneModel<P> m;
    lane_model_from_theta<P>(theta, q, pr, ignore_prior, m);
    const LaneFactorsRing<P, NP, UNROLLED && !REPDT> src{ring};
    double ll = lane_filter<P, LaneFactorsRing<P, NP, UNROLLED && !REPDT>, false, REPDT>(m, series, n, src);
    ll += log_prior(m.scale, pr.measerr_dof);
    if (m.sing || !m.valid) ll = -1.0 / 0.0;
    return ll;
}
#endif

}  // namespace carma
