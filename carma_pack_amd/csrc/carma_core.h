// carma_core.h -- the batched CARMA(p,q) / CAR(1) Kalman log-density, written once for a
// "lane group" of G lanes per evaluation.  Included by carma_kernels.hip (gfx950, grp_device.h)
// and by tests/emu/emu_core.cpp (CPU lane emulator, test harness only).
//
// What it computes (reference file:line under /root/reference):
//   * theta -> AR roots            src/carpack.cpp:137-172   (CARp::ARRoots)
//   * theta -> MA coefficients     src/carpack.cpp:522-580, :742-756 (CARMA::ExtractMA, polycoefs)
//   * sigma^2 = theta0^2/Variance  src/carpack.cpp:377-409, src/include/carpack.hpp:316-319,391-395
//   * prior bounds / log prior     src/carpack.cpp:314-374, :709-732, src/include/carpack.hpp:118-126
//   * Kalman Reset                 src/kfilter.cpp:138-186
//   * Kalman Update (n-1 times)    src/kfilter.cpp:189-215
//   * log-likelihood sum           src/include/carpack.hpp:167-171
//   * CAR(1) Reset/Update          src/kfilter.cpp:19-48
//
// Layout: lane r (< P) of a group owns ROW r of the p x p matrix D = P - V (prediction covariance
// minus stationary covariance, in REAL modal coordinates: see filter_loop_real), its own root omega_r,
// rotated MA coefficient b_r, state z_r and c_r = (V b^H)_r.  Algebraically identical to the reference
// recursion but restructured so that the stationary matrix V never has to be kept:
//     u      = P b^H            = D b^H + c
//     var_k  = Re(b P b^H)+e_k  = s0 + Re(b D b^H) + e_k,      s0 = Re(b V b^H)
//     D     <- (rho rho^H) o (D - u u^H / var)                 (kfilter.cpp:197,204)
//     x     <- rho o (x + u innov / var)                        (kfilter.cpp:194,201)
// sum log(var) is accumulated as a mantissa product + integer exponent sum, so the loop has no
// log(); 1/var is the only division.  The per-evaluation constants (b, c, s0, sigma^2) come from
// closed forms instead of the reference's Vandermonde solve: see struct Model.
#pragma once
#include "carma_types.h"
#include "carma_math.h"

namespace carma {

// series record k: {dt_k = t_k - t_{k-1} (dt_0 = 0), y_k, yerr_k^2, t_k}
// (double4 so that one 32-byte scalar load fetches a step)

constexpr double TWO_PI = 6.283185307179586476925286766559;
constexpr double LN2 = 0.693147180559945309417232121458;

CARMA_DEV Cx cmul(Cx a, Cx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
CARMA_DEV Cx cmulc(Cx a, Cx b) { return {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; }  // a*conj(b)
CARMA_DEV Cx cadd(Cx a, Cx b) { return {a.re + b.re, a.im + b.im}; }
CARMA_DEV Cx csub(Cx a, Cx b) { return {a.re - b.re, a.im - b.im}; }
CARMA_DEV Cx cscale(Cx a, double s) { return {a.re * s, a.im * s}; }
// 1/v to ~1 ulp: v_rcp_f64 + two Newton steps on the GPU (the IEEE division expands to 12
// instructions), plain division on the host.
CARMA_DEV double recip(double v)
{
#ifdef __HIPCC__
    // v_rcp_f64 is good to 2^-24.4 (measured); one cubic step s0 (1 + e + e^2), e = 1 - v s0, leaves
    // e^3 ~ 1e-22 plus one rounding: 0.5 ulp on 4M random inputs, one instruction less than two
    // Newton steps and a shorter dependent chain
    const double s0 = __builtin_amdgcn_rcp(v);
    const double e = fma(-v, s0, 1.0);
    return fma(s0, fma(e, e, e), s0);
#else
    return 1.0 / v;
#endif
}
CARMA_DEV Cx cdiv(Cx a, Cx b)
{
#ifdef CARMA_EXACT_CDIV
    const double den = b.re * b.re + b.im * b.im;
    return {(a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den};
#else
    const double s = recip(b.re * b.re + b.im * b.im);       // one reciprocal instead of two IEEE divisions
    return {(a.re * b.re + a.im * b.im) * s, (a.im * b.re - a.re * b.im) * s};
#endif
}
CARMA_DEV Cx csel(bool m, Cx a, Cx b) { return {m ? a.re : b.re, m ? a.im : b.im}; }

// ---------------------------------------------------------------------------------------------
// One quadratic factor -> its two roots (carpack.cpp:143-163); which = 0 / 1 picks the member.
CARMA_DEV Cx quad_root(double lq1, double lq2, int which)
{
    double q1 = exp(lq1), q2 = exp(lq2);
    // The discriminant as the reference rounds it: the product q2 q2 rounded BEFORE the subtraction, no fused multiply-add.
    // It matters in one place: where 4 q1 / q2^2 lies between 2^-53 and 2^-52 the reference's q2 - sqrt(disc) is exactly
    // zero or one ulp depending on that rounding -- and with a zero root its MA polynomial, and the log-density, is NaN (kept,
    // see below).  A fused discriminant put the sampler's stored log-posterior on the other side of that coin for about one
    // state in 1500 of the configs[2] run.  (What is left is the last bit of exp(): the two libraries agree on most arguments.)
    double q22 = q2 * q2;
#ifdef __HIPCC__
    asm volatile("" : "+v"(q22));                    // (keeps the compiler from contracting the next line into an FMA)
#endif
    double disc = q22 - 4.0 * q1;
    const double sq = sqrt(fabs(disc));              // one sqrt for both signs of the discriminant
    Cx r;
    if (disc > 0) {
        // two real roots: the larger one from the sum that does not cancel, the smaller one from the product q1 (the
        // reference's -(q2 - sq)/2 loses every digit once q2^2 >> 4 q1 -- MA parameters of the sampler reach q2 ~ 1e15
        // with the small root of order one; in quad precision the reference's form agrees with this one)
        // Where the reference's difference is EXACTLY zero (q2^2 > 2^53 * 4 q1) its MA polynomial divides by that root and
        // the log-density is NaN: kept, so that the sampler moves in the same domain as the reference's.
        const double big = -0.5 * (q2 + sq);
        r.re = which ? (q2 - sq == 0.0 ? 0.0 : q1 * recip(big)) : big;
        r.im = 0.0;
    } else {
        r.re = -0.5 * q2;
        double im = -0.5 * sq;
        r.im = which ? -im : im;
    }
    return r;
}
// compile-time loop: f(IntC<J>{}) for J = J0 .. N-1 (DPP lane selectors are instruction immediates)
template <int J>
struct IntC {
    static constexpr int value = J;
};
template <int J, int N, class F>
CARMA_DEV void static_for(F&& f)
{
    if constexpr (J < N) {
        f(IntC<J>{});
        static_for<J + 1, N>(f);
    }
}

// Root number i of a polynomial given by its m log quadratic-factor coefficients lq[0..m): the AR polynomial
// (carpack.cpp:137-172) and the MA polynomial (carpack.cpp:522-552) share this parameterisation.
CARMA_DEV Cx poly_root(const double* lq, int m, int i)
{
    if ((m & 1) && i == m - 1) return Cx{-exp(lq[m - 1]), 0.0};
    const int pair = i >> 1;
    return quad_root(lq[2 * pair], lq[2 * pair + 1], i & 1);
}

// Model quantities of one evaluation, as held by lane r of its group.
//
// Everything the recursion needs beyond the roots follows from three closed forms (alpha = monic AR polynomial with
// roots omega_k, beta = MA polynomial, real coefficients, beta(0) = 1):
//     b_r     = beta(omega_r)                                             rotated MA coefficient   kfilter.cpp:162
//     kappa_r = beta(-omega_r) / (alpha'(omega_r) alpha(-omega_r))        (V b^H)_r = sigma^2 kappa_r
//     Variance(omega, beta, sigma = 1) = sum_r b_r kappa_r                carpack.cpp:377-409
// The middle one is the partial-fraction identity  sum_j beta(omega_j) / (alpha'(omega_j) (s + omega_j)) = -beta(-s) /
// alpha(-s)  applied to  (V b^H)_r = -sigma^2 J_r sum_j conj(J_j b_j) / (omega_r + conj(omega_j)),  J_r = 1 /
// alpha'(omega_r)  (kfilter.cpp:144-172; the root set is closed under conjugation).  The reference forms these
// quantities through an LU solve of the Vandermonde system and p-term sums that cancel catastrophically when roots
// cluster (the prior admits roots 1e-4 apart, carpack.cpp:330); the products above have no cancellation beyond the
// root differences themselves.  Against 50-digit arithmetic on 27 000 prior-like parameter vectors this set-up is never
// further from the exact value than the reference's own arithmetic and 2-7 orders of magnitude closer on the
// ill-conditioned ones (tests/tools/proto/setup_v2.py).
template <int P>
struct Model {
    Cx w;            // omega_r (own AR root; idle lanes of the group hold a copy of the last root)
    Cx wall[P];      // all roots (replicated)
    Cx b;            // b_r
    Cx kap;          // kappa_r
    double sigsqr;   // driving-noise variance
    double s0;       // Re(b V b^H): the stationary variance of the process
    double mu, scale;
    bool valid;      // prior bounds satisfied (or ignored)
    bool sing;       // repeated AR root: singular eigenvector matrix (arma::solve throws, carpack.hpp:154-164)
};

// Root omega_rr of the AR polynomial encoded in theta (carpack.cpp:137-172), rr < P.
template <int P>
CARMA_DEV Cx own_ar_root(const double* theta, int rr)
{
    return poly_root(theta + 3, P, rr);
}

// kappa_r and the repeated-root flag from the roots and beta(-omega_r)
template <int P, class GrpT>
CARMA_DEV void model_kappa(const GrpT& g, Model<P>& m, const Cx bm)
{
    const int r = g.lane();
    const int rr = r < P ? r : P - 1;
    Cx ap = {1.0, 0.0}, am = {1.0, 0.0};      // alpha'(omega_r) = prod_{l != r} (omega_r - omega_l),  alpha(-omega_r)
#pragma unroll
    for (int l = 0; l < P; l++) {
        const Cx dl = csub(m.w, m.wall[l]);
        const Cx sl = {-(m.w.re + m.wall[l].re), -(m.w.im + m.wall[l].im)};
        ap = (l != rr) ? cmul(ap, dl) : ap;
        am = cmul(am, sl);
    }
    const bool zero = (ap.re == 0.0 && ap.im == 0.0);
    m.sing = g.sum((r < P && zero) ? 1.0 : 0.0) != 0.0;
    m.kap = cdiv(bm, cmul(ap, am));
}

// theta -> Model  (ARRoots, ExtractMA, ExtractSigsqr, CheckPriorBounds)
// PART selects what is evaluated -- the wave pipeline splits the set-up between its two recursion waves:
//   MODEL_ALL     everything
//   MODEL_CONSTS  roots, b, kappa, sigma^2, s0 (what the covariance wave needs); valid = true, sing as computed
//   MODEL_FLAGS   roots, mu, scale and the prior bounds (what the mean wave needs); b, kappa, sigma^2, s0 and the
//                 repeated-root flag are NOT set (the flag reaches the mean wave through the pipeline, carma_pipe3l.h)
constexpr int MODEL_ALL = 0, MODEL_CONSTS = 1, MODEL_FLAGS = 2;
template <int P, int G, int PART = MODEL_ALL, class GrpT>
CARMA_DEV void model_from_theta(const GrpT& g, const double* theta, int q, const Prior& pr, int ignore_prior,
                                Model<P>& m)
{
    const int r = g.lane();
    const int rr = r < P ? r : P - 1;
    // --- roots.  Slot s < P is AR root s (carpack.cpp:137-172), slot P + k is MA root k (carpack.cpp:522-552); lane l
    // evaluates slot l, and slot l + G in a second pass when the group is too small for all of them
    const bool ma0 = (r >= P) && (r - P < q);
    // (selections between Cx values go through csel: a ?: on the structs makes the compiler select between their
    // ADDRESSES and keeps the whole Model in scratch memory -- 5.8 MB of scratch writes per 1024-evaluation launch)
    const Cx root0 = poly_root(ma0 ? theta + 3 + P : theta + 3, ma0 ? q : P, ma0 ? r - P : rr);
#pragma unroll
    for (int j = 0; j < P; j++) {
        m.wall[j].re = g.bcast_u(root0.re, j);
        m.wall[j].im = g.bcast_u(root0.im, j);
    }
    m.w = csel(r < P, root0, m.wall[P - 1]);
    m.scale = theta[1];
    m.mu = theta[2];
    if constexpr (PART != MODEL_FLAGS) {
    constexpr int NMA = P > 1 ? P - 1 : 1;       // q <= P - 1
    constexpr int MA0 = G - P;                   // MA roots held by the first pass
    Cx root1 = {-1.0, 0.0};
    if constexpr (NMA > MA0) {
        // (only when the second pass has a slot to fill: with q <= MA0 there is none, and with q = 0 there is no MA parameter to
        // read at all -- the unconditional form read theta[d], theta[d + 1], i.e. past the END of the batch for its last
        // evaluation: a memory fault when the array ends on a page boundary, tools/fuzz_dispatch.py seed 21: CARMA(5,0),
        // 16 384 evaluations = exactly 1 MiB.  q is launch-uniform.)
        if (q > MA0) {
            const bool ma1 = MA0 + r < q;
            root1 = poly_root(theta + 3 + P, q, ma1 ? MA0 + r : 0);
        }
    }
    // --- b_r = beta(omega_r) = prod_k (mu_k - omega_r) / mu_k,  beta(-omega_r) = prod_k (mu_k + omega_r) / mu_k;
    //     prod_k mu_k is real (conjugate pairs and real roots)
    Cx pb = {1.0, 0.0}, pm = {1.0, 0.0}, pmu = {1.0, 0.0};
    const Cx wr = m.w;       // (a copy: the lambda must not capture the Model, or the whole struct is kept in scratch memory)
    static_for<0, NMA>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        Cx mu;
        if constexpr (k < MA0) {
            mu = {g.bcast_u(root0.re, P + k), g.bcast_u(root0.im, P + k)};
        } else {
            mu = {g.bcast_u(root1.re, k - MA0), g.bcast_u(root1.im, k - MA0)};
        }
        if (k < q) {
            pb = cmul(pb, csub(mu, wr));
            pm = cmul(pm, cadd(mu, wr));
            pmu = cmul(pmu, mu);
        }
    });
    const double rmu = 1.0 / pmu.re;
    m.b = cscale(pb, rmu);
    model_kappa<P>(g, m, cscale(pm, rmu));
    // --- sigma^2 = theta0^2 / Variance(omega, beta, 1)   (carpack.cpp:377-409, carpack.hpp:316-319, 391-395)
    const double var1 = g.sum(r < P ? (m.b.re * m.kap.re - m.b.im * m.kap.im) : 0.0);
    m.sigsqr = theta[0] * theta[0] / var1;
    m.s0 = theta[0] * theta[0];                  // = sigma^2 Variance(1): var_0 = sigma_y^2 + yerr_0^2 (kfilter.cpp:180-182)
    } else {
        m.sing = false;      // the repeated-root flag is the covariance wave's (model_kappa): handed over by the pipeline
    }
    // --- prior bounds (carpack.cpp:314-374, unique_roots :709-732)
    m.valid = true;
    if (PART != MODEL_CONSTS && !ignore_prior) {
        double cent = fabs(m.w.im) / 2.0 / (TWO_PI / 2.0);
        double width = -m.w.re / 2.0 / (TWO_PI / 2.0);
        bool viol = !(cent < pr.max_freq) || !(width < pr.max_freq) || !(width > pr.min_freq);
        double cent_prev = 0.0;
#pragma unroll
        for (int j = 0; j < P; j++) cent_prev = (j == rr - 1) ? fabs(m.wall[j].im) / 2.0 / (TWO_PI / 2.0) : cent_prev;
        if (rr >= 1 && (cent - cent_prev) > 1e-8) viol = true;
#pragma unroll
        for (int j = 1; j < P; j++) {
            if (j > rr) {
                // |(w - w_j) / (w + w_j)| <= 1e-4 (carpack.cpp:709-732), compared as squared moduli
                const Cx dn = csub(m.w, m.wall[j]), sm = cadd(m.w, m.wall[j]);
                const double n2 = dn.re * dn.re + dn.im * dn.im, d2 = sm.re * sm.re + sm.im * sm.im;
                if (n2 <= 1e-8 * d2) viol = true;
            }
        }
        double nviol = g.sum((r < P && viol) ? 1.0 : 0.0);
        double ysigma = theta[0], ms = theta[1];
        if (nviol != 0.0 || (ysigma > pr.max_stdev) || (ysigma < 0) || (ms < 0.5) || (ms > 2.0)) m.valid = false;
    }
}

// (sigma^2, roots, MA coefficients) -> Model: the KalmanFilterp constructor (kfilter.hpp:303-334).  om_re_im = p
// (re, im) pairs in the order ARRoots emits (conjugate pairs adjacent, the C ABI checks it), ma = p coefficients
// (zero padded, kfilter.hpp:318-320).
template <int P, int G, class GrpT>
CARMA_DEV void model_from_roots(const GrpT& g, const double* om_re_im, const double* ma, double sigsqr, Model<P>& m)
{
    const int r = g.lane();
    const int rr = r < P ? r : P - 1;
    m.w = {om_re_im[2 * rr], om_re_im[2 * rr + 1]};
#pragma unroll
    for (int j = 0; j < P; j++) m.wall[j] = {om_re_im[2 * j], om_re_im[2 * j + 1]};
    Cx b = {0.0, 0.0}, bm = {0.0, 0.0};
    const Cx nw = {-m.w.re, -m.w.im};
#pragma unroll
    for (int i = P - 1; i >= 0; i--) {           // Horner: beta(omega_r), beta(-omega_r)
        b = cadd(cmul(b, m.w), Cx{ma[i], 0.0});
        bm = cadd(cmul(bm, nw), Cx{ma[i], 0.0});
    }
    m.b = b;
    model_kappa<P>(g, m, bm);
    m.sigsqr = sigsqr;
    m.s0 = sigsqr * g.sum(r < P ? (m.b.re * m.kap.re - m.b.im * m.kap.im) : 0.0);
    m.mu = 0.0;
    m.scale = 1.0;
    m.valid = true;
}

// Running -0.5*sum(log var) - 0.5*sum(innov^2/var) without a log in the loop.
struct LogLikAcc {
    double prod;   // running product of the variances, renormalised to [0.5,1) every step
    int esum;      // binary exponent taken out of prod so far
    double chi2;
    double vmin;   // smallest var so far: var <= 0 -> NaN total (reference: log(var) = NaN); a NaN var
                   // slips through fmin but turns prod into NaN by itself
    CARMA_DEV void init()
    {
        prod = 0.5;
        esum = 1;
        chi2 = 0.0;
        vmin = 1.0;
    }
    CARMA_DEV void add_var(double var)
    {
#ifdef __HIPCC__
        asm("v_min_f64 %0, %0, %1" : "+v"(vmin) : "v"(var));     // no canonicalising v_max in front
#else
        vmin = fmin(vmin, var);
#endif
        int e;
        prod = frexp(prod * var, &e);      // v_frexp_mant_f64 + v_frexp_exp_i32_f64
        esum += e;
    }
    CARMA_DEV double total() const
    {
        double nan_ = 0.0;
        if (!(vmin > 0.0)) nan_ = (vmin - vmin) / (vmin - vmin);
        return -0.5 * (log(prod) + (double)esum * LN2) - 0.5 * chi2 + nan_;
    }
};

// Per-evaluation constants of the recursion (the part of Reset, kfilter.cpp:138-186, that the loop keeps), as held by
// lane r of the group.
template <int P>
struct FilterConsts {
    Cx b_own;     // rotated MA coefficient b_r
    Cx b_msk;     // b_r, or 0 in the idle lanes of the group
    Cx c_own;     // (V b^H)_r
    Cx ball[P];   // b_j for all j
    double s0;    // Re(b V b^H)
    bool sing;    // singular Vandermonde system (two equal roots)
};

template <int P, int G, class GrpT>
CARMA_DEV void filter_reset(const GrpT& g, const Model<P>& m, FilterConsts<P>& fc)
{
    const bool act = g.lane() < P;
    fc.b_own = m.b;
    fc.b_msk = {act ? m.b.re : 0.0, act ? m.b.im : 0.0};
    fc.c_own = cscale(m.kap, m.sigsqr);
#pragma unroll
    for (int j = 0; j < P; j++) fc.ball[j] = {g.bcast_u(m.b.re, j), g.bcast_u(m.b.im, j)};
    fc.s0 = m.s0;
    fc.sing = m.sing;
}


// Where the transition factors rho_j(k) = exp(omega_j dt_k) of a group come from.
// RhoInline: every lane computes its own factor one step ahead and shares it through the group's
// second exchange array.  (carma_ring.h has the variant fed by a producer wave.)
// DTC ("dt cache"): a step whose time step equals the previous one re-uses its factor -- a regularly sampled stretch of the
// series (dt is wave-uniform) then skips the exp/sincos evaluation, a third of the step's instructions: 65 536
// evaluations of a constant-cadence series 538 -> 426 us.  On an irregular series the test itself costs 4-8 %, so the
// host picks the variant per context (Ctx::repeated_dt).
template <int P, class GrpT, bool DTC = false>
struct RhoInline {
    static constexpr bool kRing = false;
    static constexpr bool kPaired = false;
    static constexpr int kChunk = 1 << 30;
    CARMA_DEV void chunk_begin(int) const {}
    CARMA_DEV double4 record_s(int) const { return double4{}; }
    CARMA_DEV void fetch_s(int, Cx&, Cx (&)[P]) const {}
    CARMA_DEV double4 record(int) const { return double4{}; }
    const GrpT& g;
    Cx w;            // own root
    Cx rho_next;     // factor of the upcoming step
    double dt_last = -1.0;   // DTC: the time step rho_next belongs to
    CARMA_DEV void begin(int, double dt_first)
    {
        cexp_step(w.re, w.im, dt_first, &rho_next.re, &rho_next.im);
        dt_last = dt_first;
    }
    CARMA_DEV void publish(int) const { g.publish2(rho_next.re, rho_next.im); }
    CARMA_DEV void fetch(int, Cx& rho, Cx (&rj)[P]) const
    {
        rho = rho_next;
#pragma unroll
        for (int j = 0; j < P; j++) rj[j] = g.peek2(j);
    }
    // called once the group's LDS reads are in flight: independent work that hides their latency
    CARMA_DEV void prepare(int, double dt_next)
    {
        if (!DTC || dt_next != dt_last) {
            cexp_step(w.re, w.im, dt_next, &rho_next.re, &rho_next.im);
            if (DTC) dt_last = dt_next;
        }
    }
};

// RhoPair: RhoInline for the G <= 8 loop (filter_loop_real) with the exp/sincos work SHARED inside a root pair.  The two
// lanes of a complex-conjugate pair would evaluate the same exponential and the same sine / cosine every step (their
// factors are conjugates), and the partner lane of an odd p's last, single real root is idle.  So lanes (2i, 2i+1)
// take turns: the loop runs two steps per trip; before a trip the even lane evaluates the pair's factor for its first
// step (phase 0), the odd lane the one for its second step (phase 1), both publish once, and in either phase every
// lane reads a pair's factor from the slot of the phase's owner -- conjugated for the other member, which is a
// source modifier because the phase is a compile-time constant.  Half the exp/sincos evaluations, ceil(p/2) + 1 LDS
// reads per step instead of p.  A quadratic factor with two REAL roots (positive discriminant) puts two different
// roots into a pair: filter_run uses this source only when no group of the wave holds one.
template <int P, class GrpT, bool DTC = false>
struct RhoPair {
    static constexpr bool kRing = false;
    static constexpr bool kPaired = true;
    static constexpr int kChunk = 1 << 30;
    CARMA_DEV void chunk_begin(int) const {}
    CARMA_DEV double4 record_s(int) const { return double4{}; }
    CARMA_DEV void fetch_s(int, Cx&, Cx (&)[P]) const {}
    const GrpT& g;
    Cx w;                  // own root (idle lanes: the last root, see model_from_theta)
    Cx val;                // the factor this lane evaluated last
    double dt1_last = -1.0, dt2_last = -1.0;   // the time steps `val` belongs to (RhoInline::dt_last)
    // dt1 / dt2: time steps of the coming trip's first / second step
    CARMA_DEV void begin2(double dt1, double dt2)
    {
        if (!DTC || dt1 != dt1_last || dt2 != dt2_last) {
            cexp_step(w.re, w.im, (g.lane() & 1) ? dt2 : dt1, &val.re, &val.im);
            if (DTC) {
                dt1_last = dt1;
                dt2_last = dt2;
            }
        }
    }
    template <int PH>
    CARMA_DEV void publish() const
    {
        if constexpr (PH == 0) g.publish2(val.re, val.im);
    }
    template <int PH>
    CARMA_DEV void fetch(Cx& rho, Cx (&rj)[P]) const
    {
        constexpr double se = PH ? -1.0 : 1.0, so = -se;      // sign of the imaginary part for the even / odd member
#pragma unroll
        for (int j = 0; j < P; j += 2) {
            const Cx v = g.peek2(j + PH);
            rj[j] = Cx{v.re, se * v.im};
            if (j + 1 < P) rj[j + 1] = Cx{v.re, so * v.im};
        }
        const int r = g.lane();
        const Cx v = g.peek2((r & ~1) + PH);                  // own pair
        rho = Cx{v.re, (r & 1) ? so * v.im : se * v.im};
    }
    template <int PH>
    CARMA_DEV void prepare(double dt1, double dt2)
    {
        if constexpr (PH == 1) begin2(dt1, dt2);
    }
};

// Diagnostic build only (-DCARMA_STAMPS): where a step spends its cycles (shares, not run time).
#if defined(CARMA_STAMPS) && defined(__HIPCC__)
#define CARMA_STAMP(var)                                                        \
    do {                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                      \
    } while (0)
#define CARMA_STAMP_DECL unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, sa = 0, sb = 0, sc = 0, sd = 0
#define CARMA_STAMP_ACC(acc, a, b) acc += (b) - (a)
// phase marks of a whole wave (core clock, relative to the wave's first mark), printed by workgroup 0
#define CARMA_MARK_DECL long long mark_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define CARMA_MARK(i) do { __builtin_amdgcn_sched_barrier(0); mark_[i] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define CARMA_MARK_DUMP(who, k)                                                                                   \
    do {                                                                                                            \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)                                                             \
            printf("%s %d: start %lld | +%lld +%lld +%lld +%lld +%lld +%lld +%lld | +%lld +%lld +%lld +%lld +%lld +%lld cycles\n", who, k, mark_[0],   \
                   mark_[1] ? mark_[1] - mark_[0] : 0, mark_[2] ? mark_[2] - mark_[0] : 0, mark_[3] ? mark_[3] - mark_[0] : 0,           \
                   mark_[4] ? mark_[4] - mark_[0] : 0, mark_[5] ? mark_[5] - mark_[0] : 0, mark_[6] ? mark_[6] - mark_[0] : 0,           \
                   mark_[7] ? mark_[7] - mark_[0] : 0, mark_[8] ? mark_[8] - mark_[0] : 0, mark_[9] ? mark_[9] - mark_[0] : 0,           \
                   mark_[10] ? mark_[10] - mark_[0] : 0, mark_[11] ? mark_[11] - mark_[0] : 0, mark_[12] ? mark_[12] - mark_[0] : 0,     \
                   mark_[13] ? mark_[13] - mark_[0] : 0);                                                                                 \
    } while (0)
#else
#define CARMA_STAMP(var) do { } while (0)
#define CARMA_STAMP_DECL do { } while (0)
#define CARMA_STAMP_ACC(acc, a, b) do { } while (0)
#define CARMA_MARK_DECL do { } while (0)
#define CARMA_MARK(i) do { } while (0)
#define CARMA_MARK_DUMP(who, k) do { } while (0)
#endif

// The same update loop in REAL modal coordinates (the default).
//
// The data are real, so the rotated state of a conjugate root pair (2k, 2k+1) is a conjugate pair
// itself, x_{2k+1} = conj(x_{2k}), and the p complex coordinates carry only p real degrees of
// freedom: z_{2k} = Re x_{2k}, z_{2k+1} = Im x_{2k} for a complex pair, z_r = x_r for a real root.
// In these coordinates (lane r <-> coordinate r, row r of the real symmetric D = Cov(z) - V_z):
//     y      = h.z,  h_{2k} = 2 Re b_{2k}, h_{2k+1} = -2 Im b_{2k}   (real root: h_r = b_r)
//     k      = Cov(z, y) = D h^T + c,   c_{2k} = Re (V b^H)_{2k}, c_{2k+1} = Im (V b^H)_{2k}
//     var    = s0 + h D h^T + e,        mean = h.z
//     z     <- Phi (z + k s innov),     D <- Phi (D - k k^T s) Phi^T
// where Phi is block diagonal: a rotation-scaling [[c,-s],[s,c]], c + i s = rho_{2k}, per complex pair
// and the scalar rho_r per real root.  With every lane holding its OWN rho_r = (c_r, s_r) (the odd
// member of a pair holds the conjugate, s_{2k+1} = -s_{2k}) both members use the same formulas
//     (d Phi^T)_{r,j} = d_{r,j} c_j - d_{r,j^1} s_j ,     (Phi m)_{r,j} = c_r m_{r,j} - s_r m_{r^1,j}
// and a real root (s = 0) needs no special case.  Same model, same likelihood as kfilter.cpp:189-215
// -- ~45 FP64 instructions per step for the covariance instead of ~120 in complex arithmetic, the
// partner row comes from the neighbouring lane by DPP (quad_perm xor 1).
template <int P, int G, bool WRITE_MV, class GrpT, class RhoSrc>
CARMA_DEV double filter_loop_real(const GrpT& g, const Model<P>& m, const FilterConsts<P>& fc, RhoSrc& src,
                                  const double4* __restrict__ series, int n, double* mean_out, double* var_out)
{
    const int r = g.lane();
    const bool act = r < P;
    const bool cpx = (m.w.im != 0.0) && (r < (P & ~1));     // member of a complex-conjugate pair
    const bool odd = r & 1;
    // observation row h and gain offset c in real coordinates
    const double c_im_partner = g.partner(fc.c_own.im);       // Im (V b^H) of the even member
    const double h_own = !act ? 0.0 : (cpx ? (odd ? 2.0 * fc.b_own.im : 2.0 * fc.b_own.re) : fc.b_own.re);
    const double c_own = cpx ? (odd ? c_im_partner : fc.c_own.re) : fc.c_own.re;
    double hall[P];
#pragma unroll
    for (int j = 0; j < P; j++) hall[j] = g.bcast_u(h_own, j);
    const double s0 = fc.s0;

    double D[P];
#pragma unroll
    for (int j = 0; j < P; j++) D[j] = 0.0;
    double z = 0.0;
    double k = c_own;
    double pvr = 0.0, pmr = 0.0;
    LogLikAcc acc;
    acc.init();
    double4 rprev = series[0];
    double4 rcur = series[n > 1 ? 1 : 0];
    double4 rnxt = series[n > 2 ? 2 : n - 1];
    if constexpr (RhoSrc::kPaired)
        src.begin2(rcur.x, rnxt.x);
    else
        src.begin(1, rcur.x);
    CARMA_STAMP_DECL;
    // n-1 passes; pass kk closes var_{kk-1}, mean_{kk-1} and applies Update kk.  The body is one
    // basic block (no branch on the pass index), so the var/mean butterflies and the reciprocal are
    // scheduled under the LDS round trip of the gain.  Passes come in chunks of RhoSrc::kChunk
    // (the ring's barrier period; one chunk for RhoInline).
    // RhoRing: the factors and the series record of pass kk+1 are read during pass kk (behind the
    // gain reads in the in-order LDS queue), so only the gain's round trip is ever waited for.
    Cx rho_n = {1.0, 0.0}, rj_n[P];
    double4 rec_n = rprev;
#pragma unroll
    for (int j = 0; j < P; j++) rj_n[j] = Cx{1.0, 0.0};
    if constexpr (RhoSrc::kRing) {
        if (n > 1) {
            src.chunk_begin(1);
            src.fetch_s(0, rho_n, rj_n);
            rec_n = src.record_s(0);
        }
    }
    auto pass = [&](const int kk, const int s_in_chunk, auto phase, const double4* rnn_in = nullptr) __attribute__((always_inline)) {
        constexpr int PH = decltype(phase)::value;          // RhoPair: step of the trip (0, 1); unused otherwise
        CARMA_STAMP(st0);
        double4 rnn = rnxt;
        if constexpr (!RhoSrc::kRing) rnn = rnn_in ? *rnn_in : series[(kk + 2 < n) ? kk + 2 : n - 1];
        // the gain goes through LDS (8 B per lane, read back as pairs)
        double kj[(P + 1) & ~1];
        Cx rho = rho_n, rj[P];
#pragma unroll
        for (int j = 0; j < P; j++) rj[j] = rj_n[j];
        if constexpr (RhoSrc::kRing) rprev = rec_n;              // series record kk-1
        g.publishk(k);
        if constexpr (RhoSrc::kPaired)
            src.template publish<PH>();
        else
            src.publish(kk);
#pragma unroll
        for (int i = 0; i < (P + 1) / 2; i++) g.peekk2(i, kj[2 * i], kj[2 * i + 1]);
        if constexpr (RhoSrc::kRing) {
            if (s_in_chunk == RhoSrc::kChunk - 1) {              // next pass opens a new chunk
                if (kk + 1 < n) {
                    src.chunk_begin(kk + 1);
                    src.fetch_s(0, rho_n, rj_n);
                    rec_n = src.record_s(0);
                }
            } else {
                src.fetch_s(s_in_chunk + 1, rho_n, rj_n);
                rec_n = src.record_s(s_in_chunk + 1);
            }
        } else if constexpr (RhoSrc::kPaired) {
            src.template fetch<PH>(rho, rj);
        } else {
            src.fetch(kk, rho, rj);
        }
        g.done_reading();                                        // pins the LDS issue order
        if constexpr (RhoSrc::kPaired)
            src.template prepare<PH>(rnxt.x, rnn.x);             // steps kk+1, kk+2: the next trip
        else
            src.prepare(kk + 1, rnxt.x);
        CARMA_STAMP(st1);
        // var_{kk-1} = s0 + h D h^T + e, mean_{kk-1} = h.z: DPP butterflies, bit-identical in the group
        const double pv = g.sum(pvr), pm = g.sum(pmr);
        const double var = s0 + pv + rprev.z * m.scale;      // kfilter.cpp:180-182, 209-210
        const double innov = (rprev.y - m.mu) - pm;          // kfilter.cpp:184, 207, 213
        acc.add_var(var);
        if (WRITE_MV && r == 0) {
            mean_out[kk - 1] = pm;
            var_out[kk - 1] = var;
        }
        const double s = recip(var);
        const double si = s * innov;
        acc.chi2 += innov * si;
        CARMA_STAMP(st2);
        // state (kfilter.cpp:191-194, 200-201)
        z = fma(k, si, z);
        const double zp = g.partner(z);
        z = rho.re * z - rho.im * zp;
        // covariance (kfilter.cpp:197, 204)
        const double t = k * s;
        double d[P], mm[P];
#pragma unroll
        for (int j = 0; j < P; j++) d[j] = fma(-t, kj[j], D[j]);
#pragma unroll
        for (int j = 0; j < P; j++) {
            if (j < (P & ~1))
                mm[j] = d[j] * rj[j].re - d[j ^ 1] * rj[j].im;
            else
                mm[j] = d[j] * rj[j].re;
        }
        double w0 = 0.0, w1 = 0.0;
#pragma unroll
        for (int j = 0; j < P; j++) {
            const double mp = g.partner(mm[j]);
            D[j] = rho.re * mm[j] - rho.im * mp;
            if (j & 1)
                w1 = fma(D[j], hall[j], w1);
            else
                w0 = fma(D[j], hall[j], w0);
        }
        const double w = w0 + w1;                    // (D h^T)_r
        k = w + c_own;
        pvr = h_own * w;
        pmr = h_own * z;
        rprev = rcur;
        rcur = rnxt;
        rnxt = rnn;
        CARMA_STAMP(st3);
        CARMA_STAMP_ACC(sa, st0, st1);
        CARMA_STAMP_ACC(sb, st1, st2);
        CARMA_STAMP_ACC(sc, st2, st3);
    };
    if constexpr (RhoSrc::kPaired) {
        // two steps per trip: the phase of the pair-shared factors is a compile-time constant in either half
        int kk = 1;
#pragma unroll 1
        for (; kk + 1 < n; kk += 2) {
            // the records of the two steps after this trip: adjacent in memory, one wide scalar load instead of two
            double4 ra, rb;
            if (kk + 3 < n) {
                ra = series[kk + 2];
                rb = series[kk + 3];
            } else {
                ra = series[(kk + 2 < n) ? kk + 2 : n - 1];
                rb = series[n - 1];
            }
            pass(kk, 0, IntC<0>{}, &ra);
            pass(kk + 1, 0, IntC<1>{}, &rb);
        }
        if (kk < n) pass(kk, 0, IntC<0>{});
    } else {
        for (int kk0 = 1; kk0 < n; kk0 += RhoSrc::kChunk) {
            if (RhoSrc::kRing && n - kk0 >= RhoSrc::kChunk) {
                // full chunk: constant trip count, unrolled so that ring offsets become immediates
#pragma unroll 4
                for (int s = 0; s < (RhoSrc::kRing ? RhoSrc::kChunk : 1); s++) pass(kk0 + s, s, IntC<0>{});
            } else {
                const int kend = (n - kk0 < RhoSrc::kChunk) ? n : kk0 + RhoSrc::kChunk;
#pragma unroll 1
                for (int kk = kk0; kk < kend; kk++) pass(kk, kk - kk0, IntC<0>{});
            }
        }
    }
    {   // last point: var_{n-1}, mean_{n-1}
        if constexpr (RhoSrc::kRing) rprev = series[n - 1];
        const double pv = g.sum(pvr), pm = g.sum(pmr);
        const double var = s0 + pv + rprev.z * m.scale;
        const double innov = (rprev.y - m.mu) - pm;
        acc.add_var(var);
        if (WRITE_MV && r == 0) {
            mean_out[n - 1] = pm;
            var_out[n - 1] = var;
        }
        acc.chi2 += innov * (recip(var) * innov);
    }
#if defined(CARMA_STAMPS) && defined(__HIPCC__)
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("stamps per step: exchange+wait %.1f  reduce+rcp %.1f  update %.1f cycles (n=%d)\n", (double)sa / (n - 1),
               (double)sb / (n - 1), (double)sc / (n - 1), n);
#endif
    return acc.total();
}

// Kalman filter of one evaluation (Reset + n-1 Updates) -> log-likelihood sum (no prior).
// y is centred with m.mu and yerr^2 scaled with m.scale on the fly (carpack.hpp:150-153).
// If WRITE_MV, lane 0 of the group also stores the one-step means/variances.
// *singular is set when the Vandermonde solve hits an exactly zero pivot (arma::solve throws).
template <int P, int G, bool WRITE_MV, class GrpT, bool DTC = false>
CARMA_DEV double filter_run(const GrpT& g, const Model<P>& m, const double4* __restrict__ series, int n,
                            double* mean_out, double* var_out, bool* singular)
{
    FilterConsts<P> fc;
    filter_reset<P, G>(g, m, fc);
    RhoInline<P, GrpT, DTC> src{g, m.w, Cx{1.0, 0.0}};
    double ll;
    // pair-shared factors unless some group of the wave has a quadratic factor with two real roots
    const bool real_pair = (g.lane() < (P & ~1)) && (m.w.im == 0.0);
    if (g.wave_all(!real_pair)) {
        RhoPair<P, GrpT, DTC> srcp{g, m.w, Cx{1.0, 0.0}};
        ll = filter_loop_real<P, G, WRITE_MV>(g, m, fc, srcp, series, n, mean_out, var_out);
    } else {
        ll = filter_loop_real<P, G, WRITE_MV>(g, m, fc, src, series, n, mean_out, var_out);
    }
    *singular = fc.sing;
    return ll;
}

// log prior (carpack.hpp:118-126)
CARMA_DEV double log_prior(double measerr_scale, double dof)
{
    return -0.5 * dof / measerr_scale - (1.0 + dof / 2.0) * log(measerr_scale);
}

// CARMA_Base::LogDensity (carpack.hpp:131-176) for CARp/CARMA: -inf outside the prior bounds
// or on a singular solve, else log-likelihood + log prior.
template <int P, int G, class GrpT, bool DTC = false>
CARMA_DEV double logdensity_carma(const GrpT& g, const double* theta, int q, const double4* __restrict__ series,
                                  int n, const Prior& pr, int ignore_prior)
{
    Model<P> m;
    model_from_theta<P, G>(g, theta, q, pr, ignore_prior, m);
    bool sing;
    double ll = filter_run<P, G, false, GrpT, DTC>(g, m, series, n, nullptr, nullptr, &sing);
    ll += log_prior(m.scale, pr.measerr_dof);
    const double ninf = -1.0 / 0.0;
    if (sing || !m.valid) ll = ninf;
    return ll;
}

// ---------------------------------------------------------------------------------------------
// CAR(1): one LANE per evaluation (kfilter.cpp:19-48, carpack.hpp:265,273-275, carpack.cpp:116-130)
CARMA_DEV double car1_filter(double sigsqr, double omega, double mu, double scale,
                             const double4* __restrict__ series, int n, bool write_mv, double* mean_out,
                             double* var_out)
{
    double4 rec = series[0];
    double e2 = rec.z * scale;
    double mean = 0.0;
    double var = sigsqr / (2.0 * omega) + e2;
    double yc = rec.y - mu;
    LogLikAcc acc;
    acc.init();
    acc.add_var(var);
    if (write_mv) {
        mean_out[0] = mean;
        var_out[0] = var;
    }
    // (round 4: one reciprocal of var per step instead of two IEEE divisions, the short exponential of carma_math.h instead of
    // the library's -- the step is one lane's dependent chain, 480 cycles of it before: 54 us for a 270-point series whatever
    // the batch, more than any CARMA(p >= 2) order took at the same size)
    // (evaluating rho of step k + 1 during step k -- it does not depend on the state -- was measured too: 46 instead of 44 us;
    // the step is bound by its ~50 instructions, not by the exponential's chain: profiles/r04/ab_car1_v2.txt)
    const double stat_var = sigsqr / (2.0 * omega);
    for (int k = 1; k < n; k++) {
        double innov = yc - mean;
        const double s = recip(var);
        acc.chi2 += innov * (innov * s);
        rec = series[k];
        double rho = exp_neg(-1.0 * omega * rec.x);
        double previous_var = var - e2;
        double var_ratio = previous_var * s;
        mean = rho * mean + rho * var_ratio * innov;
        var = stat_var * (1.0 - rho * rho) + rho * rho * previous_var * (1.0 - var_ratio);
        e2 = rec.z * scale;
        var += e2;
        yc = rec.y - mu;
        acc.add_var(var);
        if (write_mv) {
            mean_out[k] = mean;
            var_out[k] = var;
        }
    }
    double innov = yc - mean;
    acc.chi2 += innov * (innov * recip(var));
    return acc.total();
}

CARMA_DEV double logdensity_car1(const double* theta, const double4* __restrict__ series, int n, const Prior& pr)
{
    double ysigma = theta[0], ms = theta[1], mu = theta[2];
    double omega = exp(theta[3]);
    double sigsqr = 2.0 * ysigma * ysigma * exp(theta[3]);
    bool ok = !((omega > pr.max_freq) || (omega < pr.min_freq) || (ysigma > pr.max_stdev) || (ysigma < 0) ||
                (ms < 0.5) || (ms > 2.0));
    double ll = car1_filter(sigsqr, omega, mu, ms, series, n, false, nullptr, nullptr);
    ll += log_prior(ms, pr.measerr_dof);
    const double ninf = -1.0 / 0.0;
    return ok ? ll : ninf;
}

}  // namespace carma
