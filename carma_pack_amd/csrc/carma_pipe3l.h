// carma_pipe3l.h -- the wave pipeline of the latency regime (<= 3072 evaluations), in a CO-ROTATING FRAME (gfx950 only).
// One evaluation per 16-lane DPP row, four evaluations per workgroup; the step is cut between the covariance recursion
// (which does not depend on the data) and the mean recursion, and the two halves run on different SIMDs one chunk apart.
//
// The step of the covariance wave is  D <- Phi_k (D - k k^T / var) Phi_k^T  (kfilter.cpp:197, 204) with Phi_k the
// block-diagonal rotation exp(omega dt_k) of the real modal coordinates; 24 of its 56 issue slots (p = 5) are that
// rotation.  Write D = A S A^T with A the transition accumulated since the last RE-BASE (all Phi commute):
//     w~ = S h~ ,  h~ = A^T h        var = s0 + e + h~.w~        k~ = w~ + c~ ,  c~ = A^-1 c
//     S <- S - k~ k~^T / var         and for the mean  z = A z~ :  innov = y - mu - h~.z~ ,  z~ <- z~ + k~ innov / var
// -- no rotation of the matrix at all; the per-step vectors h~, c~ depend only on omega and on the time since the
// re-base and are computed by the PRODUCER waves (two of them: the covariance wave got 1.6x faster).  Element by
// element the arithmetic is the same as the rotated recursion up to the scale factors e^{+-Re(omega) dt}, which
// cancel in every product (numpy prototype tests/tools/proto/lazy_frame.py: the same error against the reference
// restatement as the stepwise rotation, 1e-12 at worst over the bench batch).
// RE-BASE: before |Re omega| dt_acc could overflow the scale factors the accumulated rotation is applied for real
// (S <- A S A^T, z <- A z~; column mix by DPP, row mix with the pair partner) and the frame restarts -- not at the identity
// but HALF A WINDOW AHEAD, at the scale g_r = e^{-Re omega_r W/2}, so that a root's scale factor runs from e^+200 to e^-200
// over a window instead of from 1 to e^-200: windows twice as long for the same bound on S.  (The phase does not limit the
// window: the producers recover the rounding of Im omega x dt_acc with an FMA.)  The modal coordinates themselves are
// rescaled by exact powers of two so that |h_r| is of order one -- the sampler's unconstrained MA parameters otherwise
// give h_r ~ 1e115, c_r ~ 1e-115.  The schedule is a time grid per evaluation: datum j is a re-base datum when
// floor(t_j 2^ex) != floor(t_{j-1} 2^ex), 2^-ex <= min over the roots of (LIM_RE / |Re omega|, LIM_IM / |Im omega|);
// the accumulated time of a non-re-base datum is therefore < 2^-ex.  (Dyadic cells nest: the waves branch on the
// union of the four evaluations' masks -- that of the finest grid -- and a row without a re-base of its own at such a
// datum rotates by the identity, so an evaluation's result does not depend on its neighbours in the batch.)  A
// re-base datum's ring entry holds the accumulated (E cos, E sin) instead of (h~, c~) -- there h~ = g h, c~ = c / g,
// published once by the producers -- and a 16-bit mask per chunk tells the recursion waves which data those are; a
// chunk whose mask is zero (the rule for posterior-like parameters) runs a copy of the passes without any check.
// A last chunk of 11..15 data is completed to 16 with neutral pad data (carma_types.h, p3l_pad).  The producers need h and c: they
// evaluate the exp/sincos of chunk 0 while the recursion waves set the model up, wait for the covariance wave to
// publish (h_r, c_r), and only then form the entries of chunk 0.
//
//   waves P0, P1 (producers)   ring entry per (datum, evaluation, root), re-base masks
//   (y_j and yerr_j^2 are wave-uniform: the recursion waves read them with scalar loads from the series itself)
//   wave A (covariance)        [re-base]; w~, var, k~ -> link ring; S -= k~ k~^T / var          (lane r = row r of S)
//   wave B (mean)              [re-base]; innov, chi2, sum log var; z~ += k~ innov / var        -> log-likelihood
// One __syncthreads() per 16-datum chunk for all four waves: after barrier b the producers write chunk b+1 (ring
// buffer (b+1)%3), A works on chunk b (reads ring b%3, writes link b%2), B on chunk b-1 -- all distinct buffers.
#pragma once
#include <hip/hip_runtime.h>

#include "carma_core.h"
#include "grp_device.h"
#include "carma_ring.h"

namespace carma {

// real-coordinate constants of one evaluation, as held by lane r of its row (see filter_loop_real)
template <int P>
struct RowConsts {
    double h_own, c_own, s0;
};
template <int P>
__device__ __forceinline__ void row_consts(const Grp<16>& g, const Model<P>& m, const FilterConsts<P>& fc, RowConsts<P>& rc)
{
    const int r = g.lane();
    const bool act = r < P;
    const bool cpx = (m.w.im != 0.0) && (r < (P & ~1));
    const bool odd = r & 1;
    const double c_im_partner = g.partner(fc.c_own.im);
    rc.h_own = !act ? 0.0 : (cpx ? (odd ? 2.0 * fc.b_own.im : 2.0 * fc.b_own.re) : fc.b_own.re);
    rc.c_own = cpx ? (odd ? c_im_partner : fc.c_own.re) : fc.c_own.re;
    rc.s0 = fc.s0;
}

// wave priorities when workgroups share a CU (A/B builds may override; measured: profiles/r03/priorities_ab_v*.txt)
#ifndef CARMA_PRIO_A
#define CARMA_PRIO_A 3
#endif
#ifndef CARMA_PRIO_B
#define CARMA_PRIO_B 1
#endif
#ifndef CARMA_PRIO_P
#define CARMA_PRIO_P 1
#endif

template <int P>
struct Pipe3LGeom {
    static constexpr int C = 16;
    // LDS entries (16 B each) per datum: only lanes 0..P-1 of a 16-lane row carry an evaluation's values, so a row gets
    // ROWW = 8 entries (P <= 7) and the idle lanes 8..15 of all four rows share ONE dump entry.  42 KiB per workgroup
    // instead of 82: three workgroups fit the 160 KiB of a CU, which is what lets launches of 1025..3072 evaluations
    // (and sampler grids beyond one workgroup per CU) keep this kernel.
    static constexpr int ROWW = 8, SLOT = 4 * ROWW + 1;
    static constexpr int RING_OFF = 0;                          // double2[3][C][SLOT]
    static constexpr int LINK_OFF = 3 * C * SLOT;               // double2 {k~_r, var}[2][C][SLOT]
    static constexpr int CONST_OFF = LINK_OFF + 2 * C * SLOT;   // double2 {h_r, c_r}[SLOT]
    static constexpr int CONST2_OFF = CONST_OFF + SLOT;         // double2 {g_r h_r, c_r / g_r}[SLOT]: (h~, c~) at a re-base datum
    static constexpr int FLAG_OFF = CONST2_OFF + SLOT;          // u64[3]: re-base data of a chunk, bit 16 row + slot; u64[3]: rows with a repeated AR root
    static constexpr int TAIL_OFF = FLAG_OFF + 2;               // double[2][C]: yerr^2 (wave A) and y (wave B) of the last, partial chunk
    static constexpr int ENTRIES = TAIL_OFF + C;
    static constexpr size_t BYTES = (size_t)ENTRIES * sizeof(Cx);   // 41.8 KiB
    static constexpr double LIM_RE = 600.0;                     // |Re omega| x window: the scale factor of a root runs from
                                                                // e^+300 at a re-base to e^-300 at the end of the window (the
                                                                // frame starts HALF a window ahead), so S carries factors within
                                                                // e^+-600 (4e260) -- three times the window of a frame that
                                                                // starts at the identity and stops at e^-200.  The modal
                                                                // coordinates are rescaled so that S itself is of order one
                                                                // (times the conditioning of the modal basis, < 1e13) in any units
    static constexpr double LIM_IM = 262144.0;                  // |Im omega| dt_acc: the producers recover the rounding of the phase
                                                                // product (cexp_step<true>), so the window is set by the decay of
                                                                // the scale factors alone unless Q = |Im|/|Re| exceeds 1300; the
                                                                // limit keeps the phase inside the fast argument reduction (2^20)
    static_assert(P < ROWW, "one row of entries per evaluation");
    // entry of a lane inside a [SLOT] array
    static __device__ __forceinline__ int entry(int lane)
    {
        const int l = lane & 15;
        return l < ROWW ? (lane >> 4) * ROWW + l : 4 * ROWW;
    }
    static __device__ __forceinline__ int row_base(int lane) { return (lane >> 4) * ROWW; }
};

// DPP move of a double with an explicit `old` value for the lanes the pattern leaves unwritten
template <int CTRL>
CARMA_DEV double dpp_mov_old(double old, double src)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// waves P0 / P1 (pw = 0, 1)
// `tail(pw)` runs once all chunks are produced, while the recursion waves work through the last two of them: the
// sampler kernel draws the next iteration's random numbers there.
template <int P, class Tail>
__device__ __forceinline__ void pipe3l_produce(const Grp<16>& g, int pw, const double* __restrict__ theta,
                                               const double4* __restrict__ series, int n, int npad, Cx* __restrict__ ring,
                                               Tail&& tail)
{
    // n counts the npad neutral pad data at the end (carma_types.h, p3l_pad): their entries are exact zeros
    using Geo = Pipe3LGeom<P>;
    constexpr int C = Geo::C;
    if (CARMA_PRIO_P != 0) __builtin_amdgcn_s_setprio(CARMA_PRIO_P);
    // a lane works on a conjugate PAIR of roots (2 pr, 2 pr + 1): both share |E|, cos and sin, so one exp/sincos
    // evaluation serves two ring entries; an odd order's last lane holds the single real root
    constexpr int NPAIR = (P + 1) / 2, PPL = 16 / NPAIR;
    const int lane = g.lane64, l = lane & 15;
    const int sub = l / NPAIR, jr = 2 * (l - sub * NPAIR);
    const bool worker = sub < PPL, two = jr + 1 < P;
    const Cx w = own_ar_root<P>(theta, jr);
    // a quadratic factor with positive discriminant puts two different REAL roots into the pair: the second one then
    // gets its own exponential (no sine / cosine either way)
    const Cx w1 = two ? own_ar_root<P>(theta, jr + 1) : w;
    int esig = 0;                                             // binary exponent of sigma_y = theta[0] (see the rescaling below)
    {
        const double sg = fabs(theta[0]);
        (void)frexp(sg, &esig);
        if (!(sg > 0.0 && sg < 1.0 / 0.0)) esig = 0;
    }
    const bool realpair = two && w.im == 0.0;
    const int nc = (n + C - 1) / C;
    if (pw == 0 && l >= P) {                                  // entries of the idle lanes: exact zeros
#pragma unroll 4
        for (int i = 0; i < 3 * C; i++) ring[(size_t)i * Geo::SLOT + Geo::entry(lane)] = Cx{0.0, 0.0};
        ring[Geo::CONST2_OFF + Geo::entry(lane)] = Cx{0.0, 0.0};
    }
    // grid of this evaluation's re-base schedule: cells of width 2^-ex <= min over its roots of (LIM_RE / |Re omega|,
    // LIM_IM / |Im omega|).  Dyadic widths nest, so the re-base data of the evaluation with the finest grid contain
    // those of the other three evaluations of the workgroup and the waves branch on that one mask.
    double wl = fmax(fmax(fabs(w.re), fabs(w1.re)) * (1.0 / Geo::LIM_RE), fabs(w.im) * (1.0 / Geo::LIM_IM));
    wl = (wl < 1e12) ? wl : ((wl == wl && wl < 1.0 / 0.0) ? 1e12 : 0.0);
    wl = Grp<16>::max(wl);
    int wex;
    (void)frexp(wl, &wex);
    const double sc = wl > 0.0 ? ldexp(1.0, wex) : 0.0;
    // the frame of a window starts half a window ahead: scale factor g_r e^{Re omega_r dt}, g_r = e^{-Re omega_r W / 2}
    const double halfw = sc > 0.0 ? 0.5 / sc : 0.0;
    // (this wave reaches the first barrier about when the covariance wave does: short forms, no library calls)
    const double rg0 = exp_neg(w.re * halfw), g0 = recip(rg0);
    double rg1 = rg0, g1 = g0;
    if (realpair) {
        rg1 = exp_neg(w1.re * halfw);
        g1 = recip(rg1);
    }
    const double ninf = -1.0 / 0.0;
    auto clampi = [n](int i) { return i < n ? (i < 0 ? 0 : i) : n - 1; };
    double4 rec_n = series[clampi(l)];                       // records are fetched one chunk ahead
    double carry = __shfl(rec_n.w, 0, 64);                   // time of the current base datum
    double t_last = carry;                                   // time of the datum before this chunk
    double2 hc_own = make_double2(0.0, 0.0), hc_par = make_double2(0.0, 0.0);
    constexpr int NIT = (C + 2 * PPL - 1) / (2 * PPL);        // exp/sincos evaluations per lane and chunk
    double ec0[NIT], es0[NIT], e10[NIT];                              // chunk 0: evaluated while the other waves set the model up
    for (int c = 0; c < nc; c++) {
        const int j0 = c * C;
        // --- schedule of this chunk: lane s (of every row) looks at datum j0 + s
        const double4 rec = rec_n;
        rec_n = series[clampi(j0 + C + l)];
        const double tj = rec.w, tjm = dpp_mov_old<0x111>(t_last, tj);      // row_shr:1, lane 0 <- last datum of the previous chunk
        t_last = __shfl(tj, 15, 64);
        const bool fl = floor(tj * sc) != floor(tjm * sc);
        double M = fl ? tj : ninf;                            // inclusive max-scan over the 16 lanes: latest re-base time
        M = fmax(M, dpp_mov_old<0x111>(ninf, M));             // row_shr:1
        M = fmax(M, dpp_mov_old<0x112>(ninf, M));             // row_shr:2
        M = fmax(M, dpp_mov_old<0x114>(ninf, M));             // row_shr:4
        M = fmax(M, dpp_mov_old<0x118>(ninf, M));             // row_shr:8
        const double Mx = dpp_mov_old<0x111>(ninf, M);        // exclusive
        // time since the base this datum is expressed in.  (The difference of two time stamps is exact unless the base is
        // much the smaller of the two, and then off by at most half an ulp of t_j: 1e-13 rad for |Im omega| = 1 at t = 1000.)
        const double dta_l = tj - fmax(carry, Mx);
        carry = fmax(carry, __shfl(M, (lane & ~15) + 15, 64));
        const unsigned long long fmask = __ballot(fl && j0 + l < n);      // bit 16 row + s: datum j0 + s of that row's evaluation
        if (pw == 0 && lane == 0) reinterpret_cast<unsigned long long*>(ring + Geo::FLAG_OFF)[c % 3] = fmask;
        Cx* buf = ring + Geo::RING_OFF + (size_t)(c % 3) * C * Geo::SLOT + Geo::row_base(lane) + jr;
        auto slot_of = [&](int it) { return it * 2 * PPL + pw * PPL + sub; };
        auto rotation = [&](int it, double& ec, double& es, double& e1) {   // accumulated (E cos, E sin) of the lane's slot
            const int slot = slot_of(it);
            const double dta = __shfl(dta_l, (lane & ~15) + (slot < C ? slot : C - 1), 64);
            ec = 1.0;
            es = 0.0;
            if (worker && slot < C && j0 + slot < n) cexp_step<true>(w.re, w.im, dta, &ec, &es);
            e1 = ec;                                          // the partner: same modulus ...
            if (realpair && worker && slot < C && j0 + slot < n) {
                double z;
                cexp_step<true>(w1.re, 0.0, dta, &e1, &z);         // ... unless it is another real root
            }
        };
        auto entry = [&](int it, double ec, double es, double e1) {
            const int slot = slot_of(it);
            const bool flag = __shfl((int)fl, (lane & ~15) + (slot < C ? slot : C - 1), 64) != 0;
            if (worker && slot < C && j0 + slot < n) {
                // h~_r = (A^T h)_r = E (cos h_r + sin h_partner) ;  c~_r = (A^-1 c)_r = (cos c_r + sin c_partner) / E
                // (the partner root is the conjugate: same cos, sin of the opposite sign)
                // a re-base datum's entry is the rotation accumulated over the closing window (the offsets g cancel)
                const double gc = ec * g0, gs = es * g0;
                const double inv = recip(fma(gc, gc, gs * gs));
                const double ht = fma(gc, hc_own.x, gs * hc_par.x);
                const double ct = fma(gc, hc_own.y, gs * hc_par.y) * inv;
                const bool pad = j0 + slot >= n - npad;
                buf[(size_t)slot * Geo::SLOT] = pad ? Cx{0.0, 0.0} : (flag ? Cx{ec, es} : Cx{ht, ct});
                if (two) {
                    const double gc1 = e1 * g1, gs1 = es * g1;
                    const double inv1 = realpair ? recip(gc1 * gc1) : inv;
                    const double hp = fma(gc1, hc_par.x, -gs1 * hc_own.x);
                    const double cp = fma(gc1, hc_par.y, -gs1 * hc_own.y) * inv1;
                    buf[(size_t)slot * Geo::SLOT + 1] = pad ? Cx{0.0, 0.0} : (flag ? Cx{e1, -es} : Cx{hp, cp});
                }
            }
        };
        if (c == 0) {
#pragma unroll
            for (int it = 0; it < NIT; it++) rotation(it, ec0[it], es0[it], e10[it]);
            __syncthreads();                                  // the covariance wave has published (h_r, c_r)
            const double2* cst = reinterpret_cast<const double2*>(ring + Geo::CONST_OFF) + Geo::row_base(lane);
            hc_own = cst[jr];
            hc_par = cst[jr + 1];
            // Modal coordinates come in whatever scale the MA polynomial gives them: the sampler's unconstrained MA
            // parameters reach h_r ~ 1e115, c_r ~ 1e-115 with s0 = h.V.h of order one, and the frame's scale factors
            // (e^+-200) would push such products out of range.  So every coordinate is rescaled by an exact power of two
            // that makes |h_r| of order one (one power for both members of a complex pair: the rescaling must commute
            // with their rotation).  Everything the recursion waves see -- ring entries, the re-base constants -- comes
            // from here, so they work in the rescaled coordinates without knowing; h.D.h, h.z and with them var and
            // the innovation are unchanged bit for bit.
            {
                const double m0 = fmax(fabs(hc_own.x), realpair ? 0.0 : fabs(hc_par.x)), m1 = realpair ? fabs(hc_par.x) : m0;
                int e0, e1x;
                (void)frexp(m0, &e0);
                (void)frexp(m1, &e1x);
                if (!(m0 > 0.0 && m0 < 1.0 / 0.0)) e0 = 0;
                if (!(m1 > 0.0 && m1 < 1.0 / 0.0)) e1x = 0;
                // ... and by the power of two of sigma_y = sqrt(s0) on top: |h_r| ~ sigma_y, c_r ~ sigma_y, D of order one
                // whatever units the data come in, so that the frame's e^+-LIM_RE has the whole exponent range to itself
                hc_own = make_double2(ldexp(hc_own.x, esig - e0), ldexp(hc_own.y, e0 - esig));
                hc_par = make_double2(ldexp(hc_par.x, esig - e1x), ldexp(hc_par.y, e1x - esig));
            }
            if (pw == 0 && l < NPAIR) {                       // (h~, c~) right after a re-base, for the recursion waves
                double2* cst2 = reinterpret_cast<double2*>(ring + Geo::CONST2_OFF) + Geo::row_base(lane);
                cst2[jr] = make_double2(g0 * hc_own.x, hc_own.y * rg0);
                if (two) cst2[jr + 1] = make_double2(g1 * hc_par.x, hc_par.y * rg1);
            }
#pragma unroll
            for (int it = 0; it < NIT; it++) entry(it, ec0[it], es0[it], e10[it]);
        } else {
#if defined(CARMA_AB_NOPROD)                                  // timing-only A/B build: what the producers' work costs the others
            if (c < 3)
#endif
#pragma unroll 1
            for (int it = 0; it < NIT; it++) {
                double ec, es, e1;
                rotation(it, ec, es, e1);
                entry(it, ec, es, e1);
            }
        }
        __syncthreads();                                      // barrier c: chunk c is in the ring
    }
    tail(pw);
    __syncthreads();                                          // barrier nc (wave B's last chunk)
}

// wave A: lane r of a 16-lane row holds row r of S (the row sum w~ = S h~ wants the whole row in one lane).
// The step is a dependent chain  w~ -> t -> var -> 1/var -> S -> w~ ...; the row sums run as two accumulation chains
// and the reciprocal refinement is folded into the gain so that the chain, not the issue rate, stays short.
template <int P>
__device__ __forceinline__ void pipe3l_cov(const Grp<16>& g, const Model<P>& m, const RowConsts<P>& rc,
                                           const double4* __restrict__ series, int n, int npad, Cx* __restrict__ ring)
{
    const double* __restrict__ e_arr = reinterpret_cast<const double*>(series + (n - npad + P3L_PAD_RECORDS));   // yerr^2[]
    using Geo = Pipe3LGeom<P>;
    using RA = RowAsm<P>;
    constexpr int C = Geo::C;
    const int lane = g.lane64;
    const int nc = (n + C - 1) / C;
    // When workgroups share a CU this wave's instruction stream is the one the launch waits for: it issues ahead of the
    // mean waves and the producers (priority 1 both: 3 / 1 / 0 was 4 % slower) of the other workgroups on its SIMD (2048
    // evaluations 39.0 -> 35.2 us, 3072: 51.5 -> 47.4 us,
    // 16 x 128 ladders 20.6k -> 22.3k iterations/s; nothing to arbitrate with one workgroup per CU)
    __builtin_amdgcn_s_setprio(CARMA_PRIO_A);
    const bool act = (lane & 15) < P;
    const double h_row = act ? rc.h_own : 0.0, c_row = act ? rc.c_own : 0.0;     // idle lanes carry exact zeros
    reinterpret_cast<double2*>(ring + Geo::CONST_OFF)[Geo::entry(lane)] = make_double2(h_row, c_row);
    {
        // ONE definition of "repeated AR root" for the whole launch shape: this wave's (alpha'(omega_r) == 0, model_kappa --
        // the test of the other kernels), handed to the mean wave, which forms the result, through the spare flag word
        const unsigned long long sing_rows = __ballot(m.sing);
        if (lane == 0) reinterpret_cast<unsigned long long*>(ring + Geo::FLAG_OFF)[3] = sing_rows;
    }
    __syncthreads();                                          // the producers finish chunk 0 with these constants
    const double one = 1.0;
    double S[P];
#pragma unroll
    for (int j = 0; j < P; j++) S[j] = 0.0;
    const double2* ring_b = nullptr;
    double2* link_b = nullptr;
    double2 hc_n = make_double2(0.0, 0.0), hc0 = make_double2(0.0, 0.0);
    unsigned rowm = 0;
    int j0 = 0;                                               // first datum of the current chunk
    auto pass = [&](const int s, const bool more, const bool rebase, const double e) __attribute__((always_inline)) {
        const double2 hc = hc_n;
        if (more) hc_n = ring_b[(size_t)(s + 1) * Geo::SLOT];
        __builtin_amdgcn_sched_barrier(0);                    // the next pass's entry is requested HERE, a pass ahead
        double ht = hc.x, ct = hc.y;
        if (__builtin_expect(rebase, 0)) {
            // S <- A S A^T with the accumulated rotation (kfilter.cpp:204 for the whole window); a row whose own
            // schedule has no re-base here rotates by the identity and keeps its (h~, c~)
            const bool mine = (rowm >> s) & 1u;
            const double rc_ = mine ? hc.x : 1.0, rs_ = mine ? hc.y : 0.0;
            double mm[P];
            g.template row_colmix<P>(mm, rc_, rs_, S);
#pragma unroll
            for (int j = 0; j < P; j++) {
                const double mp = g.partner(mm[j]);
                S[j] = fma(rc_, mm[j], -(rs_ * mp));
            }
            ht = mine ? hc0.x : hc.x;
            ct = mine ? hc0.y : hc.y;
        }
        // w~ = S h~ ; var_j = s0 + e + h~.w~ (kfilter.cpp:180-182, 209-210) ; k~ = w~ + c~
        double w, var, k;
        RA::lazy_front(w, var, k, ht, ct, e, m.scale, rc.s0, one, S);
        link_b[(size_t)s * Geo::SLOT] = make_double2(k, var);
        // S_j -= (k~ / var) k~_j   (kfilter.cpp:197);  1/var = r0 (1 + e) (1 + O(e^2)), e = 1 - var r0, folded into
        // nt = -k~ / var.  v_rcp_f64 is good to 2^-24.4, so this Newton step leaves 2e-15 relative in the rank-1 term of
        // ONE step of a contracting recursion (parity unchanged: profiles/r02/parity_sweep_v3.txt) -- the cubic step
        // nt = kr (1 + e + e^2) of round 1 was one instruction and 8 cycles of dependent latency more, every datum.
        const double r0 = __builtin_amdgcn_rcp(var);
        const double kr = -k * r0;
        const double er = fma(-var, r0, 1.0);
        const double nt = fma(kr, er, kr);
        RA::gain_nt(S, k, nt);
    };
    const unsigned long long* flag_b = reinterpret_cast<const unsigned long long*>(ring + Geo::FLAG_OFF);
#if defined(CARMA_CHUNK_STAMPS)
    // diagnostic build: where a chunk of the covariance wave spends its cycles -- arriving at the barrier (drain of the last
    // link write), released from it, first pass's operands in registers, end of the sixteen passes
    unsigned long long cs_bar = 0, cs_head = 0, cs_body = 0, cs_t0 = 0, cs_t1 = 0, cs_t2 = 0, cs_t3 = 0;
#define CHUNK_STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
#endif
    for (int c = 0; c < nc; c++) {
#if defined(CARMA_CHUNK_STAMPS)
        __builtin_amdgcn_sched_barrier(0);
        CHUNK_STAMP(cs_t0);
        if (c > 1) cs_body += cs_t0 - cs_t3;
#endif
        __syncthreads();                                      // barrier c
#if defined(CARMA_CHUNK_STAMPS)
        __builtin_amdgcn_sched_barrier(0);
        CHUNK_STAMP(cs_t1);
        if (c > 1) cs_bar += cs_t1 - cs_t0;
#endif
        // yerr_j^2 is wave-uniform: the chunk's sixteen values in two wide scalar loads from the plain array behind the
        // records (scalar-memory and LDS returns share a counter, so a scalar load inside a pass would drain the LDS
        // prefetch every step).  Requested FIRST, so that their latency runs under that of the LDS reads below.
        j0 = c * C;
        const int len = (n - c * C < C) ? n - c * C : C;
        double ev[C];
        if (len == C) {
#if defined(CARMA_AB_NOEV)                                    // timing-only A/B build: what the chunk's scalar loads cost the head
#pragma unroll
            for (int s = 0; s < C; s++) ev[s] = 0.01 * (s + 1);
#else
#pragma unroll
            for (int s = 0; s < C; s++) ev[s] = e_arr[j0 + s];
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        ring_b = reinterpret_cast<const double2*>(ring + Geo::RING_OFF + (size_t)(c % 3) * C * Geo::SLOT) + Geo::entry(lane);
        link_b = reinterpret_cast<double2*>(ring + Geo::LINK_OFF) + (size_t)(c & 1) * C * Geo::SLOT + Geo::entry(lane);
        hc_n = ring_b[0];
        if (c == 0) hc0 = reinterpret_cast<const double2*>(ring + Geo::CONST2_OFF)[Geo::entry(lane)];
        const unsigned long long fm64 = flag_b[c % 3];
        rowm = (unsigned)(fm64 >> (16 * (lane >> 4))) & 0xffffu;                   // this row's evaluation
        const unsigned fm_lo = __builtin_amdgcn_readfirstlane((unsigned)fm64), fm_hi = __builtin_amdgcn_readfirstlane((unsigned)(fm64 >> 32));
        const unsigned fm = (fm_lo | (fm_lo >> 16) | fm_hi | (fm_hi >> 16)) & 0xffffu;     // any row (= the finest grid's)
#if defined(CARMA_CHUNK_STAMPS)
        __builtin_amdgcn_sched_barrier(0);
        CHUNK_STAMP(cs_t3);                                   // (waits for the head's LDS reads and scalar loads)
        if (c > 1) cs_head += cs_t3 - cs_t1;
        (void)cs_t2;
#endif
        if (len == C) {
            if (fm == 0) {
                // no re-base in this chunk (the rule for posterior-like parameters): a copy of the passes without the
                // sixteen skip-branches -- a TAKEN branch over the re-base block costs the wave ~30 cycles, every datum
#pragma unroll
                for (int s = 0; s < C; s++) pass(s, s + 1 < C, false, ev[s]);
            } else {
#pragma unroll
                for (int s = 0; s < C; s++) pass(s, s + 1 < C, (fm >> s) & 1u, ev[s]);
            }
        } else {
            // The last, shorter chunk runs as a rolled loop.  Its yerr_j^2 come through LDS (one vector load, lane s <->
            // datum s, staged once; read back one pass ahead like the ring entries): a scalar load inside the pass made
            // every LDS wait drain an L2 round trip as well -- 600 instead of 230 cycles per datum, 2 us per evaluation
            // of a 270-point series.
            double* tail = reinterpret_cast<double*>(ring + Geo::TAIL_OFF);
            const int jt = j0 + (lane & 15);
            const double ez = series[jt < n ? jt : n - 1].z;
            if (lane < C) tail[lane] = ez;
            g.sync();
            double e_n = tail[0];
#pragma unroll 1
            for (int s = 0; s < len; s++) {
                const double e = readlane_f64(e_n, 0);
                e_n = tail[s + 1 < C ? s + 1 : s];
                pass(s, true, (fm >> s) & 1u, e);
            }
        }
    }
#if defined(CARMA_CHUNK_STAMPS)
    if (blockIdx.x == 0 && lane == 0)
        printf("p3l covariance wave, %d chunks: per chunk (cycles) sixteen passes %llu | barrier (incl. draining the link write) %llu | head: pointers, "
               "flag word, first entry, yerr^2 loads %llu\n", nc, cs_body / (unsigned long long)(nc > 2 ? nc - 2 : 1),
               cs_bar / (unsigned long long)(nc > 2 ? nc - 2 : 1), cs_head / (unsigned long long)(nc > 2 ? nc - 2 : 1));
#endif
    __syncthreads();                                          // barrier nc
}

// What npad pad data (carma_types.h, p3l_pad) add to the sums of the mean wave, to be taken out again: each of them has
// var = 1 * scale + s0 (one FMA, as the covariance wave forms it) and innov = y_last - mu exactly, i.e. contributes
// -0.5 (log var + innov^2 / var).
__device__ __forceinline__ double pipe3l_pad_correction(int npad, double sigma_y, double scale, double y_last, double mu)
{
    const double s0 = sigma_y * sigma_y;                      // as Model::s0
    const double v = fma(1.0, scale, s0);
    const double dd = y_last - mu;
    return npad ? 0.5 * npad * (log(v) + dd * (recip(v) * dd)) : 0.0;
}

// wave B.  Of the model it needs mu only (and the flags its caller checks): the observation row h_r -- used at the
// re-base data -- is read from what the covariance wave published for the producers, so this wave does not repeat that
// part of the set-up (model_from_theta<MODEL_FLAGS>).
template <int P>
__device__ __forceinline__ double pipe3l_mean(const Grp<16>& g, double mu, const double4* __restrict__ series, int n,
                                              int npad, const Cx* __restrict__ ring, bool* sing)
{
    const double* __restrict__ y_arr = reinterpret_cast<const double*>(series + (n - npad + P3L_PAD_RECORDS)) +
                                       (n - npad + P3L_PAD_RECORDS);                                              // y[]
    using Geo = Pipe3LGeom<P>;
    using RA = RowAsm<P>;
    constexpr int C = Geo::C;
    const int lane = g.lane64;
    const int nc = (n + C - 1) / C;
    const double one = 1.0;
    double z = 0.0, h_own = 0.0;                              // h_own: read from LDS once the covariance wave has published it
    LogLikAcc acc;
    acc.init();
    __builtin_amdgcn_s_setprio(CARMA_PRIO_B);                 // see pipe3l_cov
    const double2* ring_b = nullptr;
    const double2* link_b = nullptr;
    double2 hc_n = make_double2(0.0, 0.0), lk_n = make_double2(0.0, 1.0);
    unsigned rowm = 0;
    int j0 = 0;
    auto pass = [&](const int s, const bool more, const bool rebase, const double yj) __attribute__((always_inline)) {
        const double2 hc = hc_n, lk = lk_n;                   // lk = {k~_r, var_j}
        if (more) {
            hc_n = ring_b[(size_t)(s + 1) * Geo::SLOT];
            lk_n = link_b[(size_t)(s + 1) * Geo::SLOT];
        }
        __builtin_amdgcn_sched_barrier(0);
        double ht = hc.x;
        if (__builtin_expect(rebase, 0)) {                                         // z <- A z~ (kfilter.cpp:200-201 for the whole window)
            const bool mine = (rowm >> s) & 1u;
            const double rc_ = mine ? hc.x : 1.0, rs_ = mine ? hc.y : 0.0;
            const double zp = g.partner(z);
            z = fma(rc_, z, -(rs_ * zp));
            ht = mine ? h_own : hc.x;
        }
        // innov_j = (y - mu) - h~.z~   (kfilter.cpp:184, 207, 213); log-likelihood terms (carpack.hpp:167-171)
        double innov;
        RA::innov_t2(innov, yj, mu, z, ht, one);
        acc.add_var(lk.y);
        const double si = recip(lk.y) * innov;
        acc.chi2 += innov * si;
        z = fma(lk.x, si, z);                                 // z~ += k~ innov / var (kfilter.cpp:191-194)
    };
    const unsigned long long* flag_b = reinterpret_cast<const unsigned long long*>(ring + Geo::FLAG_OFF);
    __syncthreads();                                          // (h_r, c_r) published
    __syncthreads();                                          // barrier 0
    h_own = reinterpret_cast<const double2*>(ring + Geo::CONST2_OFF)[Geo::entry(lane)].x;    // g_r h_r: h~ at a re-base datum
    *sing = ((flag_b[3] >> (16 * (lane >> 4))) & 1ull) != 0ull;     // the covariance wave's repeated-root flag of this row
    for (int c = 0; c < nc; c++) {
        __syncthreads();                                      // barrier c + 1: wave A has finished chunk c
        // y_j: two wide scalar loads, requested first (as yerr_j^2 in the covariance wave)
        j0 = c * C;
        const int len = (n - c * C < C) ? n - c * C : C;
        double yv[C];
        if (len == C) {
#pragma unroll
            for (int s = 0; s < C; s++) yv[s] = y_arr[j0 + s];
        }
        __builtin_amdgcn_sched_barrier(0);
        ring_b = reinterpret_cast<const double2*>(ring + Geo::RING_OFF + (size_t)(c % 3) * C * Geo::SLOT) + Geo::entry(lane);
        link_b = reinterpret_cast<const double2*>(ring + Geo::LINK_OFF) + (size_t)(c & 1) * C * Geo::SLOT + Geo::entry(lane);
        hc_n = ring_b[0];
        lk_n = link_b[0];
        const unsigned long long fm64 = flag_b[c % 3];
        rowm = (unsigned)(fm64 >> (16 * (lane >> 4))) & 0xffffu;                   // this row's evaluation
        const unsigned fm_lo = __builtin_amdgcn_readfirstlane((unsigned)fm64), fm_hi = __builtin_amdgcn_readfirstlane((unsigned)(fm64 >> 32));
        const unsigned fm = (fm_lo | (fm_lo >> 16) | fm_hi | (fm_hi >> 16)) & 0xffffu;     // any row (= the finest grid's)
#if defined(CARMA_AB_NOMEAN)                                  // timing-only A/B build: what the mean wave's work costs the others
        if (c > 0) continue;
#endif
        if (len == C) {
            if (fm == 0) {
#pragma unroll
                for (int s = 0; s < C; s++) pass(s, s + 1 < C, false, yv[s]);
            } else {
#pragma unroll
                for (int s = 0; s < C; s++) pass(s, s + 1 < C, (fm >> s) & 1u, yv[s]);
            }
        } else {
            // last, shorter chunk: y_j staged through LDS (see the covariance wave)
            double* tail = const_cast<double*>(reinterpret_cast<const double*>(ring + Geo::TAIL_OFF)) + C;
            const int jt = j0 + (lane & 15);
            const double yz = series[jt < n ? jt : n - 1].y;
            if (lane < C) tail[lane] = yz;
            g.sync();
            double y_n = tail[0];
#pragma unroll 1
            for (int s = 0; s < len; s++) {
                const double yj = readlane_f64(y_n, 0);
                y_n = tail[s + 1 < C ? s + 1 : s];
                pass(s, true, (fm >> s) & 1u, yj);
            }
        }
    }
    return acc.total();
}

}  // namespace carma
