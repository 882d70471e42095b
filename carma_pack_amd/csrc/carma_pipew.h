// carma_pipew.h -- the WINDOWED wave pipeline (round 5, gfx950 only): the co-rotating-frame recursion of carma_pipe3l.h
// applied a CHUNK of data at a time instead of one datum at a time -- a chunk is one symmetric elimination (LDL^T of the
// chunk's predictive covariance; numpy prototype tests/tools/proto/blocked_window.py, loglik_window) -- and with the mean
// recursion riding in the same instructions, so that there is no mean wave and no link ring.
//
// One evaluation per 16-lane DPP row, four per workgroup.  In the recursion wave the lanes of a row are
//     lanes 0 .. ND-1 (ND = 16 - P)   the data of the chunk:  hh[r] = h~_r, kk[r] = would-be gain (S h~ + c~)_r,
//                                     m = would-be variance, nu = would-be innovation            (kfilter.cpp:191, 209-213)
//     lanes ND .. 15                  the P columns of S ("virtual data": hh = e_s, kk[r] = S_rs, nu = -z~_s)
// and pivot j (the datum in lane j, its m and nu now final: var_j, innov_j) does, in every later lane,
//     G = kk@j . hh ;  t = -G / m@j ;  m += G t ;  kk += kk@j t ;  nu += nu@j t                  (x@j: DPP row broadcast)
// which for a data lane is the elimination step of the chunk's covariance and for a virtual lane IS the rank-1 downdate
// of its column of S (kfilter.cpp:197) and the update of z~_s (kfilter.cpp:194): S and z~ never leave the lanes.  The next
// chunk starts from  kk' = c~' + sum_s kk@v_s hh'_s ,  nu' = (y' - mu) + sum_s nu@v_s hh'_s ,  m' = scale yerr'^2 + hh'.kk'
// (s0 = h~.c~ comes in through c~).  19 + 2 P instructions per datum for covariance AND mean (the one-datum pass of
// carma_pipe3l.h: 30 for the covariance alone); measured in isolation 145-153 cycles per datum at P = 5 against 198
// (tools/ubench/gen_ub11.py, profiles/r05/ub11_window_v1.txt).
//
// Frame as in carma_pipe3l.h (window of 2^-ex per evaluation, frame half a window ahead, coordinates rescaled by exact
// powers of two); the RE-BASE is always the first thing of a chunk (S <- A S A^T, z~ <- A z~ on the virtual lanes, then the
// start above): a chunk opens with one when the last datum it could take would leave the window, and only data further
// than a window from the chunk's first datum are cut off (pipew_produce).  Every evaluation has its own chunk schedule
// -- a cut chunk is completed with NEUTRAL slots (h~ = c~ = 0, variance 1, innovation 0: the state does not move, the
// sums get exactly 0) -- so an evaluation's result does not depend on its neighbours in the workgroup; the workgroup
// runs until its slowest row is through.
// STATUS: the default for launches of up to one workgroup per CU (CARMA_TUNE_WIN_ROWS moves the boundary) and for the row
// sampler where the whole ladder's grid is that small: 27.4 us against the one-datum pipeline's 29.5 per 1024 evaluations,
// 32.6 against 31.1 * 10^3 sampler iterations/s at 16 x 64 (profiles/r05/window_pipeline_v1.txt); with more workgroups per
// CU the one-datum pipeline is ahead and keeps those sizes.
//
//   waves P0, P1 (producers)   per chunk and row: schedule, exp / sincos, entries {h~_r, c~_r}, {scale yerr^2, y - mu}
//   wave A (recursion)         [re-base]; start of the chunk; ND pivots; log-likelihood terms of the chunk
//   wave B (set-up)            prior bounds and log prior, then it ends
// One __syncthreads() per chunk: after barrier c chunk c is in ring buffer c % 2.  The recursion wave loads chunk c + 1
// into registers while it eliminates chunk c.
#pragma once
#include <hip/hip_runtime.h>

#include "carma_core.h"
#include "grp_device.h"
#include "carma_pipe3l.h"
#include "carma_win_asm.h"

namespace carma {

#if defined(CARMA_STAMPS)
#define PIPEW_MARK(i) do { if (mk) { __builtin_amdgcn_sched_barrier(0); mk[i] = clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define PIPEW_MARK(i) do { } while (0)
#endif

template <int P>
struct PipeWGeom {
    static constexpr int ND = 16 - P;                         // data lanes of a row
    static constexpr int NB = 2;                              // ring buffers
    static constexpr int ENT = P + 1;                         // double2 per lane: {h~_r, c~_r}, r < P; {scale yerr^2, y - mu}
    static constexpr int RING_OFF = 0;                        // double2[NB][ENT][64]
    static constexpr int ROT_OFF = NB * ENT * 64;             // double2[NB][P][4]: (c_r, s_r) of the re-base rotation, per row
    static constexpr int HDR_OFF = ROT_OFF + NB * P * 4;      // u64[NB]: bits 0-3 rows that re-base, bit 8 last chunk
    static constexpr int CONST_OFF = HDR_OFF + NB;            // double2 {h_r, c_r}[64]
    static constexpr int OUT_OFF = CONST_OFF + 64;            // double2 {log prior, valid}[4] from the set-up wave
    static constexpr int TAB_OFF = OUT_OFF + 4;               // double[MATH_TAB_N]: tables of the short exp / sincos (carma_math.h)
    static constexpr int MRG_OFF = TAB_OFF + MATH_TAB_N / 2;  // two-sided kernels, per evaluation (2): the merge's exchange area (pipew_merge)
    static constexpr int MRG_STRIDE = P + 1 + ((P + 1) & 1);  // (doubles per column: an even count, so that columns are double2 aligned)
    static constexpr int MRG_X = 0;                           //   double[P][ST]  the forward row's columns of Da and -a
    static constexpr int MRG_COL = MRG_X + P * MRG_STRIDE;    //   double[ST]     the pivot column of a step; [P]: 1 / sqrt(pivot)
    static constexpr int MRG_L = MRG_COL + MRG_STRIDE;        //   double[P][ST]  L, row i = lane ND + i's
    static constexpr int MRG_T = MRG_L + P * MRG_STRIDE;      //   double[P][ST]  T = Db L, row i = lane ND + i's; [i][P] = u_i
    static constexpr int MRG_DOUBLES = MRG_T + P * MRG_STRIDE;
    static constexpr int TH_OFF = MRG_OFF + (2 * MRG_DOUBLES + 1) / 2;   // two-sided log-density kernel: double[4 waves][4 rows][16], theta per row
    static constexpr int SCH_OFF = TH_OFF + 128;              // two-sided kernels: double2[NB][4 rows][2], the schedule's hand-over (pipew_produce)
    static constexpr int ENTRIES = SCH_OFF + NB * 4 * 2;
    // two-sided log-density kernel: the series behind everything else, double2 {y, yerr^2}[n] then double t[n]
    static constexpr int SER_OFF = ENTRIES;
    static __host__ __device__ size_t bytes_with_series(int n) { return BYTES + (size_t)(n + (n & 1)) * 24; }
    static constexpr int NPROD = 3;                           // producer waves: P0, P1 and the set-up wave once it is through
    static constexpr size_t BYTES = (size_t)ENTRIES * sizeof(double2);
    static constexpr double LIM_RE = Pipe3LGeom<P>::LIM_RE, LIM_IM = Pipe3LGeom<P>::LIM_IM;
};

// producer waves (pw = 0 .. NPROD - 1).  `tail(pw)` runs once all chunks are produced (the sampler kernel draws there).
// TS (two-sided, round 6): rows 0 / 2 of the workgroup run the recursion FORWARD over the first (n + 1) / 2 data, rows 1 / 3 run it
// BACKWARD over the rest -- the same recursion on the reversed series with h and c exchanged and the rotation sense reversed (the
// state is a stationary Gauss-Markov process: reversed in time it is Markov with transition V F^T V^-1, diagonal again in the dual
// coordinates u = V^-1 z, where the observation vector is c = V h and "V h" is h; tests/tools/proto/two_sided.py) -- and one more
// chunk after the data, the FINAL chunk, carries nothing but the rotation of both states to the meeting time in true
// coordinates (frame factor g included), where pipew_recur merges them.
// SLDS (the two-sided log-density kernel): the series is in LDS -- times lds_t[n_all], {y, yerr^2} lds_yz[n_all], copied by the kernel,
// visible behind the first barrier -- and a chunk's records are read where they are needed; elsewhere each row keeps a window of 64
// records in registers (below).
template <int P, bool TS = false, bool SLDS = false, bool HANDOVER = false, class Tail>
__device__ __forceinline__ void pipew_produce(const Grp<16>& g, int pw, const double* __restrict__ theta,
                                              const double4* __restrict__ series, int n_all, double2* __restrict__ ring, Tail&& tail,
                                              long long* mk = nullptr, const double* __restrict__ lds_t = nullptr,
                                              const double2* __restrict__ lds_yz = nullptr)
{
    using Geo = PipeWGeom<P>;
    constexpr int ND = Geo::ND, NB = Geo::NB, ENT = Geo::ENT;
    if (CARMA_PRIO_P != 0) __builtin_amdgcn_s_setprio(CARMA_PRIO_P);
    // a lane works on a conjugate PAIR of roots (jr, jr + 1) of one slot: both share |E|, cos and sin
    constexpr int NPAIR = (P + 1) / 2, PPL = 16 / NPAIR;
    constexpr int NPROD = Geo::NPROD;
    // which producer forms the data slots {scale yerr^2, y - mu}: in the two-sided log-density kernel the third one -- the set-up wave,
    // which has nothing else in front of the pipeline there -- so that producer 0 (header, the virtual lanes' constant entries) is
    // not the one the others wait for; elsewhere producer 0
    constexpr int PWD = TS ? 2 : 0;
    constexpr int NIT = (ND + 1 + NPROD * PPL - 1) / (NPROD * PPL);   // slot ND is the re-base rotation
    const double* tab = reinterpret_cast<const double*>(ring + Geo::TAB_OFF);
    const int lane = g.lane64, l = lane & 15, rowb = lane & ~15, q = lane >> 4;
    const int sub = l / NPAIR, jr = 2 * (l - sub * NPAIR);
    const bool worker = sub < PPL, two = jr + 1 < P;
    // the part of the series this row works on
    const bool bwd = TS && (q & 1);
    const int nfwd = (n_all + 1) / 2;
    const int n = TS ? (bwd ? n_all - nfwd : nfwd) : n_all;
    // (a backward row walks the series from its end, in the negated time)
    auto recat = [=](int j) {
        const int jj = j < n ? j : n - 1;
        double4 r = series[bwd ? n_all - 1 - jj : jj];
        if (bwd) r.w = -r.w;
        return r;
    };
    // The chunk schedule is data dependent (a chunk ends in front of a re-base datum), so the records of a chunk cannot be
    // requested by index a chunk ahead as in carma_pipe3l.h -- and a global load at the head of every chunk would put an L2
    // round trip on every chunk (measured: 45 us per launch instead of 22).  So each row keeps a WINDOW of 64 records in
    // registers, lane l holding records jw + l, jw + 16 + l, jw + 32 + l, jw + 48 + l; a chunk's sixteen records lie in the
    // first two and are fetched from their lanes (ds_bpermute); when the row has moved past the first sixteen the window
    // shifts and the next sixteen are requested, two shifts before they are needed.
    // (Requested FIRST: nothing in front of the first barrier depends on them, and behind the roots' exponentials their L2 round
    // trip was one more serial latency of the prologue.)
    int jw = 0;
    double4 rw0, rw1, rw2, rw3;
    if constexpr (!SLDS) {
        rw0 = recat(l);
        rw1 = recat(16 + l);
        rw2 = recat(32 + l);
        rw3 = recat(48 + l);
    }
    // SLDS: record j of this row (clamped) -- series index and the sign of its time
    auto sidx = [=](int j) {
        const int jj = j < n ? j : n - 1;
        return bwd ? n_all - 1 - jj : jj;
    };
    const double tsgn = bwd ? -1.0 : 1.0;
    double t_first = 0.0, t_meet = 0.0;
    if constexpr (!SLDS) {
        t_first = recat(0).w;
        t_meet = TS ? (bwd ? -series[nfwd - 1].w : series[nfwd - 1].w) : 0.0;    // the last forward datum's time
    }
    Cx w = own_ar_root<P>(theta, jr);
    Cx w1 = two ? own_ar_root<P>(theta, jr + 1) : w;
    if (bwd) {                                                // F^T instead of F: the conjugate roots
        w.im = -w.im;
        w1.im = -w1.im;
    }
    int esig = 0;                                             // binary exponent of sigma_y (rescaling, carma_pipe3l.h)
    {
        const double sg = fabs(theta[0]);
        (void)frexp(sg, &esig);
        if (!(sg > 0.0 && sg < 1.0 / 0.0)) esig = 0;
    }
    const double mu = theta[2], scale = theta[1];
    const bool realpair = two && w.im == 0.0;
    // the virtual lanes' entries never change: hh = e_s, c~ = 0, {1, 0} in the data slot
    if (pw == 0 && l >= ND) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
#pragma unroll
            for (int r = 0; r < P; r++) ring[Geo::RING_OFF + (b * ENT + r) * 64 + lane] = make_double2(r == l - ND ? 1.0 : 0.0, 0.0);
            ring[Geo::RING_OFF + (b * ENT + P) * 64 + lane] = make_double2(1.0, 0.0);
        }
    }
    // grid of this evaluation's re-base schedule (carma_pipe3l.h): cells of width 2^-ex
    double wl = fmax(fmax(fabs(w.re), fabs(w1.re)) * (1.0 / Geo::LIM_RE), fabs(w.im) * (1.0 / Geo::LIM_IM));
    wl = (wl < 1e12) ? wl : ((wl == wl && wl < 1.0 / 0.0) ? 1e12 : 0.0);
    wl = Grp<16>::max(wl);
    int wex;
    (void)frexp(wl, &wex);
    const double sc = wl > 0.0 ? ldexp(1.0, wex) : 0.0;
    const double halfw = sc > 0.0 ? 0.5 / sc : 0.0;
    const double rg0 = exp_neg(w.re * halfw), g0 = recip(rg0);
    double rg1 = rg0, g1 = g0;
    if (realpair) {
        rg1 = exp_neg(w1.re * halfw);
        g1 = recip(rg1);
    }
    // schedule state of this row (row-uniform)
    int j0 = 0;
    double base = t_first;                                    // (SLDS: read behind the first barrier, below)
    // ---- A chunk in three parts: its schedule, its exponentials (neither needs h, c) and the entries.  (Round 6 tried schedule and
    // exponentials of chunk 0 in front of the first barrier and every later one's behind the barrier of the chunk before: the producers
    // start up to 1.4 k cycles after the recursion wave and reach that barrier no earlier than it does, so nothing was hidden --
    // 18.55 against 18.51 us per launch, the row sampler 31.7 against 32.4 k it/s on one box, profiles/r06/ab_rotation_v1.txt.)
    // SSCHED (the two-sided kernels, series in LDS, instantiated with HANDOVER where workgroups SHARE a CU): the schedule of chunk c + 1 is formed by ONE
    // producer -- the one that also forms the data slots -- while chunk c is in the making, and reaches the other two through LDS
    // behind chunk c's barrier (chunk 0's by everyone); three copies of it were 225 of a workgroup's 780 producer instructions per
    // chunk, and with two or three workgroups on a CU its VALU issue slots are what the launch runs out of: 24.2 -> 22.3 us per 1024
    // evaluations, 29.6 -> 27.8 per 1536, the row sampler at 16 x 64 39.6 -> 41.5 k it/s.  With a CU to itself a workgroup waits for
    // its slowest producer instead, and the scheduling one is then longer than the three equal ones were (17.8 -> 18.5 us; as a RUN-TIME
    // switch the code alone cost that much): not there, a template parameter (profiles/r06/ab_ssched_v1.txt).
    constexpr bool SSCHED = TS && SLDS && HANDOVER;
    int len = 0, jc = 0;                                      // jc: first datum of the chunk
    bool rot = false, last = false, fin = false;              // fin (TS): the final chunk (no data, the rotation to the meeting time)
    double dt_rot = 0.0, dta_keep = 0.0;                      // time from the closing window's base to the chunk's first datum
    double2 dslot = make_double2(1.0, 0.0);                   // {scale yerr^2, y - mu} of this lane's datum (producer PWD)
    unsigned long long rbits = 0ull;
    double ecv[NIT], esv[NIT], e1v[NIT];
    auto sched = [&](bool finflag) __attribute__((always_inline)) {
        // --- schedule of this row's chunk: lane s looks at datum j0 + s
        jc = j0;
        const int jl = j0 + l;
        double4 rec;
        if constexpr (SLDS) {
            const int si = sidx(jl);
            rec.x = rec.y = rec.z = 0.0;
            rec.w = tsgn * lds_t[si];
            if (pw == PWD) {                                  // (wave-uniform: the data slots are ONE producer's)
                const double2 yz = lds_yz[si];
                rec.y = yz.x;
                rec.z = yz.y;
            }
        } else {
            const int pos = jl - jw, src = rowb + (pos & 15);
            const bool hi = pos >= 16;
            // (both fetches by every lane: a ds_bpermute under a divergent branch would read inactive lanes)
            const double w0 = __shfl(rw0.w, src, 64), w1_ = __shfl(rw1.w, src, 64);
            rec.x = rec.y = rec.z = 0.0;
            rec.w = hi ? w1_ : w0;
            if (pw == PWD) {                                  // (wave-uniform: the data slots are ONE producer's)
                const double y0 = __shfl(rw0.y, src, 64), y1 = __shfl(rw1.y, src, 64);
                const double z0 = __shfl(rw0.z, src, 64), z1 = __shfl(rw1.z, src, 64);
                rec.y = hi ? y1 : y0;
                rec.z = hi ? z1 : z0;
            }
        }
        const double tj = rec.w;
        // RE-BASE RULE (round 5, second version): look a chunk ahead.  The frame of a window carries scale factors e^(+-LIM_RE / 2)
        // as long as a datum is no further than W = 2^-ex from the window's base.  If the LAST datum this chunk could take is
        // beyond W, the chunk OPENS with a re-base at its first datum (rotation over the time since the old base, any length:
        // a decayed coordinate underflows to 0) -- and only data further than W from THAT are cut off (a gap inside the chunk).
        // The first version ended a chunk in front of every cell boundary of a dyadic time grid, as the one-datum pipeline
        // re-bases: a row with a short window then ran 36 chunks where its neighbours ran 25, and the launch waits for the
        // slowest row (34.1 us per 1024 evaluations, profiles/r05/window_pipeline_v1.txt).
        const int left = n - j0;
        const int ncand = left < ND ? (left > 0 ? left : 0) : ND;
        const double t0 = finflag ? t_meet : Grp<16>::template bcast_c<0>(tj);
        const double t_lastc = SLDS ? tsgn * lds_t[sidx(j0 + (ncand > 0 ? ncand - 1 : 0))] : __shfl(tj, rowb + (ncand > 0 ? ncand - 1 : 0), 64);
        const double W = sc > 0.0 ? 1.0 / sc : 1.0 / 0.0;
        rot = finflag || (ncand > 0 && (t_lastc - base) > W);
        dt_rot = t0 - base;
        base = rot ? t0 : base;
        // (the difference of two time stamps is exact unless the base is much the smaller of the two: carma_pipe3l.h)
        const double dta_l = tj - base;
        dta_keep = dta_l;
        const unsigned long long flb = __ballot(dta_l > W && l < ncand);
        const unsigned cut = (unsigned)(flb >> (16 * q)) & 0xffffu;            // (bit 0 never: dta = 0 or <= W at slot 0)
        len = cut ? __builtin_ctz(cut) : ncand;
        const bool row_done = j0 + len >= n;
        last = __ballot(!row_done) == 0ull;
        if (pw == PWD) dslot = l < len ? make_double2(rec.z * scale, rec.y - mu) : make_double2(1.0, 0.0);
        j0 += len;
        if (!SLDS && j0 - jw >= 16) {                         // (at most one shift per chunk: len < 16)
            rw0 = rw1;
            rw1 = rw2;
            rw2 = rw3;
            jw += 16;
            rw3 = recat(jw + 48 + l);
        }
    };
    // the schedule's hand-over (SSCHED): per ring buffer and row {(first datum | len << 16 | rot << 24 | last << 25), new base}, {dt_rot, -}
    double2* schb = ring + Geo::SCH_OFF;
    auto sched_put = [&](int b) __attribute__((always_inline)) {
        if (l == 0) {
            const int bits = jc | (len << 16) | (rot ? 1 << 24 : 0) | (last ? 1 << 25 : 0);
            schb[(b * 4 + q) * 2] = make_double2(__hiloint2double(0, bits), base);
            schb[(b * 4 + q) * 2 + 1] = make_double2(dt_rot, 0.0);
        }
    };
    auto sched_get = [&](int b) __attribute__((always_inline)) {
        const double2 s0 = schb[(b * 4 + q) * 2], s1 = schb[(b * 4 + q) * 2 + 1];
        const int bits = __double2loint(s0.x);
        jc = bits & 0xffff;
        len = (bits >> 16) & 0xff;
        rot = ((bits >> 24) & 1) != 0;
        last = ((bits >> 25) & 1) != 0;
        base = s0.y;
        dt_rot = s1.x;
    };
    auto exps = [&]() __attribute__((always_inline)) {
        if (pw == 0) rbits = __ballot(rot);
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int slot = it * NPROD * PPL + pw * PPL + sub;
            const bool is_rot = slot == ND;
            double dts;
            if constexpr (SLDS) {
                dts = tsgn * lds_t[sidx(jc + (slot < ND ? slot : 0))] - base;
            } else {
                dts = __shfl(dta_keep, rowb + (slot < ND ? slot : 0), 64);    // (lane `slot` of the row looked at that datum in sched())
            }
            const double dt = is_rot ? dt_rot : dts;
            const bool live = worker && (is_rot ? rot : slot < len);
            double ec = 1.0, es = 0.0, e1 = 1.0;
            if (live) cexp_step_tab<true>(w.re, w.im, dt, &ec, &es, tab);
            e1 = ec;
            if (realpair && live) e1 = exp_neg_tab(w1.re * dt, tab);
            ecv[it] = ec;
            esv[it] = es;
            e1v[it] = e1;
        }
    };
    PIPEW_MARK(1);
    __syncthreads();                                          // the recursion wave has published (h_r, c_r)
    PIPEW_MARK(2);
    if constexpr (SLDS) {
        base = tsgn * lds_t[sidx(0)];
        t_meet = tsgn * lds_t[nfwd - 1];
    }
    double2 hc_own, hc_par;
    {
        const double2* cst = ring + Geo::CONST_OFF + rowb;
        hc_own = cst[jr];
        hc_par = cst[jr + 1 < 16 ? jr + 1 : 15];
        // every coordinate rescaled by an exact power of two so that |h_r| ~ sigma_y (carma_pipe3l.h)
        const double m0 = fmax(fabs(hc_own.x), realpair ? 0.0 : fabs(hc_par.x)), m1 = realpair ? fabs(hc_par.x) : m0;
        int e0, e1x;
        (void)frexp(m0, &e0);
        (void)frexp(m1, &e1x);
        if (!(m0 > 0.0 && m0 < 1.0 / 0.0)) e0 = 0;
        if (!(m1 > 0.0 && m1 < 1.0 / 0.0)) e1x = 0;
        hc_own = make_double2(ldexp(hc_own.x, esig - e0), ldexp(hc_own.y, e0 - esig));
        hc_par = make_double2(ldexp(hc_par.x, esig - e1x), ldexp(hc_par.y, e1x - esig));
        if (bwd) {                                            // dual coordinates: h' = c, c' = V^-1 c = h (the same powers of two)
            hc_own = make_double2(hc_own.y, hc_own.x);
            hc_par = make_double2(hc_par.y, hc_par.x);
        }
    }
#if defined(CARMA_WIN_STAMPS)
    unsigned long long ps_work = 0, ps_wait = 0, ps_t0 = 0, ps_t1 = 0;
    int ps_n = 0;
#define WIN_STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
#endif
    for (int c = 0;; c++) {
        const int b = c % NB;
#if defined(CARMA_WIN_STAMPS)
        WIN_STAMP(ps_t0);
#endif
        // --- schedule (chunk 0, or no hand-over: everyone's; else the scheduling producer has it from the chunk before) and exponentials
        if (!SSCHED || c == 0)
            sched(fin);
        else if (pw != PWD)
            sched_get(b);
        exps();
        // --- the chunk's header, data slots and entries
        if (pw == 0 && lane == 0)
            reinterpret_cast<unsigned long long*>(ring + Geo::HDR_OFF)[b] =
                ((rbits & 1ull) | ((rbits >> 15) & 2ull) | ((rbits >> 30) & 4ull) | ((rbits >> 45) & 8ull)) | ((TS ? fin : last) ? 256ull : 0ull);
        if (pw == PWD && l < ND) ring[Geo::RING_OFF + (b * ENT + P) * 64 + lane] = dslot;
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int slot = it * NPROD * PPL + pw * PPL + sub;
            const bool is_rot = slot == ND;
            const bool live = worker && (is_rot ? rot : slot < len);
            const double ec = ecv[it], es = esv[it], e1 = e1v[it];
            if (worker && slot < ND) {
                // h~_r = (A^T h)_r = E (cos h_r + sin h_partner) ;  c~_r = (A^-1 c)_r = (cos c_r + sin c_partner) / E
                const double gc = ec * g0, gs = es * g0;
                const double inv = recip(fma(gc, gc, gs * gs));
                const double ht = fma(gc, hc_own.x, gs * hc_par.x);
                const double ct = fma(gc, hc_own.y, gs * hc_par.y) * inv;
                double2* dst = ring + Geo::RING_OFF + (b * ENT + jr) * 64 + rowb + slot;
                dst[0] = live ? make_double2(ht, ct) : make_double2(0.0, 0.0);
                if (two) {
                    const double gc1 = e1 * g1, gs1 = es * g1;
                    const double inv1 = realpair ? recip(gc1 * gc1) : inv;
                    const double hp = fma(gc1, hc_par.x, -gs1 * hc_own.x);
                    const double cp = fma(gc1, hc_par.y, -gs1 * hc_own.y) * inv1;
                    dst[64] = live ? make_double2(hp, cp) : make_double2(0.0, 0.0);
                }
            }
            if (worker && is_rot) {                           // rotation over the closing window (identity when there is none)
                double2* dst = ring + Geo::ROT_OFF + (b * P + jr) * 4 + q;
                // (the final chunk leaves the frame: its rotation carries the frame factor, S -> the true D = (G A) S (G A)^T)
                const double f0 = fin ? g0 : 1.0, f1 = fin ? g1 : 1.0;
                dst[0] = make_double2(ec * f0, es * f0);
                if (two) dst[4] = make_double2(e1 * f1, -es * f1);
            }
        }
#if defined(CARMA_WIN_STAMPS)
        WIN_STAMP(ps_t1);
        ps_work += ps_t1 - ps_t0;
#endif
        const bool last_c = last, fin_c = fin;
        if (SSCHED && pw == PWD && !(TS ? fin_c : last_c)) {  // the next chunk's schedule, for everyone
            sched(last_c);
            sched_put((c + 1) % NB);
        }
        if (c < 3) PIPEW_MARK(3 + 2 * c - (c == 2));           // chunk 0, 1: arrival at the barrier (marks 3, 5); chunk 2: mark 6
        __syncthreads();                                      // barrier c: chunk c is in the ring
        if (c == 0) PIPEW_MARK(4);
#if defined(CARMA_WIN_STAMPS)
        WIN_STAMP(ps_t0);
        ps_wait += ps_t0 - ps_t1;
        ps_n++;
#endif
        if (TS ? fin_c : last_c) break;
        fin = last_c;
    }
#if defined(CARMA_WIN_STAMPS)
    if (blockIdx.x == 0 && lane == 0)
        printf("window pipeline, producer %d: %d chunks, per chunk %llu cycles of work, %llu at the barrier\n", pw, ps_n,
               ps_work / (unsigned long long)ps_n, ps_wait / (unsigned long long)ps_n);
#endif
    tail(pw);
    // the recursion wave's barrier in front of its last chunk (TS: that barrier was the final chunk's)
    if (!TS) __syncthreads();
}

// The merge of a two-sided evaluation (rows 2k: forward over the first half, 2k + 1: backward over the second half of the series).
// After the final chunk's start the virtual lanes of the forward row hold the columns of Da and -a  (z_m | first half ~ N(a, V + Da)),
// those of the backward row the columns of Db and -beta  (u_m = V^-1 z_m | second half ~ N(beta, V^-1 + Db)).  With X = -Da, Y = -Db
// (both positive semidefinite; alpha ~ N(0, X), beta ~ N(0, Y), E[alpha beta^T] = Da Db, sufficient statistics of the two halves)
//     log p(y) = l_a + l_b - 1/2 log det W + beta.a - 1/2 a.Y a - 1/2 |C^-1 L^T (Y a - beta)|^2 ,   X = L L^T,  W = I - L^T Y L = C C^T
// (tests/tools/proto/two_sided.py: merge_chol / lane_merge_chol; V does not appear, and the diagonal rescaling of the coordinates by
// powers of two -- forward z / 2^e, backward u 2^e -- cancels in every term).  Why not simply N = I - Da Db and an LU (the first
// version): in modal coordinates of nearly coincident roots X has entries ~ 1 / separation^2 that cancel in the product, and N loses
// what the recursions kept (roots 1e-6 apart: 1.5e-3 against the one-pass filter's 3e-6, this form 6e-9; two real roots 6e-4 apart,
// inside the prior's bounds: 3e-5 against 6e-8).  Why X = L L^T with DIAGONAL PIVOTING: X is numerically rank deficient as a rule -- a
// half of the series says nothing about the modes that have decayed by the meeting time, and the two coordinates of a pair can carry
// one direction only -- and in any fixed order a pivot at rounding level met before an informative one grows into it (pivots
// 1, 0.93, 1e-13, 1e-2, 1e-2 in root order against 1, 0.93, 0.1, 1e-2, 1e-14 pivoted: errors up to 7e-4 on the parity sweeps'
// prior-like entries, profiles/r06/merge_pivots_v1.txt); with the largest remaining diagonal as the pivot whatever is left when
// the pivots reach rounding level is at that level too, and is dropped.
// Lanes ND + j of the BACKWARD row hold column j throughout.  The pivot of a step is found by a 4-step DPP maximum over a 32-bit
// key (upper half of the diagonal, lane number in the low bits), its column goes through LDS -- the one place where a register
// index would be a run-time value --, the update X_ij -= l_i l_j takes both factors from that column, so that the Schur complement
// stays symmetric bit for bit.  T = Db L by broadcasts, W and the border v = L^T (Y a - beta) through LDS again (lane m takes
// column m: pivot order, not coordinate order), W = C C^T without pivoting (positive definite).  The pivots of W go into the row's
// accumulators as "variances", the quadratic terms as chi^2.
CARMA_DEV double rsqrt_pos(double d)
{
#ifdef __HIPCC__
    const double y0 = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y0, y0, 1.0);                   // 1 - d y0^2
    return fma(y0, fma(0.375 * e, e, 0.5 * e), y0);           // one cubic step: e^3 ~ 1e-22
#else
    return 1.0 / sqrt(d);
#endif
}
// (release by the writing lanes, acquire by the reading ones: the LDS operations of ONE wave execute in order, but without the pair
// the compiler moves the other rows' loads above the stores -- they are different threads to it)
CARMA_DEV void merge_lds_sync()
{
#ifdef __HIPCC__
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
template <int P>
__device__ __forceinline__ void pipew_merge(int lane, const double (&kf)[P], double nu, double2* __restrict__ ring, LogLikAcc& acc,
                                            long long* mk = nullptr)
{
    using Geo = PipeWGeom<P>;
    constexpr int ND = Geo::ND, ST = Geo::MRG_STRIDE;
    const int l = lane & 15, q = lane >> 4;
    const bool bwd = (q & 1) != 0, virt = l >= ND;
    const int j = virt ? l - ND : 0;                          // this lane's column (the other lanes' results are not used)
    double* mrg = reinterpret_cast<double*>(ring + Geo::MRG_OFF) + (q >> 1) * Geo::MRG_DOUBLES;
    // forward row: column s of Da and -a_s to LDS
    if (!bwd && virt) {
        double* dst = mrg + Geo::MRG_X + j * ST;
#pragma unroll
        for (int r = 0; r < P; r++) dst[r] = kf[r];
        dst[P] = nu;
    }
    merge_lds_sync();
    // (nu holds the NEGATED means on both sides: every term below is bilinear in (a, beta) jointly, so "a" and "beta" are simply
    // the registers' values)
    double S[P], av[P];
#pragma unroll
    for (int i = 0; i < P; i++) {
        // X_ij from ONE of the two stored elements (i, j), (j, i): the recursion's S is symmetric up to rounding only, and the
        // elimination below must see a matrix that is symmetric bit for bit
        S[i] = -mrg[Geo::MRG_X + (i >= j ? j * ST + i : i * ST + j)];
        av[i] = mrg[Geo::MRG_X + i * ST + P];                 // a_i
    }
    double aj = mrg[Geo::MRG_X + j * ST + P];
    // Equilibration by exact powers of two, X <- D X D, Y <- D^-1 Y D^-1, a <- D a, beta <- D^-1 beta with D_kk ~ 1 / sqrt(X_kk): every
    // term below is invariant, and "a pivot at rounding level" gets a meaning (the diagonal of X is in [1, 4) wherever it is positive).
    // The recursions' coordinates are scaled for THEIR sums (carma_pipe3l.h): a coordinate can sit at 1e-300 in X and 1e+300 in Y.
    int sj = 0;
    double dg = -mrg[Geo::MRG_X + j * ST + j];                // own diagonal, kept beside S[] (where its index is the lane's number)
    {
        int ex;
        (void)frexp(dg, &ex);
        if (dg > 0.0 && dg < 1.0 / 0.0) sj = -(ex >> 1);
    }
    dg = ldexp(dg, 2 * sj);
    double kfs[P];
    static_for<0, P>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        const int si = Grp<16>::template bcast_c<ND + i>(sj);
        S[i] = ldexp(S[i], si + sj);
        kfs[i] = ldexp(kf[i], -si - sj);
        av[i] = ldexp(av[i], si);
    });
    aj = ldexp(aj, sj);
    nu = ldexp(nu, -sj);
    PIPEW_MARK(8);
    // ---- X = L L^T with diagonal pivoting.  Lane j keeps ROW j of L: Lr[m] = L_jm, m the step.
    double Lr[P];
#pragma unroll
    for (int i = 0; i < P; i++) Lr[i] = 0.0;
    bool done = !virt;
    // (P copies of the step, on purpose: as a rolled loop -- nothing in a step depends on its number but the register L's entry goes
    // to -- it took 3.3 k cycles instead of 2.6 k; the step is a chain of dependent latencies, not instruction fetch:
    // profiles/r06/merge_variants_v1.txt)
    bool exhausted = false;
    static_for<0, P>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        if (exhausted) return;
        // the largest remaining diagonal (a key of its upper 28 bits and the lane: the lowest lane wins a tie); nothing above
        // rounding level left: no pivot, a zero column.  Every candidate takes 1 / sqrt of its own diagonal meanwhile -- off the
        // step's serial chain -- and the pivot's lane sends its column already divided: l = (pivot column) / sqrt(pivot)
        const bool cand = !done && dg > 4e-15;
        unsigned key = cand ? (((unsigned)__double2hiint(dg) & ~0xFu) | (unsigned)(15 - l)) : 0u;
        const double r_own = cand ? rsqrt_pos(dg) : 0.0;
#ifdef __HIPCC__
        key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, DPP_QUAD_XOR1, 0xf, 0xf, true));
        key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, DPP_QUAD_XOR2, 0xf, 0xf, true));
        key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, DPP_ROW_HALF_MIRROR, 0xf, 0xf, true));
        key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, DPP_ROW_MIRROR, 0xf, 0xf, true));
#endif
        const bool any = key != 0u;
        // (no row of the wave has anything left above rounding level -- the numerical rank of X is below P as a rule --: the
        // remaining steps are zero columns everywhere)
        if (__builtin_amdgcn_ballot_w64(any) == 0ull) {
            exhausted = true;
            return;
        }
        const int pl = 15 - (int)(key & 15u);                 // the pivot's lane in the row (row-uniform)
        const bool isp = any && l == pl;
        // the pivot's column, already divided, from the pivot's lane: ds_bpermute (the lane is a run-time value; every lane offers its
        // own column times its own 1 / sqrt -- zero where it is no candidate).  Through LDS -- the pivot's lane writes, everybody reads --
        // a step took 510 cycles, the write -> read round trip on its serial chain.
        const int src = (lane & ~15) + pl;
        double cb[P];
#pragma unroll
        for (int i = 0; i < P; i++) cb[i] = __shfl(S[i] * r_own, src, 64);
        // X_ij -= l_i l_j with l_j = X_j,pivot / sqrt(pivot) from the lane's own row entry (the same bits as the pivot column's j-th
        // entry: the Schur complement is symmetric bit for bit) -- picked while the column is on its way
        const double r_piv = __shfl(r_own, src, 64);
        const int pc = pl - ND;
        double sp = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) sp = pc == i ? S[i] : sp;
        const double lj = (any && !done) ? sp * r_piv : 0.0;  // (the pivot's own: sqrt(pivot))
        Lr[m] = lj;
#pragma unroll
        for (int i = 0; i < P; i++) S[i] = fma(-cb[i], lj, S[i]);
        dg = fma(-lj, lj, dg);
        done = done || isp;
    });
    PIPEW_MARK(9);
    // ---- T = Db L  (= -Y L):  T_im = sum_k Db_ik L_km, row i in lane i (Db_ik = this lane's kfs[k]: symmetric), L_km by broadcast
    double T[P];
#pragma unroll
    for (int m = 0; m < P; m++) T[m] = 0.0;
    static_for<0, P>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
#pragma unroll
        for (int m = 0; m < P; m++) T[m] = fma(kfs[k], Grp<16>::template bcast_c<ND + k>(Lr[m]), T[m]);
    });
    // u = Y a - beta (lane j: u_j, from its own column of Db)
    double dba = 0.0;
#pragma unroll
    for (int k = 0; k < P; k++) dba = fma(kfs[k], av[k], dba);                 // (Db a)_j
    const double uj = -nu - dba;
    PIPEW_MARK(10);
    // ---- W = I - L^T Y L = I + L^T T and the border v = L^T u, column m to lane m (pivot order) through LDS
    if (bwd && virt) {
        double* lb = mrg + Geo::MRG_L + j * ST;
        double* tb = mrg + Geo::MRG_T + j * ST;
#pragma unroll
        for (int m = 0; m < P; m++) {
            lb[m] = Lr[m];
            tb[m] = T[m];
        }
        tb[P] = uj;
    }
    merge_lds_sync();
    double Wc[P + 1];
    {
        double tm[P], lm[P], v = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) {
            tm[i] = mrg[Geo::MRG_T + i * ST + j];             // T_ij: row i, this lane's column
            lm[i] = mrg[Geo::MRG_L + i * ST + j];             // L_ij
            v = fma(lm[i], mrg[Geo::MRG_T + i * ST + P], v);  // v_j = sum_i L_ij u_i
        }
#pragma unroll
        for (int k = 0; k < P; k++) {
            double w = j == k ? 1.0 : 0.0;
#pragma unroll
            for (int i = 0; i < P; i++) w = fma(mrg[Geo::MRG_L + i * ST + k], tm[i], w);      // W_kj = delta + sum_i L_ik T_ij
            Wc[k] = w;
        }
        Wc[P] = v;
    }
    PIPEW_MARK(11);
    // ---- W = C C^T as L D L^T with the border row riding along: s_k^2 = (border entry of column k)^2 / d_k
    double piv = 1.0, s2 = 0.0;
    static_for<0, P>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        const double dk = Grp<16>::template bcast_c<ND + k>(Wc[k]);
        const double rk = recip(dk);
        if (l == ND + k) {
            piv = Wc[k];
            s2 = Wc[P] * Wc[P] * rk;
        }
        const double f = l > ND + k ? Wc[k] * rk : 0.0;
#pragma unroll
        for (int i = k + 1; i < P + 1; i++) Wc[i] = fma(-Grp<16>::template bcast_c<ND + k>(Wc[i]), f, Wc[i]);
    });
#if defined(CARMA_MERGE_DEBUG)
    if (blockIdx.x == 0 && q < 2 && virt) {
        printf("merge %s lane %d: in kf %.17g %.17g %.17g nu %.17g | sj %d Lr %.6g %.6g %.6g | T %.6g %.6g %.6g | W %.6g %.6g %.6g border %.6g | piv %.17g s2 %.6g dba %.6g aj %.6g\n",
               bwd ? "bwd" : "fwd", j, kf[0], kf[1], kf[P - 1], nu, sj, Lr[0], Lr[1], Lr[P - 1], T[0], T[1], T[P - 1], Wc[0], Wc[1], Wc[P - 1], Wc[P], piv, s2, dba, aj);
    }
#endif
    if (bwd && virt) {
        acc.add_var(piv);                                                       // -1/2 log det W
        // beta.a - 1/2 a.Y a = sum_j a_j (beta_j + 1/2 (Db a)_j);  - 1/2 s^2;   l += Q  <=>  chi2 -= 2 Q
        acc.chi2 += s2 - 2.0 * aj * fma(0.5, dba, nu);
    }
}

template <int P, bool TS = false>
__device__ __forceinline__ double pipew_recur(const Grp<16>& g, const RowConsts<P>& rc, double2* __restrict__ ring, long long* mk = nullptr)
{
    using Geo = PipeWGeom<P>;
    using WA = WinAsm<P>;
    constexpr int ND = Geo::ND, NB = Geo::NB, ENT = Geo::ENT;
    const int lane = g.lane64, l = lane & 15, q = lane >> 4;
    __builtin_amdgcn_s_setprio(CARMA_PRIO_A);
    const bool act = l < P;
    ring[Geo::CONST_OFF + lane] = make_double2(act ? rc.h_own : 0.0, act ? rc.c_own : 0.0);
    __syncthreads();                                          // the producers take the constants from here
    // two register sets for (kk, hh): the start of chunk c + 1 reads the columns of S out of chunk c's set while it writes the
    // other one, so the sets alternate and no copy is needed (the loop below is unrolled by two)
    double ka[P], ha[P], kb[P], hb[P];
#pragma unroll
    for (int r = 0; r < P; r++) ka[r] = ha[r] = kb[r] = hb[r] = 0.0;
    double mA = 1.0, mB = 1.0, nuA = 0.0, nuB = 0.0;
    LogLikAcc acc;
    acc.init();
    const bool data = l < ND, evn = (l & 1) == 0;
    double2 en[ENT];
    unsigned long long hdr = 0;
    auto load = [&](int b, double2(&e)[ENT], unsigned long long& h) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < ENT; r++) e[r] = ring[Geo::RING_OFF + (b * ENT + r) * 64 + lane];
        h = reinterpret_cast<const unsigned long long*>(ring + Geo::HDR_OFF)[b];
    };
    // start of a chunk (into kn, hn) from the columns of S in the virtual lanes of kold (all zero in front of the first chunk)
    auto start = [&](int b, const double(&kold)[P], double(&kn)[P], double(&hn)[P]) __attribute__((always_inline)) {
        const double nuF = (ND & 1) ? nuB : nuA;              // virtual lanes: what the last pivot wrote
        double nun = en[P].y, mn = en[P].x;
        double cn[P];
#pragma unroll
        for (int r = 0; r < P; r++) {
            hn[r] = en[r].x;
            cn[r] = en[r].y;
        }
        if (__builtin_expect((hdr & 0xfull) != 0ull, 0)) {
            // RE-BASE: S <- A S A^T, z~ <- A z~ with the rotation A accumulated over the closing window (kfilter.cpp:200-204 for the
            // whole window; a row without a re-base of its own gets the identity).  S is not rotated where it sits: the start of the
            // chunk needs S_new h~ = A (S_old (A^T h~)) only, so every lane rotates ITS OWN h~ by A^T (in the lane: the two members
            // of a pair are two registers), takes S_old (A^T h~) through the same broadcast-FMAs as ever, and rotates the result by A
            // -- for a virtual lane (h~ = e_s) that IS column s of A S A^T, and its nu' = (A^T e_s) . nu_old = -(A z~)_s.  No
            // cross-lane traffic beyond the broadcasts the start has anyway (the first version mixed the columns of S between the
            // virtual lanes through ds_bpermute: 330-600 cycles at every third chunk start of a row with a short window).
            double cr[P], sr[P], hx[P];
            const bool mine = ((hdr >> q) & 1ull) != 0ull;
#pragma unroll
            for (int r = 0; r < P; r++) {
                const double2 v = ring[Geo::ROT_OFF + (b * P + r) * 4 + q];
                cr[r] = v.x;
                sr[r] = (r ^ 1) < P ? v.y : 0.0;                  // (an odd order's last root is real)
            }
#pragma unroll
            for (int r = 0; r < P; r++) {
                const int rp = (r ^ 1) < P ? (r ^ 1) : r;
                hx[r] = fma(cr[r], hn[r], sr[r] * hn[rp]);        // (A^T h~)_r = c_r h~_r + s_r h~_partner
                // (a row without a re-base of its own has c = 1, s = 0: every operation below is then exact, and its
                // accumulation starts from c~ as in the other branch -- the same bits whatever its neighbours do)
                kn[r] = mine ? 0.0 : cn[r];
            }
            WA::init(kn, nun, kold, nuF, hx);
            double t[P];
#pragma unroll
            for (int r = 0; r < P; r++) t[r] = kn[r];
#pragma unroll
            for (int r = 0; r < P; r++) {
                const int rp = (r ^ 1) < P ? (r ^ 1) : r;
                kn[r] = fma(cr[r], t[r], -(sr[r] * t[rp])) + (mine ? cn[r] : 0.0);   // (A .)_r = c_r x_r - s_r x_partner, + c~
            }
        } else {
#pragma unroll
            for (int r = 0; r < P; r++) kn[r] = cn[r];
            WA::init(kn, nun, kold, nuF, hn);
        }
        // var' = scale yerr^2 + h~ . k~'  (h~ . c~ = s0; kfilter.cpp:209-210)
#pragma unroll
        for (int r = 0; r < P; r++) mn = fma(hn[r], kn[r], mn);
        mA = mB = mn;
        nuA = nuB = nun;
    };
    // one chunk: the pivots, the log-likelihood terms, the start of the next chunk.  Returns true behind the last chunk (TS: when
    // the chunk it has just started is the final one, whose start is all there is to it).
    auto chunk = [&](double(&kk)[P], double(&hh)[P], double(&kn)[P], double(&hn)[P], int c) __attribute__((always_inline)) -> bool {
        const bool last = !TS && (hdr & 256ull) != 0ull;
        double2 en2[ENT];
        unsigned long long hdr2 = 0;
        // barrier c + 1 (chunk c + 1 is in the ring) and its loads: in FRONT of the pivots, so that the loads have a chunk to
        // arrive -- except for chunk 0, whose pivots need not wait for the production of chunk 1
        if (c > 0) {
            __syncthreads();
            if (!last) load((c + 1) % NB, en2, hdr2);
        }
        __builtin_amdgcn_sched_barrier(0);
        WA::chunk(kk, hh, mA, mB, nuA, nuB);
        __builtin_amdgcn_sched_barrier(0);
        if (c == 0) {
            __syncthreads();
            if (!last) load((c + 1) % NB, en2, hdr2);
        }
        // log-likelihood terms of the chunk (carpack.hpp:167-171): a data lane's variance and innovation are in the register its
        // own pivot read
        {
            const double varF = data ? (evn ? mA : mB) : 1.0, innF = data ? (evn ? nuA : nuB) : 0.0;
            acc.add_var(varF);
            acc.chi2 += innF * (recip(varF) * innF);
        }
        if (last) return true;
#pragma unroll
        for (int r = 0; r < ENT; r++) en[r] = en2[r];
        hdr = hdr2;
        start((c + 1) % NB, kk, kn, hn);
        return TS && (hdr & 256ull) != 0ull;
    };
    __syncthreads();                                          // barrier 0
    PIPEW_MARK(3);
    load(0, en, hdr);
    start(0, ka, kb, hb);
    PIPEW_MARK(4);
    bool fin_in_a = true;                                     // which register set the final chunk's start wrote (wave-uniform)
    for (int c = 0;; c += 2) {
        if (chunk(kb, hb, ka, ha, c)) break;
        if (chunk(ka, ha, kb, hb, c + 1)) {
            fin_in_a = false;
            break;
        }
    }
    PIPEW_MARK(5);
    if constexpr (TS) {
        // (ONE call site: the merge runs once per launch out of a cold instruction cache, and two copies of it were 11 KB of code)
        double kf[P];
#pragma unroll
        for (int r = 0; r < P; r++) kf[r] = fin_in_a ? ka[r] : kb[r];
        pipew_merge<P>(lane, kf, nuA, ring, acc, mk);
    }
    PIPEW_MARK(6);
    double ll = Grp<16>::sum(acc.total());
    if constexpr (TS) ll += __shfl_xor(ll, 16, 64);           // forward + backward row (+ the merge, in the backward row's sums)
    return ll;
}

}  // namespace carma
