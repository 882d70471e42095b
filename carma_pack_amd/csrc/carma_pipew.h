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
    static constexpr int MRG_OFF = TAB_OFF + MATH_TAB_N / 2;  // two-sided kernels: double[2][P][P + 1], the forward rows' (D, -a) for the merge
    static constexpr int MRG_STRIDE = P + 1 + ((P + 1) & 1);  // (doubles per column: an even count, so that columns are double2 aligned)
    static constexpr int ENTRIES = MRG_OFF + (2 * P * MRG_STRIDE + 1) / 2;
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
template <int P, bool TS = false, class Tail>
__device__ __forceinline__ void pipew_produce(const Grp<16>& g, int pw, const double* __restrict__ theta,
                                              const double4* __restrict__ series, int n_all, double2* __restrict__ ring, Tail&& tail)
{
    using Geo = PipeWGeom<P>;
    constexpr int ND = Geo::ND, NB = Geo::NB, ENT = Geo::ENT;
    if (CARMA_PRIO_P != 0) __builtin_amdgcn_s_setprio(CARMA_PRIO_P);
    // a lane works on a conjugate PAIR of roots (jr, jr + 1) of one slot: both share |E|, cos and sin
    constexpr int NPAIR = (P + 1) / 2, PPL = 16 / NPAIR;
    constexpr int NPROD = Geo::NPROD;
    constexpr int NIT = (ND + 1 + NPROD * PPL - 1) / (NPROD * PPL);   // slot ND is the re-base rotation
    const double* tab = reinterpret_cast<const double*>(ring + Geo::TAB_OFF);
    const int lane = g.lane64, l = lane & 15, rowb = lane & ~15, q = lane >> 4;
    const int sub = l / NPAIR, jr = 2 * (l - sub * NPAIR);
    const bool worker = sub < PPL, two = jr + 1 < P;
    // the part of the series this row works on
    const bool bwd = TS && (q & 1);
    const int nfwd = (n_all + 1) / 2;
    const int n = TS ? (bwd ? n_all - nfwd : nfwd) : n_all;
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
    // (a backward row walks the series from its end, in the negated time)
    auto recat = [=](int j) {
        const int jj = j < n ? j : n - 1;
        double4 r = series[bwd ? n_all - 1 - jj : jj];
        if (bwd) r.w = -r.w;
        return r;
    };
    double base = recat(0).w;
    const double t_meet = TS ? (bwd ? -series[nfwd - 1].w : series[nfwd - 1].w) : 0.0;    // the last forward datum's time
    // The chunk schedule is data dependent (a chunk ends in front of a re-base datum), so the records of a chunk cannot be
    // requested by index a chunk ahead as in carma_pipe3l.h -- and a global load at the head of every chunk would put an L2
    // round trip on every chunk (measured: 45 us per launch instead of 22).  So each row keeps a WINDOW of 64 records in
    // registers, lane l holding records jw + l, jw + 16 + l, jw + 32 + l, jw + 48 + l; a chunk's sixteen records lie in the
    // first two and are fetched from their lanes (ds_bpermute); when the row has moved past the first sixteen the window
    // shifts and the next sixteen are requested, two shifts before they are needed.
    int jw = 0;
    double4 rw0 = recat(l), rw1 = recat(16 + l), rw2 = recat(32 + l), rw3 = recat(48 + l);
    __syncthreads();                                          // the recursion wave has published (h_r, c_r)
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
    bool fin = false;                                         // TS: the final chunk (no data, the rotation to the meeting time)
    for (int c = 0;; c++) {
        const int b = c % NB;
#if defined(CARMA_WIN_STAMPS)
        WIN_STAMP(ps_t0);
#endif
        // --- schedule of this row's chunk: lane s looks at datum j0 + s
        const int jl = j0 + l;
        double4 rec;
        {
            const int pos = jl - jw, src = rowb + (pos & 15);
            const bool hi = pos >= 16;
            // (both fetches by every lane: a ds_bpermute under a divergent branch would read inactive lanes)
            const double y0 = __shfl(rw0.y, src, 64), y1 = __shfl(rw1.y, src, 64);
            const double z0 = __shfl(rw0.z, src, 64), z1 = __shfl(rw1.z, src, 64);
            const double w0 = __shfl(rw0.w, src, 64), w1_ = __shfl(rw1.w, src, 64);
            rec.x = 0.0;
            rec.y = hi ? y1 : y0;
            rec.z = hi ? z1 : z0;
            rec.w = hi ? w1_ : w0;
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
        const double t0 = fin ? t_meet : __shfl(tj, rowb, 64);
        const double t_lastc = __shfl(tj, rowb + (ncand > 0 ? ncand - 1 : 0), 64);
        const double W = sc > 0.0 ? 1.0 / sc : 1.0 / 0.0;
        const bool rot = fin || (ncand > 0 && (t_lastc - base) > W);
        const double base_old = base;
        base = rot ? t0 : base;
        // (the difference of two time stamps is exact unless the base is much the smaller of the two: carma_pipe3l.h)
        const double dta_l = tj - base;
        const unsigned long long flb = __ballot(dta_l > W && l < ncand);
        const unsigned cut = (unsigned)(flb >> (16 * q)) & 0xffffu;            // (bit 0 never: dta = 0 or <= W at slot 0)
        const int len = cut ? __builtin_ctz(cut) : ncand;
        const bool row_done = j0 + len >= n;
        const bool last = __ballot(!row_done) == 0ull;
        if (pw == 0) {
            const unsigned long long rb = __ballot(rot);
            if (lane == 0)
                reinterpret_cast<unsigned long long*>(ring + Geo::HDR_OFF)[b] =
                    ((rb & 1ull) | ((rb >> 15) & 2ull) | ((rb >> 30) & 4ull) | ((rb >> 45) & 8ull)) | ((TS ? fin : last) ? 256ull : 0ull);
            if (l < ND)
                ring[Geo::RING_OFF + (b * ENT + P) * 64 + lane] = l < len ? make_double2(rec.z * scale, rec.y - mu) : make_double2(1.0, 0.0);
        }
        // --- entries
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int slot = it * NPROD * PPL + pw * PPL + sub;
            const bool is_rot = slot == ND;
            const double dts = __shfl(dta_l, rowb + (slot < ND ? slot : 0), 64);
            const double dt = is_rot ? t0 - base_old : dts;
            const bool live = worker && (is_rot ? rot : slot < len);
            double ec = 1.0, es = 0.0, e1 = 1.0;
            if (live) cexp_step_tab<true>(w.re, w.im, dt, &ec, &es, tab);
            e1 = ec;
            if (realpair && live) e1 = exp_neg_tab(w1.re * dt, tab);
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
        j0 += len;
        if (j0 - jw >= 16) {                                  // (at most one shift per chunk: len < 16)
            rw0 = rw1;
            rw1 = rw2;
            rw2 = rw3;
            jw += 16;
            rw3 = recat(jw + 48 + l);
        }
#if defined(CARMA_WIN_STAMPS)
        WIN_STAMP(ps_t1);
        ps_work += ps_t1 - ps_t0;
#endif
        __syncthreads();                                      // barrier c: chunk c is in the ring
#if defined(CARMA_WIN_STAMPS)
        WIN_STAMP(ps_t0);
        ps_wait += ps_t0 - ps_t1;
        ps_n++;
#endif
        if (TS ? fin : last) break;
        fin = last;
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
// those of the backward row the columns of Db and -beta  (u_m = V^-1 z_m | second half ~ N(beta, V^-1 + Db)), and
//     log p(y) = l_a + l_b - 1/2 log det N + w.N^-1 a + (beta / 2).N^-1 (Da beta) ,   N = I - Da Db ,  w = beta + Db a / 2
// (tests/tools/proto/two_sided.py; V does not appear, and the diagonal rescaling of the coordinates by powers of two -- forward
// z / 2^e, backward u 2^e -- is a similarity of N).  The lanes of the BACKWARD row hold the COLUMNS of the bordered matrix
//     [ N  a  Da beta ]      lanes ND .. 15: columns of N;  lane ND - 1: a;  lane ND - 2: Da beta
//     [ w    0    0   ]      (rows P and P + 1 are never pivot rows)
//     [ beta/2  0  0  ]
// and eliminate its first P columns with row pivoting -- the pivot search is a compare chain inside lane ND + k, the row exchange
// a select in every lane, the multipliers come by DPP broadcast --, after which the border holds -w.N^-1 a and -(beta/2).N^-1 Da beta
// and the pivots the determinant.  Both go into the row's accumulators: |pivot| as a "variance", -2 x the quadratic terms as chi^2.
template <int P>
__device__ __forceinline__ void pipew_merge(int lane, const double (&kf)[P], double nu, double2* __restrict__ ring, LogLikAcc& acc)
{
    using Geo = PipeWGeom<P>;
    constexpr int ND = Geo::ND, ST = Geo::MRG_STRIDE;
    const int l = lane & 15, q = lane >> 4;
    const bool bwd = (q & 1) != 0, virt = l >= ND;
    double* mrg = reinterpret_cast<double*>(ring + Geo::MRG_OFF) + (q >> 1) * (P * ST);
    // forward row: column s of Da and -a_s to LDS (the wave's own LDS operations execute in order: no barrier)
    if (!bwd && virt) {
        double* dst = mrg + (l - ND) * ST;
#pragma unroll
        for (int r = 0; r < P; r++) dst[r] = kf[r];
        dst[P] = nu;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double av[P];                                             // a
#pragma unroll
    for (int k = 0; k < P; k++) av[k] = mrg[k * ST + P];
    // x: what this lane's column is the product of Da with;  col_i = base_i - sum_k Da_ik x_k
    //   lane ND + j:  x = Db[:, j] (own registers), base = e_j      -> column j of N = I - Da Db
    //   lane ND - 2:  x = -beta, base = 0                           -> Da beta
    //   lane ND - 1:  x = 0,     base = a                           -> a
    // (nu holds the NEGATED means on both sides; the quadratic terms are bilinear in (a, beta) jointly, so the two signs cancel and
    // "a", "beta" below are simply the registers' values)
    double x[P], col[P + 2];
    double betak[P];
    static_for<0, P>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        betak[k] = Grp<16>::template bcast_c<ND + k>(nu);     // beta_k (backward rows)
    });
#pragma unroll
    for (int k = 0; k < P; k++) x[k] = virt ? kf[k] : (l == ND - 2 ? -betak[k] : 0.0);
#pragma unroll
    for (int i = 0; i < P; i++) {
        double c0 = virt ? (l - ND == i ? 1.0 : 0.0) : (l == ND - 1 ? av[i] : 0.0);
#pragma unroll
        for (int k = 0; k < P; k++) c0 = fma(-mrg[k * ST + i], x[k], c0);      // Da_ik = Da_ki: column k, row i
        col[i] = c0;
    }
    // border rows (columns of N only): w_j = beta_j + 1/2 sum_k Db_kj a_k,  beta_j / 2
    {
        double wj = nu;
#pragma unroll
        for (int k = 0; k < P; k++) wj = fma(0.5 * kf[k], av[k], wj);
        col[P] = virt ? wj : 0.0;
        col[P + 1] = virt ? 0.5 * nu : 0.0;
    }
    double piv = 1.0;
    static_for<0, P>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        // pivot row: the largest |col_i|, i >= k, of column k (lane ND + k decides)
        int idx = k;
        double best = fabs(col[k]);
#pragma unroll
        for (int i = k + 1; i < P; i++) {
            const double v = fabs(col[i]);
            const bool gt = v > best;
            best = gt ? v : best;
            idx = gt ? i : idx;
        }
        idx = Grp<16>::template bcast_c<ND + k>(idx);
        double ck = col[k];
#pragma unroll
        for (int i = k + 1; i < P; i++) {
            const bool sel = idx == i;
            const double ci = col[i];
            col[i] = sel ? col[k] : ci;
            ck = sel ? ci : ck;
        }
        col[k] = ck;
        if (l == ND + k) piv = ck;
        const double r = -recip(ck);
#pragma unroll
        for (int i = k + 1; i < P + 2; i++) {
            const double li = Grp<16>::template bcast_c<ND + k>(col[i] * r);    // -(multiplier of row i), from the pivot column's lane
            col[i] = fma(li, ck, col[i]);
        }
    });
    if (bwd) {
        if (virt) acc.add_var(fabs(piv));
        // border: col[P] in lane ND - 1 = -(w.N^-1 a), col[P + 1] in lane ND - 2 = -(beta/2).N^-1 (Da beta);  l += Q  <=>  chi2 -= 2 Q
        acc.chi2 += l == ND - 1 ? 2.0 * col[P] : (l == ND - 2 ? 2.0 * col[P + 1] : 0.0);
    }
}

// wave A.  Returns the log-likelihood of the row's evaluation (row-uniform; TS: of the two rows of an evaluation together).
template <int P, bool TS = false>
__device__ __forceinline__ double pipew_recur(const Grp<16>& g, const RowConsts<P>& rc, double2* __restrict__ ring)
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
    load(0, en, hdr);
    start(0, ka, kb, hb);
    for (int c = 0;; c += 2) {
        if (chunk(kb, hb, ka, ha, c)) {
            if constexpr (TS) pipew_merge<P>(lane, ka, nuA, ring, acc);
            break;
        }
        if (chunk(ka, ha, kb, hb, c + 1)) {
            if constexpr (TS) pipew_merge<P>(lane, kb, nuA, ring, acc);
            break;
        }
    }
    double ll = Grp<16>::sum(acc.total());
    if constexpr (TS) ll += __shfl_xor(ll, 16, 64);           // forward + backward row (+ the merge, in the backward row's sums)
    return ll;
}

}  // namespace carma
