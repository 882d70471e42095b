// carma_lane_frame.h -- one evaluation per LANE in a CO-ROTATING FRAME (round 5): the recursion of carma_lane.h without the
// rotation of the matrix.
//
// lane_filter (carma_lane.h) spends 58 of the ~146 arithmetic instructions of a step (p = 5) on  D <- Phi (D - k k^T / var) Phi^T.
// The wave pipelines of the latency regime do not rotate at all (carma_pipe3l.h:5-14): with D = A S A^T, A the transition
// accumulated since the last RE-BASE (all Phi commute),
//     w~ = S h~ ,  h~ = A^T h       var = s0 + e + h~.w~       k~ = w~ + c~ ,  c~ = A^-1 c
//     S <- S - k~ k~^T / var        z = A z~ :  innov = y - mu - h~.z~ ,  z~ <- z~ + k~ innov / var
// (kfilter.cpp:191-213 element by element, up to the scale factors e^{+-Re(omega) dt}, which cancel in every product).  The
// same frame per lane: the step is ~80 instructions, and what it needs per datum -- the 2 p numbers (h~_r, c~_r) -- depends on
// omega and on the time since the re-base only: the lane computes them in line (k_logdens_carma_lanef: one complex exponential
// per PAIR as ever, ~20 instructions per pair for the two vectors), or three PRODUCER waves do (k_logdens_carma_lpcf: the
// consumer is the 80 instructions).
// RE-BASE (per lane; carma_pipe3l.h:15-20 for the frame that starts HALF A WINDOW AHEAD and for the rescaling of the modal
// coordinates by exact powers of two): a datum further than a window W from the lane's base opens a new frame -- S <- A S A^T,
// z~ <- A z~ with the rotation accumulated over the closing window (any length: a decayed coordinate underflows to 0), then
// h~ = g h, c~ = c / g.  W = 2^-ex <= min over the roots of (LIM_RE / |Re omega|, LIM_IM / |Im omega|).  The re-base runs
// under the lane's own predicate, the step itself is the same instructions for every lane: an evaluation's result does not
// depend on its neighbours in the wave.  Whether ANY lane re-bases inside a chunk of data is one comparison per chunk (times
// increase); a chunk without runs a loop that has no test in it.
// Plain C++ over doubles: compiled for the host by the test harness as well (tests/emu).
#pragma once
#include "carma_lane.h"

namespace carma {

// a value the compiler knows nothing about (device): the in-line kernel and the producer-wave kernel must contract the same
// products, and one of them sees these values come out of LDS
#if defined(__HIP_DEVICE_COMPILE__)
#define CARMA_LF_OPAQUE(x) asm("" : "+v"(x))
#else
#define CARMA_LF_OPAQUE(x) (void)(x)
#endif

struct LaneFrameLim {
    static constexpr double LIM_RE = 600.0;                   // carma_pipe3l.h: scale factors within e^+-300 around the frame's middle
    static constexpr double LIM_IM = 65536.0;                 // |Im omega| x window: inside the table form's own argument reduction
};

template <int P>
struct LaneFrame {
    // (h, c in coordinates rescaled by exact powers of two, |h_r| ~ sigma_y; g_r = e^{-Re omega_r W / 2} the frame's offset)
    double g[P];              // g_r
    double gh[P], gc[P];      // g h, g c: the entries are (ec gh_a + es gh_b, (ec gc_a + es gc_b) / |g A|^2) -- three vectors, not five
    double W;                 // window (inf: never re-base)
    // (h~, c~) at a re-base datum: g h and c / g -- the latter formed from gc where it is needed (lane_frame_cg)
};
template <int P>
CARMA_DEV double lane_frame_cg(const LaneFrame<P>& f, int r)
{
    const double rg = recip(f.g[r]);
    return (f.gc[r] * rg) * rg;
}

template <int P>
CARMA_DEV void lane_frame_setup(const LaneModel<P>& m, LaneFrame<P>& f)
{
    double wl = 0.0;
#pragma unroll
    for (int r = 0; r < P; r++) wl = fmax(wl, fmax(fabs(m.wre[r]) * (1.0 / LaneFrameLim::LIM_RE), fabs(m.wim[r]) * (1.0 / LaneFrameLim::LIM_IM)));
    wl = (wl < 1e12) ? wl : ((wl == wl && wl < 1.0 / 0.0) ? 1e12 : 0.0);
    int wex;
    (void)frexp(wl, &wex);
    const double sc = wl > 0.0 ? ldexp(1.0, wex) : 0.0;       // 2^ex > wl
    f.W = sc > 0.0 ? 1.0 / sc : 1.0 / 0.0;
    const double halfw = sc > 0.0 ? 0.5 / sc : 0.0;
    int esig = 0;                                             // binary exponent of sigma_y = sqrt(s0)
    {
        int e2;
        (void)frexp(m.s0, &e2);
        if (m.s0 > 0.0 && m.s0 < 1.0 / 0.0) esig = e2 / 2;
    }
    auto expo = [](double v) {
        int e;
        (void)frexp(v, &e);
        return (v > 0.0 && v < 1.0 / 0.0) ? e : 0;
    };
#pragma unroll
    for (int i = 0; i < (P + 1) / 2; i++) {
        const int a = 2 * i, b = 2 * i + 1;
        const bool two = b < P;
        const bool cpx = two && !m.realpair[i];
        // one power of two for both members of a complex pair (the rescaling must commute with their rotation)
        const double ma = cpx ? fmax(fabs(m.h[a]), fabs(m.h[two ? b : a])) : fabs(m.h[a]);
        const int ea = expo(ma);
        const double ga = recip(exp_neg(m.wre[a] * halfw));
        f.g[a] = ga;
        f.gh[a] = ga * ldexp(m.h[a], esig - ea);
        f.gc[a] = ga * ldexp(m.c[a], ea - esig);
        if (two) {
            const int eb = cpx ? ea : expo(fabs(m.h[b]));
            const double gb = cpx ? ga : recip(exp_neg(m.wre[b] * halfw));
            f.g[b] = gb;
            f.gh[b] = gb * ldexp(m.h[b], esig - eb);
            f.gc[b] = gb * ldexp(m.c[b], eb - esig);
        }
    }
#pragma unroll
    for (int r = 0; r < P; r++) {
        CARMA_LF_OPAQUE(f.g[r]);
        CARMA_LF_OPAQUE(f.gh[r]);
        CARMA_LF_OPAQUE(f.gc[r]);
    }
}

// exp(a dt) with the rounding of the product recovered (the frame's scale factor must not depend on how a dt rounds: it is
// the SAME factor that h~ is multiplied and c~ divided by, but S remembers the factors of earlier data)
CARMA_DEV double exp_step_tab_exact(double a, double dt, const double* tab)
{
    const double x = a * dt;
    const double n = rint(x * INV_LN2_32);
    double r = fma3(-n, LN2_32_HI, x);
    r = fma3(-n, LN2_32_LO, r);
    r += fma3(a, dt, -x);
    const int i = (int)fmin(fmax(n, -70400.0), 70400.0);
    const double e = tab[i & 31];
    double q = 1.0 / 720.0;
    q = fma3(q, r, 1.0 / 120.0);
    q = fma3(q, r, 1.0 / 24.0);
    q = fma3(q, r, 1.0 / 6.0);
    q = fma3(q, r, 0.5);
    q = fma3(q, r, 1.0);
    return ldexp(fma3(e, q * r, e), i >> 5);
}

// What a datum at time dta after the lane's base needs: the accumulated rotation (cr, sr: lane_filter's convention, for a
// re-base) and h~ = (g A)^T h, c~ = (g A)^-1 c.  One complex exponential per PAIR; pair members: h~_e = ec (g h_e) + es (g h_o),
// h~_o = e1 (g h_o) - es (g h_e), the same for c with 1 / |g A|^2 (carma_pipe3l.h, entry()).
template <int P, bool CHECK, bool ANYREAL>
CARMA_DEV void lane_frame_entries_impl(const LaneModel<P>& m, const LaneFrame<P>& f, double dta, const double* tab, double (&cr)[P],
                                       double (&sr)[P], double (&ht)[P], double (&ct)[P])
{
#pragma unroll
    for (int i = 0; i < P / 2; i++) {
        const int a = 2 * i, b = 2 * i + 1;
        double ec, es;
        cexp_step_tab_impl<CHECK, true>(m.wre[a], m.wim[a], dta, &ec, &es, tab);
        double e1 = ec;
        if (ANYREAL) {
            // a quadratic factor with two real roots: the second member has its own modulus (and no phase: es = 0 exactly)
            const double x = exp_step_tab_exact(m.wre[b], dta, tab);
            if (m.realpair[i]) e1 = x;
        }
        cr[a] = ec;
        sr[a] = es;
        cr[b] = e1;
        sr[b] = -es;
        const double xc = ec * f.g[a], xs = es * f.g[a];
        const double inv = recip(fma(xc, xc, xs * xs));
        ht[a] = fma(ec, f.gh[a], es * f.gh[b]);
        ct[a] = fma(ec, f.gc[a], es * f.gc[b]) * inv;
        double inv1 = inv;
        if (ANYREAL) {
            const double x1 = e1 * f.g[b];
            const double x = recip(x1 * x1);
            if (m.realpair[i]) inv1 = x;
        }
        ht[b] = fma(e1, f.gh[b], -(es * f.gh[a]));
        ct[b] = fma(e1, f.gc[b], -(es * f.gc[a])) * inv1;
    }
    if (P & 1) {
        constexpr int a = P - 1;
        const double e = exp_step_tab_exact(m.wre[a], dta, tab);
        cr[a] = e;
        sr[a] = 0.0;
        const double x = e * f.g[a];
        ht[a] = e * f.gh[a];
        ct[a] = (e * f.gc[a]) * recip(x * x);
    }
#pragma unroll
    for (int r = 0; r < P; r++) {
        CARMA_LF_OPAQUE(cr[r]);
        CARMA_LF_OPAQUE(sr[r]);
        CARMA_LF_OPAQUE(ht[r]);
        CARMA_LF_OPAQUE(ct[r]);
    }
}
// The huge-phase test (library reduction, carma_math.h) is made once per datum for all pairs and the whole wave, and the
// "is there a real pair in this wave" test picks a copy of the code: a datum's exponentials are then ONE basic block, whose
// two or three polynomial chains the compiler interleaves -- the producers' run time is that chain, not its instruction count.
template <int P>
CARMA_DEV void lane_frame_entries(const LaneModel<P>& m, const LaneFrame<P>& f, bool anyreal, double dta, const double* tab,
                                  double (&cr)[P], double (&sr)[P], double (&ht)[P], double (&ct)[P])
{
    bool slow = false;
#pragma unroll
    for (int i = 0; i < P / 2; i++) slow = slow || !(fabs(m.wim[2 * i] * dta) < CEXP_TAB_MAXPHASE);
    if (lane_any(slow)) {
        lane_frame_entries_impl<P, true, true>(m, f, dta, tab, cr, sr, ht, ct);
    } else if (anyreal) {
        lane_frame_entries_impl<P, false, true>(m, f, dta, tab, cr, sr, ht, ct);
    } else {
        lane_frame_entries_impl<P, false, false>(m, f, dta, tab, cr, sr, ht, ct);
    }
}

// S <- A S A^T, z~ <- A z~   (cr, sr per coordinate: (d A^T)_ij = d_ij c_j - d_{i,j^1} s_j, (A m)_ij = c_i m_ij - s_i m_{i^1,j})
template <int P>
CARMA_DEV void lane_frame_rebase(double (&S)[P * (P + 1) / 2], double (&z)[P], const double (&cr)[P], const double (&sr)[P])
{
    constexpr int PE = P & ~1;
    double zu[P];
#pragma unroll
    for (int r = 0; r < P; r++) zu[r] = z[r];
#pragma unroll
    for (int r = 0; r < P; r++) z[r] = (r < PE) ? fma(cr[r], zu[r], -(sr[r] * zu[r ^ 1])) : cr[r] * zu[r];
    double mm[P][P];
#pragma unroll
    for (int i = 0; i < P; i++) {
#pragma unroll
        for (int j = 0; j < P; j++) {
            const bool need = (j >= i) || (j == i - 1 && (i & 1) && i < PE);
            if (need) {
                if (j < PE)
                    mm[i][j] = fma(S[tri<P>(i, j)], cr[j], -(S[tri<P>(i, j ^ 1)] * sr[j]));
                else
                    mm[i][j] = S[tri<P>(i, j)] * cr[j];
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
                S[tri<P>(i, j)] = fma(cr[i], mm[i][j], -(sr[i] * mm[i ^ 1][j]));
            else
                S[tri<P>(i, j)] = cr[i] * mm[i][j];
        }
    }
}

// one datum: var and innovation into acc, then the update of z~ and the downdate of S (kfilter.cpp:191-213 in the frame)
// ev = s0 + scale yerr^2, yc = y - mu
template <int P>
CARMA_DEV void lane_frame_step(double (&S)[P * (P + 1) / 2], double (&z)[P], const double (&ht)[P], const double (&ct)[P], double ev,
                               double yc, LogLikAcc& acc)
{
    double w[P];
#pragma unroll
    for (int i = 0; i < P; i++) {
        double a = S[tri<P>(i, 0)] * ht[0];
#pragma unroll
        for (int j = 1; j < P; j++) a = fma(S[tri<P>(i, j)], ht[j], a);
        w[i] = a;
    }
    double var = ev, innov = yc;
#pragma unroll
    for (int r = 0; r < P; r++) {
        var = fma(ht[r], w[r], var);
        innov = fma(-ht[r], z[r], innov);
    }
    acc.add_var(var);
    const double s = recip(var);
    const double si = s * innov;
    acc.chi2 = fma(innov, si, acc.chi2);
    double k[P];
#pragma unroll
    for (int r = 0; r < P; r++) {
        k[r] = w[r] + ct[r];
        z[r] = fma(k[r], si, z[r]);
    }
#pragma unroll
    for (int i = 0; i < P; i++) {
        const double t = k[i] * s;
#pragma unroll
        for (int j = i; j < P; j++) S[tri<P>(i, j)] = fma(-t, k[j], S[tri<P>(i, j)]);
    }
}

// Reset + n - 1 Updates -> log-likelihood sum (no prior), everything in line
template <int P>
CARMA_DEV double lane_filter_frame(const LaneModel<P>& m, const double4* __restrict__ series, int n, bool anyreal, const double* tab)
{
    constexpr int NT = P * (P + 1) / 2;
    constexpr int CH = 4;                                     // data per "does any lane re-base" test
    LaneFrame<P> f;
    lane_frame_setup<P>(m, f);
    double S[NT], z[P];
#pragma unroll
    for (int i = 0; i < NT; i++) S[i] = 0.0;
#pragma unroll
    for (int r = 0; r < P; r++) z[r] = 0.0;
    LogLikAcc acc;
    acc.init();
    double base = series[0].w;
    int j0 = 0;
    // chunk by chunk: the loop without a test where no lane leaves its window, the per-datum test elsewhere (and for the tail)
    while (j0 < n) {
        const int j1 = j0 + CH < n ? j0 + CH : n;
        if (j1 - j0 == CH && !lane_any(series[j1 - 1].w - base > f.W)) {
#pragma unroll 1
            for (int j = j0; j < j1; j++) {
                const double4 rec = series[j];
                double cr[P], sr[P], ht[P], ct[P];
                lane_frame_entries<P>(m, f, anyreal, rec.w - base, tab, cr, sr, ht, ct);
                lane_frame_step<P>(S, z, ht, ct, fma(rec.z, m.scale, m.s0), rec.y - m.mu, acc);
            }
        } else {
#pragma unroll 1
            for (int j = j0; j < j1; j++) {
                const double4 rec = series[j];
                const double dta = rec.w - base;
                const bool fl = dta > f.W;
                double cr[P], sr[P], ht[P], ct[P];
                lane_frame_entries<P>(m, f, anyreal, dta, tab, cr, sr, ht, ct);
#ifndef CARMA_LF_X_NOSLOW
                if (lane_any(fl)) {
                    if (fl) {
                        lane_frame_rebase<P>(S, z, cr, sr);
#pragma unroll
                        for (int r = 0; r < P; r++) {
                            ht[r] = f.gh[r];
                            ct[r] = lane_frame_cg<P>(f, r);
                        }
                        base = rec.w;
                    }
                }
#endif
                lane_frame_step<P>(S, z, ht, ct, fma(rec.z, m.scale, m.s0), rec.y - m.mu, acc);
            }
        }
        j0 = j1;
    }
    return acc.total();
}

// CARMA_Base::LogDensity (carpack.hpp:131-176), as logdensity_lane
template <int P>
CARMA_DEV double logdensity_lane_frame(const double* theta, int q, const double4* __restrict__ series, int n, const Prior& pr,
                                       int ignore_prior, const double* tab)
{
    LaneModel<P> m;
    lane_model_from_theta<P>(theta, q, pr, ignore_prior, m);
    bool anyreal = false;
#pragma unroll
    for (int i = 0; i < P / 2; i++) anyreal = anyreal || m.realpair[i];
    double ll = lane_filter_frame<P>(m, series, n, lane_any(anyreal), tab);
    ll += log_prior(m.scale, pr.measerr_dof);
    if (m.sing || !m.valid) ll = -1.0 / 0.0;
    return ll;
}

#if defined(__HIPCC__)
// ---------------------------------------------------------------------------------------------------------------------
// PRODUCER WAVES (k_logdens_carma_lpcf): a workgroup of four waves is ONE consumer (64 evaluations: lane_frame_step and, rarely,
// lane_frame_rebase -- nothing else) and NP = 3 producers (lane l serves lane l of the consumer; producer k takes the data
// s = k mod NP of a chunk of CH).  Two buffers, one workgroup barrier per chunk.  A datum's slot holds double2 {h~_r, c~_r} per
// coordinate -- at a re-base datum of the lane {cr_r, sr_r}, the rotation over the closing window, and a per-lane flag says
// so; one word per chunk says whether any lane of the wave re-bases in it (then the consumer reads the flags).
template <int P, int NP, int CH>
struct LaneFrameRingGeom {
    static constexpr int SLOT2 = P * 64;                                    // double2 per datum
    static constexpr size_t FLAG_OFF = (size_t)2 * CH * SLOT2 * 2;          // (in doubles) int flag[2][CH][64]
    static constexpr size_t ANY_OFF = FLAG_OFF + (size_t)2 * CH * 32;       // int any[2]
    static constexpr size_t DOUBLES = ANY_OFF + 1;
    static constexpr size_t BYTES = DOUBLES * sizeof(double);               // p = 5, CH = 6: 63.0 KiB; p = 7, CH = 3: 43.5 KiB
};
template <int P, int NP>
struct LaneFrameCH {
    static constexpr int value = (P <= 5 || NP == 6) ? 6 : 3;               // (p >= 6, NP = 3: within the 64 KiB a launch gets without asking)
};

template <int P, int NP, int CH>
__device__ __forceinline__ void lane_frame_produce(int k, const double* theta, int q, const Prior& pr, int ignore_prior, double* ring,
                                                   const double4* __restrict__ series, int n, const double* tab)
{
    using Geo = LaneFrameRingGeom<P, NP, CH>;
    const int lane = threadIdx.x & 63;
    LaneModel<P> m;
    lane_model_from_theta<P>(theta, q, pr, ignore_prior, m);
    bool anyreal = false;
#pragma unroll
    for (int i = 0; i < P / 2; i++) anyreal = anyreal || m.realpair[i];
    anyreal = __builtin_amdgcn_ballot_w64(anyreal) != 0;
    LaneFrame<P> f;
    lane_frame_setup<P>(m, f);
    double base = series[0].w;
    const int nc = (n + CH - 1) / CH;
    // the times of a chunk are requested a chunk ahead (scalar loads: a dependent load per datum costs more than its exponentials)
    double tn[CH];
#pragma unroll
    for (int s = 0; s < CH; s++) tn[s] = series[s < n ? s : n - 1].w;
    for (int c = 0; c < nc; c++) {
        const int j0 = c * CH;
        double tc[CH];
#pragma unroll
        for (int s = 0; s < CH; s++) {
            tc[s] = tn[s];
            const int jn = j0 + CH + s;
            tn[s] = series[jn < n ? jn : n - 1].w;            // (beyond the end: the last time again)
        }
        double2* ent = reinterpret_cast<double2*>(ring) + (size_t)(c & 1) * CH * Geo::SLOT2 + lane;
        int* flg = reinterpret_cast<int*>(ring + Geo::FLAG_OFF) + (c & 1) * CH * 64 + lane;
        const bool any = __builtin_amdgcn_ballot_w64(tc[CH - 1] - base > f.W) != 0;      // every producer: the same word
        if (k == 0 && lane == 0) reinterpret_cast<int*>(ring + Geo::ANY_OFF)[c & 1] = any ? 1 : 0;
#if defined(CARMA_AB_LF_NOPROD)                               // timing-only A/B build: the consumer alone
        if (c >= 2) {
            __syncthreads();
            continue;
        }
#endif
        // the lane's base through the chunk (a re-base datum moves it), and what this producer's own data see of it
        constexpr int NOWN = CH / NP;
        double dta_own[NOWN];
        bool fl_own[NOWN];
#pragma unroll
        for (int u = 0; u < NOWN; u++) {
            dta_own[u] = 0.0;
            fl_own[u] = false;
        }
#pragma unroll
        for (int s = 0; s < CH; s++) {
            const double dta = tc[s] - base;
            const bool fl = any && dta > f.W && j0 + s < n;
#pragma unroll
            for (int u = 0; u < NOWN; u++) {
                if (s == k + u * NP) {                         // (wave-uniform)
                    dta_own[u] = dta;
                    fl_own[u] = fl;
                }
            }
            if (fl) base = tc[s];
        }
#pragma unroll
        for (int u = 0; u < NOWN; u++) {
            const int so = k + u * NP;
            if (j0 + so < n) {
                double cr[P], sr[P], ht[P], ct[P];
                lane_frame_entries<P>(m, f, anyreal, dta_own[u], tab, cr, sr, ht, ct);
                const bool fl = fl_own[u];
#pragma unroll
                for (int r = 0; r < P; r++) ent[(size_t)(so * P + r) * 64] = fl ? make_double2(cr[r], sr[r]) : make_double2(ht[r], ct[r]);
                if (any) flg[so * 64] = fl ? 1 : 0;
            }
        }
        __syncthreads();                                      // barrier c: chunk c is in the ring
    }
}

template <int P, int NP, int CH>
__device__ __forceinline__ double logdensity_lane_frame_ring(const double* theta, int q, const double4* __restrict__ series, int n,
                                                             const Prior& pr, int ignore_prior, const double* ring)
{
    using Geo = LaneFrameRingGeom<P, NP, CH>;
    constexpr int NT = P * (P + 1) / 2;
    const int lane = threadIdx.x & 63;
    LaneModel<P> m;
    lane_model_from_theta<P>(theta, q, pr, ignore_prior, m);
    LaneFrame<P> f;
    lane_frame_setup<P>(m, f);
    double S[NT], z[P];
#pragma unroll
    for (int i = 0; i < NT; i++) S[i] = 0.0;
#pragma unroll
    for (int r = 0; r < P; r++) z[r] = 0.0;
    LogLikAcc acc;
    acc.init();
    const int nc = (n + CH - 1) / CH;
    // (y, yerr^2) of a chunk are requested a chunk ahead (scalar loads)
    double yn[CH], en[CH];
#pragma unroll
    for (int s = 0; s < CH; s++) {
        const double4 rec = series[s < n ? s : n - 1];
        yn[s] = rec.y;
        en[s] = rec.z;
    }
    for (int c = 0; c < nc; c++) {
        const int j0 = c * CH, j1 = j0 + CH < n ? j0 + CH : n;
        double yc[CH], ec[CH];
#pragma unroll
        for (int s = 0; s < CH; s++) {
            yc[s] = yn[s];
            ec[s] = en[s];
            const int jn = j0 + CH + s;
            const double4 rec = series[jn < n ? jn : n - 1];
            yn[s] = rec.y;
            en[s] = rec.z;
        }
        __syncthreads();                                      // barrier c
        const double2* ent = reinterpret_cast<const double2*>(ring) + (size_t)(c & 1) * CH * Geo::SLOT2 + lane;
        const int* flg = reinterpret_cast<const int*>(ring + Geo::FLAG_OFF) + (c & 1) * CH * 64 + lane;
        const int any = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(ring + Geo::ANY_OFF)[c & 1]);
#if defined(CARMA_AB_LF_NOCONS)                               // timing-only A/B build: the producers alone
        if (c >= 2) continue;
#endif
        if (!any && j1 - j0 == CH) {
#pragma unroll
            for (int s = 0; s < CH; s++) {
                double ht[P], ct[P];
#pragma unroll
                for (int r = 0; r < P; r++) {
                    const double2 e = ent[(size_t)(s * P + r) * 64];
                    ht[r] = e.x;
                    ct[r] = e.y;
                }
                lane_frame_step<P>(S, z, ht, ct, fma(ec[s], m.scale, m.s0), yc[s] - m.mu, acc);
            }
        } else {
#pragma unroll
            for (int s = 0; s < CH; s++) {
                if (j0 + s < n) {
                    const bool fl = any ? flg[s * 64] != 0 : false;
                    double ht[P], ct[P];
#pragma unroll
                    for (int r = 0; r < P; r++) {
                        const double2 e = ent[(size_t)(s * P + r) * 64];
                        ht[r] = e.x;
                        ct[r] = e.y;
                    }
                    if (__builtin_amdgcn_ballot_w64(fl) != 0) {
                        if (fl) {
                            double cr[P], sr[P];
#pragma unroll
                            for (int r = 0; r < P; r++) {
                                cr[r] = ht[r];
                                sr[r] = ct[r];
                            }
                            lane_frame_rebase<P>(S, z, cr, sr);
#pragma unroll
                            for (int r = 0; r < P; r++) {
                                ht[r] = f.gh[r];
                                ct[r] = lane_frame_cg<P>(f, r);
                            }
                        }
                    }
                    lane_frame_step<P>(S, z, ht, ct, fma(ec[s], m.scale, m.s0), yc[s] - m.mu, acc);
                }
            }
        }
    }
    double ll = acc.total();
    ll += log_prior(m.scale, pr.measerr_dof);
    if (m.sing || !m.valid) ll = -1.0 / 0.0;
    return ll;
}
#endif

}  // namespace carma
