// carma_kernels.hip -- gfx950 kernels for the batched CARMA / CAR(1) Kalman log-density.
//
// K1  k_logdens_carma<P,G,WAVES> : B independent CARMA_Base::LogDensity evaluations
//      (src/include/carpack.hpp:131-176 over src/kfilter.cpp:138-215).  G lanes per evaluation
//      (G = 8 for p = 5..7, 4 for p = 3..4, 2 for p = 2), i.e. 8/16/32 evaluations per wave64;
//      lane r owns row r of the Hermitian state covariance (registers), the per-step all-gather
//      of (u_r, rho_r) goes through 2 KiB of LDS per wave, var/mean through a DPP butterfly.
//      The series record of step k is wave-uniform -> one 32-byte scalar load per step.
// K2  k_logdens_car1            : CAR(1), one lane per evaluation (src/kfilter.cpp:19-48).
// K1m k_kfilter_carma<P,G>      : one evaluation that also stores mean[n], var[n]
//      (KalmanFilterp::Filter, src/include/kfilter.hpp:126-132).
//
// FP64 vector ALU + software exp/sincos; not a contraction, so no MFMA.  HBM traffic is the
// series (24 B/datum, shared by every evaluation and L2/scalar-cache resident) + 8(d+1) B/eval.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "grp_device.h"
#include "carma_core.h"
#include "carma_ring.h"
#include "carma_predict.h"
#include "carma_simulate.h"
#include "carma_pipe3l.h"
#include "carma_pipew.h"
#include "carma_lane.h"
#include "carma_launch.h"

namespace carma {

template <int P>
struct GroupOf {
    static constexpr int value = P <= 2 ? 2 : (P <= 4 ? 4 : 8);
};

template <int P, int G, int WAVES, bool DTC = false>
__global__ __launch_bounds__(64 * WAVES) void k_logdens_carma(const double* __restrict__ theta, int B, int d, int q,
                                                             const double4* __restrict__ series, int n, Prior pr,
                                                             int ignore_prior, double* __restrict__ out)
{
    __shared__ double4 xch[64 * WAVES];
    __shared__ double2 xch2[64 * WAVES];
    const int tid = threadIdx.x;
    Grp<G> g{xch + (tid & ~63), tid & 63, xch2 + (tid & ~63)};
    long e = ((long)blockIdx.x * (64 * WAVES) + tid) / G;
    const bool live = e < B;
    if (!live) e = B - 1;
    double ll = logdensity_carma<P, G, Grp<G>, DTC>(g, theta + e * d, q, series, n, pr, ignore_prior);
    if (live && g.lane() == 0) out[e] = ll;
}

// Latency-regime variant (carma_ring.h): a workgroup holds PAIRS x (consumer wave + rho-producer wave),
// each pair working on 64/G evaluations.  PAIRS = 2 fills the four SIMDs of a CU from ONE workgroup
// (two 2-wave workgroups on a CU were observed to share SIMDs).
template <int P, int G, int PAIRS>
__global__ __launch_bounds__(128 * PAIRS) void k_logdens_carma_pc(const double* __restrict__ theta, int B, int d, int q,
                                                                  const double4* __restrict__ series, int n, Prior pr,
                                                                  int ignore_prior, double* __restrict__ out)
{
    extern __shared__ double4 smem4[];
    const int tid = threadIdx.x, wave = tid >> 6, pair = wave >> 1, role = wave & 1, lane64 = tid & 63;
    Grp<G> g{smem4 + wave * 64, lane64, nullptr};
    Cx* ring = reinterpret_cast<Cx*>(smem4 + 128 * PAIRS) + (size_t)pair * RingGeom<P>::ENTRIES;
    long e = (((long)blockIdx.x * PAIRS + pair) * 64 + lane64) / G;
    const bool live = e < B;
    if (!live) e = B - 1;
    if (role == 1) {
        const int r = g.lane();
        ring_produce<P, G>(g, own_ar_root<P>(theta + e * d, r < P ? r : P - 1), series, n, ring);
        return;
    }
    CARMA_STAMP_DECL;
    CARMA_STAMP(st0);
    Model<P> m;
    model_from_theta<P, G>(g, theta + e * d, q, pr, ignore_prior, m);
    CARMA_STAMP(st1);
    bool sing;
    double ll = ring_consume<P, G>(g, m, series, n, ring, &sing);
    CARMA_STAMP(st2);
    ll += log_prior(m.scale, pr.measerr_dof);
    const double ninf = -1.0 / 0.0;
    if (sing || !m.valid) ll = ninf;
    if (live && g.lane() == 0) out[e] = ll;
    CARMA_STAMP(st3);
#if defined(CARMA_STAMPS)
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("kernel stamps (100 MHz ticks): model %llu  reset+loop %llu  tail %llu\n", st1 - st0, st2 - st1, st3 - st2);
#endif
}

// Smallest launches (<= 1024 evaluations): covariance wave + mean wave + two producer waves per 4 evaluations, in a
// co-rotating frame (carma_pipe3l.h).  256 threads, 82 KiB of LDS: one workgroup per CU.
template <int P>
__global__ __launch_bounds__(256) void k_logdens_carma_p3l(const double* __restrict__ theta, int B, int d, int q,
                                                           const double4* __restrict__ series, int n, Prior pr,
                                                           int ignore_prior, double* __restrict__ out, int ncu, int npad)
{
    // n includes npad neutral pad data at the end (carma_types.h, p3l_pad)
    extern __shared__ double4 smem4[];
    const int tid = threadIdx.x, lane64 = tid & 63;
    // Which wave plays which part.  Workgroups i, i + ncu, i + 2 ncu share a CU, and the waves of successive workgroups
    // of a CU land on SIMDs (s, s+2, s+1, s+3), (s+2, s+1, s+3, s), (s+1, s+3, s, s+2) (tools/ubench/wave_placement.hip).
    // With the same assignment everywhere the second workgroup's covariance wave -- the critical one -- would share
    // its SIMD with the first one's mean wave; rotated, each covariance and each mean wave gets a producer for company.
    //                 part of wave:  0  1  2  3      (0 covariance, 1 mean, 2 / 3 producers)
    //   first  workgroup of a CU     0  1  2  3
    //   second                       2  0  1  3
    //   third                        2  1  3  0
    const int round = (blockIdx.x >= (unsigned)ncu) + (blockIdx.x >= 2u * (unsigned)ncu);
    const int wave = ((round == 0 ? 0xE4 : round == 1 ? 0xD2 : 0x36) >> (2 * (tid >> 6))) & 3;
    Grp<16> g{nullptr, lane64, nullptr};
    Cx* ring = reinterpret_cast<Cx*>(smem4);
    long e = ((long)blockIdx.x * 64 + lane64) / 16;
    const bool live = e < B;
    if (!live) e = B - 1;
    CARMA_MARK_DECL;
    CARMA_MARK(0);
    if (wave >= 2) {
        pipe3l_produce<P>(g, wave - 2, theta + e * d, series, n, npad, ring, [](int) {});
        CARMA_MARK(3);
        CARMA_MARK_DUMP("producer", wave - 2);
        return;
    }
    Model<P> m;
    if (wave == 0) {
        model_from_theta<P, 16, MODEL_CONSTS>(g, theta + e * d, q, pr, ignore_prior, m);
        CARMA_MARK(1);
        FilterConsts<P> fc;
        filter_reset<P, 16>(g, m, fc);
        RowConsts<P> rc;
        row_consts<P>(g, m, fc, rc);
        CARMA_MARK(2);
        pipe3l_cov<P>(g, m, rc, series, n, npad, ring);
        CARMA_MARK(3);
        CARMA_MARK_DUMP("covariance", 0);
        return;
    }
    // the set-up is split: the covariance wave forms the constants of the recursion, this wave checks the prior bounds
    model_from_theta<P, 16, MODEL_FLAGS>(g, theta + e * d, q, pr, ignore_prior, m);
    CARMA_MARK(1);
    // the log prior is evaluated HERE, while the mean wave would otherwise wait for the pipeline to fill, not after the
    // recursion (a serial chain of ~1000 cycles on the critical path of the launch)
    double lpri = log_prior(m.scale, pr.measerr_dof) + pipe3l_pad_correction(npad, theta[e * d], m.scale, series[n - npad - 1].y, m.mu);
    asm volatile("" : "+v"(lpri));
    CARMA_MARK(2);
    bool sing;
    double ll = pipe3l_mean<P>(g, m.mu, series, n, npad, ring, &sing);
    CARMA_MARK(3);
    ll += lpri;
    const double ninf = -1.0 / 0.0;
    if (sing || !m.valid) ll = ninf;
    if (live && g.lane() == 0) out[e] = ll;
    CARMA_MARK(4);
    CARMA_MARK_DUMP("mean", 0);
}

// The windowed wave pipeline (carma_pipew.h, round 5): recursion wave (covariance AND mean, a chunk of 16 - P data per
// elimination) + set-up wave + two producer waves per 4 evaluations.  13-17 KiB of LDS.
template <int P>
__global__ __launch_bounds__(256) void k_logdens_carma_w(const double* __restrict__ theta, int B, int d, int q,
                                                         const double4* __restrict__ series, int n, Prior pr,
                                                         int ignore_prior, double* __restrict__ out, int ncu)
{
    extern __shared__ double4 smem4[];
    const int tid = threadIdx.x, lane64 = tid & 63;
    // which wave plays which part: as k_logdens_carma_p3l (0 recursion, 1 set-up, 2 / 3 producers)
    const int round = (blockIdx.x >= (unsigned)ncu) + (blockIdx.x >= 2u * (unsigned)ncu);
    const int wave = ((round == 0 ? 0xE4 : round == 1 ? 0xD2 : 0x36) >> (2 * (tid >> 6))) & 3;
    Grp<16> g{nullptr, lane64, nullptr};
    double2* ring = reinterpret_cast<double2*>(smem4);
    long e = ((long)blockIdx.x * 64 + lane64) / 16;
    const bool live = e < B;
    if (!live) e = B - 1;
    using Geo = PipeWGeom<P>;
    math_tab_fill(reinterpret_cast<double*>(ring + Geo::TAB_OFF));      // (visible to the producers behind the first barrier)
    if (wave >= 2) {
        pipew_produce<P>(g, wave - 2, theta + e * d, series, n, ring, [](int) {});
        return;
    }
    Model<P> m;
    if (wave == 1) {
        // prior bounds and log prior (carpack.hpp:118-126, 178-191; carpack.cpp:314-374), handed over through LDS; then this
        // wave is the third producer
        model_from_theta<P, 16, MODEL_FLAGS>(g, theta + e * d, q, pr, ignore_prior, m);
        const double lpri = log_prior(m.scale, pr.measerr_dof);
        if ((lane64 & 15) == 0) ring[Geo::OUT_OFF + (lane64 >> 4)] = make_double2(lpri, m.valid ? 1.0 : 0.0);
        pipew_produce<P>(g, 2, theta + e * d, series, n, ring, [](int) {});
        return;
    }
    model_from_theta<P, 16, MODEL_CONSTS>(g, theta + e * d, q, pr, ignore_prior, m);
    FilterConsts<P> fc;
    filter_reset<P, 16>(g, m, fc);
    RowConsts<P> rc;
    row_consts<P>(g, m, fc, rc);
    double ll = pipew_recur<P>(g, rc, ring);
    const double2 o = ring[Geo::OUT_OFF + (lane64 >> 4)];
    ll += o.x;
    const double ninf = -1.0 / 0.0;
    if (m.sing || o.y == 0.0) ll = ninf;
    if (live && g.lane() == 0) out[e] = ll;
}

// The TWO-SIDED window pipeline (round 6): an evaluation takes TWO DPP rows -- the even row filters the first half of the series
// forward, the odd row the second half backward (carma_pipew.h, TS), and the two states are merged at the meeting time -- so the
// serial chain of kfilter.cpp:189-215 is half as long.  Two evaluations per workgroup.
// Which wave plays which part: as k_logdens_carma_w.  (Measured and dropped, profiles/r06/w2_check_v2.txt: with two workgroups per
// CU, two producer waves per workgroup on the SIMDs without a recursion wave and the set-up wave idle -- one producer then makes two
// passes per chunk, and the recursion wave waits for it: 28.0 against 25.1 us per 1024 evaluations.)
// HO: the producers' schedule hand-over (carma_pipew.h, SSCHED) -- launches of more than one workgroup per CU.
// SL: the series in LDS (up to W2_MAX_N data); without, a longer series is read from global memory through the rows' register windows.
template <int P, bool HO, bool SL = true>
__global__ __launch_bounds__(256) void k_logdens_carma_w2(const double* __restrict__ theta, int B, int d, int q,
                                                          const double4* __restrict__ series, int n, Prior pr,
                                                          int ignore_prior, double* __restrict__ out, int ncu)
{
    extern __shared__ double4 smem4[];
    const int tid = threadIdx.x, lane64 = tid & 63;
    const int round = (blockIdx.x >= (unsigned)ncu) + (blockIdx.x >= 2u * (unsigned)ncu);
    const int wave = ((round == 0 ? 0xE4 : round == 1 ? 0xD2 : 0x36) >> (2 * (tid >> 6))) & 3;
    Grp<16> g{nullptr, lane64, nullptr};
    double2* ring = reinterpret_cast<double2*>(smem4);
    long e = ((long)blockIdx.x * 64 + lane64) / 32;
    const bool live = e < B;
    if (!live) e = B - 1;
    using Geo = PipeWGeom<P>;
    CARMA_MARK_DECL;
    CARMA_MARK(0);
    // The parameters through LDS: each lane requests ONE element of its row's vector -- together with the element of the math tables it
    // copies -- and everything behind reads theta from the wave's own LDS copy.  (Read where they are used, the parameters were three
    // or four dependent L2 round trips in every wave's prologue, and the tables' copy one more in front of them: a second pass over
    // the same set-up code took 2.3 k cycles where the first took 5.6 k, profiles/r06/w2_stamps_v4.txt.)
    const double* th;
    {
        double* s_th = reinterpret_cast<double*>(ring + Geo::TH_OFF) + (tid >> 6) * 64 + (lane64 & ~15);
        const int l16 = lane64 & 15;
        const double thv = l16 < d ? theta[e * d + l16] : 0.0;
        double* tab = reinterpret_cast<double*>(ring + Geo::TAB_OFF);
        const double tv = tid < MATH_TAB_N ? c_math_tab[tid] : 0.0;
        s_th[l16] = thv;
        if (tid < MATH_TAB_N) tab[tid] = tv;
        static_assert(MATH_TAB_N <= 256, "one table element per thread");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        th = s_th;
    }
    // the series into LDS (producers: carma_pipew.h, SLDS): {y, yerr^2}[n], then t[n]; visible behind the producers' first barrier
    double2* lds_yz = ring + Geo::SER_OFF;
    double* lds_t = reinterpret_cast<double*>(lds_yz + n + (n & 1));
    if constexpr (SL) {
        for (int i = tid; i < n; i += 256) {
            const double4 r = series[i];
            lds_yz[i] = make_double2(r.y, r.z);
            lds_t[i] = r.w;
        }
    }
    if (wave >= 2) {
#if defined(CARMA_STAMPS)
        pipew_produce<P, true, SL, HO>(g, wave - 2, th, series, n, ring, [](int) {}, mark_, lds_t, lds_yz);
        CARMA_MARK_DUMP("two-sided producer: barrier 1 arrival, passed, chunk 0 done, barrier passed, chunk 1 done, chunk 2 done", wave - 2);
#else
        pipew_produce<P, true, SL, HO>(g, wave - 2, th, series, n, ring, [](int) {}, nullptr, lds_t, lds_yz);
#endif
        __syncthreads();                                      // (the set-up wave's hand-over, below)
        return;
    }
    Model<P> m;
    if (wave == 1) {
        // the third producer FIRST -- prior bounds and log prior (carpack.hpp:118-126, 178-191; carpack.cpp:314-374) are wanted at the
        // very end only, and in front of the pipeline they kept the first chunk waiting (the other waves were at the first barrier
        // 3.5 k cycles before this one: profiles/r06/w2_stamps_v3.txt); now they run while the recursion wave merges
#if defined(CARMA_STAMPS)
        pipew_produce<P, true, SL, HO>(g, 2, th, series, n, ring, [](int) {}, mark_, lds_t, lds_yz);
        CARMA_MARK_DUMP("two-sided set-up wave: barrier 1 arrival, passed, chunk 0 done, barrier passed, chunk 1 done, chunk 2 done", 2);
#else
        pipew_produce<P, true, SL, HO>(g, 2, th, series, n, ring, [](int) {}, nullptr, lds_t, lds_yz);
#endif
        model_from_theta<P, 16, MODEL_FLAGS>(g, th, q, pr, ignore_prior, m);
        const double lpri = log_prior(m.scale, pr.measerr_dof);
        if ((lane64 & 15) == 0) ring[Geo::OUT_OFF + (lane64 >> 4)] = make_double2(lpri, m.valid ? 1.0 : 0.0);
        __syncthreads();
        return;
    }
    model_from_theta<P, 16, MODEL_CONSTS>(g, th, q, pr, ignore_prior, m);
    CARMA_MARK(1);
#if defined(CARMA_STAMPS) && defined(CARMA_STAMP_TWICE)
    {
        // the same set-up once more (instructions now cached, theta in the L1): what a warm pass costs
        const double* th2 = th;
        asm volatile("" : "+v"(th2));
        long long w0 = clock64();
        Model<P> m2;
        model_from_theta<P, 16, MODEL_CONSTS>(g, th2, q, pr, ignore_prior, m2);
        double sink = m2.sigsqr + m2.kap.re + m2.b.re;
        asm volatile("" : "+v"(sink));
        long long w1 = clock64();
        if (blockIdx.x == 0 && lane64 == 0) printf("set-up a second time: %lld cycles (first: %lld)\n", w1 - w0, mark_[1] - mark_[0]);
    }
#endif
    FilterConsts<P> fc;
    filter_reset<P, 16>(g, m, fc);
    RowConsts<P> rc;
    row_consts<P>(g, m, fc, rc);
    CARMA_MARK(2);
#if defined(CARMA_STAMPS)
    double ll = pipew_recur<P, true>(g, rc, ring, mark_);
#else
    double ll = pipew_recur<P, true>(g, rc, ring);
#endif
    __syncthreads();                                          // the set-up wave's {log prior, valid}
    const double2 o = ring[Geo::OUT_OFF + (lane64 >> 4)];
    ll += o.x;
    const double ninf = -1.0 / 0.0;
    if (m.sing || o.y == 0.0) ll = ninf;
    if (live && (lane64 & 31) == 0) out[e] = ll;
    CARMA_MARK(7);
    CARMA_MARK_DUMP("two-sided recursion: model, reset, barrier 0, first start, last start, merge, end", 0);
}

// Throughput regime proper (tens of thousands of evaluations): ONE EVALUATION PER LANE (carma_lane.h) -- nothing crosses
// lanes, all 64 lanes work; a wave per 64 evaluations.
// REPDT (here and below): the variant for series with repeated time steps (carma_lane.h, lane_filter)
template <int P, bool REPDT = false>
__global__ __launch_bounds__(64) void k_logdens_carma_lane(const double* __restrict__ theta, int B, int d, int q,
                                                          const double4* __restrict__ series, int n, Prior pr,
                                                          int ignore_prior, double* __restrict__ out)
{
    __shared__ double s_tab[MATH_TAB_N];                     // tables of the table-based exp / sincos (carma_math.h)
    math_tab_fill(s_tab);
    __syncthreads();
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = e < B;
    if (!live) e = B - 1;
    const double ll = logdensity_lane<P, REPDT>(theta + e * d, q, series, n, pr, ignore_prior, s_tab);
    if (live) out[e] = ll;
}

// One evaluation per lane with the transition factors from PRODUCER WAVES (carma_lane.h, LaneFactorsRing): workgroups of
// four waves -- two consumers with a producer each (NP = 1, 128 evaluations), or one consumer with three producers
// (NP = 3, 64 evaluations).  For launches in which a lone wave's instruction stream would be the run time.
// Which wave plays which part rotates with the round of workgroups (blockIdx against the CU count): the waves of the
// workgroups that share a CU land on its SIMDs in a fixed pattern (tools/ubench/wave_placement.hip), and two consumers
// on one SIMD would wait for each other.
// UNROLLED: the consumer takes a ring buffer of six steps as one basic block (carma_lane.h, LaneFactorsRing) -- launches of up to
// one workgroup per CU.
template <int P, int NP, bool REPDT = false, bool UNROLLED = false>
__global__ __launch_bounds__(256) void k_logdens_carma_lpc(const double* __restrict__ theta, int B, int d, int q,
                                                           const double4* __restrict__ series, int n, Prior pr,
                                                           int ignore_prior, double* __restrict__ out, int ncu, int rot)
{
    using Geo = LaneRingGeom<P, NP>;
    extern __shared__ double lpc_ring[];
    const int lane = threadIdx.x & 63;
    const int round = (int)(blockIdx.x / (unsigned)ncu);
    const int part = ((int)(threadIdx.x >> 6) + ((rot >> (4 * (round & 3))) & 3)) & 3;      // 0 .. NC-1: consumers
    const int cons = part < Geo::NC ? part : (part - Geo::NC) / NP;
    long e = ((long)blockIdx.x * Geo::NC + cons) * 64 + lane;
    const bool live = e < B;
    if (!live) e = B - 1;
    double* ring = lpc_ring + cons * Geo::DOUBLES + lane;
    __shared__ double s_tab[MATH_TAB_N];                     // tables of the table-based exp / sincos (carma_math.h)
    math_tab_fill(s_tab);
    __syncthreads();
    if (part >= Geo::NC) {
        lane_produce<P, NP, REPDT>((part - Geo::NC) % NP, theta + e * d, ring, series, n, s_tab);
        return;
    }
    const double ll = logdensity_lane_ring<P, NP, REPDT, UNROLLED>(theta + e * d, q, series, n, pr, ignore_prior, ring);
    if (live) out[e] = ll;
}

// KalmanFilterp::Filter() for MANY models at once (round 4): one model per lane (carma_lane.h kfilter_lane), the series shared.
// par: per model [B][3 P + 2]: P roots (re, im) in normalised order, P MA coefficients, sigsqr, mu.  mv: [2 n][B] (mean rows, then
// variance rows; a wave's stores of one row are contiguous) -- transposed into the caller's [B][n] arrays by k_transpose_mv.
template <int P>
__global__ __launch_bounds__(64) void k_kfilter_carma_lane(const double* __restrict__ par, int B, const double4* __restrict__ series,
                                                          int n, double* __restrict__ mv, int* __restrict__ singular)
{
    __shared__ double s_tab[MATH_TAB_N];
    math_tab_fill(s_tab);
    __syncthreads();
    long e = (long)blockIdx.x * 64 + threadIdx.x;
    const bool live = e < B;
    if (!live) e = B - 1;
    const double* pm = par + e * (3 * P + 2);
    // idle lanes of the last wave write to a spare column (mv has B + 64 columns)
    const bool sing = kfilter_lane<P>(pm, pm + 2 * P, pm[3 * P], pm[3 * P + 1], series, n, s_tab, mv + (live ? e : (long)B + threadIdx.x),
                                      (long)B + 64);
    if (live) singular[e] = sing ? 1 : 0;
}

// in [rows][ld] (first `cols` columns of every row) -> out [cols][rows]
__global__ __launch_bounds__(256) void k_transpose_mv(const double* __restrict__ in, long ld, int rows, int cols, double* __restrict__ out)
{
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;    // bx: column block of `in`, by: row block
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads
    for (int j = ty; j < 32; j += 8)
        if (by + j < rows && bx + tx < cols) tile[j][tx] = in[(long)(by + j) * ld + bx + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (bx + j < cols && by + tx < rows) out[(long)(bx + j) * rows + by + tx] = tile[tx][j];
}

__global__ __launch_bounds__(64) void k_logdens_car1(const double* __restrict__ theta, int B,
                                                     const double4* __restrict__ series, int n, Prior pr,
                                                     double* __restrict__ out)
{
    long e = (long)blockIdx.x * 64 + threadIdx.x;
    if (e < B) out[e] = logdensity_car1(theta + 4 * e, series, n, pr);
}

// CAR(1), PARALLEL IN TIME (round 4): one evaluation per WAVE, for launches whose evaluations do not fill the lanes.
// With one evaluation per lane the n - 1 steps of a series are one lane's dependent chain -- 44 us for 270 data whatever the
// batch, slower than any CARMA(p >= 2) order at the same size.  But both recursions of KalmanFilter1 (kfilter.cpp:19-48) are
// compositions of maps that can be multiplied out in any grouping:
//   * the predicted process variance pv_k = var_k - e_k follows  pv_k = S (1 - rho^2) + rho^2 pv e / (pv + e)
//     (S = sigsqr / 2 omega, rho = exp(-omega dt_k), e = scale yerr_{k-1}^2, pv = pv_{k-1}) -- a MOEBIUS map of pv,
//     [[S (1 - rho^2) + rho^2 e, S (1 - rho^2) e], [1, e]]; all four entries are >= 0, so products of these matrices have no
//     cancellation and compose to rounding;
//   * given the variances, the mean follows the AFFINE map  mean_k = rho (1 - r) mean + rho r (y_{k-1} - mu),  r = pv / var.
// So: lane l takes a block of ceil((n - 1) / 64) consecutive steps; exponentials and the block's matrix product in the lanes
// side by side; an exclusive scan of the 64 block products across the wave (six shuffle stages, matrices renormalised by a
// power of two); every lane then walks its block from the exact starting value with the reference's own step formulas, which
// gives var_k, r_k and the block's affine map; a second scan for the mean; a last walk for the innovations.  The steps
// inside a block are the reference's arithmetic; only the values at the 63 block boundaries come out of the scans.
// Nothing is kept between the walks but the block's starting values -- the transition factors are evaluated once per walk
// (three exponentials per step instead of one: 0.5 us of 8 at n = 270) -- so the series may be of any length: 10^5 data in
// 0.5 ms where a lane of its own takes 16.
struct M22 {
    double a, b, c, d;
};
__device__ __forceinline__ M22 m22_mul(const M22& x, const M22& y)      // x after y
{
    return M22{fma(x.a, y.a, x.b * y.c), fma(x.a, y.b, x.b * y.d), fma(x.c, y.a, x.d * y.c), fma(x.c, y.b, x.d * y.d)};
}
__device__ __forceinline__ M22 m22_norm(const M22& x)                   // the map does not change under a common factor
{
    const double mx = fmax(fmax(fabs(x.a), fabs(x.b)), fmax(fabs(x.c), fabs(x.d)));
    int e;
    (void)frexp(mx, &e);
    if (!(mx > 0.0 && mx < 1.0 / 0.0)) e = 0;
    return M22{ldexp(x.a, -e), ldexp(x.b, -e), ldexp(x.c, -e), ldexp(x.d, -e)};
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// The wave-wide part: acc holds the lane's share of the log-likelihood sums afterwards.  MV: mean_k and var_k of every datum are
// stored as well (KalmanFilter1::Filter + GetMean / GetVar, kfilter.hpp:116-117, 222-245).
template <bool MV>
__device__ __forceinline__ void car1_scan_wave(double sigsqr, double omega, double mu, double ms, const double4* __restrict__ series,
                                               int n, LogLikAcc& acc, double* __restrict__ mean_out, double* __restrict__ var_out)
{
    const int lane = threadIdx.x & 63;
    const double S = sigsqr / (2.0 * omega);
    const int m = n - 1;                                      // steps 1 .. m
    const int c = (m + 63) / 64;                              // steps per lane
    const int k0 = 1 + lane * c, k1 = (k0 + c - 1 < m) ? k0 + c - 1 : m;      // this lane's steps (none when k0 > m)
    // ---- phase A: transition factors and the block's Moebius product
    M22 L{1.0, 0.0, 0.0, 1.0};
    for (int k = k0; k <= k1; k++) {
        const double rho = exp_neg(-1.0 * omega * series[k].x);
        const double e = series[k - 1].z * ms;
        const double g = S * (1.0 - rho * rho), r2 = rho * rho;
        const M22 Mk{fma(r2, e, g), g * e, 1.0, e};
        L = m22_norm(m22_mul(Mk, L));
    }
    // ---- phase B: exclusive scan across the lanes (Kogge-Stone: lane l ends up with the product of the blocks of lanes < l)
    M22 P = L;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        M22 Q{__shfl_up(P.a, o, 64), __shfl_up(P.b, o, 64), __shfl_up(P.c, o, 64), __shfl_up(P.d, o, 64)};
        if (lane >= o) P = m22_norm(m22_mul(P, Q));
    }
    M22 E{__shfl_up(P.a, 1, 64), __shfl_up(P.b, 1, 64), __shfl_up(P.c, 1, 64), __shfl_up(P.d, 1, 64)};
    if (lane == 0) E = M22{1.0, 0.0, 0.0, 1.0};
    // ---- phase C: the block's variances by the reference's step, from the exact value at its start; the block's affine map
    const double pv_start = fma(E.a, S, E.b) * recip(fma(E.c, S, E.d));   // pv_{k0 - 1}  (pv_0 = S)
    double pv = pv_start;
    double al = 1.0, be = 0.0;                                // mean_{k1} = al mean_{k0 - 1} + be
    acc.init();
    if (lane == 0) acc.add_var(S + series[0].z * ms);         // var_0 (kfilter.cpp:21-26)
    for (int k = k0; k <= k1; k++) {
        const double4 rp = series[k - 1];
        const double e_prev = rp.z * ms;
        const double var_prev = pv + e_prev;
        const double r = pv * recip(var_prev);                // var_ratio
        const double rho = exp_neg(-1.0 * omega * series[k].x);
        pv = S * (1.0 - rho * rho) + rho * rho * pv * (1.0 - r);     // previous_var of the next step (kfilter.cpp:36-44)
        acc.add_var(pv + series[k].z * ms);
        const double a_k = rho * (1.0 - r), b_k = rho * r * (rp.y - mu);
        be = fma(a_k, be, b_k);
        al = a_k * al;
    }
    // ---- phase D: scan of the affine maps; mean at the start of the block
    double pa = al, pb = be;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double qa = __shfl_up(pa, o, 64), qb = __shfl_up(pb, o, 64);
        if (lane >= o) {
            pb = fma(pa, qb, pb);
            pa = pa * qa;
        }
    }
    double mean = __shfl_up(pb, 1, 64);                       // mean_{k0 - 1}  (mean_0 = 0)
    if (lane == 0) mean = 0.0;
    // ---- phase E: innovations
    if (lane == 0) {
        const double i0 = series[0].y - mu;
        acc.chi2 += i0 * (i0 * recip(S + series[0].z * ms));
        if constexpr (MV) {
            mean_out[0] = 0.0;
            var_out[0] = S + series[0].z * ms;
        }
    }
    pv = pv_start;                                            // the same walk once more (same operations: same values)
    for (int k = k0; k <= k1; k++) {
        const double4 rp = series[k - 1];
        const double r = pv * recip(pv + rp.z * ms);
        const double4 rk = series[k];
        const double rho = exp_neg(-1.0 * omega * rk.x);
        mean = rho * mean + rho * r * ((rp.y - mu) - mean);   // kfilter.cpp:40
        pv = S * (1.0 - rho * rho) + rho * rho * pv * (1.0 - r);
        const double innov = (rk.y - mu) - mean;
        acc.chi2 += innov * (innov * recip(pv + rk.z * ms));
        if constexpr (MV) {
            mean_out[k] = mean;
            var_out[k] = pv + rk.z * ms;
        }
    }
}

__global__ __launch_bounds__(64) void k_logdens_car1_scan(const double* __restrict__ theta, int B,
                                                          const double4* __restrict__ series, int n, Prior pr,
                                                          double* __restrict__ out)
{
    const int lane = threadIdx.x;
    const long ev = blockIdx.x;
    const double* th = theta + 4 * ev;
    const double ysigma = th[0], ms = th[1], mu = th[2];
    const double omega = exp(th[3]);
    const double sigsqr = 2.0 * ysigma * ysigma * exp(th[3]);
    const bool ok = !((omega > pr.max_freq) || (omega < pr.min_freq) || (ysigma > pr.max_stdev) || (ysigma < 0) || (ms < 0.5) ||
                      (ms > 2.0));
    LogLikAcc acc;
    car1_scan_wave<false>(sigsqr, omega, mu, ms, series, n, acc, nullptr, nullptr);
    // ---- phase F: the wave's sums
    const double lg = wave_sum(log(acc.prod) + (double)acc.esum * LN2);
    const double chi = wave_sum(acc.chi2);
    double vmin = acc.vmin;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmin = fmin(vmin, __shfl_xor(vmin, o, 64));
    if (lane == 0) {
        double ll = -0.5 * lg - 0.5 * chi;
        if (!(vmin > 0.0)) ll = (vmin - vmin) / (vmin - vmin);
        ll += log_prior(ms, pr.measerr_dof);
        out[ev] = ok ? ll : -1.0 / 0.0;
    }
}

// KalmanFilter1(time, y, yerr, sigsqr, omega).Filter(): mean[n], var[n] of one model, the series cut across ONE wave's lanes
// (round 4; a lane of its own took 0.16 us per datum: 1.6 ms for 10^4 data)
__global__ __launch_bounds__(64) void k_kfilter_car1_scan(double sigsqr, double omega, const double4* __restrict__ series, int n,
                                                          double* __restrict__ mean, double* __restrict__ var)
{
    LogLikAcc acc;
    car1_scan_wave<true>(sigsqr, omega, 0.0, 1.0, series, n, acc, mean, var);
}

template <int P, int G>
__global__ __launch_bounds__(64) void k_kfilter_carma(const double* __restrict__ om_re_im, const double* __restrict__ ma,
                                                      double sigsqr, const double4* __restrict__ series, int n,
                                                      double* __restrict__ mean, double* __restrict__ var,
                                                      int* __restrict__ singular)
{
    __shared__ double4 xch[64];
    __shared__ double2 xch2[64];
    const int tid = threadIdx.x;
    Grp<G> g{xch, tid & 63, xch2};
    Model<P> m;
    model_from_roots<P, G>(g, om_re_im, ma, sigsqr, m);
    bool sing;
    // every group of the wave runs the same evaluation and stores the same values
    filter_run<P, G, true>(g, m, series, n, mean, var, &sing);
    if (tid == 0) *singular = sing ? 1 : 0;
}

__global__ __launch_bounds__(64) void k_kfilter_car1(double sigsqr, double omega, const double4* __restrict__ series,
                                                     int n, double* __restrict__ mean, double* __restrict__ var)
{
    if (threadIdx.x == 0) car1_filter(sigsqr, omega, 0.0, 1.0, series, n, true, mean, var);
}

// KalmanFilterp::Predict for M times at once: one lane group per prediction time.
template <int P, int G>
__global__ __launch_bounds__(64) void k_predict_carma(const double* __restrict__ om_re_im, const double* __restrict__ ma,
                                                      double sigsqr, const double4* __restrict__ series, int n,
                                                      const double* __restrict__ tpred, int M, double* __restrict__ pmean,
                                                      double* __restrict__ pvar, int* __restrict__ singular)
{
    __shared__ double4 xch[64];
    __shared__ double2 xch2[64];
    const int tid = threadIdx.x;
    Grp<G> g{xch, tid & 63, xch2};
    long e = ((long)blockIdx.x * 64 + tid) / G;
    const bool live = e < M;
    if (!live) e = M - 1;
    Model<P> m;
    model_from_roots<P, G>(g, om_re_im, ma, sigsqr, m);
    double pm, pv;
    bool sing;
    predict_run<P, G>(g, m, series, n, tpred[e], &pm, &pv, &sing);
    if (live && g.lane() == 0) {
        pmean[e] = pm;
        pvar[e] = pv;
        if (sing) *singular = 1;
    }
}

__global__ __launch_bounds__(64) void k_predict_car1(double sigsqr, double omega, const double4* __restrict__ series, int n,
                                                     const double* __restrict__ tpred, int M, double* __restrict__ pmean,
                                                     double* __restrict__ pvar)
{
    const long e = (long)blockIdx.x * 64 + threadIdx.x;
    if (e < M) predict_car1(sigsqr, omega, series, n, tpred[e], pmean + e, pvar + e);
}

// carma_process for `npaths` independent paths at once: one lane group per path.
template <int P, int G>
__global__ __launch_bounds__(64) void k_simulate_carma(const double* __restrict__ om_re_im, const double* __restrict__ ma,
                                                       double sigsqr, const double* __restrict__ times, int n, int npaths,
                                                       unsigned seed0, unsigned seed1, unsigned path0, double* __restrict__ out,
                                                       int* __restrict__ singular)
{
    __shared__ double4 xch[64];
    __shared__ double2 xch2[64];
    const int tid = threadIdx.x;
    Grp<G> g{xch, tid & 63, xch2};
    long e = ((long)blockIdx.x * 64 + tid) / G;
    const bool live = e < npaths;
    if (!live) e = npaths - 1;
    Model<P> m;
    model_from_roots<P, G>(g, om_re_im, ma, sigsqr, m);
    RngKey key{seed0, seed1, path0 + (unsigned)e};
    bool sing;
    // (shadow groups past the end redo the last path and write the same values)
    simulate_run<P, G>(g, m, times, n, key, out + e * (long)n, &sing);
    if (live && g.lane() == 0 && sing) *singular = 1;
}

__global__ __launch_bounds__(64) void k_simulate_car1(double sigsqr, double omega, const double* __restrict__ times, int n,
                                                      int npaths, unsigned seed0, unsigned seed1, unsigned path0,
                                                      double* __restrict__ out)
{
    const long e = (long)blockIdx.x * 64 + threadIdx.x;
    if (e < npaths) simulate_car1(sigsqr, omega, times, n, RngKey{seed0, seed1, path0 + (unsigned)e}, out + e * (long)n);
}

// ---------------------------------------------------------------------------------------------
// CUs of the CURRENT device, cached per device (contexts may live on different, or differently partitioned, devices)
int device_cus()
{
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// Largest launch (in workgroups of four evaluations) that takes the wave pipeline of carma_pipe3l.h: three workgroups
// per CU (measured, tools/tput_probe.py).  CARMA_TUNE_P3L_ROWS overrides it for such measurements; read once.
static long p3l_max_rows()
{
    static const long tune = [] {
        const char* e = getenv("CARMA_TUNE_P3L_ROWS");
        return e ? atol(e) : -1L;
    }();
    return tune >= 0 ? tune : 3L * device_cus();      // three workgroups per CU (registers and LDS allow exactly that)
}

// Launch shape for B evaluations of order P (one table for the launcher and for carma_logdensity_kernel_name)
enum class LdShape { P3L, PC1, PC2, PLAIN1, PLAIN4, LANE, LPC, WIN, WIN2 };
// The launch-shape switches (carma_launch.h): environment read once, atomics afterwards.
static const char* const TUNE_NAMES[TUNE_COUNT] = {"CARMA_TUNE_WIN_ROWS", "CARMA_TUNE_WIN2_EVALS", "CARMA_TUNE_PT_ROW_WIN"};
static std::atomic<long> g_tune[TUNE_COUNT];
static std::atomic<int> g_tune_init{0};
static void tune_init()
{
    if (g_tune_init.load(std::memory_order_acquire) == 2) return;
    int expect = 0;
    if (g_tune_init.compare_exchange_strong(expect, 1)) {
        for (int i = 0; i < TUNE_COUNT; i++) {
            const char* e = getenv(TUNE_NAMES[i]);
            g_tune[i].store(e ? atol(e) : TUNE_UNSET, std::memory_order_relaxed);
        }
        g_tune_init.store(2, std::memory_order_release);
    } else {
        while (g_tune_init.load(std::memory_order_acquire) != 2) {}
    }
}
long tune_get(int which)
{
    tune_init();
    return which >= 0 && which < TUNE_COUNT ? g_tune[which].load(std::memory_order_relaxed) : TUNE_UNSET;
}
int tune_set(const char* name, long value)
{
    if (!name) return -1;
    tune_init();
    for (int i = 0; i < TUNE_COUNT; i++)
        if (!strcmp(name, TUNE_NAMES[i]) || !strcmp(name, TUNE_NAMES[i] + 11)) {       // (+ 11: behind "CARMA_TUNE_")
            g_tune[i].store(value, std::memory_order_relaxed);
            return 0;
        }
    return -1;
}
// Largest launch (in workgroups of four evaluations) that takes the windowed wave pipeline (carma_pipew.h): one workgroup per CU
// (measured per order at 1024 evaluations, profiles/r05/window_pipeline_v1.txt: 1-10 % ahead of the one-datum pipeline; with two
// or three workgroups per CU that one is ahead).  CARMA_TUNE_WIN_ROWS overrides (0: never, also for the two-sided kernel), so that
// one test process can run both pipelines (carma_tune_set).
static long win_max_rows()
{
    const long v = tune_get(TUNE_WIN_ROWS);
    return v != TUNE_UNSET ? v : (long)device_cus();
}
// Largest launch (in EVALUATIONS) that takes the two-sided window pipeline: three workgroups of two evaluations per CU (measured
// per order, profiles/r06/w2_sizes_v1.txt: p = 5 at 1280 / 1536 evaluations 29.6 us against the one-datum pipeline's 35.3 / 35.4,
// p = 7 34.4 against 37.4, p = 3 23.6 against 26.9; at 2048 -- four per CU -- 38.2 against 35.5, 47.8 against 37.5, 31.3 against 27.0).
// CARMA_TUNE_WIN2_EVALS overrides (0: never).
static long win2_max_evals()
{
    const long v = tune_get(TUNE_WIN2_EVALS);
    return v != TUNE_UNSET ? v : 6L * device_cus();
}
// (an override of CARMA_TUNE_WIN_ROWS also lifts the series criterion: the tests force the window pipelines onto series that fail it)
static bool win_forced() { return tune_get(TUNE_WIN_ROWS) != TUNE_UNSET; }
// longest series the two-sided kernel keeps in LDS beside its rings (24 bytes a datum; 5000: 137 KiB, one workgroup per CU)
constexpr int W2_MAX_N = 5000;
// smallest launch that takes one evaluation per lane (measured: tools/tput_probe.py; CARMA_TUNE_LANE_MIN overrides, read once)
static long lane_min_evals(int p = 5)
{
    static const long tune = [] {
        const char* e = getenv("CARMA_TUNE_LANE_MIN");
        return e ? atol(e) : -1L;
    }();
    if (tune >= 0) return tune;
    // (p = 7, one producer-wave workgroup per CU: the lane-group kernel keeps 16 385 ... 28 671 evaluations -- 257 us at 24 576
    // against 309 for the lone wave; 374 against 323 at 32 768: profiles/r04/lpc_orders_v1.txt)
    if (p == 7) return 112L * device_cus();
    return 64L * 4 * device_cus() * 3 / 8;  // 3/8 of a wave per SIMD (24 576 on 256 CUs): 206 vs 212-236 us there
}
// Launches (lpc_min, lpc_max] take the lane kernel with producer waves, k_logdens_carma_lpc<P,3> (measured per order:
// tools/lpc_probe.sh, profiles/r03/lpc_orders_v1.txt).  Below, the lane-group kernels still have at most one wave per SIMD
// and are faster; above -- more than three workgroups per CU, or more than the registers allow (two at p = 5, 6, one at
// p = 7) -- the SIMDs are full either way and the plain lane kernel does the same work without the LDS traffic.
// CARMA_TUNE_LPC_MIN / _MAX override (read once).
static long lpc_tune(const char* name, long dflt)
{
    const char* e = getenv(name);
    return e ? atol(e) : dflt;
}
static int lpc_rot()
{
    static const long v = lpc_tune("CARMA_TUNE_LPC_ROT", 0x0202);
    return (int)v;
}
template <int P>
static long lpc_min_evals()
{
    static const long v = lpc_tune("CARMA_TUNE_LPC_MIN", -1);
    if (v >= 0) return v;
    // Measured per order (profiles/r04/lpc_orders_v1.txt, round 4: with the table-based exp / sincos and the one-basic-block step
    // the consumer's stream is short enough that for the low orders this kernel wins right above the wave pipeline's range):
    //   p = 2, 3   from 12 x #CUs evaluations (3073):  50 / 62 us flat up to 16 384 against 62-68 / 67-89 (pair kernel, lane groups)
    //   p = 4      from 32 x #CUs (8193):              88 us against 90-93
    //   p >= 5     from 32 x #CUs (8193), as in round 3: below, the lane-group kernel's 104-112 us are ahead of 107-160
    // Round 5, the consumer takes a ring buffer as one basic block (LaneFactorsRing, UNROLLED: 110 -> 92 us at p = 5):
    //   p = 5      from 16 x #CUs (4097): 89-90 us flat against 99-103 for the lane groups (8192: 8.0 -> 9.1e7 evals/s)
    //   p = 4, 6, 7  as before (72 against 71-73; 118 against 104-106; 142 against 109-113: profiles/r05/lpc_min_probe_v1.txt)
    if (P <= 3) return 12L * device_cus();
    if (P == 5) return 16L * device_cus();
    return 32L * device_cus();
}
template <int P>
static long lpc_max_evals()
{
    static const long v = lpc_tune("CARMA_TUNE_LPC_MAX", -1);
    if (v >= 0) return v;
    // workgroups of this kernel a CU holds (registers, LDS), cached per device like device_cus()
    static std::atomic<int> blocks[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        (void)hipGetLastError();
        return 64L * device_cus();
    }
    int nb = blocks[dev].load(std::memory_order_relaxed);
    if (nb <= 0) {
        const void* kern = reinterpret_cast<const void*>(&k_logdens_carma_lpc<P, 3>);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, LaneRingGeom<P, 3>::BYTES) != hipSuccess || nb <= 0) {
            (void)hipGetLastError();                          // the query's error must not surface as the next launch's
            nb = 1;
        }
        blocks[dev].store(nb, std::memory_order_relaxed);
    }
    return 64L * (nb < 3 ? nb : 3) * device_cus();
}
template <int P>
static LdShape logdens_shape(long B, int n, int series_flags)
{
    constexpr int EPW = 64 / GroupOf<P>::value;       // evaluations per wave of the G-lane kernels
    const long waves = (B + EPW - 1) / EPW;
    const long rows = (B + 3) / 4;                    // workgroups with one evaluation per 16-lane DPP row
    const bool w2ok = (series_flags & SERIES_WINDOW2_OK) || ((series_flags & SERIES_WINDOW2_SMALL) && B <= 2L * device_cus());
    if (B <= win2_max_evals() && win_max_rows() > 0 && n >= 16 && (w2ok || win_forced()))
        return LdShape::WIN2;                         // (CARMA_TUNE_WIN_ROWS = 0: no window pipeline of either kind)
    if (rows <= win_max_rows() && n >= 8 && ((series_flags & SERIES_WINDOW_OK) || win_forced())) return LdShape::WIN;
    if (rows <= p3l_max_rows() && n >= 8) return LdShape::P3L;
    if (B > lpc_min_evals<P>() && B <= lpc_max_evals<P>() && n >= 8) return LdShape::LPC;
    if (B >= lane_min_evals(P)) return LdShape::LANE;
    // few evaluations in flight: one wave's instruction stream is the run time, so split it (consumer + rho producer,
    // carma_ring.h).  Beyond 512 waves (two rounds of workgroups) the plain kernel with pair-shared exp/sincos is
    // ahead: 104 vs 111 us at 6144 evaluations (tools/midrange_probe.py)
    if (waves <= 256 && n >= 8) return LdShape::PC1;
    if (waves <= 512 && n >= 8) return LdShape::PC2;
    return waves <= 2048 ? LdShape::PLAIN1 : LdShape::PLAIN4;
}

template <int P>
static hipError_t launch_logdens_p(const double* theta, int B, int d, int q, const double4* series, int n,
                                   const Prior& pr, int ignore_prior, double* out, hipStream_t st, int series_flags)
{
    const bool repeated_dt = (series_flags & SERIES_REPEATED_DT) != 0;
    constexpr int G = GroupOf<P>::value;
    constexpr int EPW = 64 / G;   // evaluations per wave
    const long waves = ((long)B + EPW - 1) / EPW;
    const long rows = ((long)B + 3) / 4;
    auto launch_pc = [&](auto kern, long npairs, int pairs) -> hipError_t {
        const size_t lds = (size_t)pairs * (128 * sizeof(double4) + RingGeom<P>::BYTES);
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            160 * 1024);
        if (ea != hipSuccess) return ea;
        hipLaunchKernelGGL(kern, dim3((unsigned)((npairs + pairs - 1) / pairs)), dim3(128 * pairs), lds, st, theta, B, d, q,
                           series, n, pr, ignore_prior, out);
        return hipGetLastError();
    };
    switch (logdens_shape<P>(B, n, series_flags)) {
        case LdShape::P3L:
            // covariance wave + mean wave + two producer waves per four evaluations, co-rotating frame
            // (carma_pipe3l.h); 42 KiB of LDS: up to three workgroups per CU
            hipLaunchKernelGGL((k_logdens_carma_p3l<P>), dim3((unsigned)rows), dim3(256), Pipe3LGeom<P>::BYTES, st, theta, B, d, q,
                               series, n + p3l_pad(n), pr, ignore_prior, out, device_cus(), p3l_pad(n));
            return hipGetLastError();
        case LdShape::WIN:
            hipLaunchKernelGGL((k_logdens_carma_w<P>), dim3((unsigned)rows), dim3(256), PipeWGeom<P>::BYTES, st, theta, B, d, q,
                               series, n, pr, ignore_prior, out, device_cus());
            return hipGetLastError();
        case LdShape::WIN2:
        {
            const bool sl = n <= W2_MAX_N;                     // (longer: from global memory, the rings alone in LDS)
            const size_t lds = sl ? PipeWGeom<P>::bytes_with_series(n) : PipeWGeom<P>::BYTES;
            const long wgs = ((long)B + 1) / 2;
            auto go = [&](auto kern) -> hipError_t {
                if (lds > 64 * 1024) {
                    hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    if (ea != hipSuccess) return ea;
                }
                hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(256), lds, st, theta, B, d, q, series, n, pr, ignore_prior, out, device_cus());
                return hipGetLastError();
            };
            if (!sl) return go(&k_logdens_carma_w2<P, false, false>);
            return wgs > device_cus() ? go(&k_logdens_carma_w2<P, true>) : go(&k_logdens_carma_w2<P, false>);
        }
        case LdShape::PC1: return launch_pc(&k_logdens_carma_pc<P, G, 1>, waves, 1);
        case LdShape::PC2: return launch_pc(&k_logdens_carma_pc<P, G, 2>, waves, 2);
        case LdShape::PLAIN1:
            // spread the waves over as many CUs as possible (1 wave per workgroup) until the chip is covered
            if (repeated_dt)
                hipLaunchKernelGGL((k_logdens_carma<P, G, 1, true>), dim3((unsigned)waves), dim3(64), 0, st, theta, B, d, q, series,
                                   n, pr, ignore_prior, out);
            else
                hipLaunchKernelGGL((k_logdens_carma<P, G, 1>), dim3((unsigned)waves), dim3(64), 0, st, theta, B, d, q, series, n,
                                   pr, ignore_prior, out);
            return hipGetLastError();
        case LdShape::LANE:
            // (half-filled waves -- 32 evaluations per wave, twice the waves -- are no faster per wave: an FP64 instruction
            // takes its four cycles whatever the execution mask; 65 536 evaluations 344 vs 233 us)
            if (repeated_dt)
                hipLaunchKernelGGL((k_logdens_carma_lane<P, true>), dim3((unsigned)(((long)B + 63) / 64)), dim3(64), 0, st, theta, B, d, q,
                                   series, n, pr, ignore_prior, out);
            else
                hipLaunchKernelGGL((k_logdens_carma_lane<P>), dim3((unsigned)(((long)B + 63) / 64)), dim3(64), 0, st, theta, B, d, q,
                                   series, n, pr, ignore_prior, out);
            return hipGetLastError();
        case LdShape::LPC: {
            using Geo = LaneRingGeom<P, 3>;
            static_assert(Geo::BYTES <= 64 * 1024, "within the LDS a launch may ask for without raising the kernel's limit");
            if (repeated_dt)
                hipLaunchKernelGGL((k_logdens_carma_lpc<P, 3, true>), dim3((unsigned)(((long)B + 63) / 64)), dim3(256), Geo::BYTES, st, theta,
                                   B, d, q, series, n, pr, ignore_prior, out, device_cus(), lpc_rot());
            else if ((long)B <= 64L * device_cus())
                hipLaunchKernelGGL((k_logdens_carma_lpc<P, 3, false, true>), dim3((unsigned)(((long)B + 63) / 64)), dim3(256), Geo::BYTES,
                                   st, theta, B, d, q, series, n, pr, ignore_prior, out, device_cus(), lpc_rot());
            else
                hipLaunchKernelGGL((k_logdens_carma_lpc<P, 3>), dim3((unsigned)(((long)B + 63) / 64)), dim3(256), Geo::BYTES, st, theta, B, d,
                                   q, series, n, pr, ignore_prior, out, device_cus(), lpc_rot());
            return hipGetLastError();
        }
        case LdShape::PLAIN4:
            if (repeated_dt)
                hipLaunchKernelGGL((k_logdens_carma<P, G, 4, true>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, theta, B,
                                   d, q, series, n, pr, ignore_prior, out);
            else
                hipLaunchKernelGGL((k_logdens_carma<P, G, 4>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, theta, B, d, q,
                                   series, n, pr, ignore_prior, out);
            return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

template <int P>
static int logdens_name_p(long B, int n, char* buf, int len, int series_flags)
{
    const bool repeated_dt = (series_flags & SERIES_REPEATED_DT) != 0;
    const char* dtc = repeated_dt ? ",true" : "";
    constexpr int G = GroupOf<P>::value;
    switch (logdens_shape<P>(B, n, series_flags)) {
        case LdShape::WIN: return snprintf(buf, len, "k_logdens_carma_w<%d>", P);
        case LdShape::WIN2: return snprintf(buf, len, "k_logdens_carma_w2<%d>", P);
        case LdShape::P3L: return snprintf(buf, len, "k_logdens_carma_p3l<%d>", P);
        case LdShape::PC1: return snprintf(buf, len, "k_logdens_carma_pc<%d,%d,1>", P, G);
        case LdShape::PC2: return snprintf(buf, len, "k_logdens_carma_pc<%d,%d,2>", P, G);
        case LdShape::PLAIN1: return snprintf(buf, len, "k_logdens_carma<%d,%d,1%s>", P, G, dtc);
        case LdShape::PLAIN4: return snprintf(buf, len, "k_logdens_carma<%d,%d,4%s>", P, G, dtc);
        case LdShape::LANE: return snprintf(buf, len, "k_logdens_carma_lane<%d%s>", P, dtc);
        // (up to one workgroup per CU its UNROLLED instantiation, <P,3,false,true>: the same kernel, the same bits)
        case LdShape::LPC: return snprintf(buf, len, "k_logdens_carma_lpc<%d,3%s>", P, dtc);
    }
    return -1;
}

int logdens_kernel_name(int p, long B, int n, char* buf, int len, int series_flags)
{
    switch (p) {
        case 1: return snprintf(buf, len, "k_logdens_car1");     // (or its parallel-in-time form: launch_logdens_car1)
        case 2: return logdens_name_p<2>(B, n, buf, len, series_flags);
        case 3: return logdens_name_p<3>(B, n, buf, len, series_flags);
        case 4: return logdens_name_p<4>(B, n, buf, len, series_flags);
        case 5: return logdens_name_p<5>(B, n, buf, len, series_flags);
        case 6: return logdens_name_p<6>(B, n, buf, len, series_flags);
        case 7: return logdens_name_p<7>(B, n, buf, len, series_flags);
        default: return -1;
    }
}

hipError_t launch_logdens_carma(int p, const double* theta, int B, int d, int q, const double4* series, int n,
                                const Prior& pr, int ignore_prior, double* out, hipStream_t st, int series_flags)
{
    (void)hipGetLastError();   // HIP's last-error is sticky: drop anything left by earlier calls
    switch (p) {
        case 2: return launch_logdens_p<2>(theta, B, d, q, series, n, pr, ignore_prior, out, st, series_flags);
        case 3: return launch_logdens_p<3>(theta, B, d, q, series, n, pr, ignore_prior, out, st, series_flags);
        case 4: return launch_logdens_p<4>(theta, B, d, q, series, n, pr, ignore_prior, out, st, series_flags);
        case 5: return launch_logdens_p<5>(theta, B, d, q, series, n, pr, ignore_prior, out, st, series_flags);
        case 6: return launch_logdens_p<6>(theta, B, d, q, series, n, pr, ignore_prior, out, st, series_flags);
        case 7: return launch_logdens_p<7>(theta, B, d, q, series, n, pr, ignore_prior, out, st, series_flags);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_logdens_car1(const double* theta, int B, const double4* series, int n, const Prior& pr, double* out,
                               hipStream_t st)
{
    (void)hipGetLastError();   // HIP's last-error is sticky: drop anything left by earlier calls
    // up to 48 evaluations per CU the series is cut across a wave's lanes (k_logdens_car1_scan); beyond, the lanes are worth more
    // as evaluations (CARMA_TUNE_CAR1_SCAN_MAX overrides: measured, tools/car1_probe.py)
    static const long scan_max = [] {
        const char* e = getenv("CARMA_TUNE_CAR1_SCAN_MAX");
        return e ? atol(e) : -1L;
    }();
    const long smax = scan_max >= 0 ? scan_max : 48L * device_cus();   // (n = 270: 39 against 45 us at 12 288, 52 against 45 at 16 384)
    // (both forms' times are proportional to the series' length: the break-even does not move with it)
    if (B <= smax && n >= 64) {
        hipLaunchKernelGGL(k_logdens_car1_scan, dim3((unsigned)B), dim3(64), 0, st, theta, B, series, n, pr, out);
        return hipGetLastError();
    }
    const unsigned blocks = (unsigned)(((long)B + 63) / 64);
    hipLaunchKernelGGL(k_logdens_car1, dim3(blocks), dim3(64), 0, st, theta, B, series, n, pr, out);
    return hipGetLastError();
}

template <int P>
static hipError_t launch_kfilter_p(const double* om, const double* ma, double sigsqr, const double4* series, int n,
                                   double* mean, double* var, int* singular, hipStream_t st)
{
    constexpr int G = GroupOf<P>::value;
    hipLaunchKernelGGL((k_kfilter_carma<P, G>), dim3(1), dim3(64), 0, st, om, ma, sigsqr, series, n, mean, var, singular);
    return hipGetLastError();
}

hipError_t launch_kfilter_carma(int p, const double* om, const double* ma, double sigsqr, const double4* series, int n,
                                double* mean, double* var, int* singular, hipStream_t st)
{
    (void)hipGetLastError();   // HIP's last-error is sticky: drop anything left by earlier calls
    switch (p) {
        case 2: return launch_kfilter_p<2>(om, ma, sigsqr, series, n, mean, var, singular, st);
        case 3: return launch_kfilter_p<3>(om, ma, sigsqr, series, n, mean, var, singular, st);
        case 4: return launch_kfilter_p<4>(om, ma, sigsqr, series, n, mean, var, singular, st);
        case 5: return launch_kfilter_p<5>(om, ma, sigsqr, series, n, mean, var, singular, st);
        case 6: return launch_kfilter_p<6>(om, ma, sigsqr, series, n, mean, var, singular, st);
        case 7: return launch_kfilter_p<7>(om, ma, sigsqr, series, n, mean, var, singular, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_kfilter_batch(int p, const double* par, int B, const double4* series, int n, double* mv, int* singular,
                                double* mean, double* var, hipStream_t st)
{
    (void)hipGetLastError();
    const unsigned blocks = (unsigned)(((long)B + 63) / 64);
    switch (p) {
#define CARMA_KFB(N) \
    case N: hipLaunchKernelGGL((k_kfilter_carma_lane<N>), dim3(blocks), dim3(64), 0, st, par, B, series, n, mv, singular); break;
        CARMA_KFB(2) CARMA_KFB(3) CARMA_KFB(4) CARMA_KFB(5) CARMA_KFB(6) CARMA_KFB(7)
#undef CARMA_KFB
        default: return hipErrorInvalidValue;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const long ld = (long)B + 64;
    const dim3 grid((unsigned)((B + 31) / 32), (unsigned)((n + 31) / 32));
    hipLaunchKernelGGL(k_transpose_mv, grid, dim3(256), 0, st, mv, ld, n, B, mean);
    hipLaunchKernelGGL(k_transpose_mv, grid, dim3(256), 0, st, mv + (size_t)n * ld, ld, n, B, var);
    return hipGetLastError();
}

template <int P>
static hipError_t launch_predict_p(const double* om, const double* ma, double sigsqr, const double4* series, int n,
                                   const double* tpred, int M, double* pmean, double* pvar, int* singular, hipStream_t st)
{
    constexpr int G = GroupOf<P>::value;
    const unsigned blocks = (unsigned)(((long)M * G + 63) / 64);
    hipLaunchKernelGGL((k_predict_carma<P, G>), dim3(blocks), dim3(64), 0, st, om, ma, sigsqr, series, n, tpred, M, pmean,
                       pvar, singular);
    return hipGetLastError();
}

hipError_t launch_predict_carma(int p, const double* om, const double* ma, double sigsqr, const double4* series, int n,
                                const double* tpred, int M, double* pmean, double* pvar, int* singular, hipStream_t st)
{
    (void)hipGetLastError();   // HIP's last-error is sticky: drop anything left by earlier calls
    switch (p) {
        case 2: return launch_predict_p<2>(om, ma, sigsqr, series, n, tpred, M, pmean, pvar, singular, st);
        case 3: return launch_predict_p<3>(om, ma, sigsqr, series, n, tpred, M, pmean, pvar, singular, st);
        case 4: return launch_predict_p<4>(om, ma, sigsqr, series, n, tpred, M, pmean, pvar, singular, st);
        case 5: return launch_predict_p<5>(om, ma, sigsqr, series, n, tpred, M, pmean, pvar, singular, st);
        case 6: return launch_predict_p<6>(om, ma, sigsqr, series, n, tpred, M, pmean, pvar, singular, st);
        case 7: return launch_predict_p<7>(om, ma, sigsqr, series, n, tpred, M, pmean, pvar, singular, st);
        default: return hipErrorInvalidValue;
    }
}

template <int P>
static hipError_t launch_simulate_p(const double* om, const double* ma, double sigsqr, const double* times, int n, int npaths,
                                    unsigned seed0, unsigned seed1, unsigned path0, double* out, int* singular, hipStream_t st)
{
    constexpr int G = GroupOf<P>::value;
    const unsigned blocks = (unsigned)(((long)npaths * G + 63) / 64);
    hipLaunchKernelGGL((k_simulate_carma<P, G>), dim3(blocks), dim3(64), 0, st, om, ma, sigsqr, times, n, npaths, seed0, seed1,
                       path0, out, singular);
    return hipGetLastError();
}

hipError_t launch_simulate_carma(int p, const double* om, const double* ma, double sigsqr, const double* times, int n,
                                 int npaths, unsigned seed0, unsigned seed1, unsigned path0, double* out, int* singular,
                                 hipStream_t st)
{
    (void)hipGetLastError();
    switch (p) {
        case 2: return launch_simulate_p<2>(om, ma, sigsqr, times, n, npaths, seed0, seed1, path0, out, singular, st);
        case 3: return launch_simulate_p<3>(om, ma, sigsqr, times, n, npaths, seed0, seed1, path0, out, singular, st);
        case 4: return launch_simulate_p<4>(om, ma, sigsqr, times, n, npaths, seed0, seed1, path0, out, singular, st);
        case 5: return launch_simulate_p<5>(om, ma, sigsqr, times, n, npaths, seed0, seed1, path0, out, singular, st);
        case 6: return launch_simulate_p<6>(om, ma, sigsqr, times, n, npaths, seed0, seed1, path0, out, singular, st);
        case 7: return launch_simulate_p<7>(om, ma, sigsqr, times, n, npaths, seed0, seed1, path0, out, singular, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_simulate_car1(double sigsqr, double omega, const double* times, int n, int npaths, unsigned seed0,
                                unsigned seed1, unsigned path0, double* out, hipStream_t st)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_simulate_car1, dim3((unsigned)(((long)npaths + 63) / 64)), dim3(64), 0, st, sigsqr, omega, times, n,
                       npaths, seed0, seed1, path0, out);
    return hipGetLastError();
}

hipError_t launch_predict_car1(double sigsqr, double omega, const double4* series, int n, const double* tpred, int M,
                               double* pmean, double* pvar, hipStream_t st)
{
    (void)hipGetLastError();   // HIP's last-error is sticky: drop anything left by earlier calls
    const unsigned blocks = (unsigned)(((long)M + 63) / 64);
    hipLaunchKernelGGL(k_predict_car1, dim3(blocks), dim3(64), 0, st, sigsqr, omega, series, n, tpred, M, pmean, pvar);
    return hipGetLastError();
}

hipError_t launch_kfilter_car1(double sigsqr, double omega, const double4* series, int n, double* mean, double* var,
                               hipStream_t st)
{
    (void)hipGetLastError();   // HIP's last-error is sticky: drop anything left by earlier calls
    if (n >= 64)
        hipLaunchKernelGGL(k_kfilter_car1_scan, dim3(1), dim3(64), 0, st, sigsqr, omega, series, n, mean, var);
    else
        hipLaunchKernelGGL(k_kfilter_car1, dim3(1), dim3(64), 0, st, sigsqr, omega, series, n, mean, var);
    return hipGetLastError();
}

}  // namespace carma
