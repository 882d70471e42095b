// carma_ring.h -- producer/consumer split of the Kalman step for the LATENCY regime (gfx950 only).
//
// With <= ~1000 evaluations in flight (BASELINE configs 2 and 3: 1024 evaluations = 128 waves on a
// chip with 1024 SIMDs) the run time is the instruction stream of ONE wave: n-1 dependent steps.
// Everything in a step that does not depend on the filter state -- the transition factors
// rho_r = exp(omega_r dt_k) (kfilter.cpp:200) and their products R_rj = rho_r conj(rho_j)
// (kfilter.cpp:204), i.e. the software exp/sincos and 4p multiplies -- is moved to a second
// wave of the same workgroup, running on another SIMD of the CU:
//
//   producer wave: per step, lane (group, r): rho_r -> LDS ring slot [buf][s][lane]  (16 B)
//   consumer wave: the state recursion only; reads the p ring entries of its group per step
//
// (Measured on MI355X: one wave issues an FP64 VALU instruction every ~4.8 cycles, 8 when
// dependent, v_rsq_f64 20, LDS write->read 134 -- so the run time of the latency regime is
// essentially 5 cycles x the instruction count of the critical wave.  A producer that also formed
// the products rho_r conj(rho_j) was slower than the consumer; the products stay in the consumer.)
//
// The ring is double buffered in chunks of C steps with one __syncthreads() per chunk: barrier c is
// passed by the producer after it wrote chunk c and by the consumer before it reads chunk c, so
// the producer always works one chunk ahead and never overwrites a chunk that is being read.
#pragma once
#include <hip/hip_runtime.h>

#include "carma_core.h"
#include "grp_device.h"

namespace carma {

template <int P>
struct RingGeom {
    static constexpr int C = 16;                                    // passes per chunk
    static constexpr int SLOT = 64;                                 // Cx entries per pass (one per lane)
    static constexpr int ENTRIES = 2 * C * SLOT + 2 * C;            // Cx-sized entries: rho ring + series records
    static constexpr size_t BYTES = (size_t)ENTRIES * sizeof(Cx);   // 32.5 KiB
    static constexpr int REC_OFF = 2 * C * SLOT;                    // series records {y, yerr^2}: double2[2][C]
};

// The loop of filter_loop_real makes n passes kk = 1..n; pass kk needs the series record kk-1
// (y, yerr^2 for var_{kk-1}) and, for kk < n, the factors rho(dt_kk).  The producer stages both, so
// the consumer loop has no scalar loads at all (scalar loads share the LDS wait counter and return
// out of order, which would force every LDS wait of the consumer to also cover an L2 round trip).
__device__ __forceinline__ double readlane_f64(double v, int lane_uniform)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane_uniform);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane_uniform);
    return __hiloint2double(hi, lo);
}

template <int P, int G>
__device__ __forceinline__ void ring_produce(const Grp<G>& g, const Cx w, const double4* __restrict__ series, int n,
                                             Cx* __restrict__ ring)
{
    constexpr int C = RingGeom<P>::C;
    const int lane = g.lane64;
    const int nchunks = (n - 1 + C - 1) / C;                 // passes 1 .. n-1
    double2* recs = reinterpret_cast<double2*>(ring + RingGeom<P>::REC_OFF);
    // lane s (< C) fetches what pass s of a chunk needs -- record kk-1 and dt_kk -- with vector loads
    // one chunk ahead, so no load latency sits in front of the exp/sincos stream
    const int ls = lane < C ? lane : C - 1;
    auto clampi = [n](int i) { return i < n ? i : n - 1; };
    double4 rec_n = series[clampi(ls)];
    double dt_n = series[clampi(1 + ls)].x;
    for (int c = 0; c < nchunks; c++) {
        Cx* buf = ring + (size_t)(c & 1) * C * RingGeom<P>::SLOT;
        const double4 rec_c = rec_n;
        const double dt_c = dt_n;
        const int kk0 = 1 + c * C;
        rec_n = series[clampi(kk0 + C + ls - 1)];
        dt_n = series[clampi(kk0 + C + ls)].x;
        if (lane < C && kk0 + lane < n) recs[(c & 1) * C + lane] = make_double2(rec_c.y, rec_c.z);
#pragma unroll 1
        for (int s = 0; s < C; s++) {
            const double dt = readlane_f64(dt_c, s);
            if (kk0 + s < n) {
                Cx rho;
                cexp_step(w.re, w.im, dt, &rho.re, &rho.im);
                buf[(size_t)s * RingGeom<P>::SLOT + lane] = rho;
            }
        }
        __syncthreads();
    }
}

// Consumer side: the RhoSrc policy of filter_loop that takes the factors from the ring.
// (Reading the entries one step ahead was measured slower: 141 vs 127 us per 1024-eval launch.)
template <int P, int G>
struct RhoRing {
    static constexpr bool kRing = true;
    static constexpr bool kPaired = false;
    static constexpr int kChunk = RingGeom<P>::C;
    const Grp<G>& g;
    const Cx* ring;
    const Cx* cbuf = nullptr;        // this group's entries of the current chunk
    const double2* crec = nullptr;   // series records of the current chunk
    int kk0 = 0;                     // first pass of the current chunk
    CARMA_DEV void begin(int, double) {}
    CARMA_DEV void publish(int) const {}
    // barrier c: the producer has written chunk c (passes kk0 .. kk0+C-1)
    CARMA_DEV void chunk_begin(int first)
    {
        constexpr int C = RingGeom<P>::C;
        __syncthreads();
        const int c = (first - 1) / C;
        kk0 = first;
        cbuf = ring + (size_t)(c & 1) * C * RingGeom<P>::SLOT + g.gbase();
        crec = reinterpret_cast<const double2*>(ring + RingGeom<P>::REC_OFF) + (c & 1) * C;
    }
    // series record of the pass in slot s of the current chunk (record kk-1 for pass kk)
    CARMA_DEV double4 record_s(int s) const
    {
        const double2 v = crec[s];
        return double4{0.0, v.x, v.y, 0.0};
    }
    CARMA_DEV double4 record(int) const { return double4{}; }
    CARMA_DEV void fetch(int, Cx&, Cx (&)[P]) const {}
    // factors of the pass in slot s of the current chunk
    CARMA_DEV void fetch_s(int s, Cx& rho, Cx (&rj)[P]) const
    {
        const double2* slot = reinterpret_cast<const double2*>(cbuf + (size_t)s * RingGeom<P>::SLOT);
        const double2 o = slot[g.lane()];
        rho = Cx{o.x, o.y};
#pragma unroll
        for (int j = 0; j < P; j++) {
            const double2 v = slot[j];
            rj[j] = Cx{v.x, v.y};
        }
    }
    CARMA_DEV void prepare(int, double) {}
};

template <int P, int G>
__device__ __forceinline__ double ring_consume(const Grp<G>& g, const Model<P>& m, const double4* __restrict__ series,
                                               int n, const Cx* __restrict__ ring, bool* singular)
{
    FilterConsts<P> fc;
#if defined(CARMA_STAMPS)
    unsigned long long r0, r1;
    CARMA_STAMP(r0);
#endif
    filter_reset<P, G>(g, m, fc);
#if defined(CARMA_STAMPS)
    CARMA_STAMP(r1);
    if (blockIdx.x == 0 && threadIdx.x == 0) printf("reset %llu ticks\n", r1 - r0);
#endif
    RhoRing<P, G> src{g, ring};
    const double ll = filter_loop_real<P, G, false>(g, m, fc, src, series, n, nullptr, nullptr);
    *singular = fc.sing;
    return ll;
}

}  // namespace carma
