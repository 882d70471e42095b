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
    static constexpr int C = 16;                                    // steps per chunk
    static constexpr int SLOT = 64;                                 // Cx entries per step (one per lane)
    static constexpr int ENTRIES = 2 * C * SLOT;                    // Cx entries in the ring
    static constexpr size_t BYTES = (size_t)ENTRIES * sizeof(Cx);   // 32 KiB
};

// Producer: all n-1 steps of the 64/G evaluations of this workgroup.
template <int P, int G>
__device__ __forceinline__ void ring_produce(const Grp<G>& g, const Cx w, const double4* __restrict__ series, int n,
                                             Cx* __restrict__ ring)
{
    constexpr int C = RingGeom<P>::C;
    const int lane = g.lane64;
    const int nsteps = n - 1;
    const int nchunks = (nsteps + C - 1) / C;
    for (int c = 0; c < nchunks; c++) {
        Cx* buf = ring + (size_t)(c & 1) * C * RingGeom<P>::SLOT;
#pragma unroll 1
        for (int s = 0; s < C; s++) {
            const int k = 1 + c * C + s;
            if (k < n) {
                Cx rho;
                cexp_step(w.re, w.im, series[k].x, &rho.re, &rho.im);
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
    const Grp<G>& g;
    const Cx* ring;
    CARMA_DEV void begin(int, double) {}
    CARMA_DEV void publish(int) const {}
    CARMA_DEV void fetch(int k, Cx& rho, Cx (&rj)[P]) const
    {
        constexpr int C = RingGeom<P>::C;
        const int c = (k - 1) / C, s = (k - 1) % C;
        if (s == 0) __syncthreads();                       // chunk c is in the ring
        const Cx* slot = ring + ((size_t)(c & 1) * C + s) * RingGeom<P>::SLOT + g.gbase();
        rho = slot[g.lane()];
#pragma unroll
        for (int j = 0; j < P; j++) rj[j] = slot[j];
    }
    CARMA_DEV void prepare(int, double) {}
};

template <int P, int G>
__device__ __forceinline__ double ring_consume(const Grp<G>& g, const Model<P>& m, const double4* __restrict__ series,
                                               int n, const Cx* __restrict__ ring, bool* singular)
{
    FilterConsts<P> fc;
    filter_reset<P, G>(g, m, fc);
    RhoRing<P, G> src{g, ring};
    double ll = filter_loop<P, G, false>(g, m, fc, src, series, n, nullptr, nullptr);
    *singular = fc.sing;
    return ll;
}

}  // namespace carma
