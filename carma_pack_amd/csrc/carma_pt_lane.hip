// carma_pt_lane.hip -- the parallel-tempered Robust-Adaptive-Metropolis sampler with ONE CHAIN PER LANE (round 4):
// the sampler for LARGE ENSEMBLES (tens of thousands of chains).
//
// k_pt gives a chain an 8-lane group, k_pt_row a 16-lane DPP row plus three helper waves per four chains: right while a
// launch has fewer chains than the chip has lane groups, wasteful beyond -- 16 x 512 ladders ran at 6.7e7 chain
// evaluations per second while the one-evaluation-per-lane log-density kernel (carma_lane.h) does 2.8-3.7e8.  Here a lane
// owns a chain outright, exactly as a lane owns an evaluation there:
//   * the Kalman log-density of the proposal is lane_filter (carma_lane.h), nothing crosses lanes;
//   * the Robust-Adaptive-Metropolis step (AdaptiveMetro::DoStep, src/steps.cpp:60-107; CholUpdateR1, :111-131) runs in
//     the lane: t8 draws, thn = th + R^T z, Metropolis decision, rank-1 update of the upper-triangular factor R.  The
//     chain's state is NOT held in registers across the filter (the filter wants all 236 of them at p = 5): the current
//     value th and the packed factor R live in a chain-minor scratch in global memory ([row][chain]: a wave's loads and
//     stores of one row are one 512-byte line set, L2 resident), the proposal and v = R^T z in LDS;
//   * the exchange sweep (ExchangeStep::DoStep, src/include/steps.hpp:318-362, wired hot -> cold in
//     src/carmcmc.cpp:147-157) runs INSIDE THE WAVE: a ladder's T chains are T consecutive lanes, floor(64 / T) ladders
//     per wave, one lane per ladder replays the serial decisions on the staged log-posteriors (exchange_decide, the
//     routine k_pt uses), then every lane fetches the parameter vector the sweep assigned to its temperature from the
//     staging in LDS.
// Same Philox keys, same formulas in the same operation order as ram_propose / ram_finish / exchange_decide of
// carma_pt_core.h: from the same seed the chains take the same decisions as k_pt's and differ in rounding only
// (tests/test_gpu_sampler.py).
// NP = 3: three PRODUCER waves per chain wave compute the transition factors of the proposals (lane_produce, the ring of
// k_logdens_carma_lpc) -- for ensembles that leave at most one chain wave per SIMD, where a lone wave's instruction
// stream is the iteration time.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "grp_device.h"
#include "carma_lane.h"
#include "carma_launch.h"
#include "carma_pt_core.h"

namespace carma {

template <int P>
struct PtLaneGeom {
    static constexpr int DM = P >= 2 ? 2 * P + 2 : 4;       // d = 3 + p + q <= 2 p + 2 (CAR(1): 4)
    static constexpr int NT = DM * (DM + 1) / 2;            // packed upper triangle of R, compile-time indices
    static constexpr int DMS = DM + 1;                      // stride of a lane's vector in LDS (odd: no bank pattern)
    static constexpr int ROW_TH = 0, ROW_R = DM, ROWS = DM + NT;   // rows of the chain-minor scratch: current value, packed factor
    static constexpr int MINW = P <= 6 ? 2 : 1;             // waves per SIMD the registers are budgeted for (the p = 7 filter needs 306)
    // LDS of one chain wave (doubles): proposal / staging [64][DMS], rank-1 vector [64][DMS], log-posterior, log-uniform,
    // 1/T differences, |z|^2 [64 each], then int src[64], unsigned nswap[64], nacc[64], the parked AR roots [64][WS]
    static constexpr int WS = 2 * P + 1;                    // stride of a lane's parked AR roots
    static constexpr int LDS_DOUBLES = 2 * 64 * DMS + 4 * 64 + 96 + 64 * WS;
};

template <int DM>
__device__ __forceinline__ constexpr int tri_pk(int i, int j)
{
    return i * DM - i * (i - 1) / 2 + (j - i);              // i <= j
}

// wave-level "LDS written above is read below": one wave's LDS instructions execute in order, this is the compiler's fence
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// What a lane derives from its lane number: which chain it is and where that chain's state lives.  Re-derived (from a
// laundered lane number) behind the filter instead of being kept: across the filter only LDS holds chain state, so that the
// filter's loop runs in the registers it has as a kernel of its own (236 at p = 5; every value kept alive across it was a
// spill INSIDE that loop, reloaded on 270 critical paths per iteration: 477 instead of 235 us per iteration, measured).
struct LaneChain {
    int c0, c, lbase;
    long lad, gi;
    bool active;
    uint32_t chain;
};
__device__ __forceinline__ LaneChain lane_chain(int lane, const PtLaunch& L)
{
    LaneChain x;
    const int T = L.T, LPW = 64 / T;
    const int lw = lane / T;
    x.c0 = lane - lw * T;                                    // temperature index inside the ladder
    const long lad0 = (long)blockIdx.x * LPW + lw;
    x.active = lw < LPW && lad0 < L.R;
    x.lad = x.active ? lad0 : (long)L.R - 1;                 // idle lanes shadow the last chain: uniform control flow, no writes
    x.c = x.active ? x.c0 : T - 1;
    x.gi = x.lad * T + x.c;                                  // chain in the state arrays, column of the scratch
    x.lbase = lane - x.c0;                                   // first lane of this lane's ladder
    x.chain = (uint32_t)((L.replica0 + x.lad) * L.T_global + L.slot0) + (uint32_t)x.c;
    return x;
}
__device__ __forceinline__ int launder(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

template <int P, int NP>
__global__ __launch_bounds__(NP ? 256 : 64, (NP ? PtLaneGeom<P>::MINW : 1)) void k_pt_lane(PtLaunch L, double* __restrict__ scratch, long ncol,
                                                           const double4* __restrict__ series, Prior pr,
                                                           const double* __restrict__ temps, double* __restrict__ theta,
                                                           double* __restrict__ logpost, double* __restrict__ chol,
                                                           unsigned* __restrict__ naccept, unsigned* __restrict__ nswap,
                                                           double* __restrict__ samples, double* __restrict__ sample_lp,
                                                           int ncu, int rot)
{
    using Geo = PtLaneGeom<P>;
    constexpr int DM = Geo::DM, DMS = Geo::DMS;
    extern __shared__ double lane_lds[];
    const int lane0 = threadIdx.x & 63;
    const int d = L.d, T = L.T;
    // ---- which wave plays which part (NP = 3: as k_logdens_carma_lpc)
    int part = 0;
    if constexpr (NP > 0) {
        const int round = (int)(blockIdx.x / (unsigned)ncu);
        part = ((int)(threadIdx.x >> 6) + ((rot >> (4 * (round & 3))) & 3)) & 3;
    }
    double* s_thn = lane_lds;                                // [64][DMS] proposal, then staging of the current value
    double* s_v = s_thn + 64 * DMS;                          // [64][DMS] v = R^T z
    double* s_lp = s_v + 64 * DMS;                           // [64] the chains' stored log-posteriors (always current)
    double* s_logu = s_lp + 64;                              // [64] log of the swap uniforms
    double* s_dbeta = s_logu + 64;                           // [64] 1/T_i - 1/T_{i-1}
    double* s_z2 = s_dbeta + 64;                             // [64] |z|^2 of the proposal in flight
    int* s_src = reinterpret_cast<int*>(s_z2 + 64);          // [64]
    unsigned* s_nswap = reinterpret_cast<unsigned*>(s_src + 64);
    unsigned* s_nacc = s_nswap + 64;
    double* s_w = reinterpret_cast<double*>(s_nacc + 64);    // [64][WS] AR roots of the proposals during the recursion
    double* ring = nullptr;
    if constexpr (NP > 0) ring = lane_lds + Geo::LDS_DOUBLES + lane0;

    if constexpr (NP > 0) {
        if (part != 0) {
            // ---- producers: the transition factors of the proposals, step by step (carma_lane.h)
            for (int it = 0; it < L.niter; it++) {
                __syncthreads();                             // proposals visible
                lane_produce<P, NP>(part - 1, s_thn + lane0 * DMS, ring, series, L.n);
            }
            return;
        }
    }
    // packed index of R_kj (k <= j) in a chain's column: run-time on purpose -- the chain's bookkeeping is a few thousand
    // instructions per iteration against ~90 000 of the filter, and rolled loops leave the registers alone
    auto Rp = [&](long gi, int k, int j) -> double* {
        return scratch + (long)(Geo::ROW_R + k * DM - k * (k - 1) / 2 + (j - k)) * ncol + gi;
    };
    const bool exch = L.do_exchange && T > 1;
    {
        // ---- chain state in: current value and factor into the chain-minor scratch
        const LaneChain x = lane_chain(lane0, L);
        s_lp[lane0] = logpost[x.gi];
        if (x.active) {
            double* col = scratch + x.gi;
#pragma unroll 1
            for (int j = 0; j < d; j++) col[(long)(Geo::ROW_TH + j) * ncol] = theta[x.gi * d + j];
#pragma unroll 1
            for (int k = 0; k < d; k++)
#pragma unroll 1
                for (int j = k; j < d; j++) *Rp(x.gi, k, j) = chol[(x.gi * d + k) * d + j];
        }
        s_dbeta[lane0] = x.c > 0 ? 1.0 / temps[x.c] - 1.0 / temps[x.c - 1] : 0.0;
        s_nswap[lane0] = 0u;
        s_nacc[lane0] = 0u;
    }
    wave_sync();

    for (int it = 0; it < L.niter; it++) {
        const uint64_t iter = L.iter0 + (uint64_t)it;
        {
            // ---- proposal: z ~ t8^d, v = R^T z, thn = th + v   (steps.cpp:60-73; ram_propose)
            // The lane's LDS slot first takes z; v_j needs z_0..z_j only, so the slot turns into the proposal from the top
            // down (j = d-1 .. 0).
            const int lane = launder(lane0);
            const LaneChain x = lane_chain(lane, L);
            const RngKey key{L.seed0, L.seed1, x.chain};
            double* thn_l = s_thn + lane * DMS;
            double* v_l = s_v + lane * DMS;
            const double* col = scratch + x.gi;
            double znorm2 = 0.0;
#pragma unroll 1
            for (int k = 0; k < d; k++) {
                const double zk = rng_student_t8(key, iter, (uint32_t)k);
                thn_l[k] = zk;
                znorm2 += zk * zk;
            }
            s_z2[lane] = znorm2;
#pragma unroll 1
            for (int j = d - 1; j >= 0; j--) {
                // column j of R: the loads first (independent), then the sum in the reference's order k = 0 .. j
                double rk[DM];
#pragma unroll
                for (int k = 0; k < DM; k++) rk[k] = *Rp(x.gi, k <= j ? k : j, j);
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < DM; k++)
                    if (k <= j) acc += rk[k] * thn_l[k];
                thn_l[j] = col[(long)(Geo::ROW_TH + j) * ncol] + acc;
                v_l[j] = acc;
            }
        }
        // ---- Accept (steps.cpp:36-56): one Kalman log-density of the proposal
        double ll;
        asm volatile("" ::: "memory");
        if constexpr (NP > 0) {
            __syncthreads();                                 // proposals visible to the producers
            ll = logdensity_lane_ring<P, NP>(s_thn + lane0 * DMS, L.q, series, L.n, pr, 0, ring);
        } else {
            wave_sync();
            ll = logdensity_lane_parked<P>(s_thn + lane0 * DMS, L.q, series, L.n, pr, 0, s_w + lane0 * Geo::WS);
        }
        asm volatile("" ::: "memory");
        const int lane = launder(lane0);
        const LaneChain x = lane_chain(lane, L);
        const RngKey key{L.seed0, L.seed1, x.chain};
        double* thn_l = s_thn + lane * DMS;
        double* v_l = s_v + lane * DMS;
        double* col = scratch + x.gi;
        const double lp_old = s_lp[lane];
        double alpha = (ll - lp_old) / temps[x.c];
        bool accept = false;
        if (!((alpha - alpha) == 0.0)) {
            alpha = 0.0;                                     // steps.cpp:41-46
        } else {
            const double u = rng_uniform(key, iter, RNG_ACCEPT, 0);
            alpha = fmin(exp(alpha), 1.0);
            accept = u < alpha;
        }
        if (accept) {                                        // parameter_.Save(new_value) (steps.cpp:77)
            s_lp[lane] = ll;
            s_nacc[lane]++;
            if (x.active) {
#pragma unroll 1
                for (int j = 0; j < d; j++) col[(long)(Geo::ROW_TH + j) * ncol] = thn_l[j];
            }
        } else if (exch || L.save_thin > 0) {
            // the lane's slot in LDS becomes the staging of its CURRENT value for the sweep / the save below
#pragma unroll 1
            for (int j = 0; j < d; j++) thn_l[j] = col[(long)(Geo::ROW_TH + j) * ncol];
        }
        // ---- adaptation of the proposal factor while niter < maxiter (steps.cpp:82-99, 111-131; ram_finish, chol_update_r1)
        if ((long)iter < (long)L.maxiter) {
            const double step = fmin(1.0, (double)d / pow((double)iter, 2.0 / 3.0));   // iter = 0 -> 1
            const double fac = sqrt(step * fabs(alpha - 0.25)) / sqrt(s_z2[lane]);
            const double sign = alpha < 0.25 ? -1.0 : 1.0;
#pragma unroll 1
            for (int j = 0; j < d; j++) v_l[j] *= fac;
            // row k of R at a time, in place in the scratch (row k is final after step k)
#pragma unroll 1
            for (int k = 0; k < d; k++) {
                double rj[DM], vj[DM];
#pragma unroll
                for (int jj = 0; jj < DM; jj++) {             // the row's loads first (independent)
                    const int j = k + jj < d ? k + jj : d - 1;
                    rj[jj] = *Rp(x.gi, k, j);
                    vj[jj] = v_l[j];
                }
                const double Rkk = rj[0], vk = vj[0];
                const double rr = sqrt(Rkk * Rkk + sign * vk * vk);
                const double cc = rr / Rkk, ss = vk / Rkk;
                if (x.active) *Rp(x.gi, k, k) = rr;
#pragma unroll
                for (int jj = 1; jj < DM; jj++) {
                    const int j = k + jj;
                    if (j < d) {
                        const double Rkj = (rj[jj] + sign * ss * vj[jj]) / cc;
                        if (x.active) *Rp(x.gi, k, j) = Rkj;
                        v_l[j] = cc * vj[jj] - ss * Rkj;
                    }
                }
            }
        }
        // ---- ExchangeStep sweep hot -> cold over the ladder's T lanes (steps.hpp:318-362)
        const bool save = L.save_thin > 0 && ((it + 1) % L.save_thin) == 0 && x.active && x.c0 == 0;
        const long sidx = L.save_thin > 0 ? L.save_offset + (it + 1) / L.save_thin - 1 : 0;
        if (exch) {
            s_src[lane] = x.c0;
            s_logu[lane] = x.c > 0 ? log(rng_uniform(key, iter, RNG_SWAP, 0)) : 0.0;   // keyed by the hotter chain's global slot
            wave_sync();
            if (x.active && x.c0 == 0)
                exchange_decide(T, s_lp + x.lbase, s_dbeta + x.lbase, s_logu + x.lbase, s_src + x.lbase, s_nswap + x.lbase);
            wave_sync();
            const int from = s_src[lane];
            const double* src = s_thn + (x.lbase + from) * DMS;
            if (x.active && from != x.c0) {
#pragma unroll 1
                for (int j = 0; j < d; j++) col[(long)(Geo::ROW_TH + j) * ncol] = src[j];
            }
            if (save && sidx < L.sample_cap) {
                // coldest chain of this ladder (Sampler::SaveValues, src/samplers.cpp:118-124)
#pragma unroll 1
                for (int j = 0; j < d; j++) samples[(x.lad * L.sample_cap + sidx) * d + j] = src[j];
                sample_lp[x.lad * L.sample_cap + sidx] = s_lp[lane];
            }
            wave_sync();                                     // the staging is read before the next proposals overwrite it
        } else if (save && sidx < L.sample_cap) {
#pragma unroll 1
            for (int j = 0; j < d; j++) samples[(x.lad * L.sample_cap + sidx) * d + j] = thn_l[j];
            sample_lp[x.lad * L.sample_cap + sidx] = s_lp[lane];
        }
    }
    // ---- chain state out
    {
        const LaneChain x = lane_chain(lane0, L);
        if (x.active) {
            const double* col = scratch + x.gi;
#pragma unroll 1
            for (int j = 0; j < d; j++) theta[x.gi * d + j] = col[(long)(Geo::ROW_TH + j) * ncol];
#pragma unroll 1
            for (int k = 0; k < d; k++)
#pragma unroll 1
                for (int j = k; j < d; j++) chol[(x.gi * d + k) * d + j] = *Rp(x.gi, k, j);
            logpost[x.gi] = s_lp[lane0];
            naccept[x.gi] += s_nacc[lane0];
            nswap[x.gi] += s_nswap[lane0];
        }
    }
}

// doubles of scratch a launch needs: PtLaneGeom<p>::ROWS rows of one column per chain
size_t pt_lane_scratch_doubles(int p, long nchain)
{
    if (p < 2 || p > 7 || nchain < 1) return 0;
    const size_t dm = 2 * (size_t)p + 2;
    return (dm + dm * (dm + 1) / 2) * (size_t)nchain;
}

template <int P>
static hipError_t launch_pt_lane_p(const PtLaunch& L, int np, double* scratch, const double4* series, const Prior& pr,
                                   const double* temps, double* theta, double* logpost, double* chol, unsigned* naccept,
                                   unsigned* nswap, double* samples, double* sample_lp, hipStream_t st)
{
    using Geo = PtLaneGeom<P>;
    const int LPW = 64 / L.T;
    if (LPW < 1) return hipErrorInvalidValue;
    const long waves = ((long)L.R + LPW - 1) / LPW;
    const long ncol = (long)L.R * L.T;
    const size_t lds0 = sizeof(double) * Geo::LDS_DOUBLES;
    if (np == 3) {
        const size_t lds = lds0 + LaneRingGeom<P, 3>::BYTES;
        static const long rot = [] {
            const char* e = getenv("CARMA_TUNE_LPC_ROT");
            return e ? strtol(e, nullptr, 0) : 0x0202L;
        }();
        hipLaunchKernelGGL((k_pt_lane<P, 3>), dim3((unsigned)waves), dim3(256), lds, st, L, scratch, ncol, series, pr, temps, theta,
                           logpost, chol, naccept, nswap, samples, sample_lp, device_cus(), (int)rot);
    } else {
        hipLaunchKernelGGL((k_pt_lane<P, 0>), dim3((unsigned)waves), dim3(64), lds0, st, L, scratch, ncol, series, pr, temps, theta,
                           logpost, chol, naccept, nswap, samples, sample_lp, device_cus(), 0);
    }
    return hipGetLastError();
}

hipError_t launch_pt_lane(int p, const PtLaunch& L, int np, double* scratch, const double4* series, const Prior& pr,
                          const double* temps, double* theta, double* logpost, double* chol, unsigned* naccept,
                          unsigned* nswap, double* samples, double* sample_lp, hipStream_t st)
{
    (void)hipGetLastError();
    switch (p) {
#ifndef CARMA_LANE_ONLY_P5
        case 2: return launch_pt_lane_p<2>(L, np, scratch, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        case 3: return launch_pt_lane_p<3>(L, np, scratch, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        case 4: return launch_pt_lane_p<4>(L, np, scratch, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
#endif
        case 5: return launch_pt_lane_p<5>(L, np, scratch, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
#ifndef CARMA_LANE_ONLY_P5
        case 6: return launch_pt_lane_p<6>(L, np, scratch, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        case 7: return launch_pt_lane_p<7>(L, np, scratch, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
#endif
        default: return hipErrorInvalidValue;
    }
}

}  // namespace carma
