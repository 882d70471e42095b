// carma_pt_lane.hip -- the parallel-tempered Robust-Adaptive-Metropolis sampler for LARGE ENSEMBLES (round 4): one chain
// per LANE, and an iteration as separate launches (two; three for the first of a chunk) instead of one persistent kernel:
//
//     k_ram_propose   t8 draws, thn = th + R^T z                          AdaptiveMetro::DoStep, src/steps.cpp:60-73
//                     (a launch of its own for the first iteration of a chunk only: afterwards the proposal of iteration
//                     i + 1 rides on the finish kernel of iteration i, the factor still in registers)
//     K1              the Kalman log-density of every proposal: the batched log-density launch of carma_kernels.hip,
//                     whichever shape serves that many evaluations (producer-wave lane kernel, lane kernel, ...)
//                                                                         Accept, src/steps.cpp:36-56 -> carpack.hpp:131-176
//     k_ram_finish    Metropolis decision, rank-1 update of the proposal factor, the ladder's exchange sweep, save
//                                                                         steps.cpp:36-56, 77-99, 111-131; steps.hpp:318-362
//
// Why not one kernel.  k_pt gives a chain an 8-lane group, k_pt_row a 16-lane DPP row plus three helper waves per four
// chains: right while a launch has fewer chains than the chip has lane groups, wasteful beyond -- 16 x 512 ladders ran at
// 6.7e7 chain evaluations per second while K1 does 1.4e8 at 16 384 evaluations and 2.8-3.7e8 beyond 65 536.  A fused
// lane-per-chain kernel was built first (the filter of carma_lane.h with the chain's bookkeeping around it in the same
// lane; git history, measured in profiles/r04/lane_sampler_fused_v1.txt): correct, and slow -- the filter's loop wants every
// register it gets as a kernel of its own (236 at p = 5), so whatever the bookkeeping keeps alive across it spills INSIDE
// that loop (477 instead of 235 us per iteration), and with a wave per SIMD to itself (no spills) the bookkeeping's memory
// round trips and 11 Box-Muller draws sit on the lone wave's critical path (330 us).  Split, the bookkeeping runs as two
// short, lean kernels with thousands of waves in flight (latency hidden, a few tens of registers), the filter is the tuned
// K1 launch at ITS best occupancy, and a launch boundary costs ~2 us of a >= 120 us iteration.  K1 improvements carry
// over for free.
//
// State.  A ladder's T <= 64 chains are T consecutive lanes of a wave, floor(64 / T) ladders per wave, so that the
// exchange sweep stays inside a wave (one lane per ladder replays the serial hot -> cold decisions on the staged
// log-posteriors: exchange_decide, the routine k_pt uses).  The chains' working state is CHAIN-MINOR in global memory
// ([row][chain]: a wave's access to one row is one contiguous 512-byte segment): current value th[d], v = R^T z [d],
// the packed upper-triangular factor R [d (d + 1) / 2], log-posterior, |z|^2; the proposals are chain-major [chain][d],
// the batch K1 reads.  k_ram_convert translates from / to the chain-major state arrays of the other sampler kernels
// (theta [R][T][d], chol [R][T][d][d]) at the ends of a chunk of iterations.
// Same Philox keys, same formulas in the same operation order as ram_propose / ram_finish / exchange_decide of
// carma_pt_core.h: from the same seed the chains take the same decisions as k_pt's and differ in rounding only
// (tests/test_gpu_sampler.py::test_lane_kernel_walks_the_ladder_kernels_trajectory).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "grp_device.h"
#include "carma_launch.h"
#include "carma_pt_core.h"
#include "carma_pt_row.h"

namespace carma {

constexpr int RAM_DMAX = 16;                                 // d = 3 + p + q <= 16
constexpr int RAM_DS = 17;                                   // stride of a lane's vector in LDS (odd)

struct RamState {                                            // chain-minor working state (device pointers)
    double* th;      // [d][nc]
    double* v;       // [d][nc]
    double* R;       // [d (d + 1) / 2][nc], row k of the upper triangle after row k - 1
    double* lp;      // [nc]
    double* z2;      // [nc]
    double* ll;      // [nc]      log-densities of the proposals
    double* thn;     // [nc][d]   proposals, the batch of K1
    long nc;         // chains = R * T
};

struct RamLane {
    int c0, c, lbase;
    long lad, gi;
    bool active;
    uint32_t chain;
};
// lane -> chain: LPW = 64 / T ladders of T consecutive lanes per wave; idle lanes shadow the last chain (uniform control
// flow, no writes)
__device__ __forceinline__ RamLane ram_lane(const PtLaunch& L)
{
    RamLane x;
    const int lane = threadIdx.x & 63, T = L.T, LPW = 64 / T;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lw = lane / T;
    x.c0 = lane - lw * T;
    const long lad0 = wave * LPW + lw;
    x.active = lw < LPW && lad0 < L.R;
    x.lad = x.active ? lad0 : (long)L.R - 1;
    x.c = x.active ? x.c0 : T - 1;
    x.gi = x.lad * T + x.c;
    x.lbase = lane - x.c0;
    x.chain = (uint32_t)((L.replica0 + x.lad) * L.T_global + L.slot0) + (uint32_t)x.c;
    return x;
}
__device__ __forceinline__ long tri_row(int d, int k) { return (long)k * d - (long)k * (k - 1) / 2; }   // index of R_kk

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// chain-major state arrays -> chain-minor working state (load != 0) or back.  what: 1 = values and log-posteriors, 2 = the factors
// (d (d + 1) / 2 of the d (d + 3) / 2 + 1 numbers of a chain, strided by d^2 on the chain-major side: they stay resident in the
// chain-minor state between calls and are converted only when somebody reads or writes the chain-major copy -- round 5)
constexpr int RAM_CONV_STATE = 1, RAM_CONV_FACTOR = 2;
__global__ __launch_bounds__(256) void k_ram_convert(PtLaunch L, RamState S, double* __restrict__ theta,
                                                     double* __restrict__ logpost, double* __restrict__ chol, int load, int what)
{
    const long gi = (long)blockIdx.x * 256 + threadIdx.x;
    if (gi >= S.nc) return;
    const int d = L.d;
    if (load) {
        if (what & RAM_CONV_STATE) {
            for (int j = 0; j < d; j++) S.th[(long)j * S.nc + gi] = theta[gi * d + j];
            S.lp[gi] = logpost[gi];
        }
        if (what & RAM_CONV_FACTOR)
            for (int k = 0; k < d; k++)
                for (int j = k; j < d; j++) S.R[(tri_row(d, k) + (j - k)) * S.nc + gi] = chol[(gi * d + k) * d + j];
    } else {
        if (what & RAM_CONV_STATE) {
            for (int j = 0; j < d; j++) theta[gi * d + j] = S.th[(long)j * S.nc + gi];
            logpost[gi] = S.lp[gi];
        }
        if (what & RAM_CONV_FACTOR)
            for (int k = 0; k < d; k++)
                for (int j = k; j < d; j++) chol[(gi * d + k) * d + j] = S.R[(tri_row(d, k) + (j - k)) * S.nc + gi];
    }
}

// z ~ t8^d, v = R^T z, thn = th + v   (steps.cpp:60-73; ram_propose of carma_pt_core.h).  D = d at compile time: the whole
// factor is requested in one go (D (D + 1) / 2 independent loads in flight, registers), no dependent memory round trips.
template <int D>
__global__ __launch_bounds__(256) void k_ram_propose(PtLaunch L, RamState S)
{
    const RamLane x = ram_lane(L);
    const uint64_t iter = L.iter0;
    const RngKey key{L.seed0, L.seed1, x.chain};
    constexpr int NT = D * (D + 1) / 2;
    double Rr[NT], th[D];
#pragma unroll
    for (int i = 0; i < NT; i++) Rr[i] = S.R[(long)i * S.nc + x.gi];
#pragma unroll
    for (int j = 0; j < D; j++) th[j] = S.th[(long)j * S.nc + x.gi];
    double z[D];
#pragma unroll
    for (int i = 0; i < D; i++) z[i] = 0.0;
    double znorm2 = 0.0;
#pragma unroll 1
    for (int k = 0; k < D; k++) {
        const double zk = rng_student_t8(key, iter, (uint32_t)k);
        // (z[k] with a run-time k: a chain of selects instead of a copy of the generator per component)
#pragma unroll
        for (int i = 0; i < D; i++) z[i] = i == k ? zk : z[i];
        znorm2 += zk * zk;
    }
    if (!x.active) return;
    S.z2[x.gi] = znorm2;
#pragma unroll
    for (int j = 0; j < D; j++) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k <= j; k++) acc += Rr[k * D - k * (k - 1) / 2 + (j - k)] * z[k];
        S.v[(long)j * S.nc + x.gi] = acc;
        S.thn[x.gi * D + j] = th[j] + acc;
    }
}

// Metropolis decision with the tempered ratio (steps.cpp:36-56), the RAM rank-1 update of the factor (steps.cpp:82-99,
// CholUpdateR1 :111-131), the exchange sweep of the ladder (steps.hpp:318-362), Sampler::SaveValues (samplers.cpp:118-124)
// NEXT: the proposal of the FOLLOWING iteration right behind (what k_ram_propose does, with the factor still in registers):
// one launch and one pass over the factor less per iteration.
// (one workgroup per CU's worth of registers: with the budget of two, D = 11 spills 836 bytes per lane)
template <int D, bool NEXT>
__global__ __launch_bounds__(256) void k_ram_finish(PtLaunch L, RamState S, const double* __restrict__ temps,
                                                    unsigned* __restrict__ naccept, unsigned* __restrict__ nswap,
                                                    double* __restrict__ samples, double* __restrict__ sample_lp)
{
    __shared__ double s_vec[256 * RAM_DS];                   // staging of the lanes' current values for the sweep
    __shared__ double s_lp[256], s_logu[256], s_dbeta[256];
    __shared__ int s_src[256];
    __shared__ unsigned long long s_mask[256];
    const RamLane x = ram_lane(L);
    constexpr int d = D, NT = D * (D + 1) / 2;
    const int T = L.T, tid = threadIdx.x;
    const uint64_t iter = L.iter0;
    const RngKey key{L.seed0, L.seed1, x.chain};
    double* vec = s_vec + tid * RAM_DS;
    const int wbase = tid & ~63;                             // this wave's first slot
    const bool adapt = (long)iter < (long)L.maxiter;
    // (the three scalars FIRST: loads return in order, and whatever is computed from them early then waits for them alone)
    const double ll_in = S.ll[x.gi];
    const double lp_in = S.lp[x.gi], z2_in = adapt ? S.z2[x.gi] : 1.0;
    // (the counters too: an increment behind the factor's stores would wait for all of them -- loads and stores share a counter)
    const bool exch = L.do_exchange && L.T > 1;
    const unsigned nacc_in = naccept[x.gi], nswap_in = exch ? nswap[x.gi] : 0u;
    // the factor and v: requested up front (independent loads), used by the adaptation below (and by the next proposal)
    double Rr[NT], v[D];
    if (adapt || NEXT) {
#pragma unroll
        for (int i = 0; i < NT; i++) Rr[i] = S.R[(long)i * S.nc + x.gi];
    }
    // v = R^T z of this iteration's proposal and the current value: the proposal K1 evaluated is th + v, bit for bit (the
    // sum k_ram_propose / the NEXT block below stored as thn[chain][.] -- re-formed here from two coalesced rows instead of a
    // strided read of that batch)
    double th[D];
#pragma unroll
    for (int j = 0; j < D; j++) v[j] = S.v[(long)j * S.nc + x.gi];
#pragma unroll
    for (int j = 0; j < D; j++) th[j] = S.th[(long)j * S.nc + x.gi];
    // NEXT: the draws of the following proposal depend on nothing but the key -- they run HERE, while the loads above are in
    // flight (a wave of this kernel has its SIMD to itself up to 65 536 chains: nothing else hides that latency)
    double z[NEXT ? D : 1];
    double znorm2_next = 0.0;
    if constexpr (NEXT) {
#pragma unroll
        for (int i = 0; i < D; i++) z[i] = 0.0;
#pragma unroll 1
        for (int k = 0; k < D; k++) {
            const double zk = rng_student_t8(key, iter + 1, (uint32_t)k);
            // (z[k] with a run-time k: a chain of selects instead of a copy of the generator per component)
#pragma unroll
            for (int i = 0; i < D; i++) z[i] = i == k ? zk : z[i];
            znorm2_next += zk * zk;
        }
    }
    const double ll = ll_in;
    double lp = lp_in;
    double alpha = (ll - lp) / temps[x.c];
    bool accept = false;
    if (!((alpha - alpha) == 0.0)) {
        alpha = 0.0;                                         // steps.cpp:41-46
    } else {
        const double u = rng_uniform(key, iter, RNG_ACCEPT, 0);
        alpha = fmin(exp(alpha), 1.0);
        accept = u < alpha;
    }
    if (accept) {                                            // parameter_.Save(new_value) (steps.cpp:77)
        lp = ll;
#pragma unroll
        for (int j = 0; j < D; j++) th[j] += v[j];
        if (x.active) naccept[x.gi] = nacc_in + 1u;
    }
    // ---- adaptation while niter < maxiter (ram_finish / chol_update_r1 of carma_pt_core.h, steps.cpp:82-99, 111-131)
    if (adapt) {
        const double step = fmin(1.0, (double)d / pow((double)iter, 2.0 / 3.0));   // iter = 0 -> 1
        const double fac = sqrt(step * fabs(alpha - 0.25)) / sqrt(z2_in);
        const double sign = alpha < 0.25 ? -1.0 : 1.0;
#pragma unroll
        for (int j = 0; j < D; j++) v[j] *= fac;
#pragma unroll
        for (int k = 0; k < D; k++) {
            const int kk = k * D - k * (k - 1) / 2;          // index of R_kk
            // (as chol_update_row of carma_pt_row.h: with rs = 1 / sqrt(R_kk^2 +- v_k^2) the reference's rr, c = rr / R_kk,
            // s = v_k / R_kk and the division by c are rr = x rs, c = rr / R_kk, s = v_k / R_kk, 1 / c = R_kk rs -- one
            // reciprocal square root and one reciprocal per step instead of a square root and D - k + 1 divisions)
            const double Rkk = Rr[kk], vk = v[k];
            const double xx = fma(sign * vk, vk, Rkk * Rkk);
            const double rs = rsqrt_nr(xx), iR = recip(Rkk);
            const double rr = xx * rs;
            const double cc = rr * iR, ss = vk * iR, ic = Rkk * rs;
            Rr[kk] = rr;
#pragma unroll
            for (int j = k + 1; j < D; j++) {
                const double Rkj = (Rr[kk + (j - k)] + sign * ss * v[j]) * ic;
                Rr[kk + (j - k)] = Rkj;
                v[j] = cc * v[j] - ss * Rkj;
            }
        }
        if (x.active) {
#pragma unroll
            for (int i = 0; i < NT; i++) S.R[(long)i * S.nc + x.gi] = Rr[i];
        }
    }
    // ---- ExchangeStep sweep hot -> cold over the ladder's T lanes: the lanes' values are staged in LDS, one lane per ladder
    // replays the serial decisions, every lane fetches the vector the sweep assigned to its temperature
    const bool save = L.save_thin > 0 && x.active && x.c0 == 0 && L.save_offset < L.sample_cap;
    bool moved = accept;                                     // the lane's value in global memory is out of date
    if (exch) {
#pragma unroll
        for (int j = 0; j < D; j++) vec[j] = th[j];
        s_lp[tid] = lp;
        s_src[tid] = x.c0;
        s_dbeta[tid] = x.c > 0 ? 1.0 / temps[x.c] - 1.0 / temps[x.c - 1] : 0.0;
        s_logu[tid] = x.c > 0 ? log(rng_uniform(key, iter, RNG_SWAP, 0)) : 0.0;   // keyed by the hotter chain's global slot
        wave_sync();
        const int lb = wbase + x.lbase;
        if (x.active && x.c0 == 0) s_mask[lb] = exchange_decide_mask(T, s_lp + lb, s_dbeta + lb, s_logu + lb, s_src + lb);
        wave_sync();
        if (x.active && ((s_mask[lb] >> x.c0) & 1ull)) nswap[x.gi] = nswap_in + 1u;   // ExchangeStep's count, per hotter temperature
        const int from = s_src[tid];
        const double* src = s_vec + (lb + from) * RAM_DS;
        moved = moved || from != x.c0;
        lp = s_lp[tid];
#pragma unroll
        for (int j = 0; j < D; j++) th[j] = src[j];
    }
    if (moved && x.active) {
        S.lp[x.gi] = lp;
#pragma unroll
        for (int j = 0; j < D; j++) S.th[(long)j * S.nc + x.gi] = th[j];
    }
    if (save) {                                              // Sampler::SaveValues: the coldest chain's value and log-posterior
#pragma unroll
        for (int j = 0; j < D; j++) samples[(x.lad * L.sample_cap + L.save_offset) * d + j] = th[j];
        sample_lp[x.lad * L.sample_cap + L.save_offset] = lp;
    }
    if constexpr (NEXT) {
        // ---- the next iteration's proposal (k_ram_propose) from the value the sweep left in this lane
        if (!x.active) return;
        S.z2[x.gi] = znorm2_next;
#pragma unroll
        for (int j = 0; j < D; j++) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k <= j; k++) acc += Rr[k * D - k * (k - 1) / 2 + (j - k)] * z[k];
            S.v[(long)j * S.nc + x.gi] = acc;
            S.thn[x.gi * D + j] = th[j] + acc;
        }
    }
}

// doubles of working state for nchain chains of dimension d (RamState)
size_t pt_lane_scratch_doubles(int d, long nchain)
{
    if (d < 1 || d > RAM_DMAX || nchain < 1) return 0;
    return ((size_t)3 * d + (size_t)d * (d + 1) / 2 + 3) * (size_t)nchain;
}

static RamState ram_state(double* scratch, int d, long nc)
{
    RamState S;
    S.nc = nc;
    S.th = scratch;
    S.v = S.th + (size_t)d * nc;
    S.R = S.v + (size_t)d * nc;
    S.lp = S.R + (size_t)d * (d + 1) / 2 * nc;
    S.z2 = S.lp + nc;
    S.ll = S.z2 + nc;
    S.thn = S.ll + nc;
    return S;
}

// f(std::integral_constant<int, d>) for the run-time d (4: CAR(1); 5 <= d <= 16: p >= 2)
template <class F>
static hipError_t ram_launch_d(int d, F&& f)
{
    switch (d) {
#define CARMA_RAM_D(N) \
    case N: f(std::integral_constant<int, N>{}); return hipGetLastError();
        CARMA_RAM_D(4) CARMA_RAM_D(5) CARMA_RAM_D(6) CARMA_RAM_D(7) CARMA_RAM_D(8) CARMA_RAM_D(9) CARMA_RAM_D(10) CARMA_RAM_D(11)
        CARMA_RAM_D(12) CARMA_RAM_D(13) CARMA_RAM_D(14) CARMA_RAM_D(15) CARMA_RAM_D(16)
#undef CARMA_RAM_D
        default: return hipErrorInvalidValue;
    }
}

// niter iterations of the sampler for large ensembles, enqueued on st: 2 launches per iteration (+ the first proposal) + one conversion at
// either end.  The iterations, the save slots and the Philox keys are those of launch_pt for the same PtLaunch.
// load_factor: the chain-minor factors are not current (first call, or the chain-major copy was written): take them from chol.  The
// chain-major copy is NOT written back here -- pt_lane_store_factor does that for whoever wants to read it.
hipError_t launch_pt_lane(int p, const PtLaunch& L, double* scratch, const double4* series, const Prior& pr,
                          const double* temps, double* theta, double* logpost, double* chol, unsigned* naccept,
                          unsigned* nswap, double* samples, double* sample_lp, int series_flags, bool load_factor, hipStream_t st)
{
    (void)hipGetLastError();
    if (p < 1 || L.T < 1 || L.T > 64 || L.d < 4 || L.d > RAM_DMAX || (p == 1) != (L.d == 4)) return hipErrorInvalidValue;
    const long nc = (long)L.R * L.T;
    const RamState S = ram_state(scratch, L.d, nc);
    const int LPW = 64 / L.T;
    const long waves = ((long)L.R + LPW - 1) / LPW;
    const unsigned grid = (unsigned)((waves + 3) / 4), gridc = (unsigned)((nc + 255) / 256);
    hipLaunchKernelGGL(k_ram_convert, dim3(gridc), dim3(256), 0, st, L, S, theta, logpost, chol, 1,
                       RAM_CONV_STATE | (load_factor ? RAM_CONV_FACTOR : 0));
    hipError_t e = hipGetLastError();
    for (int it = 0; it < L.niter && e == hipSuccess; it++) {
        PtLaunch Li = L;
        Li.iter0 = L.iter0 + (unsigned long long)it;
        Li.niter = 1;
        const bool save = L.save_thin > 0 && ((it + 1) % L.save_thin) == 0;
        Li.save_thin = save ? 1 : 0;
        Li.save_offset = save ? L.save_offset + (it + 1) / L.save_thin - 1 : 0;
        // (later proposals ride on the finish kernel of the iteration before -- where that kernel holds the factor, v, the value AND
        // the draws in registers: d <= 12.  Beyond, the fused form spills -- 296 bytes per lane at d = 13, 1136 at d = 16, where it
        // takes 88 us for 16 384 chains against 21 + 27 for the two kernels: profiles/r04/lane_sampler_kernels_p7_v1.txt)
        const bool fused = L.d <= 12;
        if (it == 0 || !fused)
            e = ram_launch_d(L.d, [&](auto dc) {
                hipLaunchKernelGGL((k_ram_propose<decltype(dc)::value>), dim3(grid), dim3(256), 0, st, Li, S);
            });
        if (e == hipSuccess)
            e = p == 1 ? launch_logdens_car1(S.thn, (int)nc, series, L.n, pr, S.ll, st)
                       : launch_logdens_carma(p, S.thn, (int)nc, L.d, L.q, series, L.n, pr, 0, S.ll, st, series_flags);
        if (e == hipSuccess) {
            const bool next = fused && it + 1 < L.niter;
            e = ram_launch_d(L.d, [&](auto dc) {
                if (next)
                    hipLaunchKernelGGL((k_ram_finish<decltype(dc)::value, true>), dim3(grid), dim3(256), 0, st, Li, S, temps, naccept,
                                       nswap, samples, sample_lp);
                else
                    hipLaunchKernelGGL((k_ram_finish<decltype(dc)::value, false>), dim3(grid), dim3(256), 0, st, Li, S, temps, naccept,
                                       nswap, samples, sample_lp);
            });
        }
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_ram_convert, dim3(gridc), dim3(256), 0, st, L, S, theta, logpost, chol, 0, RAM_CONV_STATE);
        e = hipGetLastError();
    }
    return e;
}

// the chain-minor factors -> the chain-major array chol [R][T][d][d] (enqueued on st)
hipError_t pt_lane_store_factor(int d, int T, int R, double* scratch, double* chol, hipStream_t st)
{
    (void)hipGetLastError();
    PtLaunch L{};
    L.d = d;
    L.T = T;
    L.R = R;
    const long nc = (long)R * T;
    const RamState S = ram_state(scratch, d, nc);
    hipLaunchKernelGGL(k_ram_convert, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, L, S, nullptr, nullptr, chol, 0, RAM_CONV_FACTOR);
    return hipGetLastError();
}

}  // namespace carma
