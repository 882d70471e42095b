// carma_pt.hip -- K3/K4: the parallel-tempered Robust-Adaptive-Metropolis sampler as ONE
// persistent kernel per chunk of iterations.
//
// Grid: one workgroup per REPLICA (an independent ladder of T tempered chains); inside it one
// G-lane group per chain, so a CARMA(5,3) ladder of 16 temperatures is 128 lanes = 2 wavefronts.
// Chain state (theta, proposal, RAM Cholesky factor, stored log-posterior) lives in LDS for the
// whole launch; each iteration is
//     every chain: t_8 proposal -> Kalman log-density (K1 core) -> MH accept -> RAM rank-1 update
//     __syncthreads; lane 0: exchange sweep hot -> cold; __syncthreads; optional save of chain 0
// and nothing touches HBM except the wave-uniform series records (scalar loads, L2 resident) and
// the saved samples.  Reference: src/carmcmc.cpp:79-177, src/samplers.cpp:37-124,
// src/steps.cpp:36-131, src/include/steps.hpp:318-362.
//
// Interleaving differs from the reference's strictly serial sweep (RAM(T-1), swap(T-1,T-2),
// RAM(T-2), ...): here all RAM steps of an iteration run concurrently and the swap sweep follows.
// Both are compositions of kernels that leave the tempered joint posterior invariant.
#include <hip/hip_runtime.h>

#include "grp_device.h"
#include "carma_pt_core.h"
#include "carma_launch.h"

namespace carma {

template <int P>
struct PtGroupOf {
    static constexpr int value = P <= 1 ? 4 : (P <= 2 ? 2 : (P <= 4 ? 4 : 8));
};

template <int P, int G, int MAXT>
__global__ __launch_bounds__(MAXT) void k_pt(PtLaunch L, const double4* __restrict__ series, Prior pr, const double* __restrict__ temps,
                     double* __restrict__ theta, double* __restrict__ logpost, double* __restrict__ chol,
                     unsigned* __restrict__ naccept, unsigned* __restrict__ nswap, double* __restrict__ samples,
                     double* __restrict__ sample_lp)
{
    extern __shared__ double4 smem4[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int d = L.d, T = L.T, per = 4 * d + d * d;
    double4* xch = smem4;                                   // one exchange slot per lane
    double2* xch2 = reinterpret_cast<double2*>(smem4 + nthr);   // second exchange array (rho)
    double* base = reinterpret_cast<double*>(smem4 + nthr + nthr / 2); // per chain: th, thn, z, v, R
    double* s_lp = base + (size_t)T * per;
    double* s_temps = s_lp + T;
    unsigned* s_nswap = reinterpret_cast<unsigned*>(s_temps + T);
    const long b = blockIdx.x;

    for (int i = tid; i < T * d; i += nthr) base[(i / d) * per + (i % d)] = theta[b * T * d + i];
    for (int i = tid; i < T * d * d; i += nthr) base[(i / (d * d)) * per + 4 * d + (i % (d * d))] = chol[b * T * d * d + i];
    for (int i = tid; i < T; i += nthr) {
        s_lp[i] = logpost[b * T + i];
        s_temps[i] = temps[i];
        s_nswap[i] = 0;
    }
    __syncthreads();

    Grp<G> g{xch + (tid & ~63), tid & 63, xch2 + (tid & ~63)};
    const int c = tid / G;
    const bool active = c < T;
    const int cc = active ? c : 0;
    ChainScratch cs;
    cs.th = base + (size_t)cc * per;
    cs.thn = cs.th + d;
    cs.z = cs.th + 2 * d;
    cs.v = cs.th + 3 * d;
    cs.R = cs.th + 4 * d;
    const uint32_t chain_base = (uint32_t)((L.replica0 + b) * L.T_global + L.slot0);
    RngKey key{L.seed0, L.seed1, chain_base + (uint32_t)cc};
    double lp = s_lp[cc];
    const double temperature = s_temps[cc];
    unsigned nacc = 0;

    for (int it = 0; it < L.niter; it++) {
        const uint64_t iter = L.iter0 + (uint64_t)it;
        if (active) {
            if (ram_step<P, G>(g, cs, d, L.q, temperature, iter, L.maxiter, key, series, L.n, pr, &lp)) nacc++;
            if (g.lane() == 0) s_lp[c] = lp;
        }
        if (L.do_exchange && T > 1) {
            __syncthreads();
            if (tid == 0) exchange_sweep(T, d, per, base, s_lp, s_temps, key, chain_base, iter, s_nswap);
            __syncthreads();
            if (active) lp = s_lp[c];
        }
        if (L.save_thin > 0 && ((it + 1) % L.save_thin) == 0 && tid < G) {
            // coldest chain of this replica (Sampler::SaveValues, src/samplers.cpp:118-124)
            const long s = L.save_offset + (it + 1) / L.save_thin - 1;
            if (s < L.sample_cap) {
                for (int j = tid; j < d; j += G) samples[(b * L.sample_cap + s) * d + j] = base[j];
                if (tid == 0) sample_lp[b * L.sample_cap + s] = s_lp[0];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < T * d; i += nthr) theta[b * T * d + i] = base[(i / d) * per + (i % d)];
    for (int i = tid; i < T * d * d; i += nthr) chol[b * T * d * d + i] = base[(i / (d * d)) * per + 4 * d + (i % (d * d))];
    for (int i = tid; i < T; i += nthr) {
        logpost[b * T + i] = s_lp[i];
        nswap[b * T + i] += s_nswap[i];
    }
    if (active && g.lane() == 0) naccept[b * T + c] += nacc;
}

size_t pt_lds_bytes(int P, int d, int T, int* nthreads_out)
{
    const int G = P <= 1 ? 4 : (P <= 2 ? 2 : (P <= 4 ? 4 : 8));
    int nthr = ((T * G + 63) / 64) * 64;
    if (nthreads_out) *nthreads_out = nthr;
    size_t per = 4 * (size_t)d + (size_t)d * d;
    return (size_t)nthr * 48 + ((size_t)T * per + 2 * (size_t)T) * 8 + (size_t)T * 4 + 16;
}

template <int P>
static hipError_t launch_pt_p(const PtLaunch& L, const double4* series, const Prior& pr, const double* temps,
                              double* theta, double* logpost, double* chol, unsigned* naccept, unsigned* nswap,
                              double* samples, double* sample_lp, hipStream_t st)
{
    constexpr int G = PtGroupOf<P>::value;
    int nthr = 0;
    size_t lds = pt_lds_bytes(P, L.d, L.T, &nthr);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pt<P, G, 256>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pt<P, G, 1024>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (nthr <= 256)
        hipLaunchKernelGGL((k_pt<P, G, 256>), dim3((unsigned)L.R), dim3((unsigned)nthr), lds, st, L, series, pr, temps,
                           theta, logpost, chol, naccept, nswap, samples, sample_lp);
    else
        hipLaunchKernelGGL((k_pt<P, G, 1024>), dim3((unsigned)L.R), dim3((unsigned)nthr), lds, st, L, series, pr, temps,
                           theta, logpost, chol, naccept, nswap, samples, sample_lp);
    return hipGetLastError();
}

hipError_t launch_pt(int p, const PtLaunch& L, const double4* series, const Prior& pr, const double* temps,
                     double* theta, double* logpost, double* chol, unsigned* naccept, unsigned* nswap, double* samples,
                     double* sample_lp, hipStream_t st)
{
    switch (p) {
        case 1: return launch_pt_p<1>(L, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        case 2: return launch_pt_p<2>(L, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        case 3: return launch_pt_p<3>(L, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        case 4: return launch_pt_p<4>(L, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        case 5: return launch_pt_p<5>(L, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        case 6: return launch_pt_p<6>(L, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        case 7: return launch_pt_p<7>(L, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace carma
