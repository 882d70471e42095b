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
#include <atomic>
#include <cstdlib>

#include "grp_device.h"
#include "carma_pt_core.h"
#include "carma_pt_row.h"
#include "carma_ring.h"
#include "carma_pipe3l.h"
#include "carma_pipew.h"
#include "carma_launch.h"

namespace carma {

template <int P>
struct PtGroupOf {
    static constexpr int value = P <= 1 ? 4 : (P <= 2 ? 2 : (P <= 4 ? 4 : 8));
};

// PC = true: latency-regime variant.  The workgroup carries one rho-producer wave per consumer wave
// (carma_ring.h): threads [0, nthr/2) are the chains, threads [nthr/2, nthr) compute the transition
// factors of the same chains' proposals into per-wave LDS rings.
template <int P, int G, int MAXT, bool PC>
__global__ __launch_bounds__(MAXT) void k_pt(PtLaunch L, const double4* __restrict__ series, Prior pr, const double* __restrict__ temps,
                     double* __restrict__ theta, double* __restrict__ logpost, double* __restrict__ chol,
                     unsigned* __restrict__ naccept, unsigned* __restrict__ nswap, double* __restrict__ samples,
                     double* __restrict__ sample_lp)
{
    extern __shared__ double4 smem4[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int nthr_c = PC ? nthr / 2 : nthr;                // consumer (chain) threads
    const int d = L.d, T = L.T, per = 4 * d + d * d;
    double4* xch = smem4;                                   // one exchange slot per chain lane
    double2* xch2 = reinterpret_cast<double2*>(smem4 + nthr_c);   // second exchange array (rho)
    Cx* rings = reinterpret_cast<Cx*>(smem4 + nthr_c + nthr_c / 2);   // PC: one ring per consumer wave
    const size_t ring_entries = PC ? (size_t)(nthr_c / 64) * RingGeom<P>::ENTRIES : 0;
    double* base = reinterpret_cast<double*>(rings + ring_entries);    // per chain: th, thn, z, v, R
    double* s_lp = base + (size_t)T * per;
    double* s_temps = s_lp + T;
    double* s_dbeta = s_temps + T;                           // 1/T_i - 1/T_{i-1}
    double* s_logu = s_dbeta + T;                            // log of the swap uniform of pair (i, i-1)
    double* s_stage = s_logu + T;                            // [T][d] theta staging for the swap
    unsigned* s_nswap = reinterpret_cast<unsigned*>(s_stage + (size_t)T * d);
    int* s_src = reinterpret_cast<int*>(s_nswap + T);
    const long b = blockIdx.x;

    for (int i = tid; i < T * d; i += nthr) base[(i / d) * per + (i % d)] = theta[b * T * d + i];
    for (int i = tid; i < T * d * d; i += nthr) base[(i / (d * d)) * per + 4 * d + (i % (d * d))] = chol[b * T * d * d + i];
    for (int i = tid; i < T; i += nthr) {
        s_lp[i] = logpost[b * T + i];
        s_temps[i] = temps[i];
        s_dbeta[i] = i > 0 ? 1.0 / temps[i] - 1.0 / temps[i - 1] : 0.0;
        s_nswap[i] = 0;
    }
    __syncthreads();

    const bool producer = PC && tid >= nthr_c;
    const int ctid = producer ? tid - nthr_c : tid;         // lane of the chain this thread serves
    Grp<G> g{xch + (ctid & ~63), ctid & 63, xch2 + (ctid & ~63)};
    Cx* ring = rings + (size_t)(ctid >> 6) * RingGeom<P>::ENTRIES;
    const int c = ctid / G;
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
        if constexpr (PC) {
            // every wave passes: barrier A (proposals visible), then one barrier per ring chunk
            // (a chain wave always holds at least one active group, so wrapping the chain work in
            // `active` never lets a whole wave skip a barrier)
            double znorm2 = 0.0;
            if (!producer && active) znorm2 = ram_propose<G>(g, cs, d, iter, key);
            __syncthreads();
            if (producer) {
                const int rr = g.lane() < P ? g.lane() : P - 1;
                ring_produce<P, G>(g, own_ar_root<P>(cs.thn, rr), series, L.n, ring);
            } else if (active) {
                Model<P> m;
                model_from_theta<P, G>(g, cs.thn, L.q, pr, 0, m);
                bool sing;
                double ll = ring_consume<P, G>(g, m, series, L.n, ring, &sing);
                ll += log_prior(m.scale, pr.measerr_dof);
                if (sing || !m.valid) ll = -1.0 / 0.0;
                if (ram_finish<G>(g, cs, d, temperature, iter, L.maxiter, key, ll, znorm2, &lp)) nacc++;
                if (g.lane() == 0) s_lp[c] = lp;
            }
        } else if (active) {
            if (ram_step<P, G>(g, cs, d, L.q, temperature, iter, L.maxiter, key, series, L.n, pr, &lp)) nacc++;
            if (g.lane() == 0) s_lp[c] = lp;
        }
        if (L.do_exchange && T > 1) {
            // ExchangeStep sweep (steps.hpp:318-362): parallel draws + staging, serial decisions on
            // the log-posteriors only, parallel theta moves.
            if (active && !producer) {
                for (int j = g.lane(); j < d; j += G) s_stage[c * d + j] = cs.th[j];
                if (g.lane() == 0) {
                    RngKey k2 = key;                      // keyed by the hotter chain's global slot
                    s_logu[c] = c > 0 ? log(rng_uniform(k2, iter, RNG_SWAP, 0)) : 0.0;
                    s_src[c] = c;
                }
            }
            __syncthreads();
            if (tid == 0) exchange_decide(T, s_lp, s_dbeta, s_logu, s_src, s_nswap);
            __syncthreads();
            if (active && !producer) {
                const int from = s_src[c];
                if (from != c)
                    for (int j = g.lane(); j < d; j += G) cs.th[j] = s_stage[from * d + j];
                lp = s_lp[c];
            }
            __syncthreads();
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
    if (active && !producer && g.lane() == 0) naccept[b * T + c] += nacc;
}

// LDS bytes and threads of one workgroup; *pc_out tells whether the producer/consumer variant fits
// (and pays: it needs p >= 2 and a series long enough to amortise the chunk barriers).
size_t pt_lds_bytes(int P, int d, int T, int* nthreads_out, int* pc_out)
{
    const int G = P <= 1 ? 4 : (P <= 2 ? 2 : (P <= 4 ? 4 : 8));
    const int nthr_c = ((T * G + 63) / 64) * 64;
    const size_t per = 4 * (size_t)d + (size_t)d * d;
    const size_t state = ((size_t)T * per + 4 * (size_t)T + (size_t)T * d) * 8 + (size_t)T * 8 + 16;
    const size_t plain = (size_t)nthr_c * 48 + state;
    const size_t ring_bytes = RingGeom<2>::BYTES;            // same for every P
    const size_t with_ring = plain + (size_t)(nthr_c / 64) * ring_bytes;
    const bool pc = P >= 2 && 2 * nthr_c <= 1024 && with_ring <= 150 * 1024;
    if (pc_out) *pc_out = pc ? 1 : 0;
    if (nthreads_out) *nthreads_out = pc ? 2 * nthr_c : nthr_c;
    return pc ? with_ring : plain;
}

template <int P, int G, int MAXT, bool PC>
static hipError_t launch_pt_k(const PtLaunch& L, int nthr, size_t lds, const double4* series, const Prior& pr,
                              const double* temps, double* theta, double* logpost, double* chol, unsigned* naccept,
                              unsigned* nswap, double* samples, double* sample_lp, hipStream_t st)
{
    // before every launch (see launch_logdens_p: once per process is not enough)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pt<P, G, MAXT, PC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_pt<P, G, MAXT, PC>), dim3((unsigned)L.R), dim3((unsigned)nthr), lds, st, L, series, pr, temps,
                       theta, logpost, chol, naccept, nswap, samples, sample_lp);
    return hipGetLastError();
}

template <int P>
static hipError_t launch_pt_p(const PtLaunch& L, const double4* series, const Prior& pr, const double* temps,
                              double* theta, double* logpost, double* chol, unsigned* naccept, unsigned* nswap,
                              double* samples, double* sample_lp, hipStream_t st)
{
    constexpr int G = PtGroupOf<P>::value;
    int nthr = 0, pc = 0;
    const size_t lds = pt_lds_bytes(P, L.d, L.T, &nthr, &pc);
    if constexpr (P >= 2) {
        // The producer/consumer variant halves the instruction stream of the chain waves but doubles the wave count: it
        // wins while a CU holds at most one ladder (12.3k vs 8.0k it/s at 16 x 256) and loses once the ladders queue up
        // for the SIMDs (3.1k vs 4.0k it/s at 16 x 1024, tools/mcmc_bigR_probe.py; the plain chain waves share the
        // exp/sincos inside root pairs, RhoPair).  CARMA_PT_PLAIN=0/1 overrides.
        bool plain = L.R > (long)device_cus();
        static const int force_plain = [] {                 // CARMA_PT_PLAIN=0/1, read once
            const char* fp = getenv("CARMA_PT_PLAIN");
            return fp ? (fp[0] == '1' ? 1 : 0) : -1;
        }();
        if (force_plain >= 0) plain = force_plain == 1;
        if (pc && L.n >= 32 && !plain) {
            if (nthr <= 256)
                return launch_pt_k<P, G, 256, true>(L, nthr, lds, series, pr, temps, theta, logpost, chol, naccept, nswap,
                                                    samples, sample_lp, st);
            return launch_pt_k<P, G, 1024, true>(L, nthr, lds, series, pr, temps, theta, logpost, chol, naccept, nswap,
                                                 samples, sample_lp, st);
        }
    }
    // plain variant: every chain wave computes its own rho
    int nthr_plain = pc ? nthr / 2 : nthr;
    const size_t lds_plain = pc ? lds - (size_t)(nthr_plain / 64) * RingGeom<2>::BYTES : lds;
    if (nthr_plain <= 256)
        return launch_pt_k<P, G, 256, false>(L, nthr_plain, lds_plain, series, pr, temps, theta, logpost, chol, naccept,
                                             nswap, samples, sample_lp, st);
    return launch_pt_k<P, G, 1024, false>(L, nthr_plain, lds_plain, series, pr, temps, theta, logpost, chol, naccept,
                                          nswap, samples, sample_lp, st);
}

// ---------------------------------------------------------------------------------------------
// Row variant: ONE CHAIN PER 16-LANE DPP ROW (the wave pipeline of carma_pipe3l.h), 4 chains per
// workgroup, a ladder of T chains spread over wpl = ceil(T/4) workgroups on different CUs.
//
// Why: with one ladder per workgroup (k_pt) a CU hosts 2 chain waves + 2 producer waves, and a Kalman
// step costs the CU ~2200 SIMD-cycles -- the kernel is bound by the issue rate of ONE CU while 3/4 of
// the chip idles (16 temperatures x 64 replicas use 64 of 256 CUs).  The row loop needs ~200 cycles per
// step and chain instead of ~560, and spreading the ladder gives every group of 4 chains its own CU.
// The price is that the swap sweep crosses workgroups: once per iteration the ladder's workgroups
// publish (theta, log-posterior) to global memory, meet at an arrival counter (all of them are
// resident: the launch is cooperative), and every workgroup replays the same hot->cold decisions
// (exchange_decide, same counter-based uniforms) for its ladder.  Staging is double buffered, so a
// workgroup can run at most one exchange ahead of the slowest one.  A rendezvous that nevertheless does
// not complete within ~seconds sets abort_flag and the launch ends -- the host restores the chunk and
// re-runs it with the ladder kernel -- instead of hanging the GPU.
//
// Round 3: EVERY PART HAS ITS OWN LOOP.  The four waves of a workgroup play four parts (covariance recursion, mean
// recursion, two producers).  In round 2 they shared one iteration loop and branched on the part inside it, so every
// part's values were live across every other part's code: 256 VGPRs + 108 AGPRs, one workgroup per CU.  Now each part
// runs its own loop (the waves meet at barriers only, which count arrivals, not program counters), and the register
// count is that of the hungriest part alone: the chains (theta, the proposal, column j of the RAM Cholesky factor in
// lane j: carma_pt_row.h) stay with the covariance wave, the leanest one (72 registers for the recursion + 40 for the
// chain), and the kernel fits the 168 registers that let THREE workgroups share a CU -- what 16 x 65..192 ladders need
// to keep this kernel.  The swap step is split between the two recursion waves: the chain wave accepts, publishes its
// chains, then adapts the proposal factor and forms the next proposal's (z, R^T z); MEANWHILE the mean wave -- idle
// once it has handed the log-density over -- waits for the ladder at the rendezvous, fetches the staged log-posteriors
// and parameter vectors and replays the sweep.  The two meet at a barrier, the chain wave picks up what the sweep
// assigned to its rows, adds R^T z and the next filter starts.
// TAGGED STAGING (round 3).  The chains a ladder's workgroups exchange in the swap step travel as self-validating words:
// every 64-bit value v is stored twice, as v ^ tag1 and v ^ tag2 with tags derived from (iteration, launch counter), by
// relaxed agent-scope stores with nothing to wait for; a reader loads both words and takes the value when
// (a ^ tag1) == (b ^ tag2) -- a word left over from an earlier iteration (the buffers alternate) or never written fails
// that test (for a wrong pair to pass, two independent 64-bit hashes would have to collide).  Every 8-byte access is
// atomic by itself, so no ordering between the words is needed: the reader simply polls the ladder's (T (d + 1)) pairs
// until all of them validate -- ONE round trip after the last store has landed, instead of round 2's "stores, wait for
// their acknowledgement, bump an arrival counter | poll the counter, then fetch" (two waits on the publishing side, two
// round trips on the reading side): 34.1 -> 33.0 us per iteration at 16 x 64.
__device__ __forceinline__ unsigned long long pt_tag(unsigned long long iter, unsigned long long epoch, unsigned long long salt)
{
    unsigned long long z = (iter + 1) * 0x9E3779B97F4A7C15ull + epoch * 0xD1B54A32D192ED03ull + salt;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// MINW = workgroups of this kernel a CU is to hold (waves per SIMD): 2 -> up to 256 registers, nothing spilled; 3 -> 168
// registers, where the cold code of the swap step and of the random-number tails spills a few values around itself (the
// recursion loops do not).  The host takes MINW = 3 only for grids of more than two workgroups per CU.
// WIN (round 5): the log-density through the WINDOWED wave pipeline (carma_pipew.h) instead of the one-datum pipeline -- the chain
// wave runs the recursion of covariance AND mean and gets the log-likelihood itself, the mean wave hands prior bounds and log prior
// over through LDS, is the third producer, and keeps the sweep.  Taken where the whole ladder's grid is at most one workgroup per
// CU (launch_pt_row_p): 16 x 64 ladders 31.1 -> 32.9 * 10^3 it/s on one box (profiles/r05/window_pipeline_v1.txt); with more
// workgroups per CU the one-datum pipeline is ahead, as for the log-density kernels.
// TWO (round 6, with WIN): the TWO-SIDED window pipeline (carma_pipew.h, TS) -- a chain takes two DPP rows, the even one filters the
// first half of the series forward, the odd one the second half backward, and the chain wave merges the two states; two chains per
// workgroup, a ladder over ceil(T / 2) of them.  Both rows carry the chain's state (the same values, formed twice), the even row
// publishes it.  The series sits in LDS for the producers (copied once per launch).
template <int P, int MINW, bool WIN = false, bool TWO = false, bool HO = false, bool SL = TWO>
__global__ __launch_bounds__(256, MINW) void k_pt_row(PtLaunch L, PtRowSync S, const double4* __restrict__ series, Prior pr,
                                                   const double* __restrict__ temps, double* __restrict__ theta,
                                                   double* __restrict__ logpost, double* __restrict__ chol,
                                                   unsigned* __restrict__ naccept, unsigned* __restrict__ nswap,
                                                   double* __restrict__ samples, double* __restrict__ sample_lp)
{
    static_assert(!TWO || WIN, "the two-sided form is the window pipeline's");
    constexpr int G = 16, CPW = TWO ? 2 : 4;               // lanes per chain row, chains per workgroup
    extern __shared__ double4 smem4[];
    const int tid = threadIdx.x, lane64 = tid & 63;
    // which wave plays which part: as in k_logdens_carma_p3l (workgroups i, i + ncu, i + 2 ncu share a CU)
    //                 part of wave:  0  1  2  3      (0 covariance + chains, 1 mean + swap sweep, 2 / 3 producers)
    const int round = (blockIdx.x >= (unsigned)S.ncu) + (blockIdx.x >= 2u * (unsigned)S.ncu);
    const int wave = ((round == 0 ? 0xE4 : round == 1 ? (S.rot & 0xff) : (S.rot >> 8)) >> (2 * (tid >> 6))) & 3;
    const int d = L.d, T = L.T;
    Cx* ring = reinterpret_cast<Cx*>(smem4);               // carma_pipe3l.h rings (WIN: carma_pipew.h's, in the same space)
    double2* ringw = reinterpret_cast<double2*>(smem4);
    static_assert(PipeWGeom<P>::ENTRIES <= Pipe3LGeom<P>::ENTRIES, "the window pipeline's LDS fits the one-datum pipeline's");
    if constexpr (WIN) math_tab_fill(reinterpret_cast<double*>(ringw + PipeWGeom<P>::TAB_OFF));     // visible behind the barrier below
    double* s_thn = reinterpret_cast<double*>(ring + Pipe3LGeom<P>::ENTRIES);  // [CPW][16] proposals
    double* s_ll = s_thn + CPW * PT_DMAX;                  // [CPW] log-density of the proposals (mean wave -> chain wave)
    double* s_ua = s_ll + CPW;                             // [CPW] acceptance uniforms of this iteration (producer wave 1)
    double* s_lp = s_ua + CPW;                             // [T] the ladder's log-posteriors after the sweep
    double* s_dbeta = s_lp + T;
    double* s_logu = s_dbeta + T;
    unsigned* s_nswap = reinterpret_cast<unsigned*>(s_logu + T);
    int* s_src = reinterpret_cast<int*>(s_nswap + T);      // [T] whose parameter vector temperature i holds after the sweep
    int* s_flag = s_src + T;                               // [1] abort seen (+ pad)
    double* s_z = reinterpret_cast<double*>(s_flag + 2);   // [64] next iteration's t8 variates, drawn by producer wave 0
    double* s_lu = s_z + 64;                               // [64] this exchange's log-uniforms (T <= 64), producer wave 1
    double* s_step = s_lu + 64;                            // [1] this iteration's adaptation step length (+ pad), same
    double* s_tha = s_step + 2;                            // [T][d + 1] the ladder's staged (theta, log-posterior), validated copy
    // TWO: the series for the producers, {y, yerr^2}[n] then t[n] (16-byte aligned behind s_tha)
    double2* lds_yz = reinterpret_cast<double2*>(reinterpret_cast<char*>(smem4) +
                                                 ((reinterpret_cast<char*>(s_tha + T * (PT_DMAX + 1)) - reinterpret_cast<char*>(smem4) + 15) & ~(ptrdiff_t)15));
    double* lds_t = reinterpret_cast<double*>(lds_yz + L.n + (L.n & 1));
    // (SL: the series in LDS; without, for series the LDS does not hold, each row's window of 64 records in registers -- carma_pipew.h)
    if constexpr (TWO && SL) {
        for (int i = tid; i < L.n; i += 256) {
            const double4 r = series[i];
            lds_yz[i] = make_double2(r.y, r.z);
            lds_t[i] = r.w;
        }
    }
    // Which ladder, which part of it.  Workgroups are dealt to the eight XCDs (each with its own L2) round-robin by
    // blockIdx, and the swap step is an exchange between the workgroups of ONE ladder: S.xcd_map = 8 puts a ladder's wpl
    // workgroups on blockIdx b, b + 8, b + 16, ... -- the same XCD -- instead of b, b + 1, ... (eight different ones).
    // Placement only: the exchange uses agent-scope operations either way, so a device that deals differently is merely
    // slower.  (The host sets it when the replicas fill whole groups of eight.)
    long lad;
    int part;
    if (S.xcd_map > 1) {
        const unsigned x = blockIdx.x % (unsigned)S.xcd_map, slot = blockIdx.x / (unsigned)S.xcd_map;
        lad = (long)(slot / (unsigned)S.wpl) * S.xcd_map + x;
        part = (int)(slot % (unsigned)S.wpl);
    } else {
        lad = blockIdx.x / S.wpl;
        part = (int)(blockIdx.x % S.wpl);
    }
    const long ch0 = lad * T;                              // first chain of the ladder in the state arrays

    for (int i = tid; i < T; i += 256) {
        s_dbeta[i] = i > 0 ? 1.0 / temps[i] - 1.0 / temps[i - 1] : 0.0;
        s_nswap[i] = 0;
    }
    if (tid == 0) *s_flag = 0;

    Grp<G> g{nullptr, lane64, nullptr};
    const int row = lane64 >> 4, j = lane64 & 15;
    const int crow = TWO ? row >> 1 : row;                 // chain of the workgroup this row works for
    const bool pub = !TWO || (row & 1) == 0;               // the row that publishes / stores the chain
    const int c = part * CPW + crow;                       // chain (temperature index) of this row
    const bool active = c < T;
    // rows past the ladder's end shadow its last chain: uniform control flow, nothing written back
    const int cc = active ? c : T - 1;
    double* thn_lds = s_thn + crow * PT_DMAX;
    const uint32_t chain_base = (uint32_t)((L.replica0 + lad) * L.T_global + L.slot0);
    const RngKey key{L.seed0, L.seed1, chain_base + (uint32_t)cc};
    const int npad = p3l_pad(L.n);                         // neutral pad data completing the last chunk (carma_types.h)
    const bool exch = L.do_exchange && T > 1;
    const size_t nchain_all = (size_t)L.R * T;
    __syncthreads();
    // Barriers of one iteration, the same for every part: "proposals visible", the pipeline's own (carma_pipe3l.h),
    // "log-densities visible", "sweep done".

    if (wave >= 2) {
        // ---- producers.  Behind the pipeline's last barrier -- the mean wave still has a chunk to go, then come the
        // Metropolis decisions -- they draw the random numbers that are functions of key and iteration only: P0 the NEXT
        // iteration's proposal variates and this iteration's adaptation step length, P1 the logs of the swap uniforms of
        // THIS iteration's exchange.  (Outside pipe3l_produce: nothing of the pipeline is live across these draws, which
        // is what keeps the 168-register build of this loop free of spills.)
        for (int it = 0; it < L.niter; it++) {
            const uint64_t iter = L.iter0 + (uint64_t)it;
            __syncthreads();                               // proposals visible
            if constexpr (WIN)
                pipew_produce<P, TWO, TWO && SL, HO>(g, wave - 2, thn_lds, series, L.n, ringw, [](int) {}, nullptr, lds_t, lds_yz);
            else
                pipe3l_produce<P>(g, wave - 2, thn_lds, series, L.n + npad, npad, ring, [](int) {});
            if (wave == 2) {
                s_z[lane64] = rng_student_t8(key, iter + 1, (uint32_t)(j < d ? j : 0));
                if (lane64 == 0) *s_step = ram_adapt_step(d, iter);
            } else {
                if (exch && T <= 64) {
                    const int i = lane64 < T ? lane64 : T - 1;
                    RngKey k2{L.seed0, L.seed1, chain_base + (uint32_t)i};
                    s_lu[lane64] = i > 0 ? log(rng_uniform(k2, iter, RNG_SWAP, 0)) : 0.0;
                }
                const double ua = rng_uniform(key, iter, RNG_ACCEPT, 0);     // the rows' Metropolis uniforms
                if (j == 0) s_ua[crow] = ua;
            }
            __syncthreads();                               // log-densities (and these draws) visible
            __syncthreads();                               // sweep done
            if (*s_flag) break;
        }
        return;
    }

    if (wave == 1) {
        // ---- mean recursion of the four proposals, then the ladder's swap sweep (steps.hpp:318-362)
        int buf = 0;
        for (int it = 0; it < L.niter; it++) {
            const uint64_t iter = L.iter0 + (uint64_t)it;
            __syncthreads();                               // proposals visible
            {
                Model<P> m;
                model_from_theta<P, G, MODEL_FLAGS>(g, thn_lds, L.q, pr, 0, m);
                if constexpr (WIN) {
                    const double lpri = log_prior(m.scale, pr.measerr_dof);
                    if (j == 0) ringw[PipeWGeom<P>::OUT_OFF + row] = make_double2(lpri, m.valid ? 1.0 : 0.0);
                    pipew_produce<P, TWO, TWO && SL, HO>(g, 2, thn_lds, series, L.n, ringw, [](int) {}, nullptr, lds_t, lds_yz);
                } else {
                double lpri = log_prior(m.scale, pr.measerr_dof) + pipe3l_pad_correction(npad, thn_lds[0], m.scale, series[L.n - 1].y, m.mu);
                asm volatile("" : "+v"(lpri));
                bool sing;
                double ll = pipe3l_mean<P>(g, m.mu, series, L.n + npad, npad, ring, &sing);
                ll += lpri;
                if (sing || !m.valid) ll = -1.0 / 0.0;
                if (j == 0) s_ll[crow] = ll;
                }
            }
            __syncthreads();                               // log-densities visible to the chain wave
            __builtin_amdgcn_s_setprio(3);                 // the ladder waits for this sweep (back to 1 in pipe3l_mean)
#if defined(CARMA_STAMPS)
            unsigned long long b0 = 0, b1 = 0, b2 = 0;
            CARMA_STAMP(b0);
#endif
            if (exch) {
                // the ladder's staged chains: poll the tagged words until every pair validates (see "tagged staging")
                const size_t nval = nchain_all * (size_t)(d + 1);
                const unsigned long long* st_a = S.stage + (size_t)buf * 2 * nval + (size_t)ch0 * (d + 1);
                const unsigned long long* st_b = st_a + nval;
                const unsigned long long tg1 = pt_tag(iter, S.epoch, 1), tg2 = pt_tag(iter, S.epoch, 2);
                const int NV = T * (d + 1);
                int aborted = 0;
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
                    for (int v = lane64; v < NV; v += 64) {
                        const unsigned long long a = __hip_atomic_load(&st_a[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ tg1;
                        const unsigned long long b = __hip_atomic_load(&st_b[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ tg2;
                        ok = ok && (a == b);
                        s_tha[v] = __longlong_as_double((long long)a);
                    }
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                    if (lane64 == 0 && (__hip_atomic_load(S.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || ++spins > 5000000u)) {
                        __hip_atomic_store(S.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *s_flag = 1;
                        aborted = 1;
                    }
                    aborted = __builtin_amdgcn_readfirstlane(aborted);
                    if (aborted) break;
                    __builtin_amdgcn_s_sleep(2);
                }
#if defined(CARMA_STAMPS)
                CARMA_STAMP(b1);
#endif
                if (!aborted) {
                    g.sync();                              // this wave's LDS writes above, read back below
                    if (T <= 64) {
                        // lane i owns temperature i, the sweep runs through v_readlane
                        const int i = lane64 < T ? lane64 : T - 1;
                        double lp_i = s_tha[i * (d + 1) + d];
                        const double logu_i = s_lu[lane64];           // drawn by producer wave 1 behind the pipeline
                        int src_i = i;
                        bool sw;
                        exchange_decide_wave(T, lane64, lp_i, s_dbeta[i], logu_i, src_i, &sw);
                        if (lane64 < T) {
                            if (sw) s_nswap[lane64]++;
                            s_lp[lane64] = lp_i;
                            s_src[lane64] = src_i;
                        }
                    } else {
                        for (int i = lane64; i < T; i += 64) {
                            s_lp[i] = s_tha[i * (d + 1) + d];
                            s_src[i] = i;
                            RngKey k2{L.seed0, L.seed1, chain_base + (uint32_t)i};
                            s_logu[i] = i > 0 ? log(rng_uniform(k2, iter, RNG_SWAP, 0)) : 0.0;
                        }
                        g.sync();
                        if (lane64 == 0) exchange_decide(T, s_lp, s_dbeta, s_logu, s_src, s_nswap);
                    }
                }
                buf ^= 1;
            }
#if defined(CARMA_STAMPS)
            CARMA_STAMP(b2);
            if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && lane64 == 0 && it == L.niter - 1)
                printf("pt_row mean wave, workgroup %u (cycles): ll handed over at %llu | rendezvous wait %llu  fetch + sweep %llu\n", blockIdx.x,
                       b0 % 100000000ull, b1 - b0, b2 - b1);
#endif
            __syncthreads();                               // sweep done
            if (*s_flag) break;
        }
        if (part == 0)
            for (int i = lane64; i < T; i += 64) nswap[ch0 + i] += s_nswap[i];
        return;
    }

    // ---- covariance recursion + the chains: state in registers (carma_pt_row.h), lane j owns component j and column j of R
    RowChain ch;
    ch.th = j < d ? theta[(ch0 + cc) * d + j] : 0.0;
    ch.thn = ch.th;
#pragma unroll
    for (int k = 0; k < PT_DMAX; k++) ch.Rc[k] = (j < d && k <= j) ? chol[(ch0 + cc) * d * d + (size_t)k * d + j] : 0.0;
    double lp = logpost[ch0 + cc];
    const double temperature = temps[cc];
    unsigned nacc = 0;
    int buf = 0;
    // (z, v = R^T z) of the first proposal; those of the later ones are formed while the ladder meets for the swap
    double znorm2 = ram_draw_row(g, ch, d, rng_student_t8(key, L.iter0, (uint32_t)(j < d ? j : 0)));

    CARMA_STAMP_DECL;
#if defined(CARMA_STAMPS)
    unsigned long long st5 = 0, st6 = 0, st7 = 0;
#endif
    for (int it = 0; it < L.niter; it++) {
        const uint64_t iter = L.iter0 + (uint64_t)it;
        CARMA_STAMP(st0);
        ch.thn = ch.th + ch.v;                             // steps.cpp:72-73
        if (j < d) thn_lds[j] = ch.thn;
        __syncthreads();                                   // proposals visible to the other three waves
        CARMA_STAMP(st1);
        {
            Model<P> m;
            model_from_theta<P, G, MODEL_CONSTS>(g, thn_lds, L.q, pr, 0, m);
            FilterConsts<P> fc;
            filter_reset<P, G>(g, m, fc);
            RowConsts<P> rc;
            row_consts<P>(g, m, fc, rc);
            if constexpr (WIN) {
                double llw = pipew_recur<P, TWO>(g, rc, ringw);
                const double2 o = ringw[PipeWGeom<P>::OUT_OFF + row];
                llw += o.x;
                if (m.sing || o.y == 0.0) llw = -1.0 / 0.0;
                if (j == 0) s_ll[crow] = llw;
            } else {
                pipe3l_cov<P>(g, m, rc, series, L.n + npad, npad, ring);
            }
        }
        CARMA_STAMP(st2);
        __syncthreads();                                   // log-densities visible
        CARMA_STAMP(st3);
        double alpha;
        if (ram_accept_row(ch, temperature, iter, key, s_ll[crow], &lp, &alpha, s_ua + crow)) nacc++;
        if (exch) {
            // publish this workgroup's chains as tagged words (see "tagged staging"): fire and forget
            const size_t nval = nchain_all * (size_t)(d + 1);
            unsigned long long* st_a = S.stage + (size_t)buf * 2 * nval + (size_t)(ch0 + c) * (d + 1);
            unsigned long long* st_b = st_a + nval;
            const unsigned long long tg1 = pt_tag(iter, S.epoch, 1), tg2 = pt_tag(iter, S.epoch, 2);
            if (active && pub) {
                if (j < d) {
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(ch.th);
                    __hip_atomic_store(&st_a[j], bits ^ tg1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&st_b[j], bits ^ tg2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (j == 0) {                              // (d may be 16: the log-posterior has no lane of its own)
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(lp);
                    __hip_atomic_store(&st_a[d], bits ^ tg1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&st_b[d], bits ^ tg2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            buf ^= 1;
        }
        CARMA_STAMP(st4);
        // Neither the adaptation of the proposal factor (steps.cpp:82-99) nor the next proposal's (z, R^T z) touch the
        // state: they run while the mean wave waits for the ladder and replays the sweep.  (The producers' draws --
        // s_z, s_step -- are behind the "log-densities visible" barrier.)
        ram_adapt_row(g, ch, d, iter, L.maxiter, alpha, znorm2, *s_step);
        znorm2 = ram_draw_row(g, ch, d, s_z[lane64]);
        CARMA_STAMP(st5);
        __syncthreads();                                   // sweep done
        CARMA_STAMP(st6);
        if (*s_flag) break;
        if (exch) {
            const int from = s_src[cc];
            lp = s_lp[cc];
            if (from != cc && j < d) ch.th = s_tha[from * (d + 1) + j];
        }
        CARMA_STAMP(st7);
#if defined(CARMA_STAMPS)
        if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && lane64 == 0 && it == L.niter - 1)
            printf("pt_row chain wave, workgroup %u: iteration starts at %llu | propose+barrier %llu  model+reset+filter %llu  wait for ll %llu  accept+publish %llu  "
                   "adapt+draw %llu  wait for sweep %llu  pick up %llu\n",
                   blockIdx.x, st0 % 100000000ull, st1 - st0, st2 - st1, st3 - st2, st4 - st3, st5 - st4, st6 - st5, st7 - st6);
#endif
        if (L.save_thin > 0 && ((it + 1) % L.save_thin) == 0 && part == 0 && lane64 < G) {
            // coldest chain of the ladder (Sampler::SaveValues, src/samplers.cpp:118-124): row 0 of part 0
            const long s = L.save_offset + (it + 1) / L.save_thin - 1;
            if (s < L.sample_cap) {
                if (j < d) samples[(lad * L.sample_cap + s) * d + j] = ch.th;
                if (lane64 == 0) sample_lp[lad * L.sample_cap + s] = lp;
            }
        }
    }
    if (active && pub) {
        if (j < d) {
            theta[(ch0 + c) * d + j] = ch.th;
#pragma unroll
            for (int k = 0; k < PT_DMAX; k++)
                if (k <= j) chol[(ch0 + c) * d * d + (size_t)k * d + j] = ch.Rc[k];
        }
        if (j == 0) {
            logpost[ch0 + c] = lp;
            naccept[ch0 + c] += nacc;
        }
    }
}

static size_t pt_row_lds(int d, int T, int n_series = 0)
{
    (void)d;
    if (n_series > 0) return ((pt_row_lds(d, T) + 15) & ~(size_t)15) + 16 + (size_t)(n_series + (n_series & 1)) * 24;   // (two-sided: + the series)
    // rings, proposals [4][16], s_ll [4], s_ua [4], s_lp / s_dbeta / s_logu [T], s_nswap + s_src [T] (4 B each), flag, s_z / s_lu [64], step, s_tha
    return Pipe3LGeom<2>::BYTES + (4 * (size_t)PT_DMAX + 8 + 3 * (size_t)T) * 8 + (size_t)T * 8 + 16 + 128 * 8 + 16 +
           (size_t)T * (PT_DMAX + 1) * 8;
}

template <int P>
static const void* pt_row_fn(int minw, bool win = false, bool two = false, bool ho = false, bool sl = true)
{
    // (HO: the two-sided form's schedule hand-over, where workgroups share a CU: carma_pipew.h SSCHED; !sl: the series stays in global memory)
    if (two && minw < 3 && !sl) return reinterpret_cast<const void*>(&k_pt_row<P, 2, true, true, false, false>);
    if (two && minw < 3) return ho ? reinterpret_cast<const void*>(&k_pt_row<P, 2, true, true, true>) : reinterpret_cast<const void*>(&k_pt_row<P, 2, true, true, false>);
    if (win && minw < 3) return reinterpret_cast<const void*>(&k_pt_row<P, 2, true>);
    return minw >= 3 ? reinterpret_cast<const void*>(&k_pt_row<P, 3>) : reinterpret_cast<const void*>(&k_pt_row<P, 2>);
}

static const void* pt_row_fn_p(int p, int minw)
{
    switch (p) {
        case 2: return pt_row_fn<2>(minw);
        case 3: return pt_row_fn<3>(minw);
        case 4: return pt_row_fn<4>(minw);
        case 5: return pt_row_fn<5>(minw);
        case 6: return pt_row_fn<6>(minw);
        case 7: return pt_row_fn<7>(minw);
        default: return nullptr;
    }
}

long pt_row_capacity(int p, int d, int T, int n)
{
    const void* fn = pt_row_fn_p(p, 3);
    if (!fn || n < 32 || T < 1) return 0;
    const size_t lds = pt_row_lds(d, T);
    if (lds > 160 * 1024) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    int coop = 0;
    if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess || !coop) return 0;
    // Workgroups the device holds at once: what registers (168 -> three waves per SIMD) and LDS allow per CU, as the
    // runtime counts them -- the cooperative launch (launch_pt_row_p) checks the same thing again.  Beyond that the
    // ladder kernel (8 chains per wave) takes over.
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds) != hipSuccess || per_cu < 1) return 0;
    static const long tune = [] {
        const char* e = getenv("CARMA_TUNE_PT_ROW_WGS_PER_CU");      // measurements only; read once
        return e ? atol(e) : 0L;
    }();
    if (tune > 0 && tune < per_cu) per_cu = (int)tune;
    if (per_cu > 3) per_cu = 3;
    return (long)device_cus() * per_cu;
}

static std::atomic<int> g_row_pipeline{-1};
int pt_row_last_pipeline() { return g_row_pipeline.load(std::memory_order_relaxed); }

template <int P>
static hipError_t launch_pt_row_p(const PtLaunch& L, const PtRowSync& S, const double4* series, const Prior& pr,
                                  const double* temps, double* theta, double* logpost, double* chol, unsigned* naccept,
                                  unsigned* nswap, double* samples, double* sample_lp, hipStream_t st)
{
    // the TWO-SIDED window pipeline (a chain = two rows, ceil(T / 2) workgroups a ladder) where the WHOLE ladder set's grid in that form
    // is at most two workgroups per CU, the series suits the window pipeline and fits the LDS beside the rings; from T_global, so
    // that a sharded ladder's blocks decide as the one-GPU run does.  CARMA_TUNE_PT_ROW_WIN = 0 / 1 (one-datum / one-sided window)
    // or 2 (two-sided) overrides.
    const long grid2_global = (long)L.R * (((long)L.T_global + 1) / 2);
    const bool w2ok = (S.window_ok & SERIES_WINDOW2_OK) || ((S.window_ok & SERIES_WINDOW2_SMALL) && grid2_global <= (long)S.ncu);
    // the series sits in LDS: up to 1024 data (24 KiB) two workgroups still share a CU; longer ones -- as long as the LDS holds them
    // (~4800 data) -- where the ladder set leaves a CU to every workgroup, e.g. the single ladder of a run_mcmc call
    // -- and any longer series from global memory through the rows' register windows (18.5 against 17.7 us per launch of the README
    // series: the slower way to feed the producers, and the only one there)
    // (from T_global, like every other choice here: a block of a sharded ladder holds fewer temperatures and would fit more)
    const bool sl = L.n <= 1024 || (grid2_global <= (long)S.ncu && pt_row_lds(L.d, (int)L.T_global, L.n) <= 160 * 1024);
    bool two = grid2_global <= 2L * S.ncu && w2ok && L.n >= 32;
    if (const long ew = tune_get(TUNE_PT_ROW_WIN); ew != TUNE_UNSET) two = ew == 2 && grid2_global <= 2L * S.ncu && L.n >= 32;
    const int wpl = two ? (L.T + 1) / 2 : S.wpl;
    const size_t lds = pt_row_lds(L.d, L.T, two && sl ? L.n : 0);
    const long grid = (long)L.R * wpl;
    const int minw = grid > 2L * S.ncu ? 3 : 2;             // the 168-register build only where three workgroups share a CU
    // the windowed pipeline where the WHOLE ladder's grid is at most one workgroup per CU (from T_global: a sharded ladder's blocks
    // decide as the one-GPU run does, so both stay on one arithmetic); CARMA_TUNE_PT_ROW_WIN=0 / 1 overrides (carma_tune_set)
    const long grid_global = (long)L.R * (((long)L.T_global + 3) / 4);
    bool win = grid_global <= (long)S.ncu && (S.window_ok & SERIES_WINDOW_OK);       // (and the series suits it: carma_types.h)
    if (const long ew = tune_get(TUNE_PT_ROW_WIN); ew != TUNE_UNSET) win = ew != 0 && minw < 3;
    two = two && minw < 3;
    const void* fn = pt_row_fn<P>(minw, win, two, grid > (long)S.ncu, sl);
    g_row_pipeline.store(two ? 2 : (win && minw < 3 ? 1 : 0), std::memory_order_relaxed);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    PtLaunch La = L;
    PtRowSync Sa = S;
    // Which wave plays which part in the workgroups that share a CU (i, i + ncu, i + 2 ncu).  An ordinary launch places
    // them as k_logdens_carma_p3l's are placed (same tables); the workgroups of a COOPERATIVE launch all land with the
    // same wave -> SIMD pattern (tools/ubench/wave_placement.hip with a second argument 1,
    // profiles/r03/wave_placement_coop.txt), so the parts simply trade places: (cov, mean, P0, P1), (P0, P1, cov, mean),
    // (mean, cov, P1, P0) -- every SIMD gets one covariance or mean wave of each kind at most.
    static const long tune_rot = [] {
        const char* e = getenv("CARMA_TUNE_PT_ROW_ROT");                        // measurements only; read once
        return e ? strtol(e, nullptr, 16) : -1L;
    }();
    Sa.wpl = wpl;
    Sa.rot = wpl == 1 ? 0x36D2 : 0xB14E;
    if (tune_rot >= 0) Sa.rot = (int)tune_rot;
    static const long tune_xcd = [] {
        const char* e = getenv("CARMA_TUNE_PT_ROW_XCD_MAP");                    // measurements only; read once
        return e ? atol(e) : -1L;
    }();
    Sa.xcd_map = (wpl > 1 && L.R % 8 == 0) ? 8 : 1;
    if (tune_xcd >= 0) Sa.xcd_map = (tune_xcd > 1 && L.R % tune_xcd == 0) ? (int)tune_xcd : 1;
    Prior pra = pr;
    void* args[] = {&La, &Sa, (void*)&series, &pra, (void*)&temps, &theta, &logpost, &chol, &naccept, &nswap, &samples, &sample_lp};
    if (wpl == 1)
        // the whole ladder (block) in one workgroup: the rendezvous has a single participant, no co-residency needed --
        // an ordinary launch (a cooperative one costs ~2 ms on this stack, which matters when the ladder is sharded
        // across GPUs and every iteration is a launch of its own)
        return hipLaunchKernel(fn, dim3((unsigned)grid), dim3(256), args, lds, st);
    // COOPERATIVE launch: the swap step is a rendezvous of the ladder's workgroups through global memory, so the whole
    // grid has to be resident at once.  The runtime checks that (hipErrorCooperativeLaunchTooLarge otherwise, and the
    // host falls back to the ladder kernel) and schedules the grid as a gang, instead of this code assuming it.
    return hipLaunchCooperativeKernel(fn, dim3((unsigned)grid), dim3(256), args, (unsigned)lds, st);
}

hipError_t launch_pt_row(int p, const PtLaunch& L, const PtRowSync& S, const double4* series, const Prior& pr,
                         const double* temps, double* theta, double* logpost, double* chol, unsigned* naccept,
                         unsigned* nswap, double* samples, double* sample_lp, hipStream_t st)
{
    (void)hipGetLastError();   // HIP's last-error is sticky: drop anything left by earlier calls
    switch (p) {
        case 2: return launch_pt_row_p<2>(L, S, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        case 3: return launch_pt_row_p<3>(L, S, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        case 4: return launch_pt_row_p<4>(L, S, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        case 5: return launch_pt_row_p<5>(L, S, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        case 6: return launch_pt_row_p<6>(L, S, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        case 7: return launch_pt_row_p<7>(L, S, series, pr, temps, theta, logpost, chol, naccept, nswap, samples, sample_lp, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_pt(int p, const PtLaunch& L, const double4* series, const Prior& pr, const double* temps,
                     double* theta, double* logpost, double* chol, unsigned* naccept, unsigned* nswap, double* samples,
                     double* sample_lp, hipStream_t st)
{
    (void)hipGetLastError();   // HIP's last-error is sticky: drop anything left by earlier calls
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
