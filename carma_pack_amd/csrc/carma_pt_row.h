// carma_pt_row.h -- the Robust-Adaptive-Metropolis step of carma_pt_core.h with the chain state in
// REGISTERS of one 16-lane DPP row (k_pt_row, gfx950 only).
//
// Lane j (< d <= 16) of the row owns component j of the chain: th_j, the proposal thn_j, the unit draw
// z_j, the rank-1 vector v_j and COLUMN j of the upper-triangular Cholesky factor R (Rc[k] = R_kj,
// zero for k > j).  Everything another lane needs arrives by a DPP row broadcast (v_mov_b64_dpp
// row_newbcast:k, one instruction, no LDS round trip, no group barrier):
//     thn_j = th_j + sum_{k<=j} R_kj z_k           steps.cpp:60-73    (z_k broadcast from lane k)
//     CholUpdateR1, step k: R_kk, v_k broadcast     steps.cpp:111-131
// Same draws (Philox keys), same formulas and operation order per element as ram_propose / ram_finish;
// divisions by R_kk and c are multiplications by their reciprocals (recip(), 0.5 ulp).
#pragma once
#include "carma_pt_core.h"

namespace carma {

constexpr int PT_DMAX = 16;

struct RowChain {
    double th;              // component j of the current value (lane j)
    double thn;             // proposal
    double z, v;
    double Rc[PT_DMAX];     // column j of R
};

// Unit draw z and v = R^T z of a proposal (steps.cpp:60-73); returns |z|^2.  The proposal itself is thn = th + v: the
// sampler kernel forms (z, v) of the NEXT iteration while the ladder's other workgroups arrive at the swap rendezvous --
// the factor R is final by then (the adaptation has run), only th may still change hands -- so that one addition is all
// that is left between the swap and the next filter.
__device__ __forceinline__ double ram_draw_row(const Grp<16>& g, RowChain& ch, int d, double zj)
{
    const int j = g.lane();
    ch.z = j < d ? zj : 0.0;
    double znorm2 = 0.0, acc = 0.0;
    static_for<0, PT_DMAX>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (k < d) {
            const double zk = Grp<16>::bcast_c<k>(ch.z);
            znorm2 += zk * zk;
            acc += ch.Rc[k] * zk;                       // R_kj is zero for k > j
        }
    });
    ch.v = acc;
    return znorm2;
}

// 1 / sqrt(x) to double precision: v_rsq_f64 (2^-24) + two Newton steps.  (The rank-1 update below needs sqrt(x) AND 1 / sqrt(x);
// the library sqrt is the same sequence plus a scaling for denormal inputs, which a Cholesky diagonal never is.)
__device__ __forceinline__ double rsqrt_nr(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-x * y, y, 1.0);
    return fma(0.5 * y, e, y);
}

// CholUpdateR1 (steps.cpp:111-131) on the register-resident factor.  Step k turns on (R_kk, v_k): with rs = 1 / sqrt(R_kk^2 +-
// v_k^2) the reference's c = rr / R_kk, s = v_k / R_kk and the division by c become rr = x rs, c = rr / R_kk, s = v_k / R_kk,
// 1 / c = R_kk rs.  The diagonal entries are not touched before their own step, so every lane forms 1 / R_jj ONCE in front of
// the loop; what is left on the dependent chain v_k -> rs -> row update -> v_{k+1} is one reciprocal square root instead of a
// square root and two reciprocals (round 3: the adaptation is no longer hidden behind the swap rendezvous -- the tagged
// staging made that shorter than this loop).
__device__ __forceinline__ void chol_update_row(const Grp<16>& g, int d, RowChain& ch, bool downdate)
{
    const int j = g.lane();
    const double sign = downdate ? -1.0 : 1.0;
    double diag = 1.0;
    static_for<0, PT_DMAX>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (j == k) diag = ch.Rc[k];
    });
    const double idiag = recip(diag);                       // 1 / R_jj of this lane's own diagonal entry
    static_for<0, PT_DMAX>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (k < d) {
            const double Rkk = Grp<16>::bcast_c<k>(ch.Rc[k]);
            const double iR = Grp<16>::bcast_c<k>(idiag);
            const double vk = Grp<16>::bcast_c<k>(ch.v);
            const double x = fma(sign * vk, vk, Rkk * Rkk);
            const double rs = rsqrt_nr(x);
            const double rr = x * rs;
            const double c = rr * iR, s = vk * iR, ic = Rkk * rs;
            const double Rkj = (ch.Rc[k] + sign * s * ch.v) * ic;
            const double vj = c * ch.v - s * Rkj;
            ch.Rc[k] = (j == k) ? rr : (j > k ? Rkj : ch.Rc[k]);
            ch.v = (j > k) ? vj : ch.v;
        }
    });
}

// Metropolis accept with the tempered ratio (steps.cpp:36-56); *alpha is the acceptance probability the adaptation uses.
// `upre` (may be null): the acceptance uniform drawn ahead of time by a producer wave (same key, iteration, purpose)
__device__ __forceinline__ bool ram_accept_row(RowChain& ch, double temperature, uint64_t iter, const RngKey& key, double ll,
                                               double* lp, double* alpha_out, const double* upre = nullptr)
{
    double alpha = (ll - *lp) / temperature;
    bool accept = false;
    const bool fin = (alpha - alpha) == 0.0;
    if (!fin) {
        alpha = 0.0;                                    // steps.cpp:41-46
    } else {
        const double u = upre ? *upre : rng_uniform(key, iter, RNG_ACCEPT, 0);
        alpha = fmin(exp(alpha), 1.0);
        accept = u < alpha;
    }
    if (accept) {                                       // parameter_.Save(new_value) (steps.cpp:77)
        ch.th = ch.thn;
        *lp = ll;
    }
    *alpha_out = alpha;
    return accept;
}

// The RAM rank-1 update of the proposal factor (steps.cpp:82-99); independent of the state, so the sampler kernel runs
// it while the other workgroups of the ladder arrive at the swap rendezvous.
// step length of the adaptation at `iter` (a function of the iteration alone: a producer wave evaluates it during the
// filter, so that the cube root is off the chain wave's path)
__device__ __forceinline__ double ram_adapt_step(int d, uint64_t iter)
{
    const double cb = cbrt((double)iter);                   // iter^(2/3) without pow (iter = 0 -> step 1)
    return fmin(1.0, (double)d / (cb * cb));
}

__device__ __forceinline__ void ram_adapt_row(const Grp<16>& g, RowChain& ch, int d, uint64_t iter, int maxiter, double alpha,
                                              double znorm2, double step)
{
    if ((long)iter < (long)maxiter) {
        const double fac = sqrt(step * fabs(alpha - 0.25) * recip(znorm2));
        ch.v *= fac;
        chol_update_row(g, d, ch, alpha < 0.25);
    }
}

// ExchangeStep sweep hot -> cold (steps.hpp:318-362, same decisions as exchange_decide) executed by a
// whole wave on values held in lanes: lane i (< T <= 64) owns temperature i's log-posterior, its
// 1/T_i - 1/T_{i-1} and the log of its swap uniform.
// The serial part is the chain that travels down the ladder: the state at temperature i after pair (i + 1, i) -- `hot`,
// wave-uniform.  Per step EVERY lane forms its own pair's acceptance against that `hot` (lane i's is the one that
// counts: one bit of the ballot), and the only things carried to the next step are `hot` and the bit: v_add, v_mul,
// v_cmp and a handful of scalar instructions, ~40 cycles of dependent latency (round 2 read six lanes and wrote three
// predicated results per step: 4.3k cycles for sixteen temperatures, on the path every ladder waits for).
// What each temperature ends up with follows from the swap bits alone and is worked out by all lanes at once:
//   pair (i, i - 1) swapped  -> temperature i takes the state that was at i - 1;
//   otherwise                -> it takes the travelling chain, which started at h = i + (number of consecutive swapped pairs
//                               right above i) -- the chain that kept moving down through every one of them.
// On return lane i holds the log-posterior now sitting at temperature i and the index of the chain its state comes
// from; *swapped is set for the lanes whose pair (i, i-1) swapped.
__device__ __forceinline__ void exchange_decide_wave(int T, int lane, double& lp, double dbeta, double logu, int& src,
                                                     bool* swapped)
{
    auto rl = [](double v, int i) {
        int lo = __builtin_amdgcn_readlane(__double2loint(v), i);
        int hi = __builtin_amdgcn_readlane(__double2hiint(v), i);
        return __hiloint2double(hi, lo);
    };
    const double lp_in = lp;
    const double cold_l = __shfl_up(lp_in, 1, 64);          // lane i: the log-posterior at temperature i - 1
    double hot = rl(lp_in, T - 1);
    unsigned long long mask = 0ull;
    for (int i = T - 1; i > 0; i--) {
        const double a = (cold_l - hot) * dbeta;            // ExchangeStep's exponent, lane i's is pair (i, i - 1)
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(logu < a);
        const bool swap = ((bal >> i) & 1ull) != 0ull;      // uniform
        mask |= (unsigned long long)swap << i;
        const double cold_i = rl(lp_in, i - 1);
        hot = swap ? hot : cold_i;
    }
    const bool sw = ((mask >> lane) & 1ull) != 0ull;          // (bit 0 and the bits from T up are never set)
    const unsigned long long above = ~((mask >> lane) >> 1);  // bit k: pair (lane + 1 + k, lane + k) did NOT swap
    const int h = lane + __builtin_ctzll(above);
    const int from = sw ? lane - 1 : h;
    if (lane < T) {
        src = from;
        lp = __shfl(lp_in, from, 64);
    } else {
        (void)__shfl(lp_in, lane, 64);
    }
    *swapped = sw;
}

}  // namespace carma
