// carma_pt_core.h -- one Robust-Adaptive-Metropolis step of one tempered chain, and the
// adjacent-temperature exchange sweep.  Shared by carma_pt.hip (gfx950) and the CPU lane
// emulator (tests/emu, test harness only).
//
// Reference (file:line under /root/reference):
//   AdaptiveMetro::DoStep / Accept   src/steps.cpp:36-107   (Vihola 2012 RAM, target 0.25, gamma 2/3)
//   CholUpdateR1                     src/steps.cpp:111-131
//   StudentProposal(8,1)             src/carmcmc.cpp:139, src/random.cpp:158
//   ExchangeStep::DoStep             src/include/steps.hpp:318-362
// A chain is worked on by the same G-lane group that evaluates its log-density; vectors of
// length d (<= 16) are spread over the lanes (component j lives in lane j % G) and live in the
// chain's scratch (LDS on the GPU): th[d], thn[d], z[d], v[d], R[d*d] (upper triangular,
// row-major, Sigma = R^T R as arma::chol returns, steps.cpp:32).
#pragma once
#include "carma_core.h"
#include "carma_rng.h"

namespace carma {

struct ChainScratch {
    double* th;    // current value
    double* thn;   // proposal
    double* z;     // unit proposal
    double* v;     // scaled proposal / rank-1 vector
    double* R;     // Cholesky factor of the proposal scale matrix
};

// src/steps.cpp:111-131, columns j > k spread over the lanes.
template <int G, class GrpT>
CARMA_DEV void chol_update_r1(const GrpT& g, int d, double* R, double* v, bool downdate)
{
    const int r = g.lane();
    const double sign = downdate ? -1.0 : 1.0;
    for (int k = 0; k < d; k++) {
        const double Rkk = R[k * d + k], vk = v[k];
        const double rr = sqrt(Rkk * Rkk + sign * vk * vk);
        const double c = rr / Rkk, s = vk / Rkk;
        g.sync();                      // everybody has read R_kk, v_k
        if (r == 0) R[k * d + k] = rr;
        for (int j = k + 1 + r; j < d; j += G) {
            double Rkj = (R[k * d + j] + sign * s * v[j]) / c;
            R[k * d + j] = Rkj;
            v[j] = c * v[j] - s * Rkj;
        }
        g.sync();
    }
}

// First half of a RAM step (steps.cpp:60-73): draw the unit proposal z ~ t_8^d and form
// thn = th + R^T z.  Returns |z|^2.
template <int G, class GrpT>
CARMA_DEV double ram_propose(const GrpT& g, const ChainScratch& cs, int d, uint64_t iter, const RngKey& key)
{
    const int r = g.lane();
    for (int k = r; k < d; k += G) cs.z[k] = rng_student_t8(key, iter, (uint32_t)k);
    g.sync();
    double znorm2 = 0.0;
    for (int k = 0; k < d; k++) znorm2 += cs.z[k] * cs.z[k];
    for (int j = r; j < d; j += G) {
        double acc = 0.0;
        for (int k = 0; k <= j; k++) acc += cs.R[k * d + j] * cs.z[k];
        cs.v[j] = acc;
        cs.thn[j] = cs.th[j] + acc;
    }
    g.sync();
    return znorm2;
}

// Second half (steps.cpp:36-56, 77-99): Metropolis accept with the tempered ratio, then the RAM
// rank-1 update of the proposal factor.  ll = log-density of the proposal.
template <int G, class GrpT>
CARMA_DEV bool ram_finish(const GrpT& g, const ChainScratch& cs, int d, double temperature, uint64_t iter, int maxiter,
                          const RngKey& key, double ll, double znorm2, double* lp)
{
    const int r = g.lane();
    double alpha = (ll - *lp) / temperature;
    bool accept = false;
    const bool fin = (alpha - alpha) == 0.0;   // finite
    if (!fin) {
        alpha = 0.0;                           // steps.cpp:41-46
    } else {
        const double u = rng_uniform(key, iter, RNG_ACCEPT, 0);
        alpha = fmin(exp(alpha), 1.0);
        accept = u < alpha;
    }
    if (accept) {                              // parameter_.Save(new_value) (steps.cpp:77)
        for (int j = r; j < d; j += G) cs.th[j] = cs.thn[j];
        *lp = ll;
    }
    // adaptation of the scale matrix while niter < maxiter (steps.cpp:82-99)
    if ((long)iter < (long)maxiter) {
        const double step = fmin(1.0, (double)d / pow((double)iter, 2.0 / 3.0));   // iter = 0 -> 1
        const double fac = sqrt(step * fabs(alpha - 0.25)) / sqrt(znorm2);
        for (int j = r; j < d; j += G) cs.v[j] *= fac;
        g.sync();
        chol_update_r1<G>(g, d, cs.R, cs.v, alpha < 0.25);
    }
    g.sync();
    return accept;
}

// One RAM step (steps.cpp:60-107).  lp = stored log-posterior of the chain (updated on accept).
// Returns true when the proposal was accepted.
template <int P, int G, class GrpT>
CARMA_DEV bool ram_step(const GrpT& g, const ChainScratch& cs, int d, int q, double temperature, uint64_t iter,
                        int maxiter, const RngKey& key, const double4* __restrict__ series, int n, const Prior& pr,
                        double* lp)
{
    const double znorm2 = ram_propose<G>(g, cs, d, iter, key);
    // Accept (steps.cpp:36-56): one Kalman log-density of the proposal
    double ll;
    if constexpr (P == 1)
        ll = logdensity_car1(cs.thn, series, n, pr);
    else
        ll = logdensity_carma<P, G>(g, cs.thn, q, series, n, pr, 0);
    return ram_finish<G>(g, cs, d, temperature, iter, maxiter, key, ll, znorm2, lp);
}

// ExchangeStep sweep hot -> cold over the T chains of one replica (steps.hpp:318-362), executed by
// ONE lane.  th = T vectors of length d, `stride` doubles apart; lp = [T], temps = [T]
// (temps[i] > temps[i-1]); chain_base is the global
// slot of chain 0 of this replica (the swap uniform is keyed by the hotter chain's global slot so a
// ladder split across GPUs draws the same number on both sides).
CARMA_DEV void exchange_sweep(int T, int d, int stride, double* th, double* lp, const double* temps, RngKey key,
                              uint32_t chain_base, uint64_t iter, unsigned* nswap)
{
    for (int i = T - 1; i > 0; i--) {
        const double this_lp = lp[i], other_lp = lp[i - 1];
        double alpha = 1.0 / temps[i] * (other_lp - this_lp) + 1.0 / temps[i - 1] * (this_lp - other_lp);
        key.chain = chain_base + (uint32_t)i;
        const double u = rng_uniform(key, iter, RNG_SWAP, 0);
        alpha = fmin(exp(alpha), 1.0);
        if (!((alpha - alpha) == 0.0)) alpha = 0.0;
        if (u < alpha) {
            for (int j = 0; j < d; j++) {
                double tmp = th[i * stride + j];
                th[i * stride + j] = th[(i - 1) * stride + j];
                th[(i - 1) * stride + j] = tmp;
            }
            lp[i] = other_lp;
            lp[i - 1] = this_lp;
            nswap[i]++;
        }
    }
}

// The same sweep split for the GPU: the uniforms (as log u_i) are drawn in parallel by the chains
// before the sweep and theta is NOT moved during it -- the sweep only needs the stored
// log-posteriors, so one lane walks hot -> cold over lp[], records the resulting permutation in
// src[] (src[i] = which chain's theta ends up at temperature i) and the chains copy their new theta
// afterwards, in parallel.  u < min(exp(a), 1)  <=>  log u < a  (a NaN -> reject, as steps.hpp:336-338).
// dbeta[i] = 1/T_i - 1/T_{i-1}, so a = (lp[i-1] - lp[i]) * dbeta[i]   (steps.hpp:331-332).
// Returns the accepted swaps as a mask (bit i: temperatures i and i - 1 exchanged) -- the caller counts them.
CARMA_DEV unsigned long long exchange_decide_mask(int T, double* lp, const double* dbeta, const double* logu, int* src)
{
    unsigned long long mask = 0;
    double hot = lp[T - 1];
    int hot_src = src[T - 1];
    for (int i = T - 1; i > 0; i--) {
        const double cold = lp[i - 1];
        const int cold_src = src[i - 1];
        const double a = (cold - hot) * dbeta[i];
        if (logu[i] < a) {
            lp[i] = cold;            // temperature i now holds the colder chain's state
            src[i] = cold_src;
            mask |= 1ull << i;
            // `hot` (the state that moved down) is what temperature i-1 now holds
        } else {
            lp[i] = hot;
            src[i] = hot_src;
            hot = cold;
            hot_src = cold_src;
        }
    }
    lp[0] = hot;
    src[0] = hot_src;
    return mask;
}
CARMA_DEV void exchange_decide(int T, double* lp, const double* dbeta, const double* logu, int* src, unsigned* nswap)
{
    const unsigned long long mask = exchange_decide_mask(T, lp, dbeta, logu, src);
    for (int i = 1; i < T; i++)
        if ((mask >> i) & 1ull) nswap[i]++;
}

}  // namespace carma
