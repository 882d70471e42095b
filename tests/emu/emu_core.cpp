// emu_core.cpp -- builds carma_pack_amd/csrc/carma_core.h for the host with the lane emulator.
// TEST HARNESS ONLY (see grp_emu.h).  Entry points mirror the device launchers.
#include <algorithm>
#include <array>
#include <vector>
#include "grp_emu.h"
#include "../../carma_pack_amd/csrc/carma_core.h"
#include "../../carma_pack_amd/csrc/carma_lane.h"

using namespace carma;

template <int P>
struct GroupOf {
    static constexpr int value = P <= 2 ? 2 : (P <= 4 ? 4 : 8);
};

template <int P>
static void logdens_one(const double* theta, int q, const double4* series, int n, const Prior& pr, int ignore_prior,
                        double* out)
{
    constexpr int G = GroupOf<P>::value;
    run_group<G>([&](const Grp<G>& g) {
        double ll = logdensity_carma<P, G>(g, theta, q, series, n, pr, ignore_prior);
        if (g.lane() == 0) *out = ll;
    });
}

// one evaluation per 16-lane group: the set-up path of the wave pipeline (all roots in one pass) with the generic loop
template <int P>
static void logdens_row_one(const double* theta, int q, const double4* series, int n, const Prior& pr, int ignore_prior,
                            double* out)
{
    run_group<16>([&](const Grp<16>& g) {
        double ll = logdensity_carma<P, 16>(g, theta, q, series, n, pr, ignore_prior);
        if (g.lane() == 0) *out = ll;
    });
}

template <int P>
static void kfilter_one(const double* om_re, const double* om_im, const double* ma, double sigsqr,
                        const double4* series, int n, double* mean, double* var, double* ll, int* sing)
{
    constexpr int G = GroupOf<P>::value;
    run_group<G>([&](const Grp<G>& g) {
        Model<P> m;
        double om[2 * P];
        for (int j = 0; j < P; j++) {
            om[2 * j] = om_re[j];
            om[2 * j + 1] = om_im[j];
        }
        model_from_roots<P, G>(g, om, ma, sigsqr, m);
        bool s;
        double l = filter_run<P, G, true>(g, m, series, n, mean, var, &s);
        if (g.lane() == 0) {
            *ll = l;
            *sing = s;
        }
    });
}

extern "C" int emu_logdensity_carma_lane(int p, int q, const double* theta, int B, const double* series, int n, const double* prior,
                                         int ignore_prior, double* out)
{
    // one evaluation per lane (carma_lane.h): plain scalar code, run once per parameter vector
    Prior pr{prior[0], prior[1], prior[2], prior[3]};
    const double4* s4 = reinterpret_cast<const double4*>(series);
    const int d = 3 + p + q;
    for (int b = 0; b < B; b++) {
        const double* th = theta + (size_t)b * d;
        switch (p) {
            case 2: out[b] = logdensity_lane<2>(th, q, s4, n, pr, ignore_prior, h_math_tab); break;
            case 3: out[b] = logdensity_lane<3>(th, q, s4, n, pr, ignore_prior, h_math_tab); break;
            case 4: out[b] = logdensity_lane<4>(th, q, s4, n, pr, ignore_prior, h_math_tab); break;
            case 5: out[b] = logdensity_lane<5>(th, q, s4, n, pr, ignore_prior, h_math_tab); break;
            case 6: out[b] = logdensity_lane<6>(th, q, s4, n, pr, ignore_prior, h_math_tab); break;
            case 7: out[b] = logdensity_lane<7>(th, q, s4, n, pr, ignore_prior, h_math_tab); break;
            default: return -1;
        }
    }
    return 0;
}

extern "C" {

int emu_logdensity_carma(int p, int q, const double* theta, int B, const double* series, int n, const double* prior,
                         int ignore_prior, double* out)
{
    Prior pr{prior[0], prior[1], prior[2], prior[3]};
    const double4* s4 = reinterpret_cast<const double4*>(series);
    int d = 3 + p + q;
    for (int b = 0; b < B; b++) {
        const double* th = theta + (size_t)b * d;
        switch (p) {
            case 2: logdens_one<2>(th, q, s4, n, pr, ignore_prior, out + b); break;
            case 3: logdens_one<3>(th, q, s4, n, pr, ignore_prior, out + b); break;
            case 4: logdens_one<4>(th, q, s4, n, pr, ignore_prior, out + b); break;
            case 5: logdens_one<5>(th, q, s4, n, pr, ignore_prior, out + b); break;
            case 6: logdens_one<6>(th, q, s4, n, pr, ignore_prior, out + b); break;
            case 7: logdens_one<7>(th, q, s4, n, pr, ignore_prior, out + b); break;
            default: return -1;
        }
    }
    return 0;
}

int emu_logdensity_carma_row(int p, int q, const double* theta, int B, const double* series, int n, const double* prior,
                             int ignore_prior, double* out)
{
    Prior pr{prior[0], prior[1], prior[2], prior[3]};
    const double4* s4 = reinterpret_cast<const double4*>(series);
    int d = 3 + p + q;
    for (int b = 0; b < B; b++) {
        const double* th = theta + (size_t)b * d;
        switch (p) {
            case 2: logdens_row_one<2>(th, q, s4, n, pr, ignore_prior, out + b); break;
            case 3: logdens_row_one<3>(th, q, s4, n, pr, ignore_prior, out + b); break;
            case 4: logdens_row_one<4>(th, q, s4, n, pr, ignore_prior, out + b); break;
            case 5: logdens_row_one<5>(th, q, s4, n, pr, ignore_prior, out + b); break;
            case 6: logdens_row_one<6>(th, q, s4, n, pr, ignore_prior, out + b); break;
            case 7: logdens_row_one<7>(th, q, s4, n, pr, ignore_prior, out + b); break;
            default: return -1;
        }
    }
    return 0;
}

int emu_kfilter_carma(int p, const double* om_re, const double* om_im, const double* ma, double sigsqr,
                      const double* series, int n, double* mean, double* var, double* ll)
{
    const double4* s4 = reinterpret_cast<const double4*>(series);
    int sing = 0;
    switch (p) {
        case 2: kfilter_one<2>(om_re, om_im, ma, sigsqr, s4, n, mean, var, ll, &sing); break;
        case 3: kfilter_one<3>(om_re, om_im, ma, sigsqr, s4, n, mean, var, ll, &sing); break;
        case 4: kfilter_one<4>(om_re, om_im, ma, sigsqr, s4, n, mean, var, ll, &sing); break;
        case 5: kfilter_one<5>(om_re, om_im, ma, sigsqr, s4, n, mean, var, ll, &sing); break;
        case 6: kfilter_one<6>(om_re, om_im, ma, sigsqr, s4, n, mean, var, ll, &sing); break;
        case 7: kfilter_one<7>(om_re, om_im, ma, sigsqr, s4, n, mean, var, ll, &sing); break;
        default: return -1;
    }
    return sing ? 1 : 0;
}

int emu_logdensity_car1(const double* theta, int B, const double* series, int n, const double* prior, double* out)
{
    Prior pr{prior[0], prior[1], prior[2], prior[3]};
    const double4* s4 = reinterpret_cast<const double4*>(series);
    for (int b = 0; b < B; b++) out[b] = logdensity_car1(theta + 4 * (size_t)b, s4, n, pr);
    return 0;
}
}

// ---------------------------------------------------------------------------------------------
// Sampler emulation: T chains of one replica, each on its own G-thread team, global barrier around
// the exchange sweep -- the same structure as k_pt in carma_pt.hip.
#include "../../carma_pack_amd/csrc/carma_pt_core.h"

template <int P>
struct PtGroupOf {
    static constexpr int value = P <= 1 ? 4 : (P <= 2 ? 2 : (P <= 4 ? 4 : 8));
};

template <int P>
static void pt_run(int q, const double4* series, int n, const Prior& pr, int T, const double* temps, int maxiter,
                   long niter, int save_from, int thin, uint32_t seed0, uint32_t seed1, double* theta, double* lp,
                   double* chol, int d, double* samples, double* sample_lp, unsigned* nacc, unsigned* nswap)
{
    constexpr int G = PtGroupOf<P>::value;
    const int per = 4 * d + d * d;
    std::vector<double> base((size_t)T * per, 0.0);
    for (int c = 0; c < T; c++) {
        for (int j = 0; j < d; j++) base[(size_t)c * per + j] = theta[c * d + j];
        for (int j = 0; j < d * d; j++) base[(size_t)c * per + 4 * d + j] = chol[c * d * d + j];
    }
    std::vector<EmuShared> sh(T);
    for (auto& s : sh) s.bar.n = G;
    SpinBarrier all;
    all.n = T * G;
    std::vector<std::thread> th;
    for (int c = 0; c < T; c++) {
        for (int r = 0; r < G; r++) {
            th.emplace_back([&, c, r]() {
                Grp<G> g{&sh[c], r};
                ChainScratch cs;
                cs.th = &base[(size_t)c * per];
                cs.thn = cs.th + d;
                cs.z = cs.th + 2 * d;
                cs.v = cs.th + 3 * d;
                cs.R = cs.th + 4 * d;
                RngKey key{seed0, seed1, (uint32_t)c};
                double mylp = lp[c];
                for (long it = 0; it < niter; it++) {
                    bool acc = ram_step<P, G>(g, cs, d, q, temps[c], (uint64_t)it, maxiter, key, series, n, pr, &mylp);
                    if (r == 0) {
                        lp[c] = mylp;
                        if (acc) nacc[c]++;
                    }
                    all.wait();
                    if (c == 0 && r == 0) {
                        if (T > 1) exchange_sweep(T, d, per, base.data(), lp, temps, key, 0u, (uint64_t)it, nswap);
                        if (it >= save_from && ((it - save_from + 1) % thin) == 0) {
                            long s = (it - save_from + 1) / thin - 1;
                            for (int j = 0; j < d; j++) samples[s * d + j] = base[j];
                            sample_lp[s] = lp[0];
                        }
                    }
                    all.wait();
                    mylp = lp[c];
                }
            });
        }
    }
    for (auto& t : th) t.join();
    for (int c = 0; c < T; c++) {
        for (int j = 0; j < d; j++) theta[c * d + j] = base[(size_t)c * per + j];
        for (int j = 0; j < d * d; j++) chol[c * d * d + j] = base[(size_t)c * per + 4 * d + j];
    }
}

extern "C" int emu_pt_run(int p, int q, const double* series, int n, const double* prior, int T, const double* temps,
                          int maxiter, long niter, int save_from, int thin, unsigned seed0, unsigned seed1, double* theta,
                          double* lp, double* chol, double* samples, double* sample_lp, unsigned* nacc, unsigned* nswap)
{
    Prior pr{prior[0], prior[1], prior[2], prior[3]};
    const double4* s4 = reinterpret_cast<const double4*>(series);
    const int d = p == 1 ? 4 : 3 + p + q;
    switch (p) {
        case 1: pt_run<1>(q, s4, n, pr, T, temps, maxiter, niter, save_from, thin, seed0, seed1, theta, lp, chol, d, samples, sample_lp, nacc, nswap); break;
        case 2: pt_run<2>(q, s4, n, pr, T, temps, maxiter, niter, save_from, thin, seed0, seed1, theta, lp, chol, d, samples, sample_lp, nacc, nswap); break;
        case 3: pt_run<3>(q, s4, n, pr, T, temps, maxiter, niter, save_from, thin, seed0, seed1, theta, lp, chol, d, samples, sample_lp, nacc, nswap); break;
        default: return -1;
    }
    return 0;
}

extern "C" void emu_rng_draws(unsigned seed0, unsigned seed1, unsigned chain, long n, double* t8, double* u)
{
    RngKey key{seed0, seed1, chain};
    for (long i = 0; i < n; i++) {
        t8[i] = rng_student_t8(key, (uint64_t)i, 0);
        u[i] = rng_uniform(key, (uint64_t)i, RNG_ACCEPT, 0);
    }
}

// ---------------------------------------------------------------------------------------------
// Predict (carma_predict.h)
#include "../../carma_pack_amd/csrc/carma_predict.h"

template <int P>
static void predict_one(const double* om_re, const double* om_im, const double* ma, double sigsqr, const double4* series,
                        int n, double tp, double* pm, double* pv)
{
    constexpr int G = GroupOf<P>::value;
    run_group<G>([&](const Grp<G>& g) {
        Model<P> m;
        double om[2 * P];
        for (int j = 0; j < P; j++) {
            om[2 * j] = om_re[j];
            om[2 * j + 1] = om_im[j];
        }
        model_from_roots<P, G>(g, om, ma, sigsqr, m);
        double a, b;
        bool s;
        predict_run<P, G>(g, m, series, n, tp, &a, &b, &s);
        if (g.lane() == 0) {
            *pm = a;
            *pv = b;
        }
    });
}

extern "C" int emu_predict_carma(int p, const double* om_re, const double* om_im, const double* ma, double sigsqr,
                                 const double* series, int n, const double* tpred, int M, double* pmean, double* pvar)
{
    const double4* s4 = reinterpret_cast<const double4*>(series);
    for (int i = 0; i < M; i++) {
        switch (p) {
            case 2: predict_one<2>(om_re, om_im, ma, sigsqr, s4, n, tpred[i], pmean + i, pvar + i); break;
            case 3: predict_one<3>(om_re, om_im, ma, sigsqr, s4, n, tpred[i], pmean + i, pvar + i); break;
            case 4: predict_one<4>(om_re, om_im, ma, sigsqr, s4, n, tpred[i], pmean + i, pvar + i); break;
            case 5: predict_one<5>(om_re, om_im, ma, sigsqr, s4, n, tpred[i], pmean + i, pvar + i); break;
            case 6: predict_one<6>(om_re, om_im, ma, sigsqr, s4, n, tpred[i], pmean + i, pvar + i); break;
            case 7: predict_one<7>(om_re, om_im, ma, sigsqr, s4, n, tpred[i], pmean + i, pvar + i); break;
            default: return -1;
        }
    }
    return 0;
}

extern "C" void emu_predict_car1(double sigsqr, double omega, const double* series, int n, const double* tpred, int M,
                                 double* pmean, double* pvar)
{
    const double4* s4 = reinterpret_cast<const double4*>(series);
    for (int i = 0; i < M; i++) predict_car1(sigsqr, omega, s4, n, tpred[i], pmean + i, pvar + i);
}


