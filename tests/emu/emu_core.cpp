// emu_core.cpp -- builds carma_pack_amd/csrc/carma_core.h for the host with the lane emulator.
// TEST HARNESS ONLY (see grp_emu.h).  Entry points mirror the device launchers.
#include "grp_emu.h"
#include "../../carma_pack_amd/csrc/carma_core.h"

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

template <int P>
static void kfilter_one(const double* om_re, const double* om_im, const double* ma, double sigsqr,
                        const double4* series, int n, double* mean, double* var, double* ll, int* sing)
{
    constexpr int G = GroupOf<P>::value;
    run_group<G>([&](const Grp<G>& g) {
        Model<P> m;
        int r = g.lane() < P ? g.lane() : P - 1;
        m.w = {om_re[r], om_im[r]};
        for (int j = 0; j < P; j++) {
            m.wall[j] = {om_re[j], om_im[j]};
            m.beta[j] = ma[j];
        }
        m.sigsqr = sigsqr;
        m.mu = 0.0;
        m.scale = 1.0;
        m.valid = true;
        bool s;
        double l = filter_run<P, G, true>(g, m, series, n, mean, var, &s);
        if (g.lane() == 0) {
            *ll = l;
            *sing = s;
        }
    });
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
