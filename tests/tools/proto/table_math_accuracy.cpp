// Accuracy of the table-based exp / complex exponential of carma_math.h (host build of the very functions the kernels
// compile) against quad precision (libquadmath).  Prints the maximum error in units of 2^-53 of max(|re|, |im|).
//   g++ -O2 -std=c++17 -mfma -I carma_pack_amd/csrc tests/tools/proto/table_math_accuracy.cpp -lquadmath -o /tmp/tma && /tmp/tma
#include <quadmath.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#define CARMA_DEV static inline
#include "carma_math.h"
using namespace carma;

extern "C" int table_math_accuracy(int nsamples, unsigned seed, double* out /* [4]: exp tab, exp old, cexp tab, cexp old */)
{
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    double worst[4] = {0, 0, 0, 0};
    for (int i = 0; i < nsamples; i++) {
        // decay rates and frequencies over the decades the prior admits, time steps from 1e-3 to 1e3
        const double a = -std::exp(std::log(1e-6) + U(rng) * std::log(1e8));
        const double b = (U(rng) < 0.5 ? -1.0 : 1.0) * std::exp(std::log(1e-6) + U(rng) * std::log(1e9));
        const double dt = std::exp(std::log(1e-3) + U(rng) * std::log(1e6));
        if (!(std::fabs(b * dt) < 9.0e4) || a * dt < -700.0) continue;
        const double x = a * dt, ph = b * dt;                 // the functions see the rounded products: so does the reference
        const __float128 eq = expq((__float128)x);
        __float128 sq, cq;
        sincosq((__float128)ph, &sq, &cq);
        const double ulp = 1.0 / 9007199254740992.0;
        {
            const double e1 = exp_neg_tab(x, h_math_tab), e0 = exp_neg(x);
            const double t = (double)eq;
            worst[0] = std::fmax(worst[0], (double)fabsq((__float128)e1 - eq) / (t * ulp));
            worst[1] = std::fmax(worst[1], (double)fabsq((__float128)e0 - eq) / (t * ulp));
        }
        {
            double re, im, re0, im0;
            cexp_step_tab<false>(a, b, dt, &re, &im, h_math_tab);
            cexp_step<false>(a, b, dt, &re0, &im0);
            const __float128 rq = eq * cq, iq = eq * sq;
            const double sc = (double)eq * ulp;              // |rho| = e^x
            worst[2] = std::fmax(worst[2], std::fmax((double)fabsq((__float128)re - rq), (double)fabsq((__float128)im - iq)) / sc);
            worst[3] = std::fmax(worst[3], std::fmax((double)fabsq((__float128)re0 - rq), (double)fabsq((__float128)im0 - iq)) / sc);
        }
    }
    for (int k = 0; k < 4; k++) out[k] = worst[k];
    return 0;
}

#ifndef TABLE_MATH_NO_MAIN
int main(int argc, char** argv)
{
    double w[4];
    table_math_accuracy(argc > 1 ? atoi(argv[1]) : 2000000, 12345u, w);
    printf("max error in ulp: exp_neg_tab %.3f (exp_neg %.3f)   cexp_step_tab %.3f (cexp_step %.3f)\n", w[0], w[1], w[2], w[3]);
    return 0;
}
#endif
