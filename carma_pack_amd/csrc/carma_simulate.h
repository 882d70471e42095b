// carma_simulate.h -- batched exact simulation of CARMA(p,q) / CAR(1) paths.
//
// Reference: carma_process / car1_process (src/carmcmc/carma_pack.py:1126-1259): the path is drawn value by value from
// its one-step predictive distribution, y_i ~ N(mean_i, var_i), with the Kalman recursion of kfilter.cpp:138-215 run
// WITHOUT measurement error on the values drawn so far.  Here every path is one lane group of a batched launch (same
// row-per-lane layout and D = P - V recursion as carma_predict.h), the normal variates come from the counter-based
// generator (carma_rng.h) keyed by (seed, path, step), so a path is reproducible and independent of the batch it is in.
#pragma once
#include "carma_core.h"
#include "carma_rng.h"

namespace carma {

// times = n sorted time stamps; out = n values of this group's path (written by lane 0)
template <int P, int G, class GrpT>
CARMA_DEV void simulate_run(const GrpT& g, const Model<P>& m, const double* __restrict__ times, int n, const RngKey& key,
                            double* __restrict__ out, bool* singular)
{
    FilterConsts<P> fc;
    filter_reset<P, G>(g, m, fc);
    const Cx b = fc.b_msk, c_own = fc.c_own;
    const double s0 = fc.s0;
    Cx ball[P];
#pragma unroll
    for (int j = 0; j < P; j++) ball[j] = fc.ball[j];
    Cx D[P];
#pragma unroll
    for (int j = 0; j < P; j++) D[j] = {0.0, 0.0};
    Cx x = {0.0, 0.0};
    Cx u = c_own;
    double var = s0, mean = 0.0;                              // kalman_var = Re(b V b^H), kalman_mean = 0 (:1226-1227)
    for (int i = 0; i < n; i++) {
        const double z = rng_normal(key, (uint64_t)i, 0);
        const double sd = sqrt(fmax(var, 0.0));               // two coincident times leave var = 0 up to rounding
        const double innov = sd * z;                          // y_i - kalman_mean (:1233, :1255-1257)
        if (g.lane() == 0) out[i] = mean + innov;
        if (i + 1 == n) break;
        const double s = var > 0.0 ? 1.0 / var : 0.0;
        x = {x.re + u.re * s * innov, x.im + u.im * s * innov};                    // :1237-1239
        Cx rho;
        cexp_step(m.w.re, m.w.im, times[i + 1] - times[i], &rho.re, &rho.im);     // :1244-1246
        g.publish(u.re, u.im, rho.re, rho.im);
        Cx w = {0.0, 0.0};
#pragma unroll
        for (int j = 0; j < P; j++) {
            const double4 o = g.peek(j);
            const Cx t = cmulc(u, Cx{o.x, o.y});
            const Cx d = {fma(-t.re, s, D[j].re), fma(-t.im, s, D[j].im)};         // :1241
            D[j] = cmul(cmulc(rho, Cx{o.z, o.w}), d);                              // :1248 (minus V on both sides)
            w = cadd(w, cmulc(D[j], ball[j]));
        }
        g.done_reading();
        u = cadd(w, c_own);
        x = cmul(rho, x);
        var = s0 + g.sum(b.re * w.re - b.im * w.im);                               // :1251
        mean = g.sum(b.re * x.re - b.im * x.im);                                   // :1250
    }
    *singular = fc.sing;
}

// CAR(1): exact Ornstein-Uhlenbeck draw (car1_process, carma_pack.py:1126-1146), one LANE per path.
// sigsqr = driving-noise variance, omega = 1 / tau: stationary variance sigsqr / (2 omega).
CARMA_DEV void simulate_car1(double sigsqr, double omega, const double* __restrict__ times, int n, const RngKey& key,
                             double* __restrict__ out)
{
    const double sv = sigsqr / (2.0 * omega);
    double yv = sqrt(sv) * rng_normal(key, 0, 0);
    out[0] = yv;
    for (int i = 1; i < n; i++) {
        const double rho = exp(-(times[i] - times[i - 1]) * omega);
        yv = rho * yv + sqrt(fmax(sv * (1.0 - rho * rho), 0.0)) * rng_normal(key, (uint64_t)i, 0);
        out[i] = yv;
    }
}

}  // namespace carma
