// carma_predict.h -- batched KalmanFilterp::Predict / KalmanFilter1::Predict.
//
// Reference: src/kfilter.cpp:218-286 (Predict), :290-337 (InitializeCoefs / UpdateCoefs) for
// CARMA(p,q); :72-135 and :51-69 for CAR(1).  The reference re-runs the whole O(n p^2) filter for
// every requested time (carma_pack.py:793-803 calls Predict once per plot point); here every
// prediction time is one lane group of a batched launch, same row-per-lane layout and D = P - V
// recursion as the log-density kernel (carma_core.h).
//
// The groups of a wave predict at different times, so the walk over the series is written as ONE
// uniform loop over the n+1 "points" (the n data plus the prediction point inserted at position
// ip = #{t_k < time}); which vectors a transition updates is selected per group:
//   before the prediction point : the filter state x            (kfilter.cpp:231-234, 243-254)
//   leaving the prediction point: const = x - g ymean, slope = g (InitializeCoefs :290-314)
//   after it                    : const += g (y - yconst), slope -= g yslope  (UpdateCoefs :318-337)
// with g = P b^H / den, den = var of the point being left (ypredict_var for the prediction point).
#pragma once
#include "carma_core.h"

namespace carma {

// y must already be centred and yerr^2 scaled: series records are used as they are.
template <int P, int G, class GrpT>
CARMA_DEV void predict_run(const GrpT& g, const Model<P>& m, const double4* __restrict__ series, int n, double time,
                           double* pmean, double* pvar, bool* singular)
{
    FilterConsts<P> fc;
    filter_reset<P, G>(g, m, fc);
    const Cx b = fc.b_msk, c_own = fc.c_own;
    const double s0 = fc.s0;
    Cx ball[P];
#pragma unroll
    for (int j = 0; j < P; j++) ball[j] = fc.ball[j];

    // ip = number of data strictly before `time` (kfilter.cpp:221-229)
    int ip = 0;
    while (ip < n && time > series[ip].w) ip++;

    Cx D[P];
#pragma unroll
    for (int j = 0; j < P; j++) D[j] = {0.0, 0.0};
    Cx x = {0.0, 0.0}, cst = {0.0, 0.0}, slp = {0.0, 0.0};
    Cx u = c_own;
    // point 0
    double den, resid = 0.0, yslope = 0.0, ypm = 0.0, ypv = s0, yprec = 0.0, amean = 0.0;
    double t_a;
    if (ip == 0) {                       // backcast: the prediction point comes first (:238-241)
        den = s0;
        yprec = 1.0 / s0;
        t_a = time;
    } else {
        const double4 r0 = series[0];
        den = s0 + r0.z;                 // var(0)  (:180-182)
        resid = r0.y;                    // innovation (:184)
        t_a = r0.w;
    }
    for (int i = 1; i <= n; i++) {
        const bool a_pred = (i - 1 == ip), b_pred = (i == ip);
        const bool before = i - 1 < ip;                       // point a is a datum ahead of the prediction
        const int jb = (i < ip) ? i : i - 1;                  // datum index of point b (if it is a datum)
        const double4 rb = series[jb < n ? jb : n - 1];
        const double t_b = b_pred ? time : rb.w;
        const double s = 1.0 / den;
        // measurement update of the vectors at point a
        const Cx gk = {u.re * s, u.im * s};                   // gain (:191 / :292 / :320)
        if (before) {
            x = {x.re + gk.re * resid, x.im + gk.im * resid};              // :194, :245
        } else if (a_pred) {
            cst = {x.re - gk.re * ypm, x.im - gk.im * ypm};   // :294  (ymean = ypredict_mean)
            slp = gk;                                         // :295
        } else {
            cst = {cst.re + gk.re * resid, cst.im + gk.im * resid};        // :322
            slp = {slp.re - gk.re * yslope, slp.im - gk.im * yslope};      // :323
        }
        // covariance: D <- rho rho^H o (D - u u^H / den)       (:197,204 / :297-305 / :325-331)
        Cx rho;
        cexp_step(m.w.re, m.w.im, fabs(t_b - t_a), &rho.re, &rho.im);
        g.publish(u.re, u.im, rho.re, rho.im);
        Cx w = {0.0, 0.0};
#pragma unroll
        for (int j = 0; j < P; j++) {
            const double4 o = g.peek(j);
            const Cx t = cmulc(u, Cx{o.x, o.y});
            const Cx d = {fma(-t.re, s, D[j].re), fma(-t.im, s, D[j].im)};
            D[j] = cmul(cmulc(rho, Cx{o.z, o.w}), d);
            w = cadd(w, cmulc(D[j], ball[j]));
        }
        g.done_reading();
        u = cadd(w, c_own);
        x = cmul(rho, x);                                     // :201, :250
        cst = cmul(rho, cst);                                 // :301, :327
        slp = cmul(rho, slp);                                 // :302, :328
        const double Sw = g.sum(b.re * w.re - b.im * w.im);
        const double Sx = g.sum(b.re * x.re - b.im * x.im);
        const double Sc = g.sum(b.re * cst.re - b.im * cst.im);
        const double Ss = g.sum(b.re * slp.re - b.im * slp.im);
        if (b_pred) {                                         // arrival at the prediction time (:252-254)
            ypm = Sx;
            ypv = s0 + Sw;
            den = ypv;
            yprec = 1.0 / ypv;                                // :263-264
            amean = ypm * yprec;
        } else if (i < ip) {                                  // ordinary filter step (:207-213)
            den = s0 + Sw + rb.z;
            resid = rb.y - Sx;
        } else {                                              // linear-filter coefficients (:309-313, :332-336)
            const double var_b = s0 + Sw + rb.z;
            yslope = Ss;
            resid = rb.y - Sc;
            den = var_b;
            yprec += yslope * yslope / var_b;                 // :272-273, :277-278
            amean += yslope * resid / var_b;
        }
        t_a = t_b;
    }
    const bool forecast = (ip == n);                          // :257-261
    *pvar = forecast ? ypv : 1.0 / yprec;                     // :281-282
    *pmean = forecast ? ypm : amean / yprec;
    *singular = fc.sing;
}

// CAR(1): one LANE per prediction time (kfilter.cpp:72-135, 51-69).
CARMA_DEV void predict_car1(double sigsqr, double omega, const double4* __restrict__ series, int n, double time,
                            double* pmean, double* pvar)
{
    int ip = 0;
    while (ip < n && time > series[ip].w) ip++;
    const double sv = sigsqr / (2.0 * omega);
    double mean = 0.0, var = sv + series[0].z;
    for (int k = 1; k < ip; k++) {
        const double4 a = series[k - 1], bb = series[k];
        const double rho = exp(-1.0 * omega * (bb.w - a.w));
        const double previous_var = var - a.z;
        const double var_ratio = previous_var / var;
        mean = rho * mean + rho * var_ratio * (a.y - mean);
        var = sv * (1.0 - rho * rho) + rho * rho * previous_var * (1.0 - var_ratio) + bb.z;
    }
    double ypm, ypv;
    if (ip == 0) {
        ypm = 0.0;
        ypv = sv;
    } else {
        const double4 a = series[ip - 1];
        const double rho = exp(-(time - a.w) * omega);
        const double previous_var = var - a.z;
        const double var_ratio = previous_var / var;
        ypm = rho * mean + rho * var_ratio * (a.y - mean);
        ypv = sv * (1.0 - rho * rho) + rho * rho * previous_var * (1.0 - var_ratio);
    }
    if (ip == n) {
        *pmean = ypm;
        *pvar = ypv;
        return;
    }
    double yprec = 1.0 / ypv;
    ypm *= yprec;
    double4 cur = series[ip];
    double yconst = 0.0;
    double yslope = exp(-fabs(cur.w - time) * omega);
    var = sv * (1.0 - yslope * yslope) + cur.z;
    yprec += yslope * yslope / var;
    ypm += yslope * (cur.y - yconst) / var;
    for (int k = ip + 1; k < n; k++) {
        const double4 nx = series[k];
        const double rho = exp(-1.0 * (nx.w - cur.w) * omega);
        const double previous_var = var - cur.z;
        const double var_ratio = previous_var / var;
        yslope *= rho * (1.0 - var_ratio);
        yconst = yconst * rho * (1.0 - var_ratio) + rho * var_ratio * cur.y;
        var = sv * (1.0 - rho * rho) + rho * rho * previous_var * (1.0 - var_ratio) + nx.z;
        yprec += yslope * yslope / var;
        ypm += yslope * (nx.y - yconst) / var;
        cur = nx;
    }
    *pvar = 1.0 / yprec;
    *pmean = ypm * (1.0 / yprec);
}

}  // namespace carma
