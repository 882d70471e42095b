// carma_math.h -- lean FP64 exp / sincos for the transition factor rho = exp(omega dt)
// (src/kfilter.cpp:200 calls std::exp(std::complex<double>) once per root per step).
//
// gfx950 has no FP64 transcendental instructions; the ROCm device-library exp() + sincos() cost
// ~120 VALU instructions per call pair (40 % of the whole Kalman step).  The versions here do the
// same argument reductions with FMA and evaluate fixed polynomials: ~20 (exp) + ~30 (sincos)
// instructions, max error < 2 ulp on the ranges the filter produces:
//   exp_neg(x)  : any x (meant for x = Re(omega) dt <= 0); Cody-Waite ln2 split + degree-13 Taylor
//   sincos_cw(x): |x| < 2^20 by 3-term Cody-Waite reduction by pi/2 + fdlibm kernels; larger
//                 arguments (pathological gap/min-dt ratios) take the library path.
// Shared with the CPU lane emulator (tests/emu), hence plain C++ with fma().
#pragma once
#include "carma_math_tab.h"

namespace carma {

// Three-operand FP64 FMA.  hipcc selects v_fmac_f64 (dst tied to the addend) for fma(), which
// costs an extra v_mov_b64 whenever the addend is a loop-invariant polynomial coefficient; the asm
// form keeps one instruction per Horner step.
#ifdef __HIPCC__
CARMA_DEV double fma3(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#else
static inline double fma3(double a, double b, double c) { return fma(a, b, c); }
#endif

CARMA_DEV double exp_neg(double x)
{
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;   // 33 significant bits
    const double LN2_LO = 1.90821492927058770002e-10;
    double n = rint(x * LOG2E);
    double r = fma3(-n, LN2_HI, x);
    r = fma3(-n, LN2_LO, r);
    // exp(r), |r| <= ln2/2: Taylor to r^13 (truncation 4e-18)
    double p = 1.0 / 6227020800.0;
    p = fma3(p, r, 1.0 / 479001600.0);
    p = fma3(p, r, 1.0 / 39916800.0);
    p = fma3(p, r, 1.0 / 3628800.0);
    p = fma3(p, r, 1.0 / 362880.0);
    p = fma3(p, r, 1.0 / 40320.0);
    p = fma3(p, r, 1.0 / 5040.0);
    p = fma3(p, r, 1.0 / 720.0);
    p = fma3(p, r, 1.0 / 120.0);
    p = fma3(p, r, 1.0 / 24.0);
    p = fma3(p, r, 1.0 / 6.0);
    p = fma3(p, r, 0.5);
    p = fma3(p, r, 1.0);
    p = fma3(p, r, 1.0);
    // n is integral; beyond +-2000 the result has long over/underflowed, clamp so the int
    // conversion is well defined
    double nc = fmin(fmax(n, -2200.0), 2200.0);
    return ldexp(p, (int)nc);
}

// sin and cos of x, |x| < 2^20 (caller guarantees); fdlibm __kernel_sin/__kernel_cos polynomials.
CARMA_DEV void sincos_cw(double x, double* s_out, double* c_out)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;    // first 33 bits of pi/2
    const double PIO2_2 = 6.07710050630396597660e-11;    // next 33 bits
    const double PIO2_3 = 2.02226624871116645580e-21;    // next 33 bits
    const double PIO2_3T = 8.47842766036889956997e-32;   // tail
    double n = rint(x * TWO_OVER_PI);
    double r = fma3(-n, PIO2_1, x);
    r = fma3(-n, PIO2_2, r);
    r = fma3(-n, PIO2_3, r);
    r = fma3(-n, PIO2_3T, r);
    const int q = (int)n;
    const double z = r * r;
    // sin(r) = r + r^3 (S1 + z (S2 + ... ))
    double ps = 1.58969099521155010221e-10;
    ps = fma3(ps, z, -2.50507602534068634195e-08);
    ps = fma3(ps, z, 2.75573137070700676789e-06);
    ps = fma3(ps, z, -1.98412698298579493134e-04);
    ps = fma3(ps, z, 8.33333333332248946124e-03);
    ps = fma3(ps, z, -1.66666666666666324348e-01);
    const double sn = fma3(z * r, ps, r);
    // cos(r) = 1 - z/2 + z^2 (C1 + z (C2 + ...))
    double pc = -1.13596475577881948265e-11;
    pc = fma3(pc, z, 2.08757232129817482790e-09);
    pc = fma3(pc, z, -2.75573143513906633035e-07);
    pc = fma3(pc, z, 2.48015872894767294178e-05);
    pc = fma3(pc, z, -1.38888888888741095749e-03);
    pc = fma3(pc, z, 4.16666666666666019037e-02);
    pc = fma3(pc, z, -0.5);
    const double cs = fma3(pc, z, 1.0);
    // quadrant
    const bool swap = q & 1;
    double so = swap ? cs : sn;
    double co = swap ? sn : cs;
    so = (q & 2) ? -so : so;
    co = ((q + 1) & 2) ? -co : co;
    *s_out = so;
    *c_out = co;
}

// library sincos kept out of line: it is only reached for |phase| >= 2^20 (or NaN)
struct SinCos {
    double s, c;
};
#ifdef __HIPCC__
__device__ __attribute__((noinline)) static SinCos sincos_slow(double x)
#else
static inline SinCos sincos_slow(double x)
#endif
{
    SinCos r;
    r.s = sin(x);
    r.c = cos(x);
    return r;
}

// rho = exp((a + i b) dt) -> (re, im).  The four polynomial chains (even/odd halves of the exp
// polynomial, sin, cos) are advanced in lock-step so that a single in-order wave always has
// independent FMAs to issue between the members of each dependent chain.
// EXACT: the rounding of the products a dt and b dt (relative 2^-53: 3e-14 rad at a phase of 256 rad, 1e-11 at 1e5 rad) is
// recovered with an FMA and added back after the argument reduction, so the accuracy no longer degrades with |b dt|.
// dt_lo: the part of the time difference its double does not hold (two-difference of the two times), EXACT only.
template <bool EXACT, bool CHECK>
CARMA_DEV void cexp_step_impl(double a, double b, double dt, double* re, double* im, double dt_lo)
{
    const double x = a * dt;
    const double ph = b * dt;
    if (CHECK && !(fabs(ph) < 1048576.0)) {
        // rare: library reduction for huge phases (NaN also lands here)
        const double e = exp_neg(x);
        SinCos sc = sincos_slow(ph);
        if constexpr (EXACT) {
            const double pl = fma3(b, dt_lo, fma3(b, dt, -ph));
            const double c0 = sc.c, s0 = sc.s;
            sc.c = fma3(-s0, pl, c0);
            sc.s = fma3(c0, pl, s0);
        }
        *re = e * sc.c;
        *im = e * sc.s;
        return;
    }
    // --- argument reductions
    const double n1 = rint(x * 1.4426950408889634074);
    const double n2 = rint(ph * 6.36619772367581382433e-01);
    double r = fma3(-n1, 6.93147180369123816490e-01, x);
    double t = fma3(-n2, 1.57079632673412561417e+00, ph);
    r = fma3(-n1, 1.90821492927058770002e-10, r);
    t = fma3(-n2, 6.07710050630396597660e-11, t);
    t = fma3(-n2, 2.02226624871116645580e-21, t);
    t = fma3(-n2, 8.47842766036889956997e-32, t);
    if constexpr (EXACT) {
        r += fma3(a, dt_lo, fma3(a, dt, -x));
        t += fma3(b, dt_lo, fma3(b, dt, -ph));
    }
    const double r2 = r * r;
    const double z = t * t;
    // --- exp(r) = E(r^2) + r O(r^2) (Taylor to r^13), sin(t) = t + t^3 S(z), cos(t) = 1 + z C(z)
    double pe = 1.0 / 479001600.0;          // r^12
    double po = 1.0 / 6227020800.0;         // r^13
    double ps = 1.58969099521155010221e-10;
    double pc = -1.13596475577881948265e-11;
    pe = fma3(pe, r2, 1.0 / 3628800.0);
    po = fma3(po, r2, 1.0 / 39916800.0);
    ps = fma3(ps, z, -2.50507602534068634195e-08);
    pc = fma3(pc, z, 2.08757232129817482790e-09);
    pe = fma3(pe, r2, 1.0 / 40320.0);
    po = fma3(po, r2, 1.0 / 362880.0);
    ps = fma3(ps, z, 2.75573137070700676789e-06);
    pc = fma3(pc, z, -2.75573143513906633035e-07);
    pe = fma3(pe, r2, 1.0 / 720.0);
    po = fma3(po, r2, 1.0 / 5040.0);
    ps = fma3(ps, z, -1.98412698298579493134e-04);
    pc = fma3(pc, z, 2.48015872894767294178e-05);
    pe = fma3(pe, r2, 1.0 / 24.0);
    po = fma3(po, r2, 1.0 / 120.0);
    ps = fma3(ps, z, 8.33333333332248946124e-03);
    pc = fma3(pc, z, -1.38888888888741095749e-03);
    pe = fma3(pe, r2, 0.5);
    po = fma3(po, r2, 1.0 / 6.0);
    ps = fma3(ps, z, -1.66666666666666324348e-01);
    pc = fma3(pc, z, 4.16666666666666019037e-02);
    pe = fma3(pe, r2, 1.0);
    po = fma3(po, r2, 1.0);
    const double tz = t * z;
    pc = fma3(pc, z, -0.5);
    const double ep = fma3(po, r, pe);
    const double sn = fma3(tz, ps, t);
    const double cs = fma3(pc, z, 1.0);
    const double nc = fmin(fmax(n1, -2200.0), 2200.0);
    const double e = ldexp(ep, (int)nc);
    // --- quadrant
    const int q = (int)n2;
    const bool swap = q & 1;
    double so = swap ? cs : sn;
    double co = swap ? sn : cs;
    so = (q & 2) ? -so : so;
    co = ((q + 1) & 2) ? -co : co;
    *re = e * co;
    *im = e * so;
}

// ---------------------------------------------------------------------------------------------------------------------
// TABLE-BASED forms (round 4; the one-evaluation-per-lane kernels, where exp / sincos were 156 of the 338 instructions of a
// wave-step): arguments reduced by ln 2 / 32 and pi / 32 instead of ln 2 and pi / 2, the coarse part from a table of 32 + 64
// correctly rounded entries (carma_math_tab.h: 2^(j/32); sin, cos of k pi / 32 -- in LDS on the device, math_tab_fill), the
// fine part from SHORT polynomials: exp to r^6 on |r| <= ln 2 / 64 (truncation 3e-18), sin to t^9, cos to t^8 on
// |t| <= pi / 64 (2e-21, 2e-20) -- 13 polynomial steps instead of 26, and the table covers the full circle, so the
// quadrant selects go as well.  ~45 instead of ~65 instructions per complex exponential.  Max error against quad
// precision (tests/test_emu_core.py::test_table_math_accuracy): <= 2.5 ulp of the larger of |re|, |im|.
// tab: MATH_TAB_N doubles (CARMA_MATH_TAB_VALUES).
CARMA_DEV double exp_neg_tab(double x, const double* tab)
{
    // (|x| >~ 1e52 -- reachable only with ignore_prior and a non-physical theta -- leaves r large, the polynomial overflows and the
    // result is inf / NaN where exp() would saturate.  A NaN-preserving clamp of the argument costs 6 instructions per call, 4 % of
    // the throughput kernels' instruction stream (8.29e7 -> 8.65e7 VALU instructions per 65 536-evaluation launch, measured in
    // round 5) -- not taken: such an evaluation returns NaN, which every caller treats as a failed evaluation.)
    const double n = rint(x * INV_LN2_32);
    double r = fma3(-n, LN2_32_HI, x);
    r = fma3(-n, LN2_32_LO, r);
    // n is integral; beyond +-2200 * 32 the result has long over/underflowed, clamp so the int conversion is well defined
    const int i = (int)fmin(fmax(n, -70400.0), 70400.0);
    const double e = tab[i & 31];
    // exp(r) - 1 = r q(r): the table entry enters as e + e (r q), so its rounding is the only half-ulp that is not scaled down
    double q = 1.0 / 720.0;
    q = fma3(q, r, 1.0 / 120.0);
    q = fma3(q, r, 1.0 / 24.0);
    q = fma3(q, r, 1.0 / 6.0);
    q = fma3(q, r, 0.5);
    q = fma3(q, r, 1.0);
    return ldexp(fma3(e, q * r, e), i >> 5);
}

// phases the table form reduces itself: |n| < 2^20 keeps n PI_32_1 and n PI_32_2 exact
constexpr double CEXP_TAB_MAXPHASE = 98304.0;

// EXACT: the rounding of the products a dt and b dt is recovered with an FMA and added to the reduced arguments (as in
// cexp_step_impl): the phase then stays good to 1e-16 rad at any |b dt| the reduction accepts.
template <bool CHECK, bool EXACT = false>
CARMA_DEV void cexp_step_tab_impl(double a, double b, double dt, double* re, double* im, const double* tab)
{
    const double x = a * dt;
    const double ph = b * dt;
    if (CHECK && !(fabs(ph) < CEXP_TAB_MAXPHASE)) {
        // rare: library reduction for huge phases (NaN also lands here)
        const double e = exp_neg_tab(x, tab);
        SinCos sc = sincos_slow(ph);
        if constexpr (EXACT) {
            const double pl = fma3(b, dt, -ph);
            const double c0 = sc.c, s0 = sc.s;
            sc.c = fma3(-s0, pl, c0);
            sc.s = fma3(c0, pl, s0);
        }
        *re = e * sc.c;
        *im = e * sc.s;
        return;
    }
    const double n1 = rint(x * INV_LN2_32);
    const double n2 = rint(ph * INV_PI_32);
    double r = fma3(-n1, LN2_32_HI, x);
    double t = fma3(-n2, PI_32_1, ph);
    r = fma3(-n1, LN2_32_LO, r);
    t = fma3(-n2, PI_32_2, t);
    t = fma3(-n2, PI_32_3, t);
    if constexpr (EXACT) {
        r += fma3(a, dt, -x);
        t += fma3(b, dt, -ph);
    }
    const int i1 = (int)fmin(fmax(n1, -70400.0), 70400.0);
    const int i2 = (int)n2;
    const double e0 = tab[i1 & 31];
    const double sa = tab[MATH_TAB_SC + 2 * (i2 & 63)], ca = tab[MATH_TAB_SC + 2 * (i2 & 63) + 1];
    const double z = t * t;
    // exp(r) - 1 = r pe, sin t - t = t z ps, cos t - 1 = z pc: the table entries enter as x + x (small), see exp_neg_tab
    double pe = 1.0 / 720.0;
    double ps = 1.0 / 362880.0;
    double pc = 1.0 / 40320.0;
    pe = fma3(pe, r, 1.0 / 120.0);
    ps = fma3(ps, z, -1.0 / 5040.0);
    pc = fma3(pc, z, -1.0 / 720.0);
    pe = fma3(pe, r, 1.0 / 24.0);
    ps = fma3(ps, z, 1.0 / 120.0);
    pc = fma3(pc, z, 1.0 / 24.0);
    pe = fma3(pe, r, 1.0 / 6.0);
    ps = fma3(ps, z, -1.0 / 6.0);
    pc = fma3(pc, z, -0.5);
    pe = fma3(pe, r, 0.5);
    const double tz = t * z;
    const double cm = pc * z;                                // cos t - 1
    pe = fma3(pe, r, 1.0);
    const double st = fma3(tz, ps, t);                       // sin t
    const double e = ldexp(fma3(e0, pe * r, e0), i1 >> 5);
    const double sn = fma3(sa, cm, ca * st) + sa;            // sin(k pi / 32 + t) = sa + (sa (cos t - 1) + ca sin t)
    const double cs = fma3(ca, cm, -(sa * st)) + ca;
    *re = e * cs;
    *im = e * sn;
}

template <bool EXACT = false>
CARMA_DEV void cexp_step_tab(double a, double b, double dt, double* re, double* im, const double* tab)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const bool slow = !(fabs(b * dt) < CEXP_TAB_MAXPHASE);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(slow) != 0, 0)) {     // once per wave: see cexp_step
        cexp_step_tab_impl<true, EXACT>(a, b, dt, re, im, tab);
        return;
    }
    cexp_step_tab_impl<false, EXACT>(a, b, dt, re, im, tab);
#else
    cexp_step_tab_impl<true, EXACT>(a, b, dt, re, im, tab);
#endif
}

#if defined(__HIPCC__)
// the tables in device memory, and their copy into a kernel's LDS (all threads of the workgroup call it; the caller
// synchronises before the first use)
__device__ static const double c_math_tab[MATH_TAB_N] = {CARMA_MATH_TAB_VALUES};
__device__ __forceinline__ void math_tab_fill(double* lds_tab)
{
    for (int i = threadIdx.x; i < MATH_TAB_N; i += blockDim.x) lds_tab[i] = c_math_tab[i];
}
#else
static const double h_math_tab[MATH_TAB_N] = {CARMA_MATH_TAB_VALUES};
#endif

// The huge-phase test is made ONCE PER WAVE: a per-lane test compiles to an exec-masked block in front of the fast path
// and the branch over it is TAKEN every time -- ~30 cycles of instruction-fetch bubble per evaluation when the SIMD has
// one or two waves.  If no lane of the wave needs the library reduction (the rule) the wave falls through into the fast
// path; if one does, every lane runs the per-lane form, in which the fast lanes execute the same instructions as here.
template <bool EXACT = false>
CARMA_DEV void cexp_step(double a, double b, double dt, double* re, double* im, double dt_lo = 0.0)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const bool slow = !(fabs(b * dt) < 1048576.0);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(slow) != 0, 0)) {
        cexp_step_impl<EXACT, true>(a, b, dt, re, im, dt_lo);
        return;
    }
    cexp_step_impl<EXACT, false>(a, b, dt, re, im, dt_lo);
#else
    cexp_step_impl<EXACT, true>(a, b, dt, re, im, dt_lo);
#endif
}

}  // namespace carma
