// carma_scan_dev.h -- device side of the time-parallel filter (carma_scan.h): model set-up shared through
// LDS, the scan over the lanes of a wave with LDS exchange, the final reduction.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include "carma_scan.h"
#include "grp_device.h"

namespace carma {

template <int P>
struct ScanLds {
    static constexpr int NE = ScanDim<P>::NE;
    static constexpr int NPAIR = (NE + 1) / 2;                       // double2 slots per element
    static constexpr int MODEL = 4 * P + P * P + 8;                  // doubles of the shared model
    static constexpr size_t BYTES = (size_t)NPAIR * 64 * sizeof(double2) + (size_t)MODEL * sizeof(double);
};

// Set-up with the lane-distributed code of carma_core.h (one 16-lane row; the four rows of the wave do the
// same work) and publication of the model in real modal coordinates.  Returns false when the evaluation is
// rejected before any filtering (prior bounds, singular system).
template <int P>
__device__ __forceinline__ bool scan_setup(const double* __restrict__ theta, int q, const Prior& pr, int ignore_prior,
                                           int lane64, double* __restrict__ sh, ScanModel<P>& sm, double* logprior)
{
    Grp<16> g{nullptr, lane64, nullptr};
    const int r = g.lane();
    Model<P> m;
    model_from_theta<P, 16>(g, theta, q, pr, ignore_prior, m);
    FilterConsts<P> fc;
    filter_reset<P, 16>(g, m, fc);
    // complex row r of V (kfilter.cpp:165-172)
    Cx a[P], bq[P];
#pragma unroll
    for (int j = 0; j < P; j++) {
        const Cx num = cmulc(cscale(fc.Jown, -m.sigsqr), fc.Jall[j]);
        const Cx den = {m.w.re + m.wall[j].re, m.w.im - m.wall[j].im};
        a[j] = cdiv(num, den);
        bq[j] = Cx{g.partner(a[j].re), g.partner(a[j].im)};
    }
    const bool cpx_own = (m.w.im != 0.0) && (r < (P & ~1));
    const bool odd = r & 1;
    Cx u[P];
#pragma unroll
    for (int j = 0; j < P; j++) {
        if (!cpx_own) {
            u[j] = a[j];
        } else if (!odd) {
            u[j] = Cx{0.5 * (a[j].re + bq[j].re), 0.5 * (a[j].im + bq[j].im)};
        } else {                                       // (-i/2)(V_even - V_odd), own row is the odd one
            const Cx dlt = {bq[j].re - a[j].re, bq[j].im - a[j].im};
            u[j] = Cx{0.5 * dlt.im, -0.5 * dlt.re};
        }
    }
    double vz[P];
#pragma unroll
    for (int j = 0; j < P; j++) {
        const bool ccol = (j < (P & ~1)) && (m.wall[j & ~1].im != 0.0);
        if (!ccol) {
            vz[j] = u[j].re;
        } else if (!(j & 1)) {
            vz[j] = 0.5 * (u[j].re + u[j + 1 < P ? j + 1 : j].re);
        } else {                                       // (i/2)(u_even - u_odd)
            vz[j] = -0.5 * (u[j - 1].im - u[j].im);
        }
    }
    // (the odd lane of a pair holds the conjugate coefficient b_{2k+1} = conj(b_{2k}): -2 Im b_{2k} = 2 Im b_{2k+1})
    const double h_own = (r >= P) ? 0.0 : (cpx_own ? (odd ? 2.0 * fc.b_own.im : 2.0 * fc.b_own.re) : fc.b_own.re);
    double* sh_w = sh;                 // [2P] roots
    double* sh_h = sh + 2 * P;         // [P]
    double* sh_c = sh + 3 * P;         // [P] cpx flags as doubles
    double* sh_v = sh + 4 * P;         // [P][P]
    double* sh_s = sh + 4 * P + P * P; // mu, scale, flag, logprior
    if (lane64 < P) {
        sh_w[2 * r] = m.w.re;
        sh_w[2 * r + 1] = m.w.im;
        sh_h[r] = h_own;
        sh_c[r] = cpx_own ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < P; j++) sh_v[r * P + j] = vz[j];
    }
    if (lane64 == 0) {
        sh_s[0] = m.mu;
        sh_s[1] = m.scale;
        sh_s[2] = (fc.sing || !m.valid) ? 1.0 : 0.0;
        sh_s[3] = log_prior(m.scale, pr.measerr_dof);
    }
    g.sync();
#pragma unroll
    for (int i = 0; i < P; i++) {
        sm.wre[i] = sh_w[2 * i];
        sm.wim[i] = sh_w[2 * i + 1];
        sm.h[i] = sh_h[i];
        sm.cpx[i] = sh_c[i] != 0.0;
    }
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = i; j < P; j++) sm.Vz[sym_idx<P>(i, j)] = 0.5 * (sh_v[i * P + j] + sh_v[j * P + i]);
    sm.s0 = 0.0;
#pragma unroll
    for (int i = 0; i < P; i++) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < P; j++) acc = fma(sym_get<P>(sm.Vz, i, j), sm.h[j], acc);
        sm.c[i] = acc;
        sm.s0 = fma(sm.h[i], acc, sm.s0);
    }
    sm.mu = sh_s[0];
    sm.scale = sh_s[1];
    *logprior = sh_s[3];
    return sh_s[2] == 0.0;
}

// element <-> LDS (component-pair major: slot [pair][lane], conflict-free for any lane shift)
template <int P>
__device__ __forceinline__ void scan_store(const ScanElem<P>& e, double2* __restrict__ xch, int lane)
{
    constexpr int NE = ScanDim<P>::NE;
    const double* v = reinterpret_cast<const double*>(&e);
#pragma unroll
    for (int i = 0; i < NE / 2; i++) xch[i * 64 + lane] = make_double2(v[2 * i], v[2 * i + 1]);
    if (NE & 1) xch[(NE / 2) * 64 + lane] = make_double2(v[NE - 1], 0.0);
}
template <int P>
__device__ __forceinline__ void scan_load(ScanElem<P>& e, const double2* __restrict__ xch, int lane)
{
    constexpr int NE = ScanDim<P>::NE;
    double* v = reinterpret_cast<double*>(&e);
#pragma unroll
    for (int i = 0; i < NE / 2; i++) {
        const double2 t = xch[i * 64 + lane];
        v[2 * i] = t.x;
        v[2 * i + 1] = t.y;
    }
    if (NE & 1) v[NE - 1] = xch[(NE / 2) * 64 + lane].x;
}

// One evaluation per wave.  SMAX = steps per lane the kernel is compiled for (n <= 64 SMAX).
template <int P, int SMAX>
__device__ __forceinline__ double scan_logdensity(const double* __restrict__ theta, int q, const double4* __restrict__ series,
                                                  int n, const Prior& pr, int ignore_prior, int lane, double2* xch, double* sh)
{
    ScanModel<P> sm;
    double logprior;
    CARMA_STAMP_DECL;
    CARMA_STAMP(st0);
    const bool ok = scan_setup<P>(theta, q, pr, ignore_prior, lane, sh, sm, &logprior);
    CARMA_STAMP(st1);
    const int s = (n + 63) / 64;
    const int k0 = lane * s, k1 = (k0 + s < n) ? k0 + s : n;
    const bool has = k0 < n;
    ScanElem<P> el;
    ScanPhi<P> phis[SMAX];
    if (has)
        scan_block_element<P, SMAX>(sm, series, k0, k1, lane == 0, el, phis);
    else
        scan_identity<P>(el);
    CARMA_STAMP(st2);
    // phase 2: inclusive scan over the lanes
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) {
        scan_store<P>(el, xch, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        ScanElem<P> left, out;
        scan_load<P>(left, xch, lane >= d ? lane - d : lane);
        __builtin_amdgcn_wave_barrier();
        scan_combine<P>(left, el, out);
        if (lane >= d) el = out;
    }
    CARMA_STAMP(st3);
    // phase 3: the block's share of the log-likelihood from the prefix of the lane before
    scan_store<P>(el, xch, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ScanElem<P> prev;
    scan_load<P>(prev, xch, lane > 0 ? lane - 1 : 0);
    LogLikAcc acc;
    acc.init();
    if (has) scan_block_loglik<P, SMAX>(sm, series, k0, k1, lane == 0, prev.b, prev.C, phis, acc);
    double part = has ? acc.total() : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
#if defined(CARMA_STAMPS)
    CARMA_STAMP(st4);
    if (blockIdx.x == 0 && lane == 0)
        printf("scan stamps (cycles): setup %llu  phase1 %llu  phase2 %llu  phase3+reduce %llu\n", st1 - st0, st2 - st1, st3 - st2,
               st4 - st3);
#endif
    const double ninf = -1.0 / 0.0;
    return ok ? part + logprior : ninf;
}

}  // namespace carma
