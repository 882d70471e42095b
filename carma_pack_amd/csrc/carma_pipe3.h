// carma_pipe3.h -- the row variant (one evaluation per 16-lane DPP row, filter_loop_row) as a
// THREE-WAVE PIPELINE for the smallest launches (<= 1024 evaluations, gfx950 only).
//
// The covariance recursion (D, gain k, var) does not depend on the data or on the state mean, so the
// step is cut along that line and the two halves run on different SIMDs of the CU, one chunk apart:
//   wave P (producer)   rho_r(k) = exp(omega_r dt_k), series records          -> rho ring (3 buffers)
//   wave A (covariance) var_{k-1} = s0 + e + h.w ; s = 1/var ; nt = -k s ; d = D + nt k k^T ;
//                       D = Phi d Phi^T ; w = D h^T ; k = w + c                -> {k_r, var} ring (2 buffers)
//   wave B (mean)       innov = y - mu - h.z ; chi2 += innov^2 / var ; sum log var ;
//                       z = Phi (z + k innov / var)                            -> log-likelihood
// (kfilter.cpp:189-215 / carpack.hpp:167-171, same arithmetic per element as filter_loop_row.)
// Wave A, the critical one, issues ~64 slots per step instead of ~88: no mean sums, no state update,
// no log-var bookkeeping.  One __syncthreads() per 16-step chunk for all three waves:
//   after barrier b:  P writes rho chunk b+1 (buffer (b+1)%3), A works on chunk b (reads rho b%3,
//   writes link b%2), B works on chunk b-1 (reads rho (b-1)%3, link (b-1)%2)  -- all distinct buffers.
// The loop runs n passes: pass kk closes var_{kk-1} / innov_{kk-1} and applies Update kk (pass n only closes).
#pragma once
#include <hip/hip_runtime.h>

#include "carma_core.h"
#include "grp_device.h"
#include "carma_ring.h"

namespace carma {

template <int P>
struct Pipe3Geom {
    static constexpr int C = 16, SLOT = 64;
    static constexpr int RHO_OFF = 0;                           // Cx[3][C][SLOT]
    static constexpr int LINK_OFF = 3 * C * SLOT;               // double2 {k_r, var}[2][C][SLOT]
    static constexpr int REC_OFF = LINK_OFF + 2 * C * SLOT;     // double2 {y, yerr^2}[3][C]
    static constexpr int ENTRIES = REC_OFF + 3 * C;
    static constexpr size_t BYTES = (size_t)ENTRIES * sizeof(Cx);   // 80.8 KiB
};

// wave P
template <int P>
__device__ __forceinline__ void pipe3_produce(const Grp<16>& g, const double* __restrict__ theta,
                                              const double4* __restrict__ series, int n, Cx* __restrict__ ring)
{
    using Geo = Pipe3Geom<P>;
    constexpr int C = Geo::C;
    constexpr int PPL = 16 / P;
    const int lane = g.lane64, l = lane & 15;
    const int sub = l / P, jr = l - sub * P;
    const bool worker = sub < PPL;
    const Cx w = own_ar_root<P>(theta, jr);
    const int nc = (n + C - 1) / C;                           // passes 1 .. n
    double2* recs = reinterpret_cast<double2*>(ring + Geo::REC_OFF);
    if (l >= P) {
#pragma unroll 4
        for (int i = 0; i < 3 * C; i++) ring[(size_t)i * Geo::SLOT + lane] = Cx{0.0, 0.0};
    }
    const int ls = lane < C ? lane : C - 1;
    auto clampi = [n](int i) { return i < n ? i : n - 1; };
    double4 rec_n = series[clampi(ls)];
    double dt_n = series[clampi(1 + ls)].x;
    for (int c = 0; c < nc; c++) {
        Cx* buf = ring + (size_t)(c % 3) * C * Geo::SLOT + (lane & ~15) + jr;
        const double4 rec_c = rec_n;
        const double dt_c = dt_n;
        const int kk0 = 1 + c * C;
        rec_n = series[clampi(kk0 + C + ls - 1)];
        dt_n = series[clampi(kk0 + C + ls)].x;
        if (lane < C && kk0 + lane <= n) recs[(c % 3) * C + lane] = make_double2(rec_c.y, rec_c.z);   // record kk-1
#pragma unroll 1
        for (int s0 = 0; s0 < C; s0 += PPL) {
            const int slot = s0 + sub;
            const double dt = __shfl(dt_c, slot < C ? slot : C - 1, 64);
            if (worker && slot < C && kk0 + slot < n) {
                Cx rho;
                cexp_step(w.re, w.im, dt, &rho.re, &rho.im);
                buf[(size_t)slot * Geo::SLOT] = rho;
            }
        }
        __syncthreads();                                      // barrier c: chunk c is in the ring
    }
    __syncthreads();                                          // barrier nc (wave B's last chunk)
}

// real-coordinate constants of one evaluation, as held by lane r of its row (see filter_loop_real)
template <int P>
struct RowConsts {
    double h_own, c_own, s0;
    double hall[P];
};
template <int P>
__device__ __forceinline__ void row_consts(const Grp<16>& g, const Model<P>& m, const FilterConsts<P>& fc, RowConsts<P>& rc)
{
    const int r = g.lane();
    const bool act = r < P;
    const bool cpx = (m.w.im != 0.0) && (r < (P & ~1));
    const bool odd = r & 1;
    const double c_im_partner = g.partner(fc.c_own.im);
    rc.h_own = !act ? 0.0 : (cpx ? (odd ? 2.0 * fc.b_own.im : 2.0 * fc.b_own.re) : fc.b_own.re);
    rc.c_own = cpx ? (odd ? c_im_partner : fc.c_own.re) : fc.c_own.re;
    rc.s0 = fc.s0;
#pragma unroll
    for (int j = 0; j < P; j++) rc.hall[j] = g.bcast_u(rc.h_own, j);
}

// wave A: the covariance recursion
template <int P>
__device__ __forceinline__ void pipe3_cov(const Grp<16>& g, const Model<P>& m, const RowConsts<P>& rc, int n,
                                          Cx* __restrict__ ring)
{
    using Geo = Pipe3Geom<P>;
    constexpr int C = Geo::C;
    const int lane = g.lane64, r = g.lane();
    const int nc = (n + C - 1) / C;
    double D[P];
#pragma unroll
    for (int j = 0; j < P; j++) D[j] = 0.0;
    double w = 0.0;
    const double2* rho_b = nullptr;      // this lane's rho entries of the current chunk
    const double2* rec_b = nullptr;
    double2* link_b = nullptr;
    double2 rho_n = make_double2(1.0, 0.0), rec_n = make_double2(0.0, 0.0);
    auto pass = [&](const int s, const bool more) __attribute__((always_inline)) {
        const double2 rho = rho_n, rec = rec_n;
        if (more) {                                           // next slot of the same chunk
            rho_n = rho_b[(size_t)(s + 1) * Geo::SLOT];
            rec_n = rec_b[s + 1];
        }
        // var_{kk-1} = s0 + e + h.w   (kfilter.cpp:180-182, 209-210)
        double var, k;                                        // k = w + c, the gain Cov(z, y) of this step
        g.template row_sums_var<P>(var, k, rec.y, m.scale, rc.s0, w, rc.c_own, rc.hall);
        const double sv = recip(var);
        // d_j = D_j - (k s) k_j   (kfilter.cpp:197); the mean wave gets the gain k_r and var
        double nt;
        g.template row_gain_cov<P>(nt, D, k, sv);
        link_b[(size_t)s * Geo::SLOT] = make_double2(k, var);
        // D = Phi d Phi^T   (kfilter.cpp:204)
        double mm[P];
        g.template row_colmix<P>(mm, rho.x, rho.y, D);
        double w0 = 0.0, w1 = 0.0;
#pragma unroll
        for (int j = 0; j < P; j++) {
            const double mp = g.partner(mm[j]);
            D[j] = fma(rho.x, mm[j], -(rho.y * mp));
            if (j & 1)
                w1 = fma(D[j], rc.hall[j], w1);
            else
                w0 = fma(D[j], rc.hall[j], w0);
        }
        w = w0 + w1;                                          // (D h^T)_r
    };
    for (int c = 0; c < nc; c++) {
        __syncthreads();                                      // barrier c
        rho_b = reinterpret_cast<const double2*>(ring + (size_t)(c % 3) * C * Geo::SLOT) + lane;
        rec_b = reinterpret_cast<const double2*>(ring + Geo::REC_OFF) + (c % 3) * C;
        link_b = reinterpret_cast<double2*>(ring + Geo::LINK_OFF) + (size_t)(c & 1) * C * Geo::SLOT + lane;
        rho_n = rho_b[0];
        rec_n = rec_b[0];
        const int len = (n - c * C < C) ? n - c * C : C;      // passes of this chunk
        if (len == C) {
#pragma unroll 4
            for (int s = 0; s < C; s++) pass(s, s + 1 < C);
        } else {
#pragma unroll 1
            for (int s = 0; s < len; s++) pass(s, true);      // (slot len is inside the buffer: len < C)
        }
    }
    (void)r;
    __syncthreads();                                          // barrier nc
}

// wave A with SPLIT ROWS: row r of D lives in two lanes of the 16-lane DPP row -- lane r holds the columns
// of the even root pairs (half A), lane 8 + r those of the odd pairs (half B; RowAsm<P>::HALF / SLOT).
// The DPP instructions are per column either way (one v_fmac_f64_dpp per column, now with a bank mask
// that selects the owning half), but everything that is per ENTRY -- the zero-initialisation of the
// column mix, the row mix with its partner moves, the h-weighted row sum -- shrinks from p to
// NSLOT = ceil(p/2 pairs) entries per lane; the two partial row sums meet by one row_ror:8.
template <int P>
__device__ __forceinline__ void pipe3_cov_split(const Grp<16>& g, const Model<P>& m, const RowConsts<P>& rc, int n,
                                                Cx* __restrict__ ring)
{
    using Geo = Pipe3Geom<P>;
    using RA = RowAsm<P>;
    constexpr int C = Geo::C, NS = RA::NSLOT;
    const int lane = g.lane64;
    const bool halfB = (lane & 8) != 0;
    const int nc = (n + C - 1) / C;
    // constants of row r in BOTH of its lanes
    const double c_own = __shfl(rc.c_own, lane & ~8, 64);
    double hs[NS];                                   // h of the column held in slot i of this half
#pragma unroll
    for (int i = 0; i < NS; i++) {
        double ha = 0.0, hb = 0.0;
#pragma unroll
        for (int j = 0; j < P; j++) {
            if (RA::SLOT[j] == i && RA::HALF[j] == 0) ha = rc.hall[j];
            if (RA::SLOT[j] == i && RA::HALF[j] == 1) hb = rc.hall[j];
        }
        hs[i] = halfB ? hb : ha;
    }
    double D[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) D[i] = 0.0;
    double w = 0.0;
    const double2* rho_b = nullptr;
    const double2* rec_b = nullptr;
    double2* link_b = nullptr;
    double2 rho_n = make_double2(1.0, 0.0), rec_n = make_double2(0.0, 0.0);
    auto pass = [&](const int s, const bool more) __attribute__((always_inline)) {
        const double2 rho = rho_n, rec = rec_n;
        if (more) {
            rho_n = rho_b[(size_t)(s + 1) * Geo::SLOT];
            rec_n = rec_b[s + 1];
        }
        double var, k;
        g.template row_sums_var<P>(var, k, rec.y, m.scale, rc.s0, w, c_own, rc.hall);
        const double sv = recip(var);
        double nt;
        RA::gain_split(nt, D, k, sv);
        link_b[(size_t)s * Geo::SLOT] = make_double2(k, var);
        double mm[NS];
        RA::colmix_split(mm, rho.x, rho.y, D);
        double wp = 0.0;
#pragma unroll
        for (int i = 0; i < NS; i++) {
            const double mp = g.partner(mm[i]);
            D[i] = fma(rho.x, mm[i], -(rho.y * mp));
            wp = fma(D[i], hs[i], wp);
        }
        w = wp + __builtin_amdgcn_update_dpp(wp, wp, 0x128, 0xf, 0xf, true);      // + the other half's partial (row_ror:8)
    };
    for (int c = 0; c < nc; c++) {
        __syncthreads();                                      // barrier c
        rho_b = reinterpret_cast<const double2*>(ring + (size_t)(c % 3) * C * Geo::SLOT) + (lane & ~8);
        rec_b = reinterpret_cast<const double2*>(ring + Geo::REC_OFF) + (c % 3) * C;
        link_b = reinterpret_cast<double2*>(ring + Geo::LINK_OFF) + (size_t)(c & 1) * C * Geo::SLOT + lane;
        rho_n = rho_b[0];
        rec_n = rec_b[0];
        const int len = (n - c * C < C) ? n - c * C : C;
        if (len == C) {
#pragma unroll 4
            for (int s = 0; s < C; s++) pass(s, s + 1 < C);
        } else {
#pragma unroll 1
            for (int s = 0; s < len; s++) pass(s, true);
        }
    }
    __syncthreads();                                          // barrier nc
}

// wave B: the state mean and the log-likelihood sum
template <int P>
__device__ __forceinline__ double pipe3_mean(const Grp<16>& g, const Model<P>& m, const RowConsts<P>& rc, int n,
                                             const Cx* __restrict__ ring)
{
    using Geo = Pipe3Geom<P>;
    constexpr int C = Geo::C;
    const int lane = g.lane64;
    const int nc = (n + C - 1) / C;
    double z = 0.0;
    LogLikAcc acc;
    acc.init();
    const double2* rho_b = nullptr;
    const double2* rec_b = nullptr;
    const double2* link_b = nullptr;
    double2 rho_n = make_double2(1.0, 0.0), rec_n = make_double2(0.0, 0.0), lk_n = make_double2(0.0, 1.0);
    auto pass = [&](const int s, const bool more) __attribute__((always_inline)) {
        const double2 rho = rho_n, rec = rec_n, lk = lk_n;    // lk = {k_r, var_{kk-1}}
        if (more) {
            rho_n = rho_b[(size_t)(s + 1) * Geo::SLOT];
            rec_n = rec_b[s + 1];
            lk_n = link_b[(size_t)(s + 1) * Geo::SLOT];
        }
        // innov_{kk-1} = (y - mu) - h.z   (kfilter.cpp:184, 207, 213); log-likelihood terms (carpack.hpp:167-171)
        double innov;
        g.template row_sums_innov<P>(innov, rec.x, m.mu, z, rc.hall);
        acc.add_var(lk.y);
        const double si = recip(lk.y) * innov;
        acc.chi2 += innov * si;
        // z = Phi (z + k s innov)   (kfilter.cpp:191-194, 200-201); same operation order as filter_loop_row,
        // so the two kernels agree bit for bit
        z = fma(lk.x, si, z);
        const double zp = g.partner(z);
        z = fma(rho.x, z, -(rho.y * zp));
    };
    __syncthreads();                                          // barrier 0
    for (int c = 0; c < nc; c++) {
        __syncthreads();                                      // barrier c + 1: wave A has finished chunk c
        rho_b = reinterpret_cast<const double2*>(ring + (size_t)(c % 3) * C * Geo::SLOT) + lane;
        rec_b = reinterpret_cast<const double2*>(ring + Geo::REC_OFF) + (c % 3) * C;
        link_b = reinterpret_cast<const double2*>(ring + Geo::LINK_OFF) + (size_t)(c & 1) * C * Geo::SLOT + lane;
        rho_n = rho_b[0];
        rec_n = rec_b[0];
        lk_n = link_b[0];
        const int len = (n - c * C < C) ? n - c * C : C;
        if (len == C) {
#pragma unroll 4
            for (int s = 0; s < C; s++) pass(s, s + 1 < C);
        } else {
#pragma unroll 1
            for (int s = 0; s < len; s++) pass(s, true);
        }
    }
    return acc.total();
}

}  // namespace carma
