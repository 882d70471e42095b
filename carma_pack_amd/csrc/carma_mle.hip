// carma_mle.hip -- lock-step bounded quasi-Newton minimiser of -LogDensity for MANY starts at once.
//
// carma_pack's get_mle runs `ntrials` separate scipy L-BFGS-B searches and crosses the FFI once per function evaluation
// (reference carma_pack.py:92-129,195-260).  On the GPU one log-density costs the same as a thousand, so all starts are
// advanced together: per iteration ONE batched launch evaluates the central-difference stencils of every active start
// (B x (2d+1) points) and one more evaluates eight consecutive backtracking step lengths of every start.  This file is
// the host side of that loop in C++ (the Python prototype, carma_pack_amd/batched_opt.py, spent as long in the
// interpreter as in the launches: 5.4 of 13.9 s of choose_order(pmax=7, ntrials=100)); same algorithm, same constants.
//
// The update is a projected L-BFGS step (two-loop recursion per start; variables sitting on a bound with the gradient
// pointing outwards are frozen) with an Armijo backtracking line search; stopping rules mirror L-BFGS-B's defaults
// (projected gradient <= gtol, or `patience` consecutive iterations with a relative decrease <= ftol after one restart of
// the quasi-Newton memory).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/carma_mi355.h"
#include "carma_host.h"

using namespace carma;

namespace {

constexpr double BIG = 1e300;
constexpr int LS_K = 8;          // step lengths t, t/2, ... evaluated per line-search launch
constexpr int PATIENCE = 3;

struct Eval {
    carma_ctx* h;
    int d, ignore_prior;
    std::vector<double> out;
    // f(pts) = -LogDensity(pts); non-finite -> BIG
    int operator()(const std::vector<double>& pts, int npts)
    {
        out.resize((size_t)npts);
        if (npts == 0) return CARMA_OK;
        const int rc = carma_logdensity_batch(h, pts.data(), npts, ignore_prior, out.data());
        if (rc != CARMA_OK) return rc;
        for (int i = 0; i < npts; i++) {
            const double f = -out[i];
            out[i] = std::isfinite(f) ? f : BIG;
        }
        return CARMA_OK;
    }
};

}  // namespace

extern "C" int carma_mle_batched(carma_ctx* h, const double* x0, int B, const double* lo_in, const double* hi_in, int maxiter,
                                 int mem, double ftol, double gtol, double fd_step, int ignore_prior, double* x_out,
                                 double* fun_out, int* nit_out, int* nfev_out, int* status_out)
{
    if (!h || !x0 || B < 0 || !x_out || !fun_out || mem < 1 || mem > 64 || maxiter < 0) {
        set_error("carma_mle_batched: bad argument");
        return CARMA_EINVAL;
    }
    const int d = carma_ctx_dim(h);
    const int m = mem;
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<double> lo(d, -inf), hi(d, inf);
    for (int j = 0; j < d; j++) {
        if (lo_in && std::isfinite(lo_in[j])) lo[j] = lo_in[j];
        if (hi_in && std::isfinite(hi_in[j])) hi[j] = hi_in[j];
    }
    auto project = [&](double v, int j) { return std::min(std::max(v, lo[j]), hi[j]); };
    Eval fun{h, d, ignore_prior, {}};

    std::vector<double> x((size_t)B * d), f(B), g((size_t)B * d);
    std::vector<int> nfev(B, 0), nit(B, 0), nhist(B, 0), nsmall(B, 0), status(B, 2);   // 2 = maximum number of iterations
    std::vector<char> active(B, 1), restarted(B, 0);
    std::vector<double> S((size_t)B * m * d, 0.0), Y((size_t)B * m * d, 0.0);
    for (int b = 0; b < B; b++)
        for (int j = 0; j < d; j++) x[(size_t)b * d + j] = project(x0[(size_t)b * d + j], j);

    // central differences (one-sided at a bound) of the starts listed in `who` at the points xs: fills fo / go
    std::vector<double> pts, up, dn;
    auto f_and_g = [&](const std::vector<int>& who, const std::vector<double>& xs, std::vector<double>& fo,
                       std::vector<double>& go) -> int {
        const int n = (int)who.size(), w = 2 * d + 1;
        pts.resize((size_t)n * w * d);
        up.resize((size_t)n * d);
        dn.resize((size_t)n * d);
        for (int i = 0; i < n; i++) {
            const double* xi = &xs[(size_t)i * d];
            double* p = &pts[(size_t)i * w * d];
            for (int k = 0; k < w; k++) std::memcpy(p + (size_t)k * d, xi, sizeof(double) * d);
            for (int j = 0; j < d; j++) {
                const double hstep = fd_step * std::max(1.0, std::fabs(xi[j]));
                up[(size_t)i * d + j] = std::min(xi[j] + hstep, hi[j]);
                dn[(size_t)i * d + j] = std::max(xi[j] - hstep, lo[j]);
                p[(size_t)(1 + j) * d + j] = up[(size_t)i * d + j];
                p[(size_t)(1 + d + j) * d + j] = dn[(size_t)i * d + j];
            }
        }
        const int rc = fun(pts, n * w);
        if (rc != CARMA_OK) return rc;
        fo.resize(n);
        go.resize((size_t)n * d);
        for (int i = 0; i < n; i++) {
            const double* fi = &fun.out[(size_t)i * w];
            fo[i] = fi[0];
            for (int j = 0; j < d; j++) {
                const double fu = fi[1 + j], fd_ = fi[1 + d + j];
                double gj = (fu - fd_) / std::max(up[(size_t)i * d + j] - dn[(size_t)i * d + j], 1e-300);
                if (fu >= BIG || fd_ >= BIG) gj = 0.0;
                go[(size_t)i * d + j] = gj;
            }
            nfev[who[i]] += w;
        }
        return CARMA_OK;
    };

    {
        std::vector<int> all(B);
        for (int b = 0; b < B; b++) all[b] = b;
        std::vector<double> f0, g0;
        const int rc = f_and_g(all, x, f0, g0);
        if (rc != CARMA_OK) return rc;
        f = f0;
        g = g0;
    }

    std::vector<int> idx, need, mv;
    std::vector<double> dir, pg, slope, gam, tstep, xn, fn, cand, xmv, fnew, gnew, gspec, xmv2, fnew2, gnew2;
    std::vector<char> frozen, haveg;
    for (int iter = 0; iter < maxiter; iter++) {
        // --- projected gradient test
        idx.clear();
        for (int b = 0; b < B; b++) {
            if (!active[b]) continue;
            double pgmax = 0.0;
            for (int j = 0; j < d; j++) {
                const double xv = x[(size_t)b * d + j], gv = g[(size_t)b * d + j];
                const bool fr = (xv <= lo[j] && gv > 0) || (xv >= hi[j] && gv < 0);
                pgmax = std::max(pgmax, fr ? 0.0 : std::fabs(gv));
            }
            if (pgmax <= gtol) {
                status[b] = 0;          // converged: projected gradient <= gtol
                active[b] = 0;
            } else {
                idx.push_back(b);
            }
        }
        const int na = (int)idx.size();
        if (na == 0) break;
        // --- search directions: two-loop recursion per start
        dir.assign((size_t)na * d, 0.0);
        pg.assign((size_t)na * d, 0.0);
        frozen.assign((size_t)na * d, 0);
        slope.assign(na, 0.0);
        gam.assign(na, 1.0);
        std::vector<double> q(d), alpha(m), r(d);
        for (int i = 0; i < na; i++) {
            const int b = idx[i];
            const double* xb = &x[(size_t)b * d];
            const double* gb = &g[(size_t)b * d];
            double* pgi = &pg[(size_t)i * d];
            char* fri = &frozen[(size_t)i * d];
            double pgn2 = 0.0;
            for (int j = 0; j < d; j++) {
                fri[j] = (xb[j] <= lo[j] && gb[j] > 0) || (xb[j] >= hi[j] && gb[j] < 0);
                pgi[j] = fri[j] ? 0.0 : gb[j];
                pgn2 += pgi[j] * pgi[j];
                q[j] = pgi[j];
            }
            const int nh = nhist[b];
            const double* Sb = &S[(size_t)b * m * d];
            const double* Yb = &Y[(size_t)b * m * d];
            auto dot = [&](const double* a, const double* c) {
                double s_ = 0.0;
                for (int j = 0; j < d; j++) s_ += a[j] * c[j];
                return s_;
            };
            for (int k = nh - 1; k >= 0; k--) {
                const double rho = 1.0 / dot(Sb + (size_t)k * d, Yb + (size_t)k * d);
                alpha[k] = rho * dot(Sb + (size_t)k * d, q.data());
                for (int j = 0; j < d; j++) q[j] -= alpha[k] * Yb[(size_t)k * d + j];
            }
            double gm = 1.0 / std::max(std::sqrt(pgn2), 1e-12);
            if (nh > 0) {
                const double ys = dot(Sb + (size_t)(nh - 1) * d, Yb + (size_t)(nh - 1) * d);
                const double yy = dot(Yb + (size_t)(nh - 1) * d, Yb + (size_t)(nh - 1) * d);
                if (yy > 0) gm = ys / std::max(yy, 1e-300);
            }
            for (int j = 0; j < d; j++) r[j] = gm * q[j];
            for (int k = 0; k < nh; k++) {
                const double rho = 1.0 / dot(Sb + (size_t)k * d, Yb + (size_t)k * d);
                const double be = rho * dot(Yb + (size_t)k * d, r.data());
                for (int j = 0; j < d; j++) r[j] += (alpha[k] - be) * Sb[(size_t)k * d + j];
            }
            double sl = 0.0;
            for (int j = 0; j < d; j++) {
                dir[(size_t)i * d + j] = fri[j] ? 0.0 : -r[j];
                sl += dir[(size_t)i * d + j] * pgi[j];
            }
            if (!(sl < 0)) {            // not a descent direction: steepest descent
                sl = 0.0;
                for (int j = 0; j < d; j++) {
                    dir[(size_t)i * d + j] = -pgi[j] * gm;
                    sl -= pgi[j] * pgi[j] * gm;
                }
            }
            slope[i] = sl;
            gam[i] = gm;
        }
        // --- Armijo backtracking on the projected path: LS_K consecutive step lengths of every start per launch, the
        // FIRST that satisfies the condition is taken -- the step sequential backtracking would take, in ~1 launch.
        // The first launch also carries the difference stencils around its first KS step lengths: a start that takes one of
        // them has its new gradient from the same launch, and the iteration costs it ONE round trip to the device instead of
        // two -- the search is bound by the slowest start's iterations (up to 2000 where the mean is 300-500:
        // tools/choose_order_profile.py), i.e. by round trips, not by evaluations.  KS = 1 (the full step) while many starts
        // are active, up to all LS_K once the launch stays within ~1024 evaluations (the two-sided kernel's flat range).
        tstep.assign(na, 1.0);
        xn.assign((size_t)na * d, 0.0);
        fn.assign(na, 0.0);
        std::vector<char> needf(na, 1);
        haveg.assign(na, 0);
        gspec.resize((size_t)na * d);
        for (int ls = 0; ls < 32; ls += LS_K) {
            need.clear();
            for (int i = 0; i < na; i++)
                if (needf[i]) need.push_back(i);
            if (need.empty()) break;
            const int nn = (int)need.size();
            const int KS = ls == 0 ? std::max(1, std::min(LS_K, (1024 / nn - LS_K) / (2 * d))) : 0;
            const int wl = LS_K + KS * 2 * d;                      // points of one start in this launch
            cand.resize((size_t)nn * wl * d);
            up.resize((size_t)nn * KS * d);
            dn.resize((size_t)nn * KS * d);
            for (int a = 0; a < nn; a++) {
                const int i = need[a], b = idx[i];
                double* ca = &cand[(size_t)a * wl * d];
                double tk = tstep[i];
                for (int k = 0; k < LS_K; k++, tk *= 0.5)
                    for (int j = 0; j < d; j++)
                        ca[(size_t)k * d + j] = project(x[(size_t)b * d + j] + tk * dir[(size_t)i * d + j], j);
                for (int c = 0; c < KS; c++) {                     // the stencil of f_and_g around candidate c
                    const double* xc = ca + (size_t)c * d;
                    double* st = ca + (size_t)(LS_K + 2 * d * c) * d;
                    double* upc = &up[((size_t)a * KS + c) * d];
                    double* dnc = &dn[((size_t)a * KS + c) * d];
                    for (int k = 0; k < 2 * d; k++) std::memcpy(st + (size_t)k * d, xc, sizeof(double) * d);
                    for (int j = 0; j < d; j++) {
                        const double hstep = fd_step * std::max(1.0, std::fabs(xc[j]));
                        upc[j] = std::min(xc[j] + hstep, hi[j]);
                        dnc[j] = std::max(xc[j] - hstep, lo[j]);
                        st[(size_t)j * d + j] = upc[j];
                        st[(size_t)(d + j) * d + j] = dnc[j];
                    }
                }
            }
            const int rc = fun(cand, nn * wl);
            if (rc != CARMA_OK) return rc;
            for (int a = 0; a < nn; a++) {
                const int i = need[a], b = idx[i];
                const double* ca = &cand[(size_t)a * wl * d];
                const double* fa = &fun.out[(size_t)a * wl];
                int first = -1;
                for (int k = 0; k < LS_K && first < 0; k++) {
                    double lin = 0.0;
                    for (int j = 0; j < d; j++) lin += (ca[(size_t)k * d + j] - x[(size_t)b * d + j]) * pg[(size_t)i * d + j];
                    if (fa[k] <= f[b] + 1e-4 * lin) first = k;
                }
                if (first >= 0) {
                    nfev[b] += first + 1;          // as sequential backtracking counts
                    std::memcpy(&xn[(size_t)i * d], ca + (size_t)first * d, sizeof(double) * d);
                    fn[i] = fa[first];
                    needf[i] = 0;
                    if (first < KS) {
                        haveg[i] = 1;
                        const double* fs = fa + LS_K + 2 * d * first;
                        const double* upc = &up[((size_t)a * KS + first) * d];
                        const double* dnc = &dn[((size_t)a * KS + first) * d];
                        for (int j = 0; j < d; j++) {
                            const double fu = fs[j], fd_ = fs[d + j];
                            double gj = (fu - fd_) / std::max(upc[j] - dnc[j], 1e-300);
                            if (fu >= BIG || fd_ >= BIG) gj = 0.0;
                            gspec[(size_t)i * d + j] = gj;
                        }
                        nfev[b] += 2 * d;           // (f_and_g counts 2 d + 1 with the centre: that one is the step's own)
                    }
                } else {
                    nfev[b] += LS_K;
                    tstep[i] *= std::ldexp(1.0, -LS_K);
                }
            }
        }
        mv.clear();
        for (int i = 0; i < na; i++) {
            if (needf[i]) {
                status[idx[i]] = 3;     // line search failed
                active[idx[i]] = 0;
            } else {
                mv.push_back(i);
            }
        }
        if (mv.empty()) continue;
        // --- gradients at the new points (a launch for the starts that did not take the full step), history update, stopping rule
        const int nm = (int)mv.size();
        std::vector<int> who(nm), who2;
        xmv.resize((size_t)nm * d);
        xmv2.clear();
        for (int a = 0; a < nm; a++) {
            who[a] = idx[mv[a]];
            std::memcpy(&xmv[(size_t)a * d], &xn[(size_t)mv[a] * d], sizeof(double) * d);
            if (!haveg[mv[a]]) {
                who2.push_back(who[a]);
                xmv2.insert(xmv2.end(), &xn[(size_t)mv[a] * d], &xn[(size_t)mv[a] * d] + d);
            }
        }
        if (!who2.empty()) {
            const int rc = f_and_g(who2, xmv2, fnew2, gnew2);
            if (rc != CARMA_OK) return rc;
        }
        fnew.resize(nm);
        gnew.resize((size_t)nm * d);
        for (int a = 0, a2 = 0; a < nm; a++) {
            if (haveg[mv[a]]) {
                fnew[a] = fn[mv[a]];
                std::memcpy(&gnew[(size_t)a * d], &gspec[(size_t)mv[a] * d], sizeof(double) * d);
            } else {
                fnew[a] = fnew2[a2];
                std::memcpy(&gnew[(size_t)a * d], &gnew2[(size_t)a2 * d], sizeof(double) * d);
                a2++;
            }
        }
        for (int a = 0; a < nm; a++) {
            const int b = who[a];
            double* Sb = &S[(size_t)b * m * d];
            double* Yb = &Y[(size_t)b * m * d];
            double sy = 0.0, yy = 0.0;
            std::vector<double> sv(d), yv(d);
            for (int j = 0; j < d; j++) {
                sv[j] = xmv[(size_t)a * d + j] - x[(size_t)b * d + j];
                yv[j] = gnew[(size_t)a * d + j] - g[(size_t)b * d + j];
                sy += sv[j] * yv[j];
                yy += yv[j] * yv[j];
            }
            if (sy > 1e-10 * yy) {
                if (nhist[b] == m) {    // drop the oldest pair
                    std::memmove(Sb, Sb + d, sizeof(double) * (size_t)(m - 1) * d);
                    std::memmove(Yb, Yb + d, sizeof(double) * (size_t)(m - 1) * d);
                    nhist[b] = m - 1;
                }
                std::memcpy(Sb + (size_t)nhist[b] * d, sv.data(), sizeof(double) * d);
                std::memcpy(Yb + (size_t)nhist[b] * d, yv.data(), sizeof(double) * d);
                nhist[b]++;
            }
            const double fold = f[b];
            const double rel = (fold - fnew[a]) / std::max(std::max(std::fabs(fold), std::fabs(fnew[a])), 1.0);
            std::memcpy(&x[(size_t)b * d], &xmv[(size_t)a * d], sizeof(double) * d);
            f[b] = fnew[a];
            std::memcpy(&g[(size_t)b * d], &gnew[(size_t)a * d], sizeof(double) * d);
            nit[b]++;
            // L-BFGS-B stops at the first iteration whose relative decrease is <= ftol.  With plain backtracking a single
            // short step in a curved valley is not a reliable sign of convergence: PATIENCE such iterations in a row, and
            // on the first occasion the quasi-Newton memory is dropped before they start to count (batched_opt.py).
            const bool small = rel <= ftol;
            nsmall[b] = small ? nsmall[b] + 1 : 0;
            if (small && !restarted[b]) {
                restarted[b] = 1;
                nhist[b] = 0;
                nsmall[b] = 0;
            }
            if (nsmall[b] >= PATIENCE) {
                status[b] = 1;          // converged: relative reduction of f <= ftol
                active[b] = 0;
            }
        }
    }
    std::memcpy(x_out, x.data(), sizeof(double) * (size_t)B * d);
    std::memcpy(fun_out, f.data(), sizeof(double) * (size_t)B);
    for (int b = 0; b < B; b++) {
        if (nit_out) nit_out[b] = nit[b];
        if (nfev_out) nfev_out[b] = nfev[b];
        if (status_out) status_out[b] = status[b];
    }
    return CARMA_OK;
}
