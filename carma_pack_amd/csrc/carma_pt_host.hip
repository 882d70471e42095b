// carma_pt_host.hip -- host side of the parallel-tempering sampler behind the C ABI:
// temperature ladder, initial proposal covariance, starting values, chunked launches of the
// persistent kernel (carma_pt.hip), sample collection.
// Reference: RunCarmaSampler / RunCar1Sampler (src/carmcmc.cpp:30-177), Sampler::Run
// (src/samplers.cpp:57-115), StartingValue routines (src/carpack.cpp:38-81,175-230,268-311,416-477).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#include "../../include/carma_mi355.h"
#include "carma_host.h"

namespace carma {

void pt_state_free(Ctx* c)
{
    PtState* s = c->pt;
    if (!s) return;
    if (s->d_temps) (void)dev_free(s->d_temps);
    if (!s->ext_state) {
        if (s->d_theta) (void)dev_free(s->d_theta);
        if (s->d_lp) (void)dev_free(s->d_lp);
    }
    if (s->d_chol) (void)dev_free(s->d_chol);
    if (s->d_nacc) (void)dev_free(s->d_nacc);
    if (s->d_nswap) (void)dev_free(s->d_nswap);
    if (s->d_samples) (void)dev_free(s->d_samples);
    if (s->d_slp) (void)dev_free(s->d_slp);
    if (s->d_stage) (void)dev_free(s->d_stage);
    if (s->d_abort) (void)dev_free(s->d_abort);
    if (s->d_backup) (void)dev_free(s->d_backup);
    if (s->d_lane_scratch) (void)dev_free(s->d_lane_scratch);
    if (s->d_send) (void)dev_free(s->d_send);
    if (s->d_recv) (void)dev_free(s->d_recv);
    if (s->d_bnd_swaps) (void)dev_free(s->d_bnd_swaps);
    if (s->d_checksum) (void)dev_free(s->d_checksum);
    delete s;
    c->pt = nullptr;
}

static double pop_var(const std::vector<double>& y)
{
    // src/carmcmc.cpp:85-88
    double sum = 0, sq = 0;
    for (double v : y) {
        sum += v;
        sq += v * v;
    }
    const double mean = sum / y.size();
    return sq / y.size() - mean * mean;
}

static double sample_var(const std::vector<double>& y)   // arma::var
{
    double mean = 0;
    for (double v : y) mean += v;
    mean /= y.size();
    double s = 0;
    for (double v : y) s += (v - mean) * (v - mean);
    return s / (y.size() - 1);
}

static int chunk_iters(const Ctx* c)
{
    // Iterations per launch.  Ladder kernel: around a quarter of a second (~0.8 us per datum per iteration for p >= 5).
    // Row kernel: ~0.14 us per datum per iteration, and its COOPERATIVE launch costs ~2 ms on this stack -- 4.5 % of a
    // 44 ms chunk (measured) -- so its chunks are sized for half a second.
    if (c->pt && c->pt->use_row) {
        const double est_us = std::max(1.0, 0.14 * c->n);
        return (int)std::max(1.0, std::min(16384.0, 500000.0 / est_us));
    }
    if (c->pt && c->pt->use_lane) {                       // large ensembles: >= ~0.45 us per datum per iteration, 2 launches each
        const double est_us = std::max(1.0, 0.45 * c->n);
        return (int)std::max(1.0, std::min(4096.0, 250000.0 / est_us));
    }
    const double est_us = std::max(1.0, 0.8 * c->n * (c->p >= 5 ? 1.0 : 0.5));
    return (int)std::max(1.0, std::min(4096.0, 250000.0 / est_us));
}

// One draw from the reference's starting-value distribution.
static void draw_start(const Ctx* c, std::mt19937_64& rng, double* theta)
{
    const int n = c->n, p = c->p, q = c->q;
    std::normal_distribution<double> norm(0.0, 1.0);
    std::uniform_real_distribution<double> unif(0.0, 1.0);
    auto scaled_inv_chisq = [&](int dof, double ssqr) {       // src/random.cpp:180-186
        std::chi_squared_distribution<double> chi(dof);
        return ssqr / chi(rng) * (double)dof;
    };
    double ymean = 0;
    for (double v : c->y) ymean += v;
    ymean /= n;
    const double yvar = scaled_inv_chisq(n - 1, sample_var(c->y));
    const double mu = ymean + std::sqrt(yvar) / n * norm(rng);
    double scale = scaled_inv_chisq((int)c->pr.measerr_dof, 1.0);
    scale = std::max(std::min(scale, 1.99), 0.51);
    theta[0] = std::sqrt(yvar);
    theta[1] = scale;
    theta[2] = mu;
    if (p == 1) {
        // CAR1::StartingValue (src/carpack.cpp:38-81)
        std::vector<double> dt(n - 1);
        for (int i = 1; i < n; i++) dt[i - 1] = c->t[i] - c->t[i - 1];
        std::sort(dt.begin(), dt.end());
        const double med = (dt.size() % 2) ? dt[dt.size() / 2] : 0.5 * (dt[dt.size() / 2 - 1] + dt[dt.size() / 2]);
        double lw = -1.0 * std::log(med * (1.0 + 49.0 * unif(rng)));
        lw = std::min(lw, c->pr.max_freq);     // sic (carpack.cpp:56)
        theta[3] = lw;
        return;
    }
    // CARp::StartingAR (src/carpack.cpp:268-311)
    const double min_freq = c->pr.min_freq, max_freq = c->pr.max_freq;
    const int nc = (p + 1) / 2;
    std::vector<double> cent(nc), width(nc);
    for (int i = 0; i < nc; i++) cent[i] = std::exp(std::log(max_freq / min_freq) * unif(rng) + std::log(min_freq));
    std::sort(cent.begin(), cent.end(), std::greater<double>());
    for (int i = 0; i < nc; i++) width[i] = std::exp(std::log(max_freq / min_freq) * unif(rng) + std::log(min_freq));
    if (p % 2 == 1) {
        cent[p / 2] = 0.0;
        const double lo = std::log(min_freq), hi = std::log(cent[p / 2 - 1]);
        width[p / 2] = std::exp(lo + (hi - lo) * unif(rng));
    }
    for (int i = 0; i < p / 2; i++) {
        const double re = -2.0 * M_PI * width[i], im = 2.0 * M_PI * cent[i];
        theta[3 + 2 * i] = std::log(re * re + im * im);
        theta[3 + 2 * i + 1] = std::log(-2.0 * re);
    }
    if (p % 2 == 1) theta[3 + p - 1] = std::log(2.0 * M_PI * width[p / 2]);
    // CARMA::StartingMA (src/carpack.cpp:515-519)
    for (int i = 0; i < q; i++) theta[3 + p + i] = std::fabs(norm(rng));
}

static PtLaunch pt_launch_args(const Ctx* c, long ch, int do_exchange, int thin, long save_offset)
{
    const PtState* s = c->pt;
    PtLaunch L{};
    L.d = c->d;
    L.q = c->q;
    L.n = c->n;
    L.T = s->T;
    L.R = s->R;
    L.maxiter = s->maxiter;
    L.iter0 = s->iter;
    L.niter = (int)ch;
    L.do_exchange = do_exchange;
    L.save_thin = thin;
    L.save_offset = save_offset;
    L.sample_cap = s->cap;
    L.seed0 = (unsigned)(s->seed & 0xffffffffu);
    L.seed1 = (unsigned)(s->seed >> 32);
    L.slot0 = s->slot0;
    L.T_global = s->T_global;
    L.replica0 = s->replica0;
    return L;
}

// one launch of `ch` iterations on `st`
static int pt_enqueue_one(Ctx* c, long ch, int do_exchange, int thin, long* save_offset, hipStream_t st)
{
    PtState* s = c->pt;
    const PtLaunch L = pt_launch_args(c, ch, do_exchange, thin, save_offset ? *save_offset : 0);
    hipError_t e;
    if (s->use_row) {
        PtRowSync S{s->d_stage, s->d_abort, ++s->epoch, s->wpl, device_cus(), 1, c->series_flags(), 0};
        e = launch_pt_row(c->p, L, S, reinterpret_cast<const double4*>(c->d_series), c->pr, s->d_temps, s->d_theta, s->d_lp,
                          s->d_chol, s->d_nacc, s->d_nswap, s->d_samples, s->d_slp, st);
        if (e == hipErrorCooperativeLaunchTooLarge) {      // the grid is not co-resident on this device: ladder kernel
            (void)hipGetLastError();
            s->use_row = false;
        }
    }
    if (!s->use_row && s->use_lane) {
        e = launch_pt_lane(c->p, L, s->d_lane_scratch, reinterpret_cast<const double4*>(c->d_series), c->pr, s->d_temps, s->d_theta,
                           s->d_lp, s->d_chol, s->d_nacc, s->d_nswap, s->d_samples, s->d_slp, c->series_flags(), !s->lane_factor_loaded,
                           st);
        if (e == hipSuccess) {
            s->lane_factor_loaded = true;                   // ... and stays in the scratch: the chain-major copy is behind until
            s->chol_stale = true;                           // somebody asks for it (pt_sync_factor)
        }
    } else if (!s->use_row) {
        e = launch_pt(c->p, L, reinterpret_cast<const double4*>(c->d_series), c->pr, s->d_temps, s->d_theta, s->d_lp,
                      s->d_chol, s->d_nacc, s->d_nswap, s->d_samples, s->d_slp, st);
    }
    if (e != hipSuccess) return hip_fail(e, "launch pt kernel");
    s->iter += ch;
    s->stat_iters += ch;
    if (thin > 0 && save_offset) *save_offset += ch / thin;
    return CARMA_OK;
}

int pt_enqueue(Ctx* c, long niter, int do_exchange, int thin, long* save_offset, hipStream_t st)
{
    const int chunk0 = chunk_iters(c);
    long left = niter;
    while (left > 0) {
        long ch = std::min<long>(left, chunk0);
        if (thin > 0) {
            ch = std::max<long>(thin, (ch / thin) * thin);   // whole thinning intervals per launch
            ch = std::min(ch, left);
        }
        int rc = pt_enqueue_one(c, ch, do_exchange, thin, save_offset, st);
        if (rc != CARMA_OK) return rc;
        left -= ch;
    }
    return CARMA_OK;
}

int pt_check_abort(Ctx* c, bool* aborted)
{
    PtState* s = c->pt;
    *aborted = false;
    if (!s->use_row) return CARMA_OK;
    unsigned flag = 0;
    hipError_t e = hipMemcpy(&flag, s->d_abort, sizeof(unsigned), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "pt abort flag");
    *aborted = flag != 0;
    return CARMA_OK;
}

// The chain-major factors d_chol brought up to date with the lane sampler's working state (no-op for the other kernels): before
// anything reads d_chol.  Enqueued on st.
hipError_t pt_sync_factor(Ctx* c, hipStream_t st)
{
    PtState* s = c->pt;
    if (!s || !s->use_lane || !s->chol_stale) return hipSuccess;
    const hipError_t e = pt_lane_store_factor(c->d, s->T, s->R, s->d_lane_scratch, s->d_chol, st);
    if (e == hipSuccess) s->chol_stale = false;
    return e;
}
// ... and the other way round: d_chol was written (carma_pt_set_factor, a restored backup): the next launch reloads the factors
void pt_factor_written(Ctx* c)
{
    if (c->pt) {
        c->pt->lane_factor_loaded = false;
        c->pt->chol_stale = false;
    }
}

// chain state <-> backup (theta, logpost, chol), asynchronous on st
static hipError_t pt_backup(Ctx* c, bool restore, hipStream_t st)
{
    PtState* s = c->pt;
    const size_t nchain = (size_t)s->T * s->R, d = c->d;
    double* b = s->d_backup;
    struct Part { double* p; size_t n; } parts[3] = {{s->d_theta, nchain * d}, {s->d_lp, nchain}, {s->d_chol, nchain * d * d}};
    if (!restore) {
        const hipError_t es = pt_sync_factor(c, st);
        if (es != hipSuccess) return es;
    } else {
        pt_factor_written(c);
    }
    for (auto& pt : parts) {
        hipError_t e = restore ? hipMemcpyAsync(pt.p, b, sizeof(double) * pt.n, hipMemcpyDeviceToDevice, st)
                               : hipMemcpyAsync(b, pt.p, sizeof(double) * pt.n, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return e;
        b += pt.n;
    }
    return hipSuccess;
}

// Run `niter` iterations and wait for them.  The ladder kernel's launches are all enqueued first; the row kernel
// (cross-workgroup rendezvous) is run chunk by chunk with the chain state saved in front of every chunk: should a
// rendezvous ever time out (the launch is cooperative, so the grid is co-resident by construction; the time-out only
// guards against a wedged GPU), the chunk is restored and re-run -- like everything after it -- with the ladder kernel.
static int pt_launch_chunks(Ctx* c, long niter, int do_exchange, int thin, long* save_offset)
{
    PtState* s = c->pt;
    if (!s->use_row) {
        int rc = pt_enqueue(c, niter, do_exchange, thin, save_offset, c->stream);
        if (rc != CARMA_OK) return rc;
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return hip_fail(e, "pt kernel");
        return CARMA_OK;
    }
    const int chunk0 = chunk_iters(c);
    long left = niter;
    while (left > 0) {
        long ch = std::min<long>(left, chunk0);
        if (thin > 0) {
            ch = std::max<long>(thin, (ch / thin) * thin);
            ch = std::min(ch, left);
        }
        if (!s->use_row) return pt_launch_chunks(c, left, do_exchange, thin, save_offset);     // after a fall-back
        const unsigned long long iter_before = s->iter, stat_before = s->stat_iters;
        const long off_before = save_offset ? *save_offset : 0;
        hipError_t e = pt_backup(c, false, c->stream);
        if (e != hipSuccess) return hip_fail(e, "pt state backup");
        int rc = pt_enqueue_one(c, ch, do_exchange, thin, save_offset, c->stream);
        if (rc != CARMA_OK) return rc;
        e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return hip_fail(e, "pt kernel");
        bool aborted = false;
        rc = pt_check_abort(c, &aborted);
        if (rc != CARMA_OK) return rc;
        if (aborted) {
            e = pt_backup(c, true, c->stream);
            if (e == hipSuccess) e = hipMemsetAsync(s->d_abort, 0, sizeof(unsigned), c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) return hip_fail(e, "pt state restore");
            s->iter = iter_before;
            s->stat_iters = stat_before;
            if (save_offset) *save_offset = off_before;
            s->use_row = false;           // acceptance / swap counters of the aborted chunk stay counted: statistics only
            continue;
        }
        left -= ch;
    }
    return CARMA_OK;
}

}  // namespace carma

using namespace carma;

extern "C" {

int carma_pt_create(carma_ctx* h, int ntemps, int nreplicas, const double* temperatures, int adapt_iters, uint64_t seed)
{
    if (!h || ntemps < 1 || nreplicas < 1 || adapt_iters < 0) {
        set_error("carma_pt_create: bad argument");
        return CARMA_EINVAL;
    }
    Ctx* c = reinterpret_cast<Ctx*>(h);
    int nthr = 0;
    const size_t lds = pt_lds_bytes(c->p, c->d, ntemps, &nthr, nullptr);
    if (nthr > 1024 || lds > 160 * 1024) {
        set_error("carma_pt_create: %d temperatures do not fit one workgroup (threads %d, LDS %zu B)", ntemps, nthr, lds);
        return CARMA_EINVAL;
    }
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    pt_state_free(c);
    PtState* s = new PtState();
    c->pt = s;
    s->T = ntemps;
    s->R = nreplicas;
    s->T_global = ntemps;
    s->maxiter = adapt_iters;
    s->seed = seed;
    s->temps.resize(ntemps);
    for (int i = 0; i < ntemps; i++) {
        // src/carmcmc.cpp:92-95: exp(linspace(0, ln 100, nwalkers))
        if (temperatures)
            s->temps[i] = temperatures[i];
        else
            s->temps[i] = (ntemps == 1) ? 1.0 : std::exp(std::log(100.0) * (double)i / (double)(ntemps - 1));
    }
    const int d = c->d;
    const size_t nchain = (size_t)ntemps * nreplicas;
    // initial proposal covariance (src/carmcmc.cpp:132-136 / :50-54): diag(1e-4), [0,0]=2 var^2/n, [2,2]=var/n
    const double var = pop_var(c->y);
    std::vector<double> R0((size_t)d * d, 0.0);
    for (int i = 0; i < d; i++) R0[(size_t)i * d + i] = 0.01;
    R0[0] = std::sqrt(2.0 * var * var / c->n);
    R0[(size_t)2 * d + 2] = std::sqrt(var / c->n);
    std::vector<double> chol(nchain * d * d);
    for (size_t k = 0; k < nchain; k++) std::memcpy(&chol[k * d * d], R0.data(), sizeof(double) * d * d);
    e = dev_malloc(&s->d_temps, sizeof(double) * ntemps);
    if (e == hipSuccess) e = dev_malloc(&s->d_theta, sizeof(double) * nchain * d);
    if (e == hipSuccess) e = dev_malloc(&s->d_lp, sizeof(double) * nchain);
    if (e == hipSuccess) e = dev_malloc(&s->d_chol, sizeof(double) * nchain * d * d);
    if (e == hipSuccess) e = dev_malloc(&s->d_nacc, sizeof(unsigned) * nchain);
    if (e == hipSuccess) e = dev_malloc(&s->d_nswap, sizeof(unsigned) * nchain);
    if (e == hipSuccess) e = hipMemcpy(s->d_temps, s->temps.data(), sizeof(double) * ntemps, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->d_chol, chol.data(), sizeof(double) * chol.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(s->d_nacc, 0, sizeof(unsigned) * nchain);
    if (e == hipSuccess) e = hipMemset(s->d_nswap, 0, sizeof(unsigned) * nchain);
    // Kernel choice.  The row variant needs every workgroup of the grid resident at the same time
    // (its swap step is a cross-workgroup rendezvous); otherwise one workgroup per ladder (k_pt).
    // CARMA_PT_KERNEL=ladder|row overrides (row only where it is safe).
    // Large ensembles: ONE CHAIN PER LANE, an iteration as the batched log-density launch + a bookkeeping kernel
    // (carma_pt_lane.hip).  Which path pays is measured, per order (tools/mcmc_lane_probe.py, profiles/r04/
    // mcmc_lane_threshold_v*.txt; 16 temperatures, n = 270):
    //   * the ladder kernel k_pt gives a ladder ceil(T G / 64) waves (G = 2 / 4 / 8 lanes per chain for p = 2 / 3-4 / 5-7); its
    //     iteration takes twice as long once those are more than the chip has SIMDs -- p = 5: 8193 chains 245 us against 127,
    //     p = 3: 32 768 chains 207 against 98 -- so from there on: one chain per lane, whatever the order;
    //   * p <= 4, whose batched launch is the producer-wave kernel right above the wave pipeline's range (carma_kernels.hip,
    //     lpc_min_evals): from 16 x #CUs chains (p = 3: 78 us flat up to 16 384 chains against 103-105, p = 4: 88-107 against
    //     116; at 3200 ... 4096 chains the ladder kernel's 74 / 79 us are still ahead of 78 / 83), p = 2 from 12 x #CUs (64 us
    //     flat against 73-96); p = 5 from 16 x #CUs as well since round 5 (below).
    // CARMA_PT_KERNEL=lane forces it (T <= 64), CARMA_TUNE_PT_LANE_MIN = N replaces the table by "from N chains".
    const char* force = getenv("CARMA_PT_KERNEL");
    if (e == hipSuccess && ntemps <= 64) {
        const char* tv = getenv("CARMA_TUNE_PT_LANE_MIN");
        const long cus = device_cus();
        const int G = c->p == 2 ? 2 : (c->p <= 4 ? 4 : 8);
        const long ladder_waves = (long)nreplicas * ((ntemps * G + 63) / 64);
        bool pays = ladder_waves > 4 * cus;
        // CAR(1): k_pt gives a chain ONE lane for the whole series (44 us per iteration at n = 270 whatever the ensemble), the batched
        // launch cuts the series across a wave (k_logdens_car1_scan: 8 us for 1024 chains) -- 18.6 against 43.8 us per iteration
        // at 16 x 64, 24.8 against 44.1 at 16 x 256, 59.7 against 73.9 at 16 x 2048; only where the parallel-in-time launch has
        // just run out of waves (48 ... 64 x #CUs chains) the ladder kernel is ahead, 46.9 against 51.4 (car1_sampler_v1.txt)
        // (evidence: n = 270, ensembles from 16 x 64.  Below 64 data the batched launch has no parallel-in-time scan -- one
        // evaluation per lane, launch_logdens_car1 -- and a small ladder such as run_mcmc_car1's default ~10 chains would pay 2-3
        // launches per iteration with nothing parallel to hide them: those keep the persistent k_pt.)
        if (c->p == 1) pays = c->n >= 64 && (long)nchain >= 64 && !((long)nchain > 48 * cus && (long)nchain <= 64 * cus);
        // (p = 3 from 12 x #CUs as well since round 5: 70 us against the ladder kernel's 72.6 at 3200 / 4096 chains; p = 4: 83 against 75,
        // p = 6, 7 at 4608 ... 8192 chains: 150 / 172 against 130 / 138 -- mcmc_lane_threshold_v3.txt)
        if (c->p == 2 || c->p == 3) pays = pays || (long)nchain > 12 * cus;
        if (c->p == 4) pays = pays || (long)nchain > 16 * cus;
        // p = 5 (round 5): its batched launch is the producer-wave kernel from 4 097 evaluations since the consumer takes a ring buffer
        // at a time (89 us) -- 111-112 us per iteration flat for 4 352 ... 8 192 chains against the ladder kernel's 122-125 (which does
        // 80 us up to 4 096 chains, against 91: profiles/r05/mcmc_lane_threshold_v1.txt, _v2.txt)
        if (c->p == 5) pays = pays || (long)nchain > 16 * cus;
        if (tv) pays = (long)nchain >= atol(tv);
        const bool forced = force && std::strcmp(force, "lane") == 0;
        if (forced || (!force && pays)) {
            s->use_lane = true;
            e = dev_malloc(&s->d_lane_scratch, sizeof(double) * pt_lane_scratch_doubles(d, (long)nchain));
        }
    }
    if (e == hipSuccess && c->p >= 2 && !s->use_lane) {
        const int wpl = (ntemps + 3) / 4;
        const long cap = pt_row_capacity(c->p, d, ntemps, c->n);
        const bool want = !(force && std::strcmp(force, "ladder") == 0);
        if (want && cap >= (long)nreplicas * wpl) {
            s->use_row = true;
            s->wpl = wpl;
            // tagged staging: two buffers x two copies of (theta[d], log-posterior) per chain; zeroed once (an all-zero
            // pair of words never validates)
            e = dev_malloc(&s->d_stage, sizeof(unsigned long long) * 4 * nchain * (d + 1));
            if (e == hipSuccess) e = hipMemset(s->d_stage, 0, sizeof(unsigned long long) * 4 * nchain * (d + 1));
            if (e == hipSuccess) e = dev_malloc(&s->d_abort, sizeof(unsigned));
            if (e == hipSuccess) e = hipMemset(s->d_abort, 0, sizeof(unsigned));
            if (e == hipSuccess) e = dev_malloc(&s->d_backup, sizeof(double) * nchain * (d + 1 + (size_t)d * d));
        }
    }
    if (e != hipSuccess) {
        int rc = hip_fail(e, "carma_pt_create");
        pt_state_free(c);
        return rc;
    }
    return CARMA_OK;
}

int carma_pt_shard(carma_ctx* h, int ntemps_global, int slot0, int replica0)
{
    if (!h || !reinterpret_cast<Ctx*>(h)->pt) return CARMA_EINVAL;
    PtState* s = reinterpret_cast<Ctx*>(h)->pt;
    if (ntemps_global < s->T || slot0 < 0 || slot0 + s->T > ntemps_global || replica0 < 0) {
        set_error("carma_pt_shard: bad shard");
        return CARMA_EINVAL;
    }
    s->T_global = (unsigned)ntemps_global;
    s->slot0 = (unsigned)slot0;
    s->replica0 = (unsigned)replica0;
    return CARMA_OK;
}

int carma_pt_bind_state(carma_ctx* h, double* d_theta, double* d_logpost)
{
    if (!h || !reinterpret_cast<Ctx*>(h)->pt || !d_theta || !d_logpost) return CARMA_EINVAL;
    Ctx* c = reinterpret_cast<Ctx*>(h);
    PtState* s = c->pt;
    const size_t nchain = (size_t)s->T * s->R;
    hipError_t e = hipMemcpy(d_theta, s->d_theta, sizeof(double) * nchain * c->d, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_logpost, s->d_lp, sizeof(double) * nchain, hipMemcpyDeviceToDevice);
    if (e != hipSuccess) return hip_fail(e, "carma_pt_bind_state");
    if (!s->ext_state) {
        (void)dev_free(s->d_theta);
        (void)dev_free(s->d_lp);
    }
    s->d_theta = d_theta;
    s->d_lp = d_logpost;
    s->ext_state = true;
    return CARMA_OK;
}

// Log-posteriors of chain states (starting values, carma_pt_set_chains without them) in launches of at most START_BATCH evaluations:
// which kernel a launch takes depends on its size (and, for the two-sided kernels on series of the SMALL class, on whether it leaves
// a CU to every workgroup), and a block of a sharded ladder or a rank's share of the replicas holds fewer chains than the
// single-process run -- its log-posteriors would come from another launch shape, a rounding apart, and the chains would no longer be
// the one-GPU run's bit for bit.  Below this size the choice does not depend on the count.
static int logdensity_of_chain_states(carma_ctx* h, const double* theta, size_t n, int d, double* out)
{
    constexpr size_t START_BATCH = 256;
    for (size_t i0 = 0; i0 < n; i0 += START_BATCH) {
        const size_t nb = std::min(START_BATCH, n - i0);
        const int rc = carma_logdensity_batch(h, theta + i0 * d, (int)nb, 0, out + i0);
        if (rc != CARMA_OK) return rc;
    }
    return CARMA_OK;
}

int carma_pt_set_chains(carma_ctx* h, const double* theta, const double* logpost)
{
    if (!h || !reinterpret_cast<Ctx*>(h)->pt || !theta) return CARMA_EINVAL;
    Ctx* c = reinterpret_cast<Ctx*>(h);
    PtState* s = c->pt;
    const size_t nchain = (size_t)s->T * s->R;
    std::vector<double> lp(nchain);
    if (logpost) {
        std::memcpy(lp.data(), logpost, sizeof(double) * nchain);
    } else {
        int rc = logdensity_of_chain_states(h, theta, nchain, c->d, lp.data());
        if (rc != CARMA_OK) return rc;
    }
    hipError_t e = hipMemcpy(s->d_theta, theta, sizeof(double) * nchain * c->d, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->d_lp, lp.data(), sizeof(double) * nchain, hipMemcpyHostToDevice);
    // a new set of chains starts a new history of boundary decisions: the two sides of a boundary compare folds of the
    // decisions since this point (carma_shard.hip), so one side re-created on its own must not inherit the old sum
    if (e == hipSuccess && s->d_checksum) e = hipMemset(s->d_checksum, 0, 4 * sizeof(unsigned long long));
    if (e != hipSuccess) return hip_fail(e, "carma_pt_set_chains");
    s->bnd_check = 0;
    s->started = true;
    return CARMA_OK;
}

int carma_pt_get_chains(carma_ctx* h, double* theta, double* logpost)
{
    if (!h || !reinterpret_cast<Ctx*>(h)->pt) return CARMA_EINVAL;
    Ctx* c = reinterpret_cast<Ctx*>(h);
    PtState* s = c->pt;
    const size_t nchain = (size_t)s->T * s->R;
    hipError_t e = hipSuccess;
    if (theta) e = hipMemcpy(theta, s->d_theta, sizeof(double) * nchain * c->d, hipMemcpyDeviceToHost);
    if (e == hipSuccess && logpost) e = hipMemcpy(logpost, s->d_lp, sizeof(double) * nchain, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "carma_pt_get_chains");
    return CARMA_OK;
}

int carma_pt_start(carma_ctx* h, const double* init, int ninit)
{
    if (!h || !reinterpret_cast<Ctx*>(h)->pt) {
        set_error("carma_pt_start: call carma_pt_create first");
        return CARMA_EINVAL;
    }
    Ctx* c = reinterpret_cast<Ctx*>(h);
    PtState* s = c->pt;
    const int d = c->d;
    const size_t nchain = (size_t)s->T * s->R;
    std::vector<double> theta(nchain * d), lp(nchain, -std::numeric_limits<double>::infinity());
    std::vector<char> done(nchain, 0);
    // user-provided start is honoured only if its length is d and its posterior is finite
    // (src/samplers.cpp:75-93, src/carpack.cpp:479-490)
    if (init && ninit == d) {
        double l0 = 0;
        int rc = carma_logdensity_batch(h, init, 1, 0, &l0);
        if (rc != CARMA_OK) return rc;
        if (std::isfinite(l0)) {
            for (size_t k = 0; k < nchain; k++) {
                std::memcpy(&theta[k * d], init, sizeof(double) * d);
                lp[k] = l0;
                done[k] = 1;
            }
        }
    }
    // The draws of a chain are keyed by (seed, the chain's GLOBAL slot, attempt), not by the order in which a process
    // happens to visit its chains: a replica or a temperature gets the same starting value whichever rank holds it
    // (carma_pt_shard), so a sharded run starts -- and, the sampler's streams being keyed the same way, continues --
    // exactly as the single-process run does.
    auto chain_rng = [&](size_t k, int round) {
        const uint64_t gslot = ((uint64_t)s->replica0 + k / (size_t)s->T) * (uint64_t)s->T_global + s->slot0 + k % (size_t)s->T;
        uint64_t z = s->seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
        z ^= (gslot + 1) * 0xBF58476D1CE4E5B9ull;
        z ^= ((uint64_t)round + 1) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return std::mt19937_64(z * 0xD6E8FEB86659FD93ull + 0x2545F4914F6CDD1Dull);
    };
    for (int round = 0; round < 4000; round++) {
        std::vector<size_t> todo;
        for (size_t k = 0; k < nchain; k++)
            if (!done[k]) todo.push_back(k);
        if (todo.empty()) break;
        std::vector<double> cand(todo.size() * d), out(todo.size());
        for (size_t i = 0; i < todo.size(); i++) {
            std::mt19937_64 rng = chain_rng(todo[i], round);
            draw_start(c, rng, &cand[i * d]);
        }
        int rc = logdensity_of_chain_states(h, cand.data(), todo.size(), d, out.data());
        if (rc != CARMA_OK) return rc;
        for (size_t i = 0; i < todo.size(); i++) {
            if (std::isfinite(out[i])) {
                std::memcpy(&theta[todo[i] * d], &cand[i * d], sizeof(double) * d);
                lp[todo[i]] = out[i];
                done[todo[i]] = 1;
            }
        }
    }
    for (size_t k = 0; k < nchain; k++) {
        if (!done[k]) {
            set_error("carma_pt_start: no finite starting value found for chain %zu", k);
            return CARMA_EINVAL;
        }
    }
    return carma_pt_set_chains(h, theta.data(), lp.data());
}

int carma_pt_iterate(carma_ctx* h, long niter, int do_exchange)
{
    if (!h || !reinterpret_cast<Ctx*>(h)->pt || niter < 0) return CARMA_EINVAL;
    Ctx* c = reinterpret_cast<Ctx*>(h);
    if (!c->pt->started) {
        set_error("carma_pt_iterate: chains have no starting values");
        return CARMA_EINVAL;
    }
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    return pt_launch_chunks(c, niter, do_exchange, 0, nullptr);
}

int carma_pt_sample(carma_ctx* h, int nsamples, int thin, double* samples, double* logposts)
{
    if (!h || !reinterpret_cast<Ctx*>(h)->pt || nsamples < 1 || thin < 1 || !samples || !logposts) {
        set_error("carma_pt_sample: bad argument");
        return CARMA_EINVAL;
    }
    Ctx* c = reinterpret_cast<Ctx*>(h);
    PtState* s = c->pt;
    if (!s->started) {
        set_error("carma_pt_sample: chains have no starting values");
        return CARMA_EINVAL;
    }
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    if (s->cap < nsamples) {
        if (s->d_samples) (void)dev_free(s->d_samples);
        if (s->d_slp) (void)dev_free(s->d_slp);
        s->d_samples = s->d_slp = nullptr;
        s->cap = 0;
        e = dev_malloc(&s->d_samples, sizeof(double) * (size_t)s->R * nsamples * c->d);
        if (e == hipSuccess) e = dev_malloc(&s->d_slp, sizeof(double) * (size_t)s->R * nsamples);
        if (e != hipSuccess) return hip_fail(e, "dev_malloc(samples)");
        s->cap = nsamples;
    }
    long off = 0;
    const long cap_saved = s->cap;
    s->cap = nsamples;   // stride of this call's output
    int rc = pt_launch_chunks(c, (long)nsamples * thin, 1, thin, &off);
    if (rc == CARMA_OK) {
        e = hipMemcpy(samples, s->d_samples, sizeof(double) * (size_t)s->R * nsamples * c->d, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(logposts, s->d_slp, sizeof(double) * (size_t)s->R * nsamples, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(e, "D2H samples");
    }
    s->cap = cap_saved;
    return rc;
}

int carma_pt_stats(carma_ctx* h, double* accept_rate, double* swap_rate, int reset)
{
    if (!h || !reinterpret_cast<Ctx*>(h)->pt) return CARMA_EINVAL;
    Ctx* c = reinterpret_cast<Ctx*>(h);
    PtState* s = c->pt;
    const size_t nchain = (size_t)s->T * s->R;
    std::vector<unsigned> a(nchain), w(nchain);
    hipError_t e = hipMemcpy(a.data(), s->d_nacc, sizeof(unsigned) * nchain, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(w.data(), s->d_nswap, sizeof(unsigned) * nchain, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "carma_pt_stats");
    const double it = s->stat_iters ? (double)s->stat_iters : 1.0;
    for (size_t k = 0; k < nchain; k++) {
        if (accept_rate) accept_rate[k] = a[k] / it;
        if (swap_rate) swap_rate[k] = w[k] / it;   // entry i = swaps between temperature i and i-1
    }
    if (reset) {
        (void)hipMemset(s->d_nacc, 0, sizeof(unsigned) * nchain);
        (void)hipMemset(s->d_nswap, 0, sizeof(unsigned) * nchain);
        s->stat_iters = 0;
    }
    return CARMA_OK;
}

int carma_pt_kernel_in_use(const carma_ctx* h)
{
    if (!h || !reinterpret_cast<const Ctx*>(h)->pt) return CARMA_EINVAL;
    const PtState* s = reinterpret_cast<const Ctx*>(h)->pt;
    return s->use_row ? 1 : (s->use_lane ? 2 : 0);
}

int carma_pt_row_pipeline(void) { return pt_row_last_pipeline(); }

long carma_pt_iterations_done(const carma_ctx* h)
{
    if (!h || !reinterpret_cast<const Ctx*>(h)->pt) return CARMA_EINVAL;
    return (long)reinterpret_cast<const Ctx*>(h)->pt->iter;
}

int carma_pt_run(carma_ctx* h, int ntemps, int nreplicas, int sample_size, int burnin, int thin, const double* init,
                 int ninit, uint64_t seed, double* samples, double* logposts)
{
    int rc = carma_pt_create(h, ntemps, nreplicas, nullptr, burnin, seed);
    if (rc == CARMA_OK) rc = carma_pt_start(h, init, ninit);
    if (rc == CARMA_OK) rc = carma_pt_iterate(h, burnin, 1);                    // Sampler::Run burn-in (samplers.cpp:97)
    if (rc == CARMA_OK) rc = carma_pt_sample(h, sample_size, thin, samples, logposts);   // :101-108
    return rc;
}

}  // extern "C"
