// carma_capi.hip -- C ABI of libcarma_mi355.so (see include/carma_mi355.h for the contract and
// the reference interface each entry point replaces).  Host side only: argument checking, the
// series preparation of KalmanFilter::init, HBM residency, launches.  No CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

#include "../../include/carma_mi355.h"
#include "carma_host.h"

// ---- device allocations (carma_host.h) -------------------------------------------------------------------------
#include <map>
#include <mutex>
namespace {
struct GuardAlloc {
    hipMemGenericAllocationHandle_t handle;
    void* base;
    size_t reserved, mapped;
};
std::mutex g_guard_mu;
std::map<void*, GuardAlloc> g_guard;      // user pointer -> mapping
bool guard_mode()
{
    static const bool on = [] {
        const char* e = getenv("CARMA_DEBUG_GUARD");
        return e && e[0] == '1';
    }();
    return on;
}
}  // namespace
hipError_t carma_dev_malloc(void** p, size_t n)
{
    if (!guard_mode()) return ::hipMalloc(p, n);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    if (gran == 0) gran = 2u << 20;
    const size_t need = n ? n : 1;
    GuardAlloc a{};
    a.mapped = (need + gran - 1) / gran * gran;
    a.reserved = a.mapped + gran;                             // one granule of address space behind the buffer stays unmapped
    e = hipMemAddressReserve(&a.base, a.reserved, gran, nullptr, 0);
    if (e != hipSuccess) return e;
    e = hipMemCreate(&a.handle, a.mapped, &prop, 0);
    if (e == hipSuccess) e = hipMemMap(a.base, a.mapped, 0, a.handle, 0);
    if (e == hipSuccess) {
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(a.base, a.mapped, &acc, 1);
    }
    if (e != hipSuccess) {
        (void)hipMemAddressFree(a.base, a.reserved);
        return e;
    }
    // the buffer ENDS where the mapping ends (16-byte alignment: up to 15 bytes of slack behind odd sizes)
    const size_t user = (need + 15) / 16 * 16;
    *p = static_cast<char*>(a.base) + (a.mapped - user);
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guard[*p] = a;
    return hipSuccess;
}
hipError_t carma_dev_free(void* p)
{
    if (!p) return hipSuccess;
    if (!guard_mode()) return ::hipFree(p);
    GuardAlloc a{};
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        auto it = g_guard.find(p);
        if (it == g_guard.end()) return hipErrorInvalidValue;
        a = it->second;
        g_guard.erase(it);
    }
    (void)hipDeviceSynchronize();                             // (hipFree's implicit synchronisation)
    hipError_t e = hipMemUnmap(a.base, a.mapped);
    if (e == hipSuccess) e = hipMemRelease(a.handle);
    // (the address range stays reserved for the life of the process: an address is never handed out twice, so a pointer
    // used after its buffer was freed faults as well)
    return e;
}

namespace carma {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what)
{
    set_error("%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver) ? CARMA_ENODEV
           : (e == hipErrorOutOfMemory)                                                             ? CARMA_ENOMEM
                                                                                                    : CARMA_EHIP;
}

// KalmanFilter::init (src/include/kfilter.hpp:43-76): sort by time when any dt < 0, then keep
// sample 0 and every sample whose dt to its predecessor in the sorted series is non-zero.
void sort_dedup(std::vector<double>& t, std::vector<double>& y, std::vector<double>& e)
{
    const size_t n = t.size();
    bool need_sort = false;
    for (size_t i = 1; i < n; i++) need_sort |= (t[i] - t[i - 1] < 0);
    if (need_sort) {
        std::vector<size_t> idx(n);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return t[a] < t[b]; });
        std::vector<double> t2(n), y2(n), e2(n);
        for (size_t i = 0; i < n; i++) {
            t2[i] = t[idx[i]];
            y2[i] = y[idx[i]];
            e2[i] = e[idx[i]];
        }
        t.swap(t2);
        y.swap(y2);
        e.swap(e2);
    }
    std::vector<char> keep(n, 1);
    bool dup = false;
    for (size_t i = 1; i < n; i++) {
        if (t[i] - t[i - 1] == 0) {
            keep[i] = 0;
            dup = true;
        }
    }
    if (dup) {
        size_t m = 0;
        for (size_t i = 0; i < n; i++) {
            if (keep[i]) {
                t[m] = t[i];
                y[m] = y[i];
                e[m] = e[i];
                m++;
            }
        }
        t.resize(m);
        y.resize(m);
        e.resize(m);
    }
}

// series records {dt_k, y_k, yerr_k^2, t_k}
std::vector<double> pack_series(const std::vector<double>& t, const std::vector<double>& y, const std::vector<double>& e)
{
    const size_t n = t.size(), nt = n + P3L_PAD_RECORDS;
    std::vector<double> s(6 * nt);
    for (size_t k = 0; k < n; k++) {
        s[4 * k + 0] = k ? t[k] - t[k - 1] : 0.0;
        s[4 * k + 1] = y[k];
        s[4 * k + 2] = e[k] * e[k];
        s[4 * k + 3] = t[k];
    }
    for (size_t k = n; k < n + P3L_PAD_RECORDS; k++) {        // neutral pad records (carma_types.h, p3l_pad)
        s[4 * k + 0] = 0.0;
        s[4 * k + 1] = n ? y[n - 1] : 0.0;
        s[4 * k + 2] = 1.0;
        s[4 * k + 3] = n ? t[n - 1] : 0.0;
    }
    // behind the records: yerr^2 and y once more as plain arrays -- the recursion waves of the pipeline fetch sixteen
    // of them per chunk with scalar loads, and contiguous doubles come in two wide loads instead of sixteen
    for (size_t k = 0; k < nt; k++) {
        s[4 * nt + k] = s[4 * k + 2];
        s[5 * nt + k] = s[4 * k + 1];
    }
    return s;
}

// AR roots as the kernels expect them: complex-conjugate pairs adjacent (negative imaginary part first), real roots
// after them -- the order CARp::ARRoots emits (src/carpack.cpp:137-172).  The result of the filter does not depend on
// the order of the roots, so roots handed over in another order (carma_pack.py's get_ar_roots puts a real root wherever
// its centroid is zero) are re-ordered here; a set that is not closed under conjugation is not a real-valued process
// and is rejected.  out = p (re, im) pairs.
int normalize_roots(int p, const double* om, double* out)
{
    std::vector<int> used(p, 0);
    int k = 0;
    for (int i = 0; i < p; i++) {
        if (used[i] || om[2 * i + 1] == 0.0) continue;
        const double re = om[2 * i], im = om[2 * i + 1], tol = 1e-12 * std::hypot(re, im);
        int mate = -1;
        for (int j = i + 1; j < p && mate < 0; j++)
            if (!used[j] && std::fabs(om[2 * j] - re) <= tol && std::fabs(om[2 * j + 1] + im) <= tol) mate = j;
        if (mate < 0) return CARMA_EINVAL;
        used[i] = used[mate] = 1;
        out[2 * k] = out[2 * k + 2] = re;
        out[2 * k + 1] = -std::fabs(im);
        out[2 * k + 3] = std::fabs(im);
        k += 2;
    }
    for (int i = 0; i < p; i++) {
        if (used[i]) continue;
        out[2 * k] = om[2 * i];
        out[2 * k + 1] = 0.0;
        k++;
    }
    return CARMA_OK;
}

int select_device(int device)
{
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device visible (this library has no CPU fallback)");
        return CARMA_ENODEV;
    }
    if (device < 0 || device >= cnt) {
        set_error("device %d out of range (have %d)", device, cnt);
        return CARMA_EINVAL;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    return CARMA_OK;
}

int Ctx::ensure_staging(int B)
{
    if (B <= cap) return CARMA_OK;
    if (d_theta) (void)dev_free(d_theta);
    if (d_out) (void)dev_free(d_out);
    if (h_stage) (void)hipHostFree(h_stage);
    d_theta = d_out = h_stage = nullptr;
    cap = 0;
    int newcap = std::max(B, 1024);
    hipError_t e = dev_malloc(&d_theta, sizeof(double) * (size_t)newcap * d);
    if (e != hipSuccess) return hip_fail(e, "dev_malloc(theta)");
    e = dev_malloc(&d_out, sizeof(double) * (size_t)newcap);
    if (e != hipSuccess) return hip_fail(e, "dev_malloc(out)");
    // pinned, so that both copies are real asynchronous DMA transfers ordered on the stream (a copy from pageable memory
    // is staged by the runtime and synchronises)
    e = hipHostMalloc(reinterpret_cast<void**>(&h_stage), sizeof(double) * (size_t)newcap * (d + 1), hipHostMallocDefault);
    if (e != hipSuccess) return hip_fail(e, "hipHostMalloc(staging)");
    cap = newcap;
    return CARMA_OK;
}

}  // namespace carma

using namespace carma;

extern "C" {

const char* carma_version(void) { return "carma_mi355 0.1 (gfx950)"; }
const char* carma_last_error(void) { return g_err; }

int carma_device_count(void)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return cnt;
}

carma_ctx* carma_ctx_create(const double* time, const double* y, const double* yerr, int n, int p, int q,
                            double max_stdev, int device)
{
    if (!time || !y || !yerr || n < 2) {
        set_error("carma_ctx_create: need n >= 2 and non-null arrays");
        return nullptr;
    }
    if (p < 1 || p > CARMA_PMAX || q < 0 || (p == 1 && q != 0) || (p > 1 && q >= p)) {
        // BOOST_ASSERT_MSG(q < p, ...) src/include/carpack.hpp:377
        set_error("carma_ctx_create: need 1 <= p <= %d and q < p (got p=%d q=%d)", CARMA_PMAX, p, q);
        return nullptr;
    }
    if (select_device(device) != CARMA_OK) return nullptr;
    Ctx* c = new Ctx();
    c->device = device;
    c->p = p;
    c->q = q;
    c->d = (p == 1) ? 4 : 3 + p + q;
    c->t.assign(time, time + n);
    c->y.assign(y, y + n);
    c->yerr.assign(yerr, yerr + n);
    sort_dedup(c->t, c->y, c->yerr);
    c->n = (int)c->t.size();
    if (c->n < 2) {
        set_error("carma_ctx_create: fewer than 2 distinct times");
        delete c;
        return nullptr;
    }
    c->pr.measerr_dof = 50.0;   // src/include/carpack.hpp:63
    carma_ctx_set_prior(reinterpret_cast<carma_ctx*>(c), max_stdev);
    std::vector<double> s = pack_series(c->t, c->y, c->yerr);
    {
        int rep = 0;
        for (int k = 2; k < c->n; k++) rep += (s[4 * (size_t)k] == s[4 * (size_t)(k - 1)]);
        c->repeated_dt = c->n > 8 && 4 * rep >= c->n;
    }
    if (p >= 2) {
        // SERIES_WINDOW_OK (carma_types.h): spans of 16 - p consecutive data against the shortest window the prior admits
        const int ND = 16 - p;
        const double wmin = 0.5 * 600.0 / (6.283185307179586 * c->pr.max_freq);       // Pipe3LGeom::LIM_RE; the grid halves it at worst
        long over = 0, tot = 0;
        for (int k = 0; k + ND - 1 < c->n; k++, tot++) over += (c->t[k + ND - 1] - c->t[k]) > wmin;
        c->window_ok = tot > 0 && 10 * over <= tot;
        // SERIES_WINDOW2_OK / _SMALL: chunks of the row with the shortest window against ceil(n / ND)
        long chunks = 0;
        for (int k = 0; k < c->n; chunks++) {
            int j = k;
            while (j + 1 < c->n && j + 1 - k < ND && c->t[j + 1] - c->t[k] <= wmin) j++;
            k = j + 1;
        }
        const double r = (double)chunks / (double)((c->n + ND - 1) / ND);
        c->window2 = r <= 2.0 ? 2 : (r <= 3.5 ? 1 : 0);
    }
    hipError_t e = dev_malloc(&c->d_series, sizeof(double) * s.size());
    if (e == hipSuccess) e = hipMemcpy(c->d_series, s.data(), sizeof(double) * s.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        hip_fail(e, "carma_ctx_create");
        carma_ctx_destroy(reinterpret_cast<carma_ctx*>(c));
        return nullptr;
    }
    return reinterpret_cast<carma_ctx*>(c);
}

void carma_ctx_destroy(carma_ctx* h)
{
    if (!h) return;
    Ctx* c = reinterpret_cast<Ctx*>(h);
    (void)hipSetDevice(c->device);
    if (c->pt) pt_state_free(c);
    if (c->d_series) (void)dev_free(c->d_series);
    if (c->d_theta) (void)dev_free(c->d_theta);
    if (c->d_out) (void)dev_free(c->d_out);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int carma_ctx_n(const carma_ctx* h) { return h ? reinterpret_cast<const Ctx*>(h)->n : CARMA_EINVAL; }
int carma_ctx_dim(const carma_ctx* h) { return h ? reinterpret_cast<const Ctx*>(h)->d : CARMA_EINVAL; }

int carma_ctx_get_data(const carma_ctx* h, double* time, double* y, double* yerr)
{
    if (!h) return CARMA_EINVAL;
    const Ctx* c = reinterpret_cast<const Ctx*>(h);
    if (time) std::memcpy(time, c->t.data(), sizeof(double) * c->n);
    if (y) std::memcpy(y, c->y.data(), sizeof(double) * c->n);
    if (yerr) std::memcpy(yerr, c->yerr.data(), sizeof(double) * c->n);
    return CARMA_OK;
}

int carma_ctx_get_prior(const carma_ctx* h, double* out3)
{
    if (!h || !out3) return CARMA_EINVAL;
    const Ctx* c = reinterpret_cast<const Ctx*>(h);
    out3[0] = c->pr.max_stdev;
    out3[1] = c->pr.max_freq;
    out3[2] = c->pr.min_freq;
    return CARMA_OK;
}

int carma_ctx_set_prior(carma_ctx* h, double max_stdev)
{
    if (!h) return CARMA_EINVAL;
    Ctx* c = reinterpret_cast<Ctx*>(h);
    // SetPrior (src/include/carpack.hpp:201-207)
    c->pr.max_stdev = max_stdev;
    double dtmin = std::numeric_limits<double>::infinity();
    for (int i = 1; i < c->n; i++) dtmin = std::min(dtmin, c->t[i] - c->t[i - 1]);
    c->pr.max_freq = 1.0 / dtmin;
    c->pr.min_freq = 1.0 / (*std::max_element(c->t.begin(), c->t.end()) - *std::min_element(c->t.begin(), c->t.end()));
    return CARMA_OK;
}

int carma_logdensity_batch_dev(carma_ctx* h, const double* d_theta, int B, int ignore_prior, double* d_out, void* stream)
{
    if (!h || !d_theta || !d_out || B < 0) {
        set_error("carma_logdensity_batch_dev: bad argument");
        return CARMA_EINVAL;
    }
    if (B == 0) return CARMA_OK;
    Ctx* c = reinterpret_cast<Ctx*>(h);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t e;
    if (c->p == 1)
        e = launch_logdens_car1(d_theta, B, reinterpret_cast<const double4*>(c->d_series), c->n, c->pr, d_out, st);
    else
        e = launch_logdens_carma(c->p, d_theta, B, c->d, c->q, reinterpret_cast<const double4*>(c->d_series), c->n, c->pr,
                                 ignore_prior, d_out, st, c->series_flags());
    if (e != hipSuccess) return hip_fail(e, "launch logdensity");
    return CARMA_OK;
}

int carma_logdensity_batch(carma_ctx* h, const double* theta, int B, int ignore_prior, double* out)
{
    if (!h || !theta || !out || B < 0) {
        set_error("carma_logdensity_batch: bad argument");
        return CARMA_EINVAL;
    }
    if (B == 0) return CARMA_OK;
    Ctx* c = reinterpret_cast<Ctx*>(h);
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    int rc = c->ensure_staging(B);
    if (rc != CARMA_OK) return rc;
    // host -> pinned -> device, launch, device -> pinned: three stream-ordered operations and ONE synchronisation
    double* h_th = c->h_stage;
    double* h_out = c->h_stage + (size_t)c->cap * c->d;
    std::memcpy(h_th, theta, sizeof(double) * (size_t)B * c->d);
    if (B <= 4096) {
        // small batches: the kernel reads its parameter vectors from, and writes its results to, the pinned buffer
        // itself (device-visible host memory) -- a read of d doubles per evaluation over the link at the start of the
        // kernel instead of two copy operations around it
        rc = carma_logdensity_batch_dev(h, h_th, B, ignore_prior, h_out, c->stream);
        if (rc != CARMA_OK) return rc;
        e = hipStreamSynchronize(c->stream);              // (polling hipStreamQuery instead: 2 us slower, measured)
        if (e != hipSuccess) return hip_fail(e, "logdensity (pinned)");
        std::memcpy(out, h_out, sizeof(double) * (size_t)B);
        return CARMA_OK;
    }
    e = hipMemcpyAsync(c->d_theta, h_th, sizeof(double) * (size_t)B * c->d, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) return hip_fail(e, "H2D theta");
    rc = carma_logdensity_batch_dev(h, c->d_theta, B, ignore_prior, c->d_out, c->stream);
    if (rc != CARMA_OK) return rc;
    e = hipMemcpyAsync(h_out, c->d_out, sizeof(double) * (size_t)B, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return hip_fail(e, "D2H logdensity");
    std::memcpy(out, h_out, sizeof(double) * (size_t)B);
    return CARMA_OK;
}

int carma_logdensity_kernel_name(const carma_ctx* h, int B, char* buf, int len)
{
    if (!h || !buf || len < 1 || B < 1) return CARMA_EINVAL;
    const Ctx* c = reinterpret_cast<const Ctx*>(h);
    return logdens_kernel_name(c->p, B, c->n, buf, len, c->series_flags()) > 0 ? CARMA_OK : CARMA_EINVAL;
}

int carma_tune_set(const char* name, long value) { return tune_set(name, value) == 0 ? CARMA_OK : CARMA_EINVAL; }

int carma_normalize_roots(int p, const double* omega_re_im, double* out)
{
    if (p < 1 || p > CARMA_PMAX || !omega_re_im || !out) return CARMA_EINVAL;
    return normalize_roots(p, omega_re_im, out);
}

double carma_logprior(const carma_ctx* h, const double* theta)
{
    if (!h || !theta) return std::numeric_limits<double>::quiet_NaN();
    const Ctx* c = reinterpret_cast<const Ctx*>(h);
    // src/include/carpack.hpp:118-126
    const double s = theta[1];
    return -0.5 * c->pr.measerr_dof / s - (1.0 + c->pr.measerr_dof / 2.0) * std::log(s);
}

}  // extern "C"

namespace carma {

// KalmanFilter1 / KalmanFilterp object (kfilter.hpp:211-334): the sorted / deduplicated series and the model, resident
// in HBM for the life of the handle (the free functions below build one per call).
struct Kf {
    int device = 0, p = 0, n = 0;
    double sigsqr = 0.0, car1_omega = 0.0;
    double *d_series = nullptr, *d_par = nullptr, *d_io = nullptr;
    int* d_sing = nullptr;
    size_t io_cap = 0;
    hipStream_t stream = nullptr;
    int ensure_io(size_t nd)
    {
        if (nd <= io_cap) return CARMA_OK;
        if (d_io) (void)dev_free(d_io);
        d_io = nullptr;
        io_cap = 0;
        hipError_t e = dev_malloc(&d_io, sizeof(double) * nd);
        if (e != hipSuccess) return hip_fail(e, "carma_kf: hipMalloc");
        io_cap = nd;
        return CARMA_OK;
    }
};

static void kf_free(Kf* k)
{
    if (!k) return;
    (void)hipSetDevice(k->device);
    if (k->d_series) (void)dev_free(k->d_series);
    if (k->d_par) (void)dev_free(k->d_par);
    if (k->d_io) (void)dev_free(k->d_io);
    if (k->d_sing) (void)dev_free(k->d_sing);
    if (k->stream) (void)hipStreamDestroy(k->stream);
    delete k;
}

static Kf* kf_make(const double* time, const double* y, const double* yerr, int n, int p, double sigsqr,
                   const double* omega_re_im, const double* ma, int nma, double car1_omega, int device, const char* who,
                   int* rc_out = nullptr)
{
    int rc_local = CARMA_EINVAL;
    int& rc = rc_out ? *rc_out : rc_local;
    rc = CARMA_EINVAL;
    if (!time || !y || !yerr || n < 1) {
        set_error("%s: bad argument", who);
        return nullptr;
    }
    if (p != 1 && (p < 2 || p > CARMA_PMAX || !omega_re_im || !ma || nma < 1)) {
        set_error("%s: need 2 <= p <= %d, omega and ma", who, CARMA_PMAX);
        return nullptr;
    }
    rc = select_device(device);
    if (rc != CARMA_OK) return nullptr;
    rc = CARMA_EINVAL;
    std::vector<double> t(time, time + n), yy(y, y + n), ee(yerr, yerr + n);
    if (n >= 2) sort_dedup(t, yy, ee);
    std::vector<double> s = pack_series(t, yy, ee);
    std::vector<double> par(2 * CARMA_PMAX + CARMA_PMAX, 0.0);
    if (p > 1) {
        if (normalize_roots(p, omega_re_im, par.data()) != CARMA_OK) {
            set_error("%s: the AR roots must be real or come in complex-conjugate pairs", who);
            return nullptr;
        }
        for (int i = 0; i < p && i < nma; i++) par[2 * CARMA_PMAX + i] = ma[i];   // zero padded (kfilter.hpp:318-320)
    }
    Kf* k = new Kf();
    k->device = device;
    k->p = p;
    k->n = (int)t.size();
    k->sigsqr = sigsqr;
    k->car1_omega = car1_omega;
    hipError_t e = dev_malloc(&k->d_series, sizeof(double) * s.size());
    if (e == hipSuccess) e = dev_malloc(&k->d_par, sizeof(double) * par.size());
    if (e == hipSuccess) e = dev_malloc(&k->d_sing, sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(k->d_series, s.data(), sizeof(double) * s.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(k->d_par, par.data(), sizeof(double) * par.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&k->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        rc = hip_fail(e, who);
        kf_free(k);
        return nullptr;
    }
    rc = CARMA_OK;
    return k;
}

// Filter() + GetMean() / GetVar() (kfilter.hpp:126-132)
static int kf_filter(Kf* k, double* mean, double* var)
{
    if (!mean || !var) return CARMA_EINVAL;
    hipError_t e = hipSetDevice(k->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    const int m = k->n;
    int rc = k->ensure_io(2 * (size_t)m);
    if (rc != CARMA_OK) return rc;
    e = hipMemsetAsync(k->d_sing, 0, sizeof(int), k->stream);
    if (e == hipSuccess) {
        if (k->p == 1)
            e = launch_kfilter_car1(k->sigsqr, k->car1_omega, reinterpret_cast<const double4*>(k->d_series), m, k->d_io, k->d_io + m,
                                    k->stream);
        else
            e = launch_kfilter_carma(k->p, k->d_par, k->d_par + 2 * CARMA_PMAX, k->sigsqr,
                                     reinterpret_cast<const double4*>(k->d_series), m, k->d_io, k->d_io + m, k->d_sing, k->stream);
    }
    int sing = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(mean, k->d_io, sizeof(double) * m, hipMemcpyDeviceToHost, k->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(var, k->d_io + m, sizeof(double) * m, hipMemcpyDeviceToHost, k->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&sing, k->d_sing, sizeof(int), hipMemcpyDeviceToHost, k->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(k->stream);
    if (e != hipSuccess) {
        // copies enqueued before the failure may still be writing into the caller's buffers (and into `sing` on this
        // stack frame): drain the stream before anybody frees or leaves them
        (void)hipStreamSynchronize(k->stream);
        return hip_fail(e, "carma_kfilter");
    }
    return sing ? 1 : CARMA_OK;
}

// Predict for M times in one launch (kfilter.cpp:218-337, 72-135)
static int kf_predict(Kf* k, const double* tpred, int M, double* pmean, double* pvar)
{
    if (!tpred || !pmean || !pvar || M < 0) {
        set_error("carma_predict: bad argument");
        return CARMA_EINVAL;
    }
    if (M == 0) return CARMA_OK;
    hipError_t e = hipSetDevice(k->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    int rc = k->ensure_io(3 * (size_t)M);
    if (rc != CARMA_OK) return rc;
    double* d_io = k->d_io;
    e = hipMemcpyAsync(d_io, tpred, sizeof(double) * M, hipMemcpyHostToDevice, k->stream);
    if (e == hipSuccess) e = hipMemsetAsync(k->d_sing, 0, sizeof(int), k->stream);
    if (e == hipSuccess) {
        if (k->p == 1)
            e = launch_predict_car1(k->sigsqr, k->car1_omega, reinterpret_cast<const double4*>(k->d_series), k->n, d_io, M, d_io + M,
                                    d_io + 2 * (size_t)M, k->stream);
        else
            e = launch_predict_carma(k->p, k->d_par, k->d_par + 2 * CARMA_PMAX, k->sigsqr,
                                     reinterpret_cast<const double4*>(k->d_series), k->n, d_io, M, d_io + M, d_io + 2 * (size_t)M,
                                     k->d_sing, k->stream);
    }
    int sing = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(pmean, d_io + M, sizeof(double) * M, hipMemcpyDeviceToHost, k->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(pvar, d_io + 2 * (size_t)M, sizeof(double) * M, hipMemcpyDeviceToHost, k->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&sing, k->d_sing, sizeof(int), hipMemcpyDeviceToHost, k->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(k->stream);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(k->stream);               // see kf_filter
        return hip_fail(e, "carma_predict");
    }
    return sing ? 1 : CARMA_OK;
}

}  // namespace carma

extern "C" {

carma_kf* carma_kf_create_carma(const double* time, const double* y, const double* yerr, int n, int p, double sigsqr,
                                const double* omega_re_im, const double* ma, int nma, int device)
{
    if (p < 2) {
        set_error("carma_kf_create_carma: need 2 <= p <= %d", CARMA_PMAX);
        return nullptr;
    }
    return reinterpret_cast<carma_kf*>(kf_make(time, y, yerr, n, p, sigsqr, omega_re_im, ma, nma, 0.0, device, "carma_kf_create_carma"));
}

carma_kf* carma_kf_create_car1(const double* time, const double* y, const double* yerr, int n, double sigsqr, double omega, int device)
{
    return reinterpret_cast<carma_kf*>(kf_make(time, y, yerr, n, 1, sigsqr, nullptr, nullptr, 0, omega, device, "carma_kf_create_car1"));
}

void carma_kf_destroy(carma_kf* h) { kf_free(reinterpret_cast<Kf*>(h)); }
int carma_kf_n(const carma_kf* h) { return h ? reinterpret_cast<const Kf*>(h)->n : CARMA_EINVAL; }
int carma_kf_filter(carma_kf* h, double* mean, double* var) { return h ? kf_filter(reinterpret_cast<Kf*>(h), mean, var) : CARMA_EINVAL; }
int carma_kf_predict(carma_kf* h, const double* tpred, int M, double* pmean, double* pvar)
{
    return h ? kf_predict(reinterpret_cast<Kf*>(h), tpred, M, pmean, pvar) : CARMA_EINVAL;
}

int carma_kfilter_carma(const double* time, const double* y, const double* yerr, int n, int p, double sigsqr,
                        const double* omega_re_im, const double* ma, int nma, double* mean, double* var, int* n_out,
                        int device)
{
    int rc0 = CARMA_EINVAL;
    Kf* k = (p >= 2) ? kf_make(time, y, yerr, n, p, sigsqr, omega_re_im, ma, nma, 0.0, device, "carma_kfilter_carma", &rc0) : nullptr;
    if (!k) {
        if (p < 2) set_error("carma_kfilter_carma: need 2 <= p <= %d, omega and ma", CARMA_PMAX);
        return rc0;
    }
    if (n_out) *n_out = k->n;
    const int rc = kf_filter(k, mean, var);
    kf_free(k);
    return rc;
}

int carma_kfilter_batch_carma(const double* time, const double* y, const double* yerr, int n, int p, int nmodels, const double* sigsqr,
                              const double* omega_re_im, const double* ma, int nma, const double* mu, double* mean, double* var,
                              int* singular, int* n_out, int device)
{
    if (!time || !y || !yerr || n < 2 || p < 2 || p > CARMA_PMAX || nmodels < 1 || !sigsqr || !omega_re_im || !ma || nma < 1 ||
        nma > p || !mean || !var) {
        set_error("carma_kfilter_batch_carma: bad argument (n >= 2, 2 <= p <= %d, 1 <= nma <= p, nmodels >= 1)", CARMA_PMAX);
        return CARMA_EINVAL;
    }
    int rc = select_device(device);
    if (rc != CARMA_OK) return rc;
    std::vector<double> t(time, time + n), yy(y, y + n), ee(yerr, yerr + n);
    sort_dedup(t, yy, ee);
    const int m = (int)t.size();
    if (n_out) *n_out = m;
    const std::vector<double> s = pack_series(t, yy, ee);
    const int PW = 3 * p + 2;
    std::vector<double> par((size_t)nmodels * PW, 0.0);
    for (int b = 0; b < nmodels; b++) {
        double* pb = par.data() + (size_t)b * PW;
        if (normalize_roots(p, omega_re_im + (size_t)b * 2 * p, pb) != CARMA_OK) {
            set_error("carma_kfilter_batch_carma: model %d: the AR roots must be real or come in complex-conjugate pairs", b);
            return CARMA_EINVAL;
        }
        for (int i = 0; i < nma; i++) pb[2 * p + i] = ma[(size_t)b * nma + i];     // zero padded to p (kfilter.hpp:318-320)
        pb[3 * p] = sigsqr[b];
        pb[3 * p + 1] = mu ? mu[b] : 0.0;
    }
    double *d_s = nullptr, *d_par = nullptr, *d_mv = nullptr, *d_mean = nullptr, *d_var = nullptr;
    int* d_sing = nullptr;
    const size_t nmv = (size_t)2 * m * ((size_t)nmodels + 64), nout = (size_t)nmodels * m;
    hipError_t e = dev_malloc(&d_s, sizeof(double) * s.size());
    if (e == hipSuccess) e = dev_malloc(&d_par, sizeof(double) * par.size());
    if (e == hipSuccess) e = dev_malloc(&d_mv, sizeof(double) * nmv);
    if (e == hipSuccess) e = dev_malloc(&d_mean, sizeof(double) * nout);
    if (e == hipSuccess) e = dev_malloc(&d_var, sizeof(double) * nout);
    if (e == hipSuccess) e = dev_malloc(&d_sing, sizeof(int) * nmodels);
    if (e == hipSuccess) e = hipMemcpy(d_s, s.data(), sizeof(double) * s.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_par, par.data(), sizeof(double) * par.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = launch_kfilter_batch(p, d_par, nmodels, reinterpret_cast<const double4*>(d_s), m, d_mv, d_sing, d_mean, d_var, nullptr);
    if (e == hipSuccess) e = hipMemcpy(mean, d_mean, sizeof(double) * nout, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(var, d_var, sizeof(double) * nout, hipMemcpyDeviceToHost);
    if (e == hipSuccess && singular) e = hipMemcpy(singular, d_sing, sizeof(int) * nmodels, hipMemcpyDeviceToHost);
    for (void* q : {(void*)d_s, (void*)d_par, (void*)d_mv, (void*)d_mean, (void*)d_var, (void*)d_sing})
        if (q) (void)dev_free(q);
    if (e != hipSuccess) return hip_fail(e, "carma_kfilter_batch_carma");
    return CARMA_OK;
}

int carma_kfilter_car1(const double* time, const double* y, const double* yerr, int n, double sigsqr, double omega,
                       double* mean, double* var, int* n_out, int device)
{
    int rc0 = CARMA_EINVAL;
    Kf* k = kf_make(time, y, yerr, n, 1, sigsqr, nullptr, nullptr, 0, omega, device, "carma_kfilter_car1", &rc0);
    if (!k) return rc0;
    if (n_out) *n_out = k->n;
    const int rc = kf_filter(k, mean, var);
    kf_free(k);
    return rc;
}

int carma_predict_carma(const double* time, const double* y, const double* yerr, int n, int p, double sigsqr,
                        const double* omega_re_im, const double* ma, int nma, const double* tpred, int M, double* pmean,
                        double* pvar, int device)
{
    int rc0 = CARMA_EINVAL;
    Kf* k = (p >= 2) ? kf_make(time, y, yerr, n, p, sigsqr, omega_re_im, ma, nma, 0.0, device, "carma_predict_carma", &rc0) : nullptr;
    if (!k) {
        if (p < 2) set_error("carma_predict_carma: need 2 <= p <= %d, omega and ma", CARMA_PMAX);
        return rc0;
    }
    const int rc = kf_predict(k, tpred, M, pmean, pvar);
    kf_free(k);
    return rc;
}

int carma_predict_car1(const double* time, const double* y, const double* yerr, int n, double sigsqr, double omega,
                       const double* tpred, int M, double* pmean, double* pvar, int device)
{
    int rc0 = CARMA_EINVAL;
    Kf* k = kf_make(time, y, yerr, n, 1, sigsqr, nullptr, nullptr, 0, omega, device, "carma_predict_car1", &rc0);
    if (!k) return rc0;
    const int rc = kf_predict(k, tpred, M, pmean, pvar);
    kf_free(k);
    return rc;
}

// carma_process / car1_process for npaths paths in one launch
static int simulate_common(const double* time, int n, int p, double sigsqr, const double* omega_re_im, const double* ma,
                           int nma, double car1_omega, int npaths, uint64_t seed, double* out, int device)
{
    if (!time || !out || n < 1 || npaths < 0) {
        set_error("carma_simulate: bad argument");
        return CARMA_EINVAL;
    }
    if (npaths == 0) return CARMA_OK;
    int rc = select_device(device);
    if (rc != CARMA_OK) return rc;
    std::vector<double> t(time, time + n);
    std::sort(t.begin(), t.end());                            // time.sort() (carma_pack.py:1176)
    std::vector<double> par(2 * CARMA_PMAX + CARMA_PMAX, 0.0);
    if (p > 1) {
        if (normalize_roots(p, omega_re_im, par.data()) != CARMA_OK) {
            set_error("carma_simulate_carma: the AR roots must be real or come in complex-conjugate pairs");
            return CARMA_EINVAL;
        }
        for (int i = 0; i < p && i < nma; i++) par[2 * CARMA_PMAX + i] = ma[i];
    }
    double *d_t = nullptr, *d_par = nullptr, *d_out = nullptr;
    int* d_sing = nullptr;
    const size_t nout = (size_t)npaths * n;
    hipError_t e = dev_malloc(&d_t, sizeof(double) * n);
    if (e == hipSuccess) e = dev_malloc(&d_par, sizeof(double) * par.size());
    if (e == hipSuccess) e = dev_malloc(&d_out, sizeof(double) * nout);
    if (e == hipSuccess) e = dev_malloc(&d_sing, sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(d_t, t.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_par, par.data(), sizeof(double) * par.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_sing, 0, sizeof(int));
    const unsigned s0 = (unsigned)(seed & 0xffffffffu), s1 = (unsigned)(seed >> 32);
    if (e == hipSuccess) {
        if (p == 1)
            e = launch_simulate_car1(sigsqr, car1_omega, d_t, n, npaths, s0, s1, 0u, d_out, nullptr);
        else
            e = launch_simulate_carma(p, d_par, d_par + 2 * CARMA_PMAX, sigsqr, d_t, n, npaths, s0, s1, 0u, d_out, d_sing, nullptr);
    }
    int sing = 0;
    if (e == hipSuccess) e = hipMemcpy(out, d_out, sizeof(double) * nout, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&sing, d_sing, sizeof(int), hipMemcpyDeviceToHost);
    if (d_t) (void)dev_free(d_t);
    if (d_par) (void)dev_free(d_par);
    if (d_out) (void)dev_free(d_out);
    if (d_sing) (void)dev_free(d_sing);
    if (e != hipSuccess) return hip_fail(e, "carma_simulate");
    return sing ? 1 : CARMA_OK;
}

int carma_simulate_carma(const double* time, int n, int p, double sigsqr, const double* omega_re_im, const double* ma, int nma,
                         int npaths, uint64_t seed, double* out, int device)
{
    if (p < 2 || p > CARMA_PMAX || !omega_re_im || !ma || nma < 1 || !(sigsqr > 0.0)) {
        set_error("carma_simulate_carma: need 2 <= p <= %d, omega, ma and sigsqr > 0", CARMA_PMAX);
        return CARMA_EINVAL;
    }
    return simulate_common(time, n, p, sigsqr, omega_re_im, ma, nma, 0.0, npaths, seed, out, device);
}

int carma_simulate_car1(const double* time, int n, double sigsqr, double omega, int npaths, uint64_t seed, double* out, int device)
{
    if (!(sigsqr > 0.0) || !(omega > 0.0)) {
        set_error("carma_simulate_car1: need sigsqr > 0 and omega > 0");
        return CARMA_EINVAL;
    }
    return simulate_common(time, n, 1, sigsqr, nullptr, nullptr, 0, omega, npaths, seed, out, device);
}

}  // extern "C"
