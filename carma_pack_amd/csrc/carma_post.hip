// carma_post.hip -- CarmaSample post-processing on the device (gfx950 only).
//
// The reference computes, in Python and per MCMC sample, the amplitude of the driving noise
// (CarmaSample._sigma_noise, src/carmcmc/carma_pack.py:513-546 == CARp::Variance with sigma = 1) and, for the credibility
// band of the power spectrum (plot_power_spectrum, :548-648; Car1Sample :950-1035), the spectrum of EVERY sample on 1000
// frequencies followed by three percentiles per frequency: nfreq x nsamples complex polynomial values and nfreq selections
// out of nsamples values (75 000 samples of BASELINE configs[2]: 7.5e7 spectrum values, 600 MB as doubles).
//
//   k_sigma_noise    one lane per sample: the reference's sum over the AR roots, complex arithmetic in registers
//   k_psd_grid       one lane per (sample, 8 frequencies): Horner's rule in i 2 pi f for alpha and delta; the coefficient
//                    arrays are sample-major in HBM ([k][ns]), so a wave's loads and its stores of psd[f][s] are coalesced
//   k_row_quantiles  one workgroup per frequency: exact order statistics of the row by a most-significant-byte-first radix
//                    SELECT on the order-preserving 64-bit image of the doubles (all requested ranks in the same passes,
//                    histograms in LDS; bytes shared by the whole row are skipped), then numpy's linear interpolation
//                    between the two neighbouring order statistics (np.percentile's default, which the reference calls)
// HBM-bound byte work: the grid is written once and read once per radix pass (at most eight).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/carma_mi355.h"
#include "carma_host.h"

namespace carma {

constexpr int POST_PMAX = CARMA_PMAX;       // AR order <= 7: alpha has <= 8 coefficients, delta <= 7
constexpr int POST_NQ = 8;                  // order statistics per row: two per percentile, four percentiles
constexpr int PSD_FT = 8;                   // frequencies per lane of k_psd_grid
constexpr int QT = 1024;                    // threads of k_row_quantiles (one workgroup per row)
constexpr int QU = 4;                       // independent loads in flight per thread and trip of its row loops

struct Cd {
    double re, im;
};
__device__ __forceinline__ Cd cmul(Cd a, Cd b) { return Cd{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Cd cadd(Cd a, Cd b) { return Cd{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ Cd csub(Cd a, Cd b) { return Cd{a.re - b.re, a.im - b.im}; }
// Smith's division (what numpy's complex division does): no overflow of |b|^2
__device__ __forceinline__ Cd cdiv(Cd a, Cd b)
{
    if (fabs(b.re) >= fabs(b.im)) {
        const double r = b.im / b.re, den = b.re + b.im * r;
        return Cd{(a.re + a.im * r) / den, (a.im - a.re * r) / den};
    }
    const double r = b.re / b.im, den = b.re * r + b.im;
    return Cd{(a.re * r + a.im) / den, (a.im * r - a.re) / den};
}

// sigma_s = sqrt(var_s / Re sum_k [delta(r_k) delta(-r_k)] / [-2 Re r_k prod_{l != k} (r_l - r_k)(conj r_l + r_k)])
// (carma_pack.py:513-546, the Python twin of CARp::Variance, src/carpack.cpp:377-409)
__global__ __launch_bounds__(256) void k_sigma_noise(int p, int nma, const double* __restrict__ roots /* [ns][p][2] */,
                                                     const double* __restrict__ ma /* [ns][nma] */,
                                                     const double* __restrict__ var, int ns, double* __restrict__ sigma)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    Cd r[POST_PMAX];
    double b[POST_PMAX];
    for (int k = 0; k < p; k++) r[k] = Cd{roots[(s * p + k) * 2], roots[(s * p + k) * 2 + 1]};
    for (int l = 0; l < nma; l++) b[l] = ma[s * nma + l];
    // Re of the sum only, accumulated with its rounding errors carried along (Neumaier's compensated sum): the p terms are of order
    // 1 / (root differences)^2 and cancel -- a plain sum of a CARMA(6,0) sample with two nearby pairs was 1.8e-10 from the exact
    // value where the reference's (numpy, another order) happened to be at 2.5e-11 (round 5 review); the terms' own rounding stays
    double tsum = 0.0, tcomp = 0.0;
    for (int k = 0; k < p; k++) {
        Cd den{-2.0 * r[k].re, 0.0};
        for (int l = 0; l < p; l++)
            if (l != k) den = cmul(den, cmul(csub(r[l], r[k]), cadd(Cd{r[l].re, -r[l].im}, r[k])));
        Cd s1{0.0, 0.0}, s2{0.0, 0.0}, pw{1.0, 0.0}, pm{1.0, 0.0};     // r_k^l and (-r_k)^l
        const Cd mr{-r[k].re, -r[k].im};
        for (int l = 0; l < nma; l++) {
            s1 = cadd(s1, Cd{b[l] * pw.re, b[l] * pw.im});
            s2 = cadd(s2, Cd{b[l] * pm.re, b[l] * pm.im});
            pw = cmul(pw, r[k]);
            pm = cmul(pm, mr);
        }
        const double x = cdiv(cmul(s1, s2), den).re;
        const double t = tsum + x;
        tcomp += fabs(tsum) >= fabs(x) ? (tsum - t) + x : (x - t) + tsum;
        tsum = t;
    }
    sigma[s] = sqrt(var[s] / (tsum + tcomp));
}

// psd[f][s] = sigma_s^2 |delta_s(i 2 pi f)|^2 / |alpha_s(i 2 pi f)|^2     (carma_pack.py:596-618)
// ar: [nar][ns] highest order first; ma: [nma][ns] lowest order first; both sample-major
__global__ __launch_bounds__(256) void k_psd_grid(int nar, int nma, const double* __restrict__ ar, const double* __restrict__ ma,
                                                  const double* __restrict__ sigma, int ns, const double* __restrict__ freq,
                                                  int nf, double* __restrict__ psd)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    double a[POST_PMAX + 1], b[POST_PMAX + 1];
    for (int k = 0; k < nar; k++) a[k] = ar[(long)k * ns + s];
    for (int k = 0; k < nma; k++) b[k] = ma[(long)k * ns + s];
    const double sg = sigma[s], s2 = sg * sg;
    const int f0 = blockIdx.y * PSD_FT;
    for (int i = 0; i < PSD_FT && f0 + i < nf; i++) {
        const double w = 2.0 * M_PI * freq[f0 + i];           // z = i w:  acc z + c = (c - acc.im w) + i (acc.re w)
        double are = 0.0, aim = 0.0;
        for (int k = 0; k < nar; k++) {
            const double t = are;
            are = fma(-aim, w, a[k]);
            aim = t * w;
        }
        double mre = 0.0, mim = 0.0;
        for (int k = nma - 1; k >= 0; k--) {
            const double t = mre;
            mre = fma(-mim, w, b[k]);
            mim = t * w;
        }
        psd[(long)(f0 + i) * ns + s] = s2 * (mre * mre + mim * mim) / (are * are + aim * aim);
    }
}

// order-preserving image of a double: unsigned comparison of the keys == numerical comparison of the values (-0 < +0)
__device__ __forceinline__ unsigned long long key_of(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return b ^ ((b >> 63) ? ~0ull : 0x8000000000000000ull);
}
__device__ __forceinline__ double value_of(unsigned long long k)
{
    const unsigned long long b = k ^ ((k >> 63) ? 0x8000000000000000ull : ~0ull);
    return __longlong_as_double((long long)b);
}

// numpy's _lerp (np.percentile, method "linear"): a + (b - a) t, from the other end for t >= 0.5
__device__ __forceinline__ double np_lerp(double a, double b, double t)
{
    const double d = b - a;
    return t >= 0.5 ? b - d * (1.0 - t) : a + d * t;
}

// One workgroup per row.  ranks[2 j], ranks[2 j + 1] = the order statistics either side of percentile j, gammas[j] its
// interpolation weight.  Radix select, most significant byte first: after the pass over byte B every rank knows the top
// (8 - B) bytes of its order statistic and its rank among the elements that share them.
__global__ __launch_bounds__(QT) void k_row_quantiles(const double* __restrict__ grid, int ns, int nq,
                                                       const int* __restrict__ ranks, const double* __restrict__ gammas,
                                                       double* __restrict__ band /* [rows][nq / 2] */)
{
    __shared__ unsigned hist[POST_NQ][256];
    __shared__ unsigned long long prefix[POST_NQ];
    __shared__ unsigned krem[POST_NQ];
    __shared__ int rep[POST_NQ];
    __shared__ unsigned long long kmin, kmax;
    __shared__ unsigned n_nan;
    const double* row = grid + (long)blockIdx.x * ns;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        kmin = ~0ull;
        kmax = 0ull;
        n_nan = 0u;
    }
    if (tid < nq) krem[tid] = (unsigned)ranks[tid];
    __syncthreads();
    {
        unsigned long long lo = ~0ull, hi = 0ull;
        unsigned nn = 0;
        for (long i0 = tid; i0 < ns; i0 += (long)QT * QU) {
            double x[QU];
#pragma unroll
            for (int u = 0; u < QU; u++) x[u] = i0 + (long)u * QT < ns ? row[i0 + (long)u * QT] : row[tid < ns ? tid : 0];   // (a repeat of an element of the row)
#pragma unroll
            for (int u = 0; u < QU; u++) {
                nn += (x[u] != x[u]) ? 1u : 0u;
                const unsigned long long k = key_of(x[u]);
                lo = k < lo ? k : lo;
                hi = k > hi ? k : hi;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long lo2 = __shfl_xor(lo, o, 64), hi2 = __shfl_xor(hi, o, 64);
            lo = lo2 < lo ? lo2 : lo;
            hi = hi2 > hi ? hi2 : hi;
            nn += __shfl_xor(nn, o, 64);
        }
        if (lane == 0) {
            atomicMin(&kmin, lo);
            atomicMax(&kmax, hi);
            atomicAdd(&n_nan, nn);
        }
    }
    __syncthreads();
    const int nperc = nq / 2;
    if (n_nan != 0u) {                                        // np.percentile: a NaN anywhere in the row makes every percentile NaN
        if (tid < nperc) band[(long)blockIdx.x * nperc + tid] = __longlong_as_double(0x7ff8000000000000ll);
        return;
    }
    // bytes above `top` are the same in every key of the row: nothing to select there
    const unsigned long long diff = kmin ^ kmax;
    const int top = diff ? (63 - __clzll((long long)diff)) / 8 : -1;
    if (tid < nq) prefix[tid] = (top >= 7 || top < 0) ? (top < 0 ? kmin : 0ull) : (kmin >> (8 * (top + 1))) << (8 * (top + 1));
    __syncthreads();
    for (int pass = top; pass >= 0; pass--) {
        const int shift = 8 * pass;
        // ranks that still share their prefix share a histogram
        if (tid < nq) {
            int r0 = tid;
            for (int r = 0; r < tid; r++)
                if (prefix[r] == prefix[tid]) {
                    r0 = r;
                    break;
                }
            rep[tid] = r0;
        }
        for (int i = tid; i < POST_NQ * 256; i += QT) (&hist[0][0])[i] = 0u;
        __syncthreads();
        unsigned long long pf[POST_NQ];
        bool own[POST_NQ];
#pragma unroll
        for (int r = 0; r < POST_NQ; r++) {
            own[r] = r < nq && rep[r] == r;
            pf[r] = own[r] ? prefix[r] : 0ull;
        }
        for (long i0 = tid; i0 < ns; i0 += (long)QT * QU) {
            double x[QU];
#pragma unroll
            for (int u = 0; u < QU; u++) x[u] = i0 + (long)u * QT < ns ? row[i0 + (long)u * QT] : 0.0;
#pragma unroll
            for (int u = 0; u < QU; u++) {
                if (i0 + (long)u * QT < ns) {
                    const unsigned long long k = key_of(x[u]);
                    const unsigned byte = (unsigned)(k >> shift) & 255u;
#pragma unroll
                    for (int r = 0; r < POST_NQ; r++)
                        if (own[r] && (pass == 7 || ((k ^ pf[r]) >> (shift + 8)) == 0ull)) atomicAdd(&hist[r][byte], 1u);
                }
            }
        }
        __syncthreads();
        // wave w resolves rank w: lane l owns bins 4 l .. 4 l + 3
        for (int r = wave; r < nq; r += QT / 64) {
            const unsigned* h = hist[rep[r]];
            const unsigned c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
            unsigned incl = c0 + c1 + c2 + c3;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = __shfl_up(incl, o, 64);
                if (lane >= o) incl += v;
            }
            const unsigned k = krem[r];
            const unsigned long long mask = __ballot(incl > k);
            const int owner = __ffsll((long long)mask) - 1;   // first lane whose cumulative count exceeds the rank
            if (lane == owner) {
                unsigned before = incl - (c0 + c1 + c2 + c3);
                unsigned bin = 4 * lane;
                if (before + c0 <= k) {
                    before += c0;
                    bin++;
                    if (before + c1 <= k) {
                        before += c1;
                        bin++;
                        if (before + c2 <= k) {
                            before += c2;
                            bin++;
                        }
                    }
                }
                krem[r] = k - before;
                prefix[r] |= (unsigned long long)bin << shift;
            }
        }
        __syncthreads();
    }
    if (tid < nperc) {
        const double a = value_of(prefix[2 * tid]), b = value_of(prefix[2 * tid + 1]);
        band[(long)blockIdx.x * nperc + tid] = np_lerp(a, b, gammas[tid]);
    }
}

struct DevBufs {                                              // frees what it holds
    std::vector<void*> p;
    ~DevBufs()
    {
        for (void* q : p)
            if (q) (void)dev_free(q);
    }
    template <class T>
    hipError_t alloc(T** out, size_t n)
    {
        void* q = nullptr;
        const hipError_t e = dev_malloc(&q, n * sizeof(T) ? n * sizeof(T) : sizeof(T));
        if (e == hipSuccess) p.push_back(q);
        *out = reinterpret_cast<T*>(q);
        return e;
    }
};

}  // namespace carma

using namespace carma;

extern "C" {

int carma_sigma_noise_batch(int p, int nma, const double* ar_roots_re_im, const double* ma_coefs, const double* var, int ns,
                            double* sigma, int device)
{
    if (p < 1 || p > CARMA_PMAX || nma < 1 || nma > p || !ar_roots_re_im || !ma_coefs || !var || !sigma || ns < 0) {
        set_error("carma_sigma_noise_batch: bad argument (1 <= p <= %d, 1 <= nma <= p)", CARMA_PMAX);
        return CARMA_EINVAL;
    }
    if (ns == 0) return CARMA_OK;
    int rc = select_device(device);
    if (rc != CARMA_OK) return rc;
    DevBufs B;
    double *d_r = nullptr, *d_m = nullptr, *d_v = nullptr, *d_s = nullptr;
    hipError_t e = B.alloc(&d_r, (size_t)ns * p * 2);
    if (e == hipSuccess) e = B.alloc(&d_m, (size_t)ns * nma);
    if (e == hipSuccess) e = B.alloc(&d_v, (size_t)ns);
    if (e == hipSuccess) e = B.alloc(&d_s, (size_t)ns);
    if (e == hipSuccess) e = hipMemcpy(d_r, ar_roots_re_im, sizeof(double) * ns * p * 2, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_m, ma_coefs, sizeof(double) * ns * nma, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_v, var, sizeof(double) * ns, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_sigma_noise, dim3((ns + 255) / 256), dim3(256), 0, nullptr, p, nma, d_r, d_m, d_v, ns, d_s);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(sigma, d_s, sizeof(double) * ns, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "carma_sigma_noise_batch");
    return CARMA_OK;
}

int carma_psd_band(int nar, int nma, const double* ar_coefs, const double* ma_coefs, const double* sigma, int ns,
                   const double* freq, int nf, const double* percentiles, int nperc, double* band, double* psd_samples,
                   int device)
{
    if (nar < 2 || nar > CARMA_PMAX + 1 || nma < 1 || nma > CARMA_PMAX || !ar_coefs || !ma_coefs || !sigma || !freq ||
        ns < 1 || nf < 1 || nperc < 0 || 2 * nperc > POST_NQ || (nperc > 0 && (!percentiles || !band))) {
        set_error("carma_psd_band: bad argument (2 <= nar <= %d, 1 <= nma <= %d, ns >= 1, nf >= 1, at most %d percentiles)",
                  CARMA_PMAX + 1, CARMA_PMAX, POST_NQ / 2);
        return CARMA_EINVAL;
    }
    for (int j = 0; j < nperc; j++)
        if (!(percentiles[j] >= 0.0 && percentiles[j] <= 100.0)) {
            set_error("carma_psd_band: percentiles must lie in [0, 100]");   // numpy: ValueError
            return CARMA_EINVAL;
        }
    int rc = select_device(device);
    if (rc != CARMA_OK) return rc;
    // sample-major copies of the coefficient arrays (coalesced loads in k_psd_grid)
    std::vector<double> art((size_t)nar * ns), mat((size_t)nma * ns);
    for (int s = 0; s < ns; s++) {
        for (int k = 0; k < nar; k++) art[(size_t)k * ns + s] = ar_coefs[(size_t)s * nar + k];
        for (int k = 0; k < nma; k++) mat[(size_t)k * ns + s] = ma_coefs[(size_t)s * nma + k];
    }
    // np.percentile(x, q) with the default method: virtual index (n - 1) q / 100, the order statistics either side of it
    std::vector<int> ranks(2 * (nperc > 0 ? nperc : 1), 0);
    std::vector<double> gam(nperc > 0 ? nperc : 1, 0.0);
    for (int j = 0; j < nperc; j++) {
        const double vi = (double)(ns - 1) * (percentiles[j] / 100.0);
        double lo = std::floor(vi);
        if (lo > ns - 1) lo = ns - 1;
        const int ilo = (int)lo, ihi = ilo + 1 < ns ? ilo + 1 : ns - 1;
        ranks[2 * j] = ilo;
        ranks[2 * j + 1] = ihi;
        gam[j] = vi - lo;
    }
    // the grid is held for `fc` frequencies at a time: at most 2^30 values (8 GiB) -- 3.2 million samples (all 64 cold chains of
    // BASELINE configs[2]) take 335 rows per round
    const int fc = (int)std::min<long>(nf, std::max<long>(1, (1L << 30) / ns));
    DevBufs B;
    double *d_a = nullptr, *d_m = nullptr, *d_s = nullptr, *d_f = nullptr, *d_g = nullptr, *d_band = nullptr, *d_gam = nullptr;
    int* d_rk = nullptr;
    hipError_t e = B.alloc(&d_a, art.size());
    if (e == hipSuccess) e = B.alloc(&d_m, mat.size());
    if (e == hipSuccess) e = B.alloc(&d_s, (size_t)ns);
    if (e == hipSuccess) e = B.alloc(&d_f, (size_t)nf);
    if (e == hipSuccess) e = B.alloc(&d_g, (size_t)fc * ns);
    if (e == hipSuccess) e = B.alloc(&d_band, (size_t)nf * (nperc > 0 ? nperc : 1));
    if (e == hipSuccess) e = B.alloc(&d_gam, gam.size());
    if (e == hipSuccess) e = B.alloc(&d_rk, ranks.size());
    if (e == hipSuccess) e = hipMemcpy(d_a, art.data(), sizeof(double) * art.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_m, mat.data(), sizeof(double) * mat.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_s, sigma, sizeof(double) * ns, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_f, freq, sizeof(double) * nf, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_gam, gam.data(), sizeof(double) * gam.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_rk, ranks.data(), sizeof(int) * ranks.size(), hipMemcpyHostToDevice);
    for (int f0 = 0; f0 < nf && e == hipSuccess; f0 += fc) {
        const int nfc = std::min(fc, nf - f0);
        hipLaunchKernelGGL(k_psd_grid, dim3((ns + 255) / 256, (nfc + PSD_FT - 1) / PSD_FT), dim3(256), 0, nullptr, nar, nma, d_a,
                           d_m, d_s, ns, d_f + f0, nfc, d_g);
        e = hipGetLastError();
        if (e == hipSuccess && nperc > 0) {
            hipLaunchKernelGGL(k_row_quantiles, dim3(nfc), dim3(QT), 0, nullptr, d_g, ns, 2 * nperc, d_rk, d_gam,
                               d_band + (size_t)f0 * nperc);
            e = hipGetLastError();
        }
        if (e == hipSuccess && psd_samples)
            e = hipMemcpy(psd_samples + (size_t)f0 * ns, d_g, sizeof(double) * (size_t)nfc * ns, hipMemcpyDeviceToHost);
    }
    if (e == hipSuccess && nperc > 0) e = hipMemcpy(band, d_band, sizeof(double) * nf * nperc, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(e, "carma_psd_band");
    return CARMA_OK;
}

}  // extern "C"
