// micro-benchmark: cost of the per-step LDS pattern of the consumer wave (one wave, latency regime)
#include <hip/hip_runtime.h>
#include <cstdio>
struct C2 { double re, im; };
template <int NK, int NR, int GSTRIDE /* entries of 16 B between groups */, bool OWN, bool REC>
__device__ long long pat(double* xk, C2* ring, int lane, double& acc)
{
    const int g = lane >> 3, r = lane & 7;
    double k = acc;
    long long t0 = clock64();
#pragma unroll 4
    for (int it = 0; it < 64; it++) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        xk[lane] = k;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NK; i++) {
            double2 v = reinterpret_cast<double2*>(xk)[g * 4 + i];
            s += v.x + v.y;
        }
        const C2* slot = ring + (it & 15) * 80 + g * GSTRIDE;
#pragma unroll
        for (int j = 0; j < NR; j++) {
            C2 v = slot[j];
            s += v.re * v.im;
        }
        if (OWN) { C2 v = slot[r]; s += v.re; }
        if (REC) { C2 v = ring[2048 + (it & 15)]; s += v.im; }
        __builtin_amdgcn_wave_barrier();
        k = s * 0.125;
    }
    long long t1 = clock64();
    acc = k;
    return t1 - t0;
}
__global__ void kern(double* out, long long* cyc, double seed)
{
    __shared__ double xk[64];
    __shared__ C2 ring[2048 + 64];
    int lane = threadIdx.x;
    for (int i = lane; i < 2048 + 64; i += 64) ring[i] = C2{seed + i * 1e-6, 1.0};
    xk[lane] = seed;
    __syncthreads();
    double acc = seed + lane * 1e-3;
    long long c;
    c = pat<1, 0, 8, false, false>(xk, ring, lane, acc); if (lane == 0) cyc[0] = c;
    c = pat<3, 0, 8, false, false>(xk, ring, lane, acc); if (lane == 0) cyc[1] = c;
    c = pat<3, 5, 8, false, false>(xk, ring, lane, acc); if (lane == 0) cyc[2] = c;
    c = pat<3, 5, 8, true, false>(xk, ring, lane, acc); if (lane == 0) cyc[3] = c;
    c = pat<3, 5, 8, true, true>(xk, ring, lane, acc); if (lane == 0) cyc[4] = c;
    c = pat<3, 5, 9, true, true>(xk, ring, lane, acc); if (lane == 0) cyc[5] = c;
    c = pat<3, 5, 10, true, true>(xk, ring, lane, acc); if (lane == 0) cyc[6] = c;
    c = pat<0, 5, 8, false, false>(xk, ring, lane, acc); if (lane == 0) cyc[7] = c;
    c = pat<0, 1, 8, false, false>(xk, ring, lane, acc); if (lane == 0) cyc[8] = c;
    out[lane] = acc;
}
int main()
{
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16 * 8);
    for (int rep = 0; rep < 2; rep++) kern<<<1, 64>>>(out, cyc, 1.5);
    long long h[16];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[] = {"w + 1 kread", "w + 3 kreads", "w + 3k + 5 ring", "w + 3k + 5 ring + own", "w + 3k + 5 ring + own + rec",
                        "same, group stride 9", "same, group stride 10", "w + 5 ring only", "w + 1 ring"};
    for (int i = 0; i < 9; i++) printf("%-32s %.1f cycles/iter\n", nm[i], h[i] / 64.0);
    return 0;
}
