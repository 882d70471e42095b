// How long does the FIRST global load of a kernel take (steady state, back-to-back launches of the same kernel on the same
// buffers)?  Each workgroup's wave 0 stamps s_memtime around (a) one load from a buffer every launch reads, (b) a second load
// from the same cache line, (c) a load from another buffer (another page); 256 workgroups x 256 threads, 100 launches.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/ub12_first_load.hip -o tools/ubench/ub12_first_load && tools/ubench/ub12_first_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void k(const double* __restrict__ a, const double* __restrict__ b, unsigned long long* out, double* sink)
{
    unsigned long long t0, t1, t2, t3;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    double x = a[blockIdx.x * 16 + (threadIdx.x & 15)];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double y = a[blockIdx.x * 16 + ((threadIdx.x + 1) & 15)];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    double z = b[blockIdx.x * 16 + (threadIdx.x & 15)];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3)::"memory");
    if (threadIdx.x == 0) {
        out[blockIdx.x * 3 + 0] = t1 - t0;
        out[blockIdx.x * 3 + 1] = t2 - t1;
        out[blockIdx.x * 3 + 2] = t3 - t2;
    }
    if (x + y + z == 1.2345) sink[threadIdx.x] = x;
}

int main()
{
    const int NB = 256;
    double *a, *b, *sink;
    unsigned long long* out;
    hipMalloc(&a, NB * 16 * 8);
    hipMalloc(&b, 1 << 22);
    hipMalloc(&sink, 4096);
    hipMalloc(&out, NB * 3 * 8);
    hipMemset(a, 0, NB * 16 * 8);
    hipMemset(b, 0, 1 << 22);
    std::vector<unsigned long long> h(NB * 3);
    for (int rep = 0; rep < 100; rep++) hipLaunchKernelGGL(k, dim3(NB), dim3(256), 0, 0, a, b + (1 << 18), out, sink);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, NB * 3 * 8, hipMemcpyDeviceToHost);
    for (int c = 0; c < 3; c++) {
        std::vector<unsigned long long> v;
        for (int i = 0; i < NB; i++) v.push_back(h[i * 3 + c]);
        std::sort(v.begin(), v.end());
        printf("%s: min %llu median %llu max %llu (100 MHz ticks x ~21 = core cycles)\n",
               c == 0 ? "first load of the kernel" : c == 1 ? "same line again" : "another buffer", v[0], v[NB / 2], v[NB - 1]);
    }
    return 0;
}
