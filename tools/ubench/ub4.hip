// micro-benchmark: issue cost of LDS reads inside an FP64 stream (single wave), results not waited for
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
__global__ void kern(double* out, long long* cyc, double seed)
{
    __shared__ double2 buf[1024];
    int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) buf[i] = make_double2(seed, 1.0);
    __syncthreads();
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = 1.0000001, c = 1e-9;
    long long t0, t1;
    unsigned addr_b = (unsigned)(size_t)(buf) + (lane & ~7) * 16;   // group-broadcast address
    unsigned addr_o = (unsigned)(size_t)(buf) + lane * 16;          // own-lane address
    unsigned addr_u = (unsigned)(size_t)(buf);                      // wave-uniform
    double2 r0, r1, r2, r3;
#define F8                                                             \
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c)); \
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c)); \
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c)); \
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c)); \
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c)); \
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c)); \
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c)); \
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
#define RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define RD64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
    double d0;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) { F8 }
    asm volatile("s_waitcnt lgkmcnt(0)");
    t1 = clock64(); if (lane == 0) cyc[0] = t1 - t0;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) { F8 RD128(r0, addr_b, 0); }
    asm volatile("s_waitcnt lgkmcnt(0)");
    t1 = clock64(); if (lane == 0) cyc[1] = t1 - t0;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) { F8 RD128(r0, addr_b, 0); RD128(r1, addr_b, 16); }
    asm volatile("s_waitcnt lgkmcnt(0)");
    t1 = clock64(); if (lane == 0) cyc[2] = t1 - t0;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) { F8 RD128(r0, addr_b, 0); RD128(r1, addr_b, 16); RD128(r2, addr_b, 32); RD128(r3, addr_b, 48); }
    asm volatile("s_waitcnt lgkmcnt(0)");
    t1 = clock64(); if (lane == 0) cyc[3] = t1 - t0;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) { F8 RD128(r0, addr_o, 0); }
    asm volatile("s_waitcnt lgkmcnt(0)");
    t1 = clock64(); if (lane == 0) cyc[4] = t1 - t0;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) { F8 RD128(r0, addr_u, 0); }
    asm volatile("s_waitcnt lgkmcnt(0)");
    t1 = clock64(); if (lane == 0) cyc[5] = t1 - t0;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) { F8 RD64(d0, addr_b, 0); }
    asm volatile("s_waitcnt lgkmcnt(0)");
    t1 = clock64(); if (lane == 0) cyc[6] = t1 - t0;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) { F8 RD64(d0, addr_o, 0); RD64(d0, addr_o, 8); }
    asm volatile("s_waitcnt lgkmcnt(0)");
    t1 = clock64(); if (lane == 0) cyc[7] = t1 - t0;
    out[lane] = a0 + a1 + a2 + a3 + r0.x + r1.x + r2.x + r3.x + d0;
}
int main()
{
    double* out; long long* cyc;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 16 * 8);
    for (int rep = 0; rep < 2; rep++) kern<<<1, 64>>>(out, cyc, 1.5);
    long long h[16];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[] = {"8 fma", "8 fma + 1 b128 bcast", "8 fma + 2 b128 bcast", "8 fma + 4 b128 bcast", "8 fma + 1 b128 own", "8 fma + 1 b128 uniform", "8 fma + 1 b64 bcast", "8 fma + 2 b64 own"};
    for (int i = 0; i < 8; i++) printf("%-26s %.1f cycles/iter\n", nm[i], h[i] / (double)REP);
    return 0;
}
