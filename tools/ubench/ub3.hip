// micro-benchmark: single-wave issue cost of scalar / s_nop / DPP instructions mixed into an FP64 stream
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
__global__ void kern(double* out, long long* cyc, double seed)
{
    int lane = threadIdx.x;
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = 1.0000001, c = 1e-9;
    long long t0, t1;
#define F4                                                             \
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c)); \
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c)); \
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c)); \
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) { F4 }
    t1 = clock64(); if (lane == 0) cyc[0] = t1 - t0;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) { F4 asm volatile("s_nop 1"); }
    t1 = clock64(); if (lane == 0) cyc[1] = t1 - t0;
    int sx = 0;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) { F4 asm volatile("s_add_i32 %0, %0, 1" : "+s"(sx)); }
    t1 = clock64(); if (lane == 0) cyc[2] = t1 - t0;
    int lo = lane, o1 = 0, o2 = 0;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) {
        F4
        asm volatile("v_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(o1), "=&v"(o2) : "v"(lo));
    }
    t1 = clock64(); if (lane == 0) cyc[3] = t1 - t0;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) {
        F4
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %2" : "=&v"(o1), "=&v"(o2) : "v"(lo));
    }
    t1 = clock64(); if (lane == 0) cyc[4] = t1 - t0;
    // 8 independent fma (no dependence within 8)
    double a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) {
        F4
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a5) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a7) : "v"(b), "v"(c));
    }
    t1 = clock64(); if (lane == 0) cyc[5] = t1 - t0;
    // v_mul_f64 / v_add_f64 independent x4
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) {
        asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(b));
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(a1) : "v"(c));
        asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a2) : "v"(b));
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(a3) : "v"(c));
    }
    t1 = clock64(); if (lane == 0) cyc[6] = t1 - t0;
    out[lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + o1 + o2 + sx;
}
int main()
{
    double* out; long long* cyc;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 16 * 8);
    for (int rep = 0; rep < 2; rep++) kern<<<1, 64>>>(out, cyc, 1.5);
    long long h[16];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[] = {"4 fma", "4 fma + s_nop 1", "4 fma + s_add", "4 fma + 2 v_mov_dpp", "4 fma + 2 v_mov", "8 fma", "2 mul + 2 add"};
    for (int i = 0; i < 7; i++) printf("%-24s %.1f cycles/iter\n", nm[i], h[i] / (double)REP);
    return 0;
}
