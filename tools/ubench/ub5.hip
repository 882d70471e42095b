// micro-benchmark: issue cost of FP64 DPP (row_newbcast) fmac vs plain fmac, v_mov_b64, v_mov_b64_dpp (single wave)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
__global__ void kern(double* out, long long* cyc, double seed)
{
    int lane = threadIdx.x;
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = 1.0000001 + lane * 1e-9, c = 1e-9;
    long long t0, t1;
#define PLAIN4                                                            \
    asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c)); \
    asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c)); \
    asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c)); \
    asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
#define DPP4                                                                                                        \
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(b), "v"(c)); \
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(a1) : "v"(b), "v"(c)); \
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(a2) : "v"(b), "v"(c)); \
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a3) : "v"(b), "v"(c));
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) { PLAIN4 }
    t1 = clock64(); if (lane == 0) cyc[0] = t1 - t0;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) { DPP4 }
    t1 = clock64(); if (lane == 0) cyc[1] = t1 - t0;
    double m0, m1, m2, m3;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) {
        asm volatile("v_mov_b64 %0, 0\n\tv_mov_b64 %1, 0\n\tv_mov_b64 %2, 0\n\tv_mov_b64 %3, 0" : "=v"(m0), "=v"(m1), "=v"(m2), "=v"(m3));
    }
    t1 = clock64(); if (lane == 0) cyc[2] = t1 - t0;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) {
        asm volatile("v_mov_b64_dpp %0, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %2, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                     : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3) : "v"(b));
    }
    t1 = clock64(); if (lane == 0) cyc[3] = t1 - t0;
    // dependent fmac_dpp chain (same accumulator)
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) {
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(b), "v"(c));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(b), "v"(c));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(b), "v"(c));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(b), "v"(c));
    }
    t1 = clock64(); if (lane == 0) cyc[4] = t1 - t0;
    // v_pk_mov_b32 zero / v_mov_b32 pairs
    int z0, z1, z2, z3, z4, z5, z6, z7;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < REP; i++) {
        asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0\n\tv_mov_b32 %4, 0\n\tv_mov_b32 %5, 0\n\tv_mov_b32 %6, 0\n\tv_mov_b32 %7, 0"
                     : "=v"(z0), "=v"(z1), "=v"(z2), "=v"(z3), "=v"(z4), "=v"(z5), "=v"(z6), "=v"(z7));
    }
    t1 = clock64(); if (lane == 0) cyc[5] = t1 - t0;
    out[lane] = a0 + a1 + a2 + a3 + m0 + m1 + m2 + m3 + z0 + z1 + z2 + z3 + z4 + z5 + z6 + z7;
}
int main()
{
    double* out; long long* cyc;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 16 * 8);
    for (int rep = 0; rep < 2; rep++) kern<<<1, 64>>>(out, cyc, 1.5);
    long long h[16];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[] = {"4 v_fmac_f64 (independent)", "4 v_fmac_f64_dpp (independent)", "4 v_mov_b64 0", "4 v_mov_b64_dpp", "4 v_fmac_f64_dpp (dependent)", "8 v_mov_b32 0"};
    for (int i = 0; i < 6; i++) printf("%-32s %.1f cycles/iter\n", nm[i], h[i] / (double)REP);
    return 0;
}
