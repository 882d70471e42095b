// micro-benchmarks: dependent-chain latencies of the ops on the Kalman critical path (one wave)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
__global__ void k(double* out, long long* cyc, double seed) {
    __shared__ double4 xch[64];
    int lane = threadIdx.x;
    double a = seed + lane * 1e-3, b = 1.0000001, c = 1e-9;
    long long t0, t1;
    // 1. dependent v_fma_f64 chain
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    t1 = clock64(); if (lane == 0) cyc[0] = t1 - t0;
    // 2. independent fma x4 interleaved
    double a1 = a, a2 = a + 1, a3 = a + 2, a4 = a + 3;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < N / 4; i++) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c));
    }
    t1 = clock64(); if (lane == 0) cyc[1] = t1 - t0;
    a = a1 + a2 + a3 + a4;
    // 3. dependent v_rsq_f64 chain
    double r = fabs(a) + 2.0;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < 64; i++) asm volatile("v_rsq_f64 %0, %0" : "+v"(r));
    t1 = clock64(); if (lane == 0) cyc[2] = t1 - t0;
    // 4. dependent v_mul_f64
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));
    t1 = clock64(); if (lane == 0) cyc[3] = t1 - t0;
    // 5. dependent v_add_f64
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));
    t1 = clock64(); if (lane == 0) cyc[4] = t1 - t0;
    // 6. LDS write -> read round trip (dependent)
    double4 v = make_double4(a, r, a, r);
    t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < 64; i++) {
        xch[lane] = v;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        double4 o = xch[(lane & ~7) + ((lane + 1) & 7)];
        v.x = o.x + 1e-9; v.y = o.y; v.z = o.z; v.w = o.w;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    t1 = clock64(); if (lane == 0) cyc[5] = t1 - t0;
    a += v.x;
    // 7. DPP stage: s_nop 1 + 2 dpp + add, dependent
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < 128; i++) {
        int lo = __double2loint(a), hi = __double2hiint(a), olo, ohi;
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(olo), "=&v"(ohi) : "v"(lo), "v"(hi));
        a = a * 0.5 + __hiloint2double(ohi, olo) * 0.5;
    }
    t1 = clock64(); if (lane == 0) cyc[6] = t1 - t0;
    // 8. 8 independent fma streams (issue rate)
    double s0=a,s1=a+1,s2=a+2,s3=a+3,s4=a+4,s5=a+5,s6=a+6,s7=a+7;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < N / 8; i++) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(s0) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(s1) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(s2) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(s3) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(s4) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(s5) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(s6) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(s7) : "v"(b), "v"(c));
    }
    t1 = clock64(); if (lane == 0) cyc[7] = t1 - t0;
    // 9. v_cndmask dependent (32-bit VALU) chain
    int q = lane;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(q) : "v"(lane));
    t1 = clock64(); if (lane == 0) cyc[8] = t1 - t0;
    // 10. scalar load latency (dependent pointer chase impossible; measure single s_load + wait)
    out[lane] = a + r + s0+s1+s2+s3+s4+s5+s6+s7 + q;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16 * 8);
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.0); hipDeviceSynchronize(); }
    long long h[16]; hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
    // clock64 = s_memtime: counts at 100MHz? print raw and per-op
    printf("fma dep chain      : %lld ticks / %d = %.2f\n", h[0], N, (double)h[0] / N);
    printf("fma 4 indep        : %lld ticks / %d = %.2f\n", h[1], N, (double)h[1] / N);
    printf("rsq dep chain      : %lld ticks / 64 = %.2f\n", h[2], (double)h[2] / 64);
    printf("mul dep chain      : %.2f\n", (double)h[3] / N);
    printf("add dep chain      : %.2f\n", (double)h[4] / N);
    printf("LDS wr->rd RTT     : %.2f\n", (double)h[5] / 64);
    printf("DPP stage (nop+2dpp+2mul/fma): %.2f\n", (double)h[6] / 128);
    printf("fma 8 indep        : %.2f\n", (double)h[7] / N);
    printf("v_add_u32 dep      : %.2f\n", (double)h[8] / N);
    return 0;
}
