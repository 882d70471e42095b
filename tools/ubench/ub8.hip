// micro-benchmark: what fits into the shadow of v_rcp_f64?  One wave; a dependent rcp -> mul chain with k INDEPENDENT FP64 FMAs
// (and DPP fmacs) placed between the rcp and its consumer: if the time per iteration does not grow with k, those
// instructions issue while the reciprocal is still in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
__global__ void kern(double* out, long long* cyc, double seed)
{
    int lane = threadIdx.x;
    double a = seed, b = 1.0000001, c = 1e-9, x0 = 0.5, x1 = 0.25, x2 = 0.125, x3 = 0.0625;
    long long t0, t1;
    int k = 0;
#define TIME(BODY)                                   \
    t0 = clock64();                                  \
    _Pragma("unroll") for (int i = 0; i < REP; i++) { BODY } \
    t1 = clock64();                                  \
    if (lane == 0) cyc[k] = t1 - t0;                 \
    k++;
#define F0 "v_fma_f64 %1, %1, %5, %6\n\t"
#define F1 "v_fma_f64 %2, %2, %5, %6\n\t"
#define F2 "v_fma_f64 %3, %3, %5, %6\n\t"
#define F3 "v_fma_f64 %4, %4, %5, %6\n\t"
#define OPS : "+v"(a), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c)
    TIME(asm volatile("v_rcp_f64 %0, %0\n\ts_nop 0\n\tv_mul_f64 %0, %0, %5" OPS);)
    TIME(asm volatile("v_rcp_f64 %0, %0\n\t" F0 "v_mul_f64 %0, %0, %5" OPS);)
    TIME(asm volatile("v_rcp_f64 %0, %0\n\t" F0 F1 "v_mul_f64 %0, %0, %5" OPS);)
    TIME(asm volatile("v_rcp_f64 %0, %0\n\t" F0 F1 F2 "v_mul_f64 %0, %0, %5" OPS);)
    TIME(asm volatile("v_rcp_f64 %0, %0\n\t" F0 F1 F2 F3 "v_mul_f64 %0, %0, %5" OPS);)
    // independent rcp's back to back: the issue cost of the instruction itself
    TIME(asm volatile("v_rcp_f64 %1, %1\n\tv_rcp_f64 %2, %2\n\tv_rcp_f64 %3, %3\n\tv_rcp_f64 %4, %4" OPS);)
    // independent FMAs back to back
    TIME(asm volatile(F0 F1 F2 F3 "s_nop 0" OPS);)
    // the dependent tail of the covariance pass: rcp -> mul -> fmac, nothing in between / two FMAs behind the rcp
    TIME(asm volatile("v_rcp_f64 %0, %0\n\ts_nop 0\n\tv_mul_f64 %1, %0, %5\n\tv_fma_f64 %2, %0, %5, %6\n\tv_fma_f64 %0, %1, %2, %1" OPS);)
    TIME(asm volatile("v_rcp_f64 %0, %0\n\t" F2 F3 "v_mul_f64 %1, %0, %5\n\tv_fma_f64 %2, %0, %5, %6\n\tv_fma_f64 %0, %1, %2, %1" OPS);)
    out[lane] = a + x0 + x1 + x2 + x3;
}
int main()
{
    double* out; long long* cyc;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 16 * 8);
    for (int rep = 0; rep < 2; rep++) kern<<<1, 64>>>(out, cyc, 1.5);
    long long h[16];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[] = {"rcp, nop, mul (dependent)", "rcp, 1 fma, mul", "rcp, 2 fma, mul", "rcp, 3 fma, mul", "rcp, 4 fma, mul", "4 independent rcp",
                        "4 independent fma", "rcp nop mul fma fma (tail)", "rcp 2fma mul fma fma (tail)"};
    for (int i = 0; i < 9; i++) printf("%-30s %.1f cycles/iter\n", nm[i], h[i] / (double)REP);
    return 0;
}
