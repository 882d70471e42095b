// micro-benchmark: DEPENDENT-chain latencies of the instructions on the critical path of the co-rotating
// covariance step (carma_pipe3l.h): one wave, each instruction consumes the previous one's result.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
__global__ void kern(double* out, long long* cyc, double seed)
{
    int lane = threadIdx.x;
    double a = seed, b = 1.0000001, c = 1e-9, one = 1.0;
    long long t0, t1;
    int k = 0;
#define TIME(BODY)                                   \
    t0 = clock64();                                  \
    _Pragma("unroll") for (int i = 0; i < REP; i++) { BODY } \
    t1 = clock64();                                  \
    if (lane == 0) cyc[k] = t1 - t0;                 \
    k++;
    TIME(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
    TIME(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));)
    TIME(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));)
    // accumulate chain: dst depends on itself, DPP source constant
    TIME(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(c), "v"(b));)
    // DPP source is the previous result (the broadcast value was just written): needs the 2 wait states
    TIME(asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(c));)
    TIME(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));)
    TIME(asm volatile("v_rcp_f64 %0, %0\n\tv_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
    // fmac (plain VOP2) chain
    TIME(asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a) : "v"(c), "v"(b));)
    // partner exchange (two v_mov_b32_dpp, hazards by the compiler) + add: dependent
    TIME(a = a + __builtin_amdgcn_update_dpp(a, a, 0xB1, 0xf, 0xf, true);)
    out[lane] = a + one;
}
int main()
{
    double* out; long long* cyc;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 16 * 8);
    for (int rep = 0; rep < 2; rep++) kern<<<1, 64>>>(out, cyc, 1.5);
    long long h[16];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[] = {"dep v_fma_f64", "dep v_add_f64", "dep v_mul_f64", "dep v_fmac_f64_dpp (acc)", "dep s_nop1+fmac_dpp (src)", "dep v_rcp_f64",
                        "dep rcp + fma", "dep v_fmac_f64", "dep nop+2 mov_dpp+add"};
    for (int i = 0; i < 9; i++) printf("%-28s %.1f cycles/iter\n", nm[i], h[i] / (double)REP);
    return 0;
}
