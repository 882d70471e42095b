// Check: bank-masked v_fmac_f64_dpp followed by another access to the same destination register with the
// other bank mask, with 0 / 1 / 2 instructions in between (see tools/gen_row_asm.py, no_adjacent_same_dst).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void kern(double* out)
{
    int lane = threadIdx.x;
    double x = 1.0 + lane, one = 1.0, d = 0.5;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, f = 3.0;
    asm volatile("s_nop 4\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xc\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0x3"
                 : "+v"(a0) : "v"(x), "v"(one));
    asm volatile("s_nop 4\n\t"
                 "v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\t"
                 "v_fmac_f64 %1, %4, %4\n\t"
                 "v_fmac_f64_dpp %0, %2, %3 row_newbcast:1 row_mask:0xf bank_mask:0xc\n\t"
                 "v_fmac_f64 %1, %4, %4\n\t"
                 "v_fmac_f64_dpp %0, %2, %3 row_newbcast:2 row_mask:0xf bank_mask:0x3"
                 : "+v"(a1), "+v"(d) : "v"(x), "v"(one), "v"(f));
    asm volatile("s_nop 4\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\t"
                 "s_nop 0\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xc\n\t"
                 "s_nop 0\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0x3"
                 : "+v"(a2) : "v"(x), "v"(one));
    out[lane] = a0; out[64 + lane] = a1; out[128 + lane] = a2 + 0 * d;
}
int main()
{
    double* d; (void)hipMalloc(&d, 192 * 8);
    kern<<<1, 64>>>(d);
    double h[192]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    // expected: lanes 0-7 of a row: x@0 + x@2 ; lanes 8-15: x@1
    int bad[3] = {0, 0, 0};
    for (int v = 0; v < 3; v++)
        for (int l = 0; l < 64; l++) {
            int row = l & ~15;
            double want = (l & 8) ? (1.0 + row + 1) : (1.0 + row + 0) + (1.0 + row + 2);
            if (h[64 * v + l] != want) bad[v]++;
        }
    printf("wrong lanes: back-to-back %d, one FP64 instruction between %d, s_nop 0 between %d\n", bad[0], bad[1], bad[2]);
    for (int l = 0; l < 16; l += 1) printf("lane %2d: %g %g %g\n", l, h[l], h[64 + l], h[128 + l]);
    return 0;
}
