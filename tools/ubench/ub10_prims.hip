// micro-benchmark (round 5): the primitives a BLOCKED form of the covariance recursion would be built from.
//   * v_mfma_f64_4x4x4_4b_f64 / v_mfma_f64_16x16x4_f64: dependent latency (C chained), issue interval of independent ones,
//     and the operand / result lane maps (one-hot probing: which lane of D receives A[la] * B[lb])
//   * independent (not chained) v_fmac_f64_dpp row_newbcast, v_fma_f64, v_readlane_b32, v_permlane{16,32}_swap_b32:
//     issue interval
// One wave; cycles per instruction from clock64 around 64 unrolled copies (ub7's method).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define REP 64
typedef double double4v __attribute__((ext_vector_type(4)));

__global__ void k_time(double* out, long long* cyc, double seed)
{
    const int lane = threadIdx.x;
    double a = seed + lane * 1e-3, b = 1.0000001, c = 1e-9;
    double x0 = 0.1, x1 = 0.2, x2 = 0.3, x3 = 0.4, x4 = 0.5, x5 = 0.6, x6 = 0.7, x7 = 0.8;
    long long t0, t1;
    int k = 0;
#define TIME(BODY)                                            \
    __builtin_amdgcn_sched_barrier(0);                        \
    asm volatile("s_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(a), "+v"(x0), "+v"(x7)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                        \
    _Pragma("unroll") for (int i = 0; i < REP; i++) { BODY }  \
    __builtin_amdgcn_sched_barrier(0);                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                        \
    if (lane == 0) cyc[k] = t1 - t0;                          \
    k++;
    // 0: dependent 4x4x4 MFMA (C = previous D)
    TIME(a = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, a, 0, 0, 0);)
    // 1: eight independent accumulators, 4x4x4 (issue interval = /8)
    TIME(x0 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, x0, 0, 0, 0); x1 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, x1, 0, 0, 0);
         x2 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, x2, 0, 0, 0); x3 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, x3, 0, 0, 0);
         x4 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, x4, 0, 0, 0); x5 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, x5, 0, 0, 0);
         x6 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, x6, 0, 0, 0); x7 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, x7, 0, 0, 0);)
    // 2: dependent through the A operand (A = previous D): the blocked recursion chains results into operands
    TIME(a = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);)
    // 3: MFMA result consumed by a VALU op and fed back as C (MFMA -> VALU -> MFMA round trip)
    TIME(a = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, a, 0, 0, 0) * b;)
    // 4: eight independent v_fmac_f64_dpp row_newbcast (issue interval = /8)
    TIME(asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                      "v_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                      "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                      "v_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                      "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                      "v_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                      "v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                      "v_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf"
                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                      : "v"(b), "v"(c));)
    // 5: eight independent v_fma_f64
    TIME(asm volatile("v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\t"
                      "v_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\tv_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7"
                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                      : "v"(b), "v"(c));)
    // 6: v_readlane_b32 pair + dependent v_fma_f64 with the SGPR pair as an operand (broadcast of one double)
    TIME(asm volatile("v_readlane_b32 s20, %1, 3\n\tv_readlane_b32 s21, %2, 3\n\tv_fma_f64 %0, s[20:21], %3, %0"
                      : "+v"(a)
                      : "v"(__double2loint(b)), "v"(__double2hiint(b)), "v"(c)
                      : "s20", "s21");)
    // 7: eight v_readlane_b32 (independent)
    TIME(asm volatile("v_readlane_b32 s20, %0, 1\n\tv_readlane_b32 s21, %0, 2\n\tv_readlane_b32 s22, %0, 3\n\tv_readlane_b32 s23, %0, 4\n\t"
                      "v_readlane_b32 s24, %0, 5\n\tv_readlane_b32 s25, %0, 6\n\tv_readlane_b32 s26, %0, 7\n\tv_readlane_b32 s27, %0, 8"
                      :
                      : "v"(__double2loint(b))
                      : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    // 8: dependent 16x16x4 MFMA
    {
        double4v acc = {a, a, a, a};
        TIME(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0); if (i == REP - 1) a += acc.x + acc.y + acc.z + acc.w;)
        // 9: two independent 16x16x4 accumulators
        double4v acc2 = {c, c, c, c};
        TIME(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc2, 0, 0, 0);
             if (i == REP - 1) a += acc.x + acc2.y;)
    }
    // 10: permlane swaps (b32), independent pairs
    {
        int p0 = lane, p1 = lane * 3, p2 = lane * 5, p3 = lane * 7;
        TIME(asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
        a += p0 + p1 + p2 + p3;
    }
    // 11: dependent chain v_mov_b64_dpp row_newbcast -> v_rcp_f64 -> 2 fma (Newton) -> v_mul: the pivot block
    TIME(asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_rcp_f64 %1, %0\n\t"
                      "v_fma_f64 %2, -%0, %1, 1.0\n\tv_fma_f64 %1, %1, %2, %1\n\tv_mul_f64 %0, %1, %3"
                      : "+v"(a), "+v"(x0), "+v"(x1)
                      : "v"(b));)
    // 12: ds_bpermute_b32 dependent (address = result)
    {
        int q = lane;
        TIME(asm volatile("ds_bpermute_b32 %0, %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(q) : "v"(lane));)
        a += q;
    }
    out[lane] = a + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

// lane maps: D lanes (and, for 16x16x4, registers) that receive A[la] * B[lb]
__global__ void k_map4(int* map)   // [64][64] -> lane of D (or -1), 4x4x4_4b
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; la++)
        for (int lb = 0; lb < 64; lb++) {
            const double A = lane == la ? 1.0 : 0.0, B = lane == lb ? 1.0 : 0.0;
            const double D = __builtin_amdgcn_mfma_f64_4x4x4f64(A, B, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(D != 0.0);
            if (lane == 0) map[la * 64 + lb] = m ? (int)__builtin_ctzll(m) + 64 * (__builtin_popcountll(m) - 1) : -1;
        }
}
__global__ void k_map16(int* map)  // [64][64] -> reg * 64 + lane of D (or -1), 16x16x4
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; la++)
        for (int lb = 0; lb < 64; lb++) {
            const double A = lane == la ? 1.0 : 0.0, B = lane == lb ? 1.0 : 0.0;
            double4v C = {0, 0, 0, 0};
            const double4v D = __builtin_amdgcn_mfma_f64_16x16x4f64(A, B, C, 0, 0, 0);
            int r = -1;
            for (int v = 0; v < 4; v++) {
                const unsigned long long m = __ballot(D[v] != 0.0);
                if (m) r = v * 64 + (int)__builtin_ctzll(m);
            }
            if (lane == 0) map[la * 64 + lb] = r;
        }
}

int main()
{
    double* out;
    long long* cyc;
    int* map;
    (void)hipMalloc(&out, 64 * 8);
    (void)hipMalloc(&cyc, 32 * 8);
    (void)hipMalloc(&map, 64 * 64 * 4);
    for (int rep = 0; rep < 3; rep++) k_time<<<1, 64>>>(out, cyc, 1.5);
    long long h[32];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[] = {"dep mfma_f64_4x4x4_4b (C chain)", "8 indep mfma_f64_4x4x4_4b (per 8)", "dep mfma_f64_4x4x4_4b (A chain)",
                        "mfma4 -> v_mul -> mfma4", "8 indep v_fmac_f64_dpp (per 8)", "8 indep v_fma_f64 (per 8)",
                        "2 readlane + dep fma(sgpr)", "8 indep v_readlane_b32 (per 8)", "dep mfma_f64_16x16x4 (C chain)",
                        "2 indep mfma_f64_16x16x4 (per 2)", "permlane16_swap + permlane32_swap", "dpp bcast+rcp+newton+mul chain",
                        "dep ds_bpermute_b32"};
    for (int i = 0; i < 13; i++) printf("%-40s %.1f cycles/iter\n", nm[i], h[i] / (double)REP);
    static int hm[64 * 64];
    k_map4<<<1, 64>>>(map);
    (void)hipMemcpy(hm, map, sizeof hm, hipMemcpyDeviceToHost);
    printf("4x4x4_4b: D lane that receives A[la]*B[lb] (rows la = 0..63, '.' = none); +64 marks more than one lane\n");
    for (int la = 0; la < 64; la++) {
        printf("la %2d:", la);
        for (int lb = 0; lb < 64; lb++)
            if (hm[la * 64 + lb] >= 0) printf(" %d>%d", lb, hm[la * 64 + lb]);
        printf("\n");
    }
    k_map16<<<1, 64>>>(map);
    (void)hipMemcpy(hm, map, sizeof hm, hipMemcpyDeviceToHost);
    printf("16x16x4: (reg,lane) of D that receives A[la]*B[lb], first 20 lanes of A\n");
    for (int la = 0; la < 20; la++) {
        printf("la %2d:", la);
        int cnt = 0;
        for (int lb = 0; lb < 64; lb++)
            if (hm[la * 64 + lb] >= 0 && cnt++ < 6) printf(" %d>(r%d,l%d)", lb, hm[la * 64 + lb] / 64, hm[la * 64 + lb] % 64);
        printf("  [%d partners]\n", cnt);
    }
    return 0;
}
