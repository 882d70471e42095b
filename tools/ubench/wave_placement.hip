// Where do the waves of co-resident workgroups land?  Grid of NWG workgroups x 256 threads with 42 KiB of dynamic LDS
// (k_logdens_carma_p3l's shape); every wave records its HW_ID / XCC_ID and spins long enough for the whole grid to be
// resident.  Prints, per (xcc, se, cu), the SIMD of every (workgroup, wave).
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/wave_placement.hip -o /tmp/wave_placement && /tmp/wave_placement 512 [1 = cooperative launch]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void k_where(unsigned* out, int spin)
{
    extern __shared__ double lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    double a = threadIdx.x;
    for (int i = 0; i < spin; i++) a = fma(a, 1.0000001, 0.5);
    if (a == 1.2345) lds[threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0) {
        out[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = hw;
        out[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = xcc;
    }
}

int main(int argc, char** argv)
{
    const int nwg = argc > 1 ? atoi(argv[1]) : 512;
    const bool coop = argc > 2 && atoi(argv[2]) != 0;         // second argument 1: cooperative launch (the row sampler's)
    unsigned* d;
    hipMalloc(&d, sizeof(unsigned) * 8 * nwg);
    if (coop) {
        int spin = 20000;
        void* args[] = {&d, &spin};
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_where), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&k_where), dim3(nwg), dim3(256), args, 47 * 1024, 0);
        printf("cooperative launch: %s\n", hipGetErrorString(e));
    } else
        hipLaunchKernelGGL(k_where, dim3(nwg), dim3(256), 42 * 1024, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(8 * nwg);
    hipMemcpy(h.data(), d, sizeof(unsigned) * 8 * nwg, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<std::pair<int, int>>> cu;      // key -> (wg*4+wave, simd)
    for (int i = 0; i < 4 * nwg; i++) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned simd = (hw >> 4) & 3, cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[(xcc << 16) | (se << 8) | (sh << 4) | cuid].push_back({i, (int)simd});
    }
    int shown = 0;
    std::map<std::string, int> pat;
    for (auto& kv : cu) {
        std::string s;
        char b[64];
        for (auto& e : kv.second) {
            snprintf(b, sizeof b, " wg%d.w%d@%d", e.first / 4, e.first % 4, e.second);
            s += b;
        }
        std::string sig;
        for (auto& e : kv.second) sig += char('0' + e.second);
        pat[sig]++;
        if (shown++ < 6) printf("xcc %u se %u sh %u cu %2u :%s\n", kv.first >> 16, (kv.first >> 8) & 0xff, (kv.first >> 4) & 0xf, kv.first & 0xf, s.c_str());
    }
    printf("%zu CUs used; SIMD patterns (waves in wg,wave order):\n", cu.size());
    for (auto& kv : pat) printf("  %s x %d\n", kv.first.c_str(), kv.second);
    // which workgroup indices share a CU
    int k = 0;
    for (auto& kv : cu) {
        if (k++ >= 4) break;
        printf("cu key %06x wgs:", kv.first);
        for (size_t i = 0; i < kv.second.size(); i += 4) printf(" %d", kv.second[i].first / 4);
        printf("\n");
    }
    return 0;
}
