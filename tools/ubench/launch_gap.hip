// Launch overhead of back-to-back kernels on one stream (grid = 256 workgroups x 256 threads, 42 KiB dynamic LDS, like
// k_logdens_carma_p3l at 1024 evaluations) and the shader clock under an FP64 load (clock64 vs the 100 MHz wall clock).
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/launch_gap.hip -o /tmp/launch_gap && /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void k_empty(double* out)
{
    extern __shared__ double lds[];
    if (out == nullptr) lds[threadIdx.x] = 1.0;
}

__global__ void k_clock(long long* out, double* sink, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.000001;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        a = fma(a, b, 0.5);
        a = fma(a, b, -0.5);
        a = fma(a, b, 0.25);
        a = fma(a, b, -0.25);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = w1 - w0;
    }
    if (a == 12345.678) *sink = a;
}

int main()
{
    double* d;
    long long* dc;
    hipMalloc(&d, 8);
    hipMalloc(&dc, 16);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int grid : {1, 256, 768}) {
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 42800, st, d);
        hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        const int N = 2000;
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 42800, st, d);
        hipStreamSynchronize(st);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("empty kernel, grid %4d x 256 threads, 42 KiB LDS: %.2f us per back-to-back launch\n", grid, us);
    }
    {   // the same 2000 launches as a captured graph of 100 kernel nodes, replayed 20 times
        hipGraph_t graph;
        hipGraphExec_t exec;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 42800, st, d);
        hipStreamEndCapture(st, &graph);
        hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        hipGraphLaunch(exec, st);
        hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 20; i++) hipGraphLaunch(exec, st);
        hipStreamSynchronize(st);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
        printf("empty kernel, grid  256, as a hipGraph of 100 nodes: %.2f us per kernel\n", us);
    }
    hipLaunchKernelGGL(k_clock, dim3(1024), dim3(64), 0, st, dc, d, 200000);
    hipStreamSynchronize(st);
    long long h[2];
    hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost);
    printf("clock64 %lld ticks over %lld wall ticks (100 MHz): shader clock %.0f MHz under a chip-wide dependent FP64 FMA chain; "
           "%.2f clocks per dependent FMA\n", h[0], h[1], 100.0 * h[0] / h[1], (double)h[0] / (4.0 * 200000));
    return 0;
}
