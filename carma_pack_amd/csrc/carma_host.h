// carma_host.h -- host-side state behind the C ABI (carma_capi.hip, carma_pt_host.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "carma_launch.h"
#include "carma_types.h"

namespace carma {

struct PtState;

struct Ctx {
    int device = 0;
    int p = 0, q = 0, d = 0, n = 0;
    std::vector<double> t, y, yerr;   // after sort/dedup
    Prior pr{};
    double* d_series = nullptr;       // [n] records {dt, y, yerr^2, t}, resident in HBM
    double* d_theta = nullptr;        // staging for the host-pointer entry points
    double* d_out = nullptr;
    int cap = 0;
    hipStream_t stream = nullptr;
    PtState* pt = nullptr;
    int ensure_staging(int B);
};

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
int select_device(int device);
void pt_state_free(Ctx* c);

}  // namespace carma
