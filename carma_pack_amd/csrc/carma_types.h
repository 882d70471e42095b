// carma_types.h -- plain structs shared by host code, kernels and the CPU lane emulator.
#pragma once

namespace carma {

// Prior bounds of CARMA_Base (src/include/carpack.hpp:241-247, set by SetPrior :201-207)
struct Prior {
    double max_stdev, max_freq, min_freq, measerr_dof;
};

}  // namespace carma
