// carma_types.h -- plain structs shared by host code, kernels and the CPU lane emulator.
#pragma once

namespace carma {

// Prior bounds of CARMA_Base (src/include/carpack.hpp:241-247, set by SetPrior :201-207)
struct Prior {
    double max_stdev, max_freq, min_freq, measerr_dof;
};

struct Cx {
    double re, im;
};

// The wave pipeline (carma_pipe3l.h) runs full 16-datum chunks unrolled and a last, shorter chunk as a rolled loop that is
// ~30 % slower per datum -- and the mean wave's last chunk is what the launch ends on.  A last chunk of 11..15 data is
// therefore completed to 16 with NEUTRAL pad data (the series in HBM always carries 16 pad records behind the real ones:
// dt = 0, y = y_last, yerr^2 = 1): the producers write zero ring entries for them, so the state does not move (k~ = 0
// exactly) and each pad adds exactly var = scale + s0, innov = y_last - mu to the sums, which the kernel takes out again
// (yerr^2 = 1 rather than 0 keeps var positive for sigma_y = 0, which the reference's bounds admit).
constexpr int P3L_PAD_RECORDS = 16;
// layout of the series in HBM: double4 records[n + 16] {dt, y, yerr^2, t}, then double yerr2[n + 16], then double y[n + 16]
inline
#if defined(__HIPCC__)
    __host__ __device__
#endif
    int p3l_pad(int n)
{
    const int r = n % 16;
    return r >= 11 ? 16 - r : 0;
}

// What the launchers know about the series of a context (carma_ctx_create), as bits:
//   SERIES_REPEATED_DT  >= 25 % of the time steps equal their predecessor (the regular-cadence variants of the throughput kernels)
//   SERIES_WINDOW_OK    the windowed wave pipeline (carma_pipew.h) suits it: a chunk of 16 - p consecutive data is, for 90 % of the
//                       chunks, no longer than the SHORTEST window the prior admits (half of LIM_RE / (2 pi max_freq)).  Otherwise the
//                       rows with short windows cut their chunks and re-base at every chunk start, and a launch -- or a ladder's
//                       rendezvous -- waits for them: BASELINE configs[3]'s series (min dt 0.1, median 1.1) ran 464 instead of 780
//                       sampler iterations/s on that pipeline.
//   SERIES_WINDOW2_OK / _SMALL   the TWO-SIDED window pipeline suits it (round 6, profiles/r06/window_criterion_v*.txt).  Measure:
//                       r = (chunks a row with the SHORTEST window needs: greedy, at most 16 - p data and at most that window per
//                       chunk) / ceil(n / (16 - p)).  README series r = 1.0-1.08, OGLE-LMC-LPV-00007 (seasons; 20-55 % of the spans
//                       over the window) 1.14-1.41: two-sided 1.3-1.8 x faster than the one-datum pipeline at every order, launch
//                       size and ensemble tried; configs[3]'s time steps r = 2.5-3.9: 17-20 % faster with a CU per workgroup, 5-15 %
//                       slower with two; one close pair of data in the README series (max_freq x 100) r = 9-13: 2-2.7 x slower.
//                       OK: r <= 2.  SMALL: r <= 3.5, used by launches of at most one workgroup per CU.
constexpr int SERIES_REPEATED_DT = 1, SERIES_WINDOW_OK = 2, SERIES_WINDOW2_OK = 4, SERIES_WINDOW2_SMALL = 8;

// Arguments of one launch of the persistent PT kernel.
struct PtLaunch {
    int d, q, n;                 // parameter dimension, MA order, series length
    int T, R;                    // temperatures / replicas held by THIS launch (= this GPU)
    int maxiter;                 // RAM adapts while iteration < maxiter (= burn-in, carmcmc.cpp:149)
    unsigned long long iter0;    // global index of the first iteration of this launch
    int niter;                   // iterations to run
    int do_exchange;             // run the local hot->cold swap sweep after every iteration
    int save_thin;               // 0: do not save; k: save chain 0 after every k-th iteration
    long save_offset;            // index of the first sample this launch writes
    long sample_cap;             // samples per replica the output buffers can hold
    unsigned seed0, seed1;
    unsigned slot0;              // global temperature index of local slot 0   (ladder sharding)
    unsigned T_global;           // global number of temperatures
    unsigned replica0;           // global index of local replica 0            (replica sharding)
};

// Cross-workgroup exchange state of the row-variant PT kernel (k_pt_row): a ladder is spread over `wpl` workgroups, which
// exchange their chains once per iteration through global memory -- as SELF-VALIDATING words (k_pt_row, "tagged staging").
struct PtRowSync {
    unsigned long long* stage;   // [2 buffers][2 copies][R*T*(d+1)] tagged 64-bit words: theta[d] and the log-posterior of every chain
    unsigned* abort_flag;    // [1] set when a rendezvous timed out (the launch then ends early)
    unsigned long long epoch;    // launch counter of this sampler (never repeats): part of the tags
    int wpl;                 // workgroups per ladder = ceil(T / 4)
    int ncu;                 // compute units of the device: workgroups i, i + ncu, i + 2 ncu share a CU
    int xcd_map;             // > 1: a ladder's workgroups sit xcd_map blocks apart (same XCD), see k_pt_row
    int window_ok;           // the context's series_flags(): SERIES_WINDOW_OK (one-sided window pipeline), SERIES_WINDOW2_OK / _SMALL
    int rot;                 // which wave plays which part in the second (bits 0-7) and third (bits 8-15) workgroup of a CU:
                             // 2 bits per wave, set by the launcher (the placement differs between launch kinds)
};

}  // namespace carma
