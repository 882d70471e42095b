// grp_device.h -- lane-group primitives for gfx950 (wave64).
//
// An "eval group" is G consecutive lanes of one wavefront (G = 2,4,8,16) that cooperate on one
// CARMA log-density evaluation; lane r of the group owns row r of the p x p state covariance.
// All cross-lane traffic is DPP (v_mov_b32_dpp) for the butterflies and ds_bpermute for
// broadcasts with a run-time source lane; nothing here uses __syncthreads().
#pragma once
#include <hip/hip_runtime.h>

#include "carma_types.h"

#define CARMA_DEV __device__ __forceinline__

namespace carma {

// DPP controls (gfx9): quad_perm[a,b,c,d] = a | b<<2 | c<<4 | d<<6
constexpr int DPP_QUAD_XOR1 = 0xB1;   // [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;   // [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;   // lane i <-> 7-i inside each 8 lanes
constexpr int DPP_ROW_MIRROR = 0x140;        // lane i <-> 15-i inside each 16 lanes

// 64-bit DPP move (two v_mov_b32_dpp).  bound_ctrl with full row/bank masks makes the `old` operand
// dead, so no extra v_mov is emitted, and the compiler owns the "VALU write -> DPP read" hazard: it
// fills the two wait states with independent instructions instead of a fixed s_nop (8 cycles each
// in a single-wave instruction stream -- measured, tools/ubench/ub3.hip).
#define CARMA_DPP_MOV64(NAME, CTRL) \
    CARMA_DEV double NAME(double v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, true); }
CARMA_DPP_MOV64(dpp_xor1, DPP_QUAD_XOR1)
CARMA_DPP_MOV64(dpp_xor2, DPP_QUAD_XOR2)
CARMA_DPP_MOV64(dpp_half_mirror, DPP_ROW_HALF_MIRROR)
CARMA_DPP_MOV64(dpp_mirror, DPP_ROW_MIRROR)

template <int G>
struct Grp {
    static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "group size");

    // scratch for the per-step all-gather: one 32-byte slot per lane, in LDS
    double4* xch;      // points at this WAVE's 64 slots
    int lane64;        // lane id inside the wave
    double2* xch2;     // second exchange array (16 B per lane), this WAVE's 64 slots

    CARMA_DEV int lane() const { return lane64 & (G - 1); }
    CARMA_DEV int gbase() const { return lane64 & ~(G - 1); }

    // Butterfly all-reduce; every lane of the group ends with the bit-identical total.
    CARMA_DEV static double sum(double v)
    {
        if (G >= 2) v += dpp_xor1(v);
        if (G >= 4) v += dpp_xor2(v);
        if (G >= 8) v += dpp_half_mirror(v);
        if (G >= 16) v += dpp_mirror(v);
        return v;
    }
    CARMA_DEV static double max(double v)
    {
        if (G >= 2) v = fmax(v, dpp_xor1(v));
        if (G >= 4) v = fmax(v, dpp_xor2(v));
        if (G >= 8) v = fmax(v, dpp_half_mirror(v));
        if (G >= 16) v = fmax(v, dpp_mirror(v));
        return v;
    }
    // value held by the neighbouring lane (lane ^ 1): the other member of a root pair
    CARMA_DEV static double partner(double v) { return dpp_xor1(v); }
    // value of v held by lane j of this group (j identical in every lane of the group)
    CARMA_DEV double bcast(double v, int j) const { return __shfl(v, gbase() + j, 64); }
    CARMA_DEV int bcast_i(int v, int j) const { return __shfl(v, gbase() + j, 64); }

    // Per-step exchange: every lane publishes 4 doubles and then reads lane j's 4 doubles.
    // LDS operations of one wave execute in issue order, so no barrier is needed; the fences
    // only stop the compiler from moving the loads above the store.
    CARMA_DEV void publish(double a, double b, double c, double d) const
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        xch[lane64] = make_double4(a, b, c, d);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    CARMA_DEV double4 peek(int j) const { return xch[gbase() + j]; }
    // Slim exchange: one double per lane (same storage), read back two lanes at a time.
    CARMA_DEV void publishk(double a) const
    {
        static_assert(G >= 2, "pairs");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        reinterpret_cast<double*>(xch)[lane64] = a;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    CARMA_DEV void peekk2(int i, double& a, double& b) const
    {
        const double2 v = reinterpret_cast<const double2*>(xch)[(gbase() >> 1) + i];
        a = v.x;
        b = v.y;
    }
    CARMA_DEV void publish2(double a, double b) const
    {
        xch2[lane64] = make_double2(a, b);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    CARMA_DEV Cx peek2(int j) const
    {
        double2 v = xch2[gbase() + j];
        return Cx{v.x, v.y};
    }
    // also a scheduling barrier: the LDS operations issued so far stay ahead of what follows
    // --- one evaluation per 16-lane DPP row (G = 16) -----------------------------------------
    // acc += (v held by lane J of this row) * mul: DPP row broadcast folded into the FP64 fmac.
    template <int J>
    CARMA_DEV void fmac_row(double& acc, double v, double mul) const
    {
        static_assert(G == 16 && J >= 0 && J < 16, "row broadcast");
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc)
                     : "v"(v), "v"(mul), "n"(J));
    }
    template <int J>
    CARMA_DEV void fnmac_row(double& acc, double v, double mul) const
    {
        static_assert(G == 16 && J >= 0 && J < 16, "row broadcast");
        asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc)
                     : "v"(v), "v"(mul), "n"(J));
    }
    // Two wait states between the VALU writes of a and b and the DPP reads that follow (the hazard
    // recogniser does not see into inline asm); the operands tie the nop behind their producers.
    CARMA_DEV void row_guard(double& a, double& b) const { asm volatile("s_nop 1" : "+v"(a), "+v"(b)); }

    CARMA_DEV void done_reading() const
    {
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // Make this wave's earlier LDS/global stores visible to its later loads (other lanes of the
    // same wave).  Hardware executes one wave's memory instructions in order; this is only a
    // compiler fence.
    CARMA_DEV void sync() const
    {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
};

}  // namespace carma
