// grp_device.h -- lane-group primitives for gfx950 (wave64).
//
// An "eval group" is G consecutive lanes of one wavefront (G = 2,4,8,16) that cooperate on one
// CARMA log-density evaluation; lane r of the group owns row r of the p x p state covariance.
// All cross-lane traffic is DPP (v_mov_b32_dpp) for the butterflies and ds_bpermute for
// broadcasts with a run-time source lane; nothing here uses __syncthreads().
#pragma once
#include <hip/hip_runtime.h>

#include "carma_types.h"
#include "carma_row_asm.h"

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
    // true when b holds in every lane of the WAVE (a scalar: branches on it are uniform)
    CARMA_DEV static bool wave_all(bool b) { return __ballot(!b) == 0ull; }
    // value held by the neighbouring lane (lane ^ 1): the other member of a root pair
    CARMA_DEV static double partner(double v) { return dpp_xor1(v); }
    // value of v held by lane J of this group, J a compile-time constant: DPP, no LDS round trip.
    //   G = 16: one row_newbcast (v_mov_b64_dpp for doubles); G = 8 / 4: one per group of the 16-lane
    //   row, selected with the bank mask (a bank is 4 lanes); G = 2: a quad permutation.
    template <int J, class T>
    CARMA_DEV static T bcast_c(T v)
    {
        static_assert(J >= 0 && J < G, "lane of the group");
        if constexpr (G == 16) {
            return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0xf, true);
        } else if constexpr (G == 8) {
            T o = __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0x3, false);
            return __builtin_amdgcn_update_dpp(o, v, 0x150 + 8 + J, 0xf, 0xc, false);
        } else if constexpr (G == 4) {
            T o = __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0x1, false);
            o = __builtin_amdgcn_update_dpp(o, v, 0x150 + 4 + J, 0xf, 0x2, false);
            o = __builtin_amdgcn_update_dpp(o, v, 0x150 + 8 + J, 0xf, 0x4, false);
            return __builtin_amdgcn_update_dpp(o, v, 0x150 + 12 + J, 0xf, 0x8, false);
        } else {
            static_assert(G == 2, "group size");
            return __builtin_amdgcn_update_dpp(v, v, J | (J << 2) | ((2 + J) << 4) | ((2 + J) << 6), 0xf, 0xf, true);
        }
    }
    // value of v held by lane j of this group (j identical in every lane of the group).  Every call
    // site sits in a fully unrolled loop, so j is a constant by the time the switch is simplified and
    // one DPP case survives; a run-time j falls through to ds_bpermute.
    template <class T>
    CARMA_DEV T bcast_any(T v, int j) const
    {
        switch (j) {
#define CARMA_BC_CASE(J) \
    case J:              \
        if constexpr (J < G) return bcast_c<(J < G ? J : 0)>(v); \
        break;
            CARMA_BC_CASE(0) CARMA_BC_CASE(1) CARMA_BC_CASE(2) CARMA_BC_CASE(3) CARMA_BC_CASE(4) CARMA_BC_CASE(5)
            CARMA_BC_CASE(6) CARMA_BC_CASE(7) CARMA_BC_CASE(8) CARMA_BC_CASE(9) CARMA_BC_CASE(10) CARMA_BC_CASE(11)
            CARMA_BC_CASE(12) CARMA_BC_CASE(13) CARMA_BC_CASE(14) CARMA_BC_CASE(15)
#undef CARMA_BC_CASE
            default: break;
        }
        return __shfl(v, gbase() + j, 64);
    }
    // bcast_u: for call sites in fully unrolled loops (constant j -> DPP); bcast: run-time j (ds_bpermute)
    CARMA_DEV double bcast_u(double v, int j) const { return bcast_any<double>(v, j); }
    CARMA_DEV int bcast_iu(int v, int j) const { return bcast_any<int>(v, j); }
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
    // --- one evaluation per 16-lane DPP row (G = 16): DPP blocks as single inline-asm statements (carma_row_asm.h,
    // generated by tools/gen_row_asm.py).  "x@j" is the value of x held by lane j of the row, applied as a DPP row
    // broadcast on the fmac's source.
    //   mm_j = c@j D_j - s@j D_{j^1}
    template <int P>
    CARMA_DEV void row_colmix(double (&mm)[P], double c, double s, const double (&D)[P]) const
    {
        static_assert(G == 16, "row broadcast");
        RowAsm<P>::colmix(mm, c, s, D);
    }

    // also a scheduling barrier: the LDS operations issued so far stay ahead of what follows
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
