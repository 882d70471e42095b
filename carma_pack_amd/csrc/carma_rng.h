// carma_rng.h -- counter-based RNG for the on-device sampler (Philox4x32-10, Salmon et al. 2011).
//
// The reference draws from one global, time-seeded boost::mt19937 (src/random.cpp:20), so its
// chains are not reproducible and trajectory-level parity is impossible; the device sampler
// instead keys every draw by (seed, global chain slot, iteration, purpose, index), which makes
// runs reproducible and lets both sides of a cross-GPU temperature swap compute the same
// uniform without exchanging it.
// Distributions needed on the path: U(0,1) (steps.cpp:48, steps.hpp:333), N(0,1) and
// Student-t with 8 dof (StudentProposal(8,1), carmcmc.cpp:139; random.cpp:158).
#pragma once
#include <stdint.h>

namespace carma {

struct Philox4 {
    uint32_t v[4];
};

CARMA_DEV uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

CARMA_DEV Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = mulhi32(M0, c0), lo0 = M0 * c0;
        uint32_t hi1 = mulhi32(M1, c2), lo1 = M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += W0;
        k1 += W1;
    }
    return Philox4{{c0, c1, c2, c3}};
}

// (0,1) with 53 random bits, never 0 or 1
CARMA_DEV double u01(uint32_t hi, uint32_t lo)
{
    uint64_t k = (((uint64_t)hi << 32) | lo) >> 11;
    return ((double)k + 0.5) * (1.0 / 9007199254740992.0);
}

struct RngKey {
    uint32_t k0, k1;      // seed
    uint32_t chain;       // global chain slot (replica * ntemps_global + temperature index)
};

// purposes
constexpr uint32_t RNG_PROPOSAL = 0, RNG_ACCEPT = 1, RNG_SWAP = 2, RNG_PATH = 3;

CARMA_DEV double rng_uniform(const RngKey& key, uint64_t iter, uint32_t purpose, uint32_t idx)
{
    Philox4 x = philox4x32_10((uint32_t)iter, (uint32_t)(iter >> 32), key.chain, (purpose << 24) | idx, key.k0, key.k1);
    return u01(x.v[0], x.v[1]);
}

// N(0,1) by Box-Muller (simulation of process paths, carma_simulate.h): one draw per (key, iter, idx)
CARMA_DEV double rng_normal(const RngKey& key, uint64_t iter, uint32_t idx)
{
    Philox4 a = philox4x32_10((uint32_t)iter, (uint32_t)(iter >> 32), key.chain, (RNG_PATH << 24) | idx, key.k0, key.k1);
    const double u1 = u01(a.v[0], a.v[1]), u2 = u01(a.v[2], a.v[3]);
    double sn, cs;
    sincos(6.283185307179586476925286766559 * u2, &sn, &cs);
    return sqrt(-2.0 * log(u1)) * cs;
}

// Student-t, nu = 8:  Z / sqrt(chi2_8 / 8), Z by Box-Muller, chi2_8 = -2 ln(U1 U2 U3 U4)
CARMA_DEV double rng_student_t8(const RngKey& key, uint64_t iter, uint32_t idx)
{
    Philox4 a = philox4x32_10((uint32_t)iter, (uint32_t)(iter >> 32), key.chain, (RNG_PROPOSAL << 24) | (2 * idx),
                              key.k0, key.k1);
    Philox4 b = philox4x32_10((uint32_t)iter, (uint32_t)(iter >> 32), key.chain, (RNG_PROPOSAL << 24) | (2 * idx + 1),
                              key.k0, key.k1);
    double u1 = u01(a.v[0], a.v[1]), u2 = u01(a.v[2], a.v[3]);
    double sn, cs;
    sincos(6.283185307179586476925286766559 * u2, &sn, &cs);
    double z = sqrt(-2.0 * log(u1)) * cs;
    const double s32 = 1.0 / 4294967296.0;
    double w = (((double)b.v[0] + 0.5) * s32) * (((double)b.v[1] + 0.5) * s32) * (((double)b.v[2] + 0.5) * s32) *
               (((double)b.v[3] + 0.5) * s32);
    double chi2 = -2.0 * log(w);
    return z / sqrt(chi2 / 8.0);
}

}  // namespace carma
