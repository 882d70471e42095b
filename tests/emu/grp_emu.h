// grp_emu.h -- CPU lane emulator for carma_core.h.  TEST HARNESS ONLY: it lets the CPU-only test
// suite execute the very same kernel-core source that carma_kernels.hip compiles for gfx950, with
// one OS thread per lane and a spin barrier at every cross-lane operation.  The product never
// loads it; it is not a fallback.
#pragma once
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <atomic>
#include <cmath>
#include <thread>
#include <vector>

#define CARMA_DEV inline

struct double4 {
    double x, y, z, w;
};
static inline double4 make_double4(double a, double b, double c, double d) { return {a, b, c, d}; }
using std::exp;
using std::fabs;
using std::fmax;
using std::fmin;
using std::fma;
using std::ldexp;
using std::rint;
static inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }
using std::pow;
using std::frexp;
using std::hypot;
using std::log;
using std::sqrt;
using std::sin;
using std::cos;

#include "../../carma_pack_amd/csrc/carma_types.h"

namespace carma {

struct SpinBarrier {
    std::atomic<int> count{0};
    std::atomic<int> gen{0};
    int n = 1;
    void wait()
    {
        int g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
            count.store(0, std::memory_order_relaxed);
            gen.fetch_add(1, std::memory_order_acq_rel);
        } else {
            int spins = 0;
            while (gen.load(std::memory_order_acquire) == g) {
                if (++spins > 2000) std::this_thread::yield();
            }
        }
    }
};

struct EmuShared {
    SpinBarrier bar;
    double slot[16];
    int islot[16];
    double4 xch[16];
    double xch2[16][2];
    double xk[16];
};

template <int G>
struct Grp {
    EmuShared* sh;
    int r;
    int lane() const { return r; }
    double xchg(double v, int partner) const
    {
        sh->slot[r] = v;
        sh->bar.wait();
        double o = sh->slot[partner];
        sh->bar.wait();
        return o;
    }
    // same butterfly partners as the DPP controls in grp_device.h
    // device: the whole wave; here a group is its own wave
    bool wave_all(bool b) const { return sum(b ? 0.0 : 1.0) == 0.0; }
    double sum(double v) const
    {
        if (G >= 2) v += xchg(v, r ^ 1);
        if (G >= 4) v += xchg(v, r ^ 2);
        if (G >= 8) v += xchg(v, (r & ~7) | (7 - (r & 7)));
        if (G >= 16) v += xchg(v, 15 - r);
        return v;
    }
    double bcast(double v, int j) const { return xchg(v, j); }
    double bcast_u(double v, int j) const { return xchg(v, j); }
    double partner(double v) const { return xchg(v, r ^ 1); }
    int bcast_iu(int v, int j) const { return bcast_i(v, j); }
    int bcast_i(int v, int j) const
    {
        sh->islot[r] = v;
        sh->bar.wait();
        int o = sh->islot[j];
        sh->bar.wait();
        return o;
    }
    void publish(double a, double b, double c, double d) const
    {
        sh->xch[r] = make_double4(a, b, c, d);
        sh->bar.wait();
    }
    double4 peek(int j) const { return sh->xch[j]; }
    void publishk(double a) const
    {
        sh->xk[r] = a;
        sh->bar.wait();
    }
    void peekk2(int i, double& a, double& b) const
    {
        a = sh->xk[2 * i];
        b = sh->xk[2 * i + 1];
    }
    void publish2(double a, double b) const
    {
        sh->xch2[r][0] = a;
        sh->xch2[r][1] = b;
        sh->bar.wait();
    }
    Cx peek2(int j) const { return Cx{sh->xch2[j][0], sh->xch2[j][1]}; }
    // DPP row blocks (device: carma_row_asm.h)
    template <int P>
    void row_colmix(double (&mm)[P], double c, double s, const double (&D)[P]) const
    {
        for (int j = 0; j < P; j++) mm[j] = fma(xchg(c, j), D[j], 0.0);
        for (int j = 0; j < (P & ~1); j++) mm[j] = fma(-xchg(s, j), D[j ^ 1], mm[j]);
    }
    void done_reading() const { sh->bar.wait(); }
    void sync() const { sh->bar.wait(); }
};

// run fn(grp) on G lane-threads
template <int G, class F>
void run_group(F fn)
{
    EmuShared sh;
    sh.bar.n = G;
    std::vector<std::thread> th;
    for (int r = 0; r < G; r++) th.emplace_back([&sh, r, &fn]() { Grp<G> g{&sh, r}; fn(g); });
    for (auto& t : th) t.join();
}

}  // namespace carma
