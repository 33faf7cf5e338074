/*
 * TEST INFRASTRUCTURE ONLY: plain-C stand-ins for spice_amd/csrc/target/spice_target.h, so that
 * the kernel sources compile with g++ for the CPU wave simulator (tests/sim/Makefile puts this
 * directory, not the product's, on the include path).  Same names, same results; what the
 * gfx950 versions do with DPP, bit-field instructions, LDS counters and packed arithmetic is
 * done here with shuffles, masks, the simulator's fiber barriers and two scalar operations.
 */
#ifndef SPICE_TARGET_H
#define SPICE_TARGET_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define SPICE_GLOBAL
#define SPICE_CONST
#define SPICE_WAVES_PER_EU(n)
#define SPICE_OPAQUE_VGPR(v) ((void)(v))
/* the product leaves these variables unset where nothing reads them; here they are set, so
 * that a compare of two unread values never depends on what the stack held */
#define SPICE_LEFT_UNSET(...) do { __VA_ARGS__ } while (0)
#define SPICE_SETPRIO(n) ((void)(n))
#define SPICE_LAUNCH_TIMED(kernel, grid, block, stream, ev_begin, ev_end, ...) \
    do {                                                                      \
        hipEventRecord(ev_begin, stream);                                     \
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);      \
        hipEventRecord(ev_end, stream);                                       \
    } while (0)

static inline uint32_t wave_scan_inclusive(uint32_t v, int lane)
{
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, (unsigned)d);
        v += lane >= d ? o : 0u;
    }
    return v;
}

static inline uint32_t wave_broadcast(uint32_t v, int src)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, src);
}

static inline uint32_t spice_low_bits(uint32_t x, int n)
{
    return x & ((1u << n) - 1);
}

static inline void spice_group_barrier(uint32_t *ctr, uint32_t &target, int lane, int nwaves)
{
    (void)target;
    (void)lane;
    sim_group_barrier(ctr, nwaves * 64);
}

static inline void spice_signal(uint32_t *ctr, int lane)
{
    if (lane == 0) {
        *ctr += 1;
        sim_note_progress();
    }
}

static inline void spice_wait_for(uint32_t *ctr, uint32_t target)
{
    while (*(volatile uint32_t *)ctr < target) {
        sim_yield();
    }
}

struct spice_f2 {
    float x, y;
    spice_f2() : x(0.f), y(0.f) {}
    spice_f2(float v) : x(v), y(v) {}
    spice_f2(float a, float b) : x(a), y(b) {}
};
static inline spice_f2 spice_fma2(spice_f2 a, spice_f2 b, spice_f2 c)
{
    /* every term of the callers' sums is exact in single precision, so rounding once or
     * twice gives the same number */
    return spice_f2(a.x * b.x + c.x, a.y * b.y + c.y);
}

#endif
