/*
 * Everything in the kernels that is written for gfx950 rather than in portable C++:
 * address-space qualifiers, occupancy attributes, DPP sequences, bit-field instructions,
 * wave priority, the LDS-counter synchronisation between wavefront groups, packed
 * single-precision pairs and self-timed launches.  The kernels include this header as
 * <spice_target.h>; the product is only ever built with this directory on the include
 * path (spice_amd/csrc/Makefile).  The CPU wave simulator the tests use has a header of the
 * same name with plain-C stand-ins (tests/sim/target), so the kernel sources themselves carry
 * no second code path.
 */
#ifndef SPICE_TARGET_H
#define SPICE_TARGET_H

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

/* Pointers that arrive inside a job record are generic to the compiler; retyped as global
 * (or constant: read through the scalar cache) memory, loads through them are plain global
 * loads and values read at a wave-uniform address stay wave-uniform. */
#define SPICE_GLOBAL __attribute__((address_space(1)))
#define SPICE_CONST __attribute__((address_space(4)))
/* exactly n wavefronts per SIMD: fixes the register budget (512 / n vector registers) */
#define SPICE_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))
/* the compiler forgets what it knows about v (it stays in a vector register) */
#define SPICE_OPAQUE_VGPR(v) asm volatile("" : "+v"(v))
/* variables that are read only where a flag says they were set are left unset here; the
 * simulator's version of this macro runs the statements, so that it compares no garbage */
#define SPICE_LEFT_UNSET(...)
/* priority of the calling wavefront when the SIMD picks whose instruction to issue (0..3) */
#define SPICE_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
/* A launch that records its own begin and end in two events (the dispatch's time stamps,
 * what a profiler reports): no event packets stand between it and its neighbours. */
#define SPICE_LAUNCH_TIMED(kernel, grid, block, stream, ev_begin, ev_end, ...) \
    hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, ev_begin, ev_end, 0, __VA_ARGS__)

/* Inclusive prefix sum over the 64 lanes: a DPP sequence, a handful of VALU instructions. */
__device__ __forceinline__ uint32_t wave_scan_inclusive(uint32_t v, int lane)
{
    (void)lane;
    int x = (int)v;
    /* rows of 16 lanes: shift right by 1, 2, 4, 8 (lanes without a source add 0) */
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    /* lane 15 of a row into the next row (rows 1 and 3), lane 31 into rows 2 and 3 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return (uint32_t)x;
}

/* The value lane `src` (a constant) holds, in every lane. */
__device__ __forceinline__ uint32_t wave_broadcast(uint32_t v, int src)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, src);
}

/* the n low bits of x (n < 32; none for n = 0): one bit-field extract */
__device__ __forceinline__ uint32_t spice_low_bits(uint32_t x, int n)
{
    return __builtin_amdgcn_ubfe(x, 0u, (uint32_t)n);
}

/* Barrier of the nwaves wavefronts that share a phase: every wavefront adds one to the
 * group's counter in LDS and waits until the counter has reached what it was before plus
 * nwaves.  The counter only grows; `target` is what the calling lane waits for next.  The
 * wavefronts of a group must all make the same barrier calls. */
__device__ __forceinline__ void spice_group_barrier(uint32_t *ctr, uint32_t &target, int lane, int nwaves)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) {
        atomicAdd(ctr, 1u);
    }
    target += (uint32_t)nwaves;
    while (*(volatile uint32_t *)ctr < target) {
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

/* One group of wavefronts waits for what another has done: the doers add to a counter in LDS
 * once their (global) stores are out, the waiters poll it. */
__device__ __forceinline__ void spice_signal(uint32_t *ctr, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) {
        atomicAdd(ctr, 1u);
    }
}
__device__ __forceinline__ void spice_wait_for(uint32_t *ctr, uint32_t target)
{
    while (*(volatile uint32_t *)ctr < target) {
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

/* two single-precision numbers worked on by one instruction (v_pk_fma_f32) */
typedef float spice_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ spice_f2 spice_fma2(spice_f2 a, spice_f2 b, spice_f2 c)
{
    return __builtin_elementwise_fma(a, b, c);
}

#endif
