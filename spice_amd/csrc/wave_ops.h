/*
 * Wavefront-wide primitives shared by the kernels.  On gfx950 they are DPP
 * sequences (a handful of VALU instructions); the CPU wave simulator used by
 * the tests sees plain shuffles with the same result.
 */
#ifndef SPICE_WAVE_OPS_H
#define SPICE_WAVE_OPS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

/* Inclusive prefix sum over the 64 lanes. */
__device__ __forceinline__ uint32_t wave_scan_inclusive(uint32_t v, int lane)
{
#ifdef SPICE_WAVE_SIM
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, (unsigned)d);
        v += lane >= d ? o : 0u;
    }
    return v;
#else
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
#endif
}

/* The value lane `src` (a constant) holds, in every lane. */
__device__ __forceinline__ uint32_t wave_broadcast(uint32_t v, int src)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, src);
}

#endif
