/*
 * GLZ parse for gfx950: one 256-lane workgroup per image, one image per
 * dictionary.
 *
 * The reference parser (server/glz_encode_tmpl.c:268-491) is a strictly serial
 * walk: at every "stop" it probes a single-slot hash table whose contents
 * depend on all earlier stops.  This kernel keeps that order -- the output is
 * byte-identical -- but splits the work on a window of GLZ_WIN consecutive
 * positions into three phases:
 *
 *   scan (all four wavefronts, one position per lane): run test and run
 *      length, 3-pixel hash, hash-table probe with reference validation
 *      (REF_SEG_IS_VALID, glz_encoder_dictionary_protected.h:176-184) and
 *      do_match's rules (glz_encode_tmpl.c:147-262) for the slot's content as
 *      it was when the window was entered, and the same for the two nearest
 *      earlier window positions with the same hash (found through an LDS
 *      bitmap keyed by the hash's low bits).  Every candidate is compared over
 *      its first GLZ_K pixels and the accept / reject decision is taken here;
 *   resolve (wavefront 0): visits the real stops in order.  The table slot a
 *      stop sees is the most recent store to it: the nearest earlier window
 *      position with that hash that really was a store (a stop or one of the
 *      two positions stored after a match), else what the scan read.  The walk
 *      goes a quarter (64 positions) at a time with the quarter's marks in
 *      64-bit scalar registers; on entering a quarter each lane settles the
 *      candidates that lie in finished quarters, so that an ordinary stop is
 *      one v_readlane plus scalar bit tests (glz_fast_stops, written in GCN
 *      assembly because one wavefront issues an instruction every ~5 cycles
 *      and each one counts).  Only candidates that outgrow GLZ_K pixels, runs
 *      that leave the window and slots whose two nearest predecessors were
 *      both skipped take glz_resolve_general with wave-wide compares;
 *   emit (all four wavefronts): sizes of the window's literals and matches are
 *      prefix-summed and every lane writes its own token; the window's table
 *      stores are issued here too, one per slot (the last store wins).
 */
#include <hip/hip_runtime.h>

#include "glz_device.h"
#include "runtime.h"
#include <spice_target.h>

#include "glz_parse_phases.h"

/* One workgroup per job: apply the window-table updates, write the header
 * (glz_encoder.c:264-276), parse, report the size. */
template <int TYPE>
/* Eight workgroups per CU, all a CU's 32 wave slots (64 VGPRs, 20 KB of LDS
 * each -- GlzShared is sized for it): a session's parse is bound by its own latency
 * chain, so the GPU's throughput is the number of sessions it holds at once. */
__global__ void __launch_bounds__(GLZ_THREADS) GLZ_EIGHT_PER_CU
glz_parse_kernel(const GlzDevJob *__restrict__ jobs, const uint32_t *__restrict__ u32_pool,
                 const GlzDevSegUpdate *__restrict__ upd_pool)
{
    GlzShared &sh = glz_sh;
    const GlzDevJob job = jobs[blockIdx.x];
    if ((int)job.type != TYPE) {
        return; /* another instantiation of this kernel handles that pixel type */
    }
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
#ifdef GLZ_STAMPS
    if (tid == 0 && blockIdx.x < 8192) {
        glz_wg_times[blockIdx.x * 2] = wall_clock64();
    }
#endif

    for (uint32_t u = (uint32_t)tid; u < job.nupd; u += GLZ_THREADS) {
        const GlzDevSegUpdate up = upd_pool[job.upd_off + u];
        glz_store_seg(job, up.index, up.seg);
    }
    for (int t = tid; t < GLZ_KEYS * 8; t += GLZ_THREADS) {
        sh.bitmap[t] = 0;
    }
    __syncthreads();
    if (tid < GLZ_SEG_CACHE) {
        sh.seg_cache[tid] = glz_load_seg(job, (uint32_t)tid);
    }

    if (tid < GLZ_HEADER_BYTES) {
        uint32_t b;
        const uint32_t id_hi = (uint32_t)(job.image_id >> 32);
        const uint32_t id_lo = (uint32_t)(job.image_id & 0xffffffu); /* glz_encoder.c:86 */
        if (lane < 4) {
            b = (0x20205a4cu >> (8 * (3 - lane))) & 0xff;
        } else if (lane < 8) {
            b = (0x00010001u >> (8 * (7 - lane))) & 0xff;
        } else if (lane == 8) {
            b = (job.type & 0x0f) | (job.top_down ? 0x10u : 0u);
        } else if (lane < 13) {
            b = (job.width >> (8 * (12 - lane))) & 0xff;
        } else if (lane < 17) {
            b = (job.height >> (8 * (16 - lane))) & 0xff;
        } else if (lane < 21) {
            b = (job.stride >> (8 * (20 - lane))) & 0xff;
        } else if (lane < 25) {
            b = (id_hi >> (8 * (24 - lane))) & 0xff;
        } else if (lane < 29) {
            b = (id_lo >> (8 * (28 - lane))) & 0xff;
        } else {
            b = (job.win_head_dist >> (8 * (32 - lane))) & 0xff;
        }
        glz_put(glz_out(job), job.out_cap, (uint32_t)lane, b);
    }
    if (tid == 0) {
        sh.op = GLZ_HEADER_BYTES;
        sh.copy = 0;
        sh.last_copy = 0;
    }
    __syncthreads();

    const uint32_t *seg_list = u32_pool + job.seg_list_off;
    if (TYPE == 6) {
        glz_parse_image<GLZ_MODE_RGB16>(job, seg_list, tid);
    } else if (TYPE == 7) {
        glz_parse_image<GLZ_MODE_RGB24>(job, seg_list, tid);
    } else {
        glz_parse_image<GLZ_MODE_RGB32>(job, seg_list, tid);
        if (TYPE == 9) {
            __syncthreads();
            glz_parse_image<GLZ_MODE_ALPHA>(job, seg_list, tid);
        }
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t op = sh.op;
        *job.result = op > job.out_cap ? GLZ_DEV_ERR_NO_SPACE : (int32_t)op;
#ifdef GLZ_STAMPS
        if (blockIdx.x < 8192) {
            glz_wg_times[blockIdx.x * 2 + 1] = wall_clock64();
        }
#endif
    }
}

/* Rows into the window arena: descs[blockIdx.y], 16 bytes per lane. */
__global__ void __launch_bounds__(256)
glz_copy_kernel(const GlzCopyDesc *descs)
{
    const GlzCopyDesc d = descs[blockIdx.y];
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)d.src | (uintptr_t)d.dst) & 15) == 0) {
        const uint64_t n16 = d.bytes >> 4;
        const uint4 *s = (const uint4 *)d.src;
        uint4 *w = (uint4 *)d.dst;
        for (uint64_t k = t; k < n16; k += stride) {
            w[k] = s[k];
        }
        for (uint64_t k = (n16 << 4) + t; k < d.bytes; k += stride) {
            d.dst[k] = d.src[k];
        }
    } else {
        for (uint64_t k = t; k < d.bytes; k += stride) {
            d.dst[k] = d.src[k];
        }
    }
}

/* type_mask: bit t set when some job has LzImageType t (6..9) */
void glz_launch_parse(const GlzDevJob *jobs, int njobs, unsigned type_mask, const uint32_t *u32_pool,
                      const GlzDevSegUpdate *upd_pool, hipStream_t stream)
{
    const dim3 grid((unsigned)njobs), block(GLZ_THREADS);
    const int t = spicegpu::timing_begin("glz_parse_kernel", stream);
    if (type_mask & (1u << 8)) {
        hipLaunchKernelGGL(glz_parse_kernel<8>, grid, block, 0, stream, jobs, u32_pool, upd_pool);
    }
    if (type_mask & (1u << 9)) {
        hipLaunchKernelGGL(glz_parse_kernel<9>, grid, block, 0, stream, jobs, u32_pool, upd_pool);
    }
    if (type_mask & (1u << 7)) {
        hipLaunchKernelGGL(glz_parse_kernel<7>, grid, block, 0, stream, jobs, u32_pool, upd_pool);
    }
    if (type_mask & (1u << 6)) {
        hipLaunchKernelGGL(glz_parse_kernel<6>, grid, block, 0, stream, jobs, u32_pool, upd_pool);
    }
    spicegpu::timing_end(t, stream);
}

void glz_launch_copy(const GlzCopyDesc *descs, int ndescs, unsigned blocks_per_desc,
                     hipStream_t stream)
{
    const int t = spicegpu::timing_begin("glz_copy_kernel", stream);
    hipLaunchKernelGGL(glz_copy_kernel, dim3(blocks_per_desc, (unsigned)ndescs), dim3(256), 0,
                       stream, descs);
    spicegpu::timing_end(t, stream);
}
