/*
 * GLZ parse for a GPU that holds few sessions: one 576-lane workgroup per image,
 * at most one or two workgroups per CU.
 *
 * The kernel of glz_kernels.hip is built for a full GPU: a session's parse is a
 * dependent chain, so throughput is the number of sessions the GPU holds, and
 * that kernel fits eight of them on a CU (20 KB of LDS, 64 registers).  With
 * fewer sessions than CUs -- a handful of busy displays, or a client whose four
 * display channels share one dictionary and must follow each other
 * (server/red_worker.c:9677-9754) -- most of the GPU idles and what counts is
 * how soon one image is done.  This kernel spends a CU's idle LDS and wave slots
 * on that: the same phases (glz_parse_phases.h) run side by side on three
 * windows, so that the walk over a window's stops, the one phase that cannot be
 * spread over lanes, never waits for the scan in front of it or the emit behind
 * it (glz_lone_seg.h has the schedule).  Same streams, byte for byte.
 */
#include <hip/hip_runtime.h>

#include "glz_device.h"
#include "runtime.h"
#include <spice_target.h>

#define GLZ_LONE 1
#ifdef GLZ_STAMPS
#define GLZ_LONE_STAMPS 1
#undef GLZ_STAMPS /* the one-window kernel's per-phase stamps */
#endif
namespace glz_lone {
#include "glz_parse_phases.h"

/* One workgroup per job: apply the window-table updates, write the header
 * (glz_encoder.c:264-276), parse, report the size. */
template <int TYPE>
__global__ void __launch_bounds__(GLZ_LONE_THREADS)
glz_parse_lone_kernel(const GlzDevJob *__restrict__ jobs, const uint32_t *__restrict__ u32_pool,
                      const GlzDevSegUpdate *__restrict__ upd_pool, int skip_blocks)
{
    GlzShared &sh = glz_sh;
    if ((int)blockIdx.x < skip_blocks) {
        return; /* a place holder: see glz_launch_parse_lone */
    }
    const GlzDevJob job = jobs[(int)blockIdx.x - skip_blocks];
    if ((int)job.type != TYPE) {
        return; /* another instantiation of this kernel handles that pixel type */
    }
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;

    for (uint32_t u = (uint32_t)tid; u < job.nupd; u += GLZ_LONE_THREADS) {
        const GlzDevSegUpdate up = upd_pool[job.upd_off + u];
        glz_store_seg(job, up.index, up.seg);
    }
    for (int s = 0; s < GLZ_SLOTS; s++) {
        for (int t = tid; t < GLZ_KEYS * 8; t += GLZ_LONE_THREADS) {
            sh.win[s].bitmap[t] = 0;
        }
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
    }
}
} // namespace glz_lone

/* type_mask: bit t set when some job has LzImageType t (6..9).  skip_blocks: that many
 * workgroups are launched in front of the jobs' and exit at once.  The hardware deals a
 * launch's workgroups out over the GPU's eight XCDs in index order, so launches that run side by
 * side on different streams with one or two images each would all start on the first XCD,
 * on CUs that other launches' images are busy on; callers with several launches in flight give
 * each a different offset. */
void glz_launch_parse_lone(const GlzDevJob *jobs, int njobs, unsigned type_mask, const uint32_t *u32_pool,
                           const GlzDevSegUpdate *upd_pool, hipStream_t stream, int skip_blocks)
{
    using namespace glz_lone;
    const dim3 grid((unsigned)(njobs + skip_blocks)), block(GLZ_LONE_THREADS);
    const int t = spicegpu::timing_begin("glz_parse_lone_kernel", stream);
    if (type_mask & (1u << 8)) {
        hipLaunchKernelGGL(glz_parse_lone_kernel<8>, grid, block, 0, stream, jobs, u32_pool, upd_pool, skip_blocks);
    }
    if (type_mask & (1u << 9)) {
        hipLaunchKernelGGL(glz_parse_lone_kernel<9>, grid, block, 0, stream, jobs, u32_pool, upd_pool, skip_blocks);
    }
    if (type_mask & (1u << 7)) {
        hipLaunchKernelGGL(glz_parse_lone_kernel<7>, grid, block, 0, stream, jobs, u32_pool, upd_pool, skip_blocks);
    }
    if (type_mask & (1u << 6)) {
        hipLaunchKernelGGL(glz_parse_lone_kernel<6>, grid, block, 0, stream, jobs, u32_pool, upd_pool, skip_blocks);
    }
    spicegpu::timing_end(t, stream);
}
