/*
 * Graduality score of bitmaps resident in HBM
 * (server/red_bitmap_utils.h:43-147, called from server/red_worker.c:5913-5967).
 *
 * One wavefront per bitmap row, a sample per lane: the sample's pixel is
 * compared with its right, lower and lower-right neighbours.  The reference
 * adds 0.5 / 1.0 / -0.25 per pair in double precision; all three are multiples
 * of 0.25, so the kernel adds 2 / 4 / -1 in integers and the host divides once,
 * which is exact.  The kernel touches 16 bytes per sample, i.e. about one
 * byte per pixel of the bitmap: it is bound by HBM sector reads.
 */
#include <hip/hip_runtime.h>
#include <spice_target.h>

#include "gradual_device.h"
#include "runtime.h"

#define GRAD_GLOBAL SPICE_GLOBAL

struct GradPixel {
    int r, g, b;
};

/* format: 6 = x1r5g5b5, 7 = b,g,r bytes, 8 / 9 = b,g,r,pad bytes (server/red_worker.c:5876-5889) */
__device__ __forceinline__ GradPixel grad_load(const GRAD_GLOBAL uint8_t *data, int format, size_t idx)
{
    GradPixel p;
    if (format == 6) {
        const uint32_t v = ((const GRAD_GLOBAL uint16_t *)data)[idx];
        p.r = (int)((v >> 10) & 0x1f);
        p.g = (int)((v >> 5) & 0x1f);
        p.b = (int)(v & 0x1f);
    } else if (format == 7) {
        const GRAD_GLOBAL uint8_t *q = data + idx * 3;
        p.b = q[0];
        p.g = q[1];
        p.r = q[2];
    } else {
        const uint32_t v = ((const GRAD_GLOBAL uint32_t *)data)[idx];
        p.b = (int)(v & 0xff);
        p.g = (int)((v >> 8) & 0xff);
        p.r = (int)((v >> 16) & 0xff);
    }
    return p;
}

/* pixelcmp, red_bitmap_utils.h:65-94, as the score x 4: equal 2, contrast 4, else -1 */
__device__ __forceinline__ int grad_pair(const GradPixel &a, const GradPixel &b, int th, bool *same)
{
    const int dr = a.r > b.r ? a.r - b.r : b.r - a.r;
    const int dg = a.g > b.g ? a.g - b.g : b.g - a.g;
    const int db = a.b > b.b ? a.b - b.b : b.b - a.b;
    if (dr >= th || dg >= th || db >= th) {
        *same = false;
        return 4;
    }
    if ((dr | dg | db) == 0) {
        return 2;
    }
    *same = false;
    return -1;
}

__global__ void __launch_bounds__(GRADUAL_THREADS)
gradual_kernel(const GradualDevJob *__restrict__ jobs, const GradualDevChunk *__restrict__ chunks,
               const uint32_t *__restrict__ u32_pool, long long *__restrict__ results)
{
    const GradualDevJob job = jobs[blockIdx.y];
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const int th = job.format == 6 ? 8 : 60; /* CONTRAST_TH, red_bitmap_utils.h:47-51 */
    const size_t width = (size_t)job.width;
    long long sum = 0, count = 0;

    for (int c = 0; c < job.nchunks; c++) {
        const GradualDevChunk ch = chunks[job.chunk_off + (uint32_t)c];
        const GRAD_GLOBAL uint8_t *data = (const GRAD_GLOBAL uint8_t *)(uintptr_t)ch.data;
        for (uint32_t row = blockIdx.x * 4u + (uint32_t)wave; row + 1 < ch.rows; row += gridDim.x * 4u) {
            const uint32_t first = u32_pool[ch.entry_off + row];
            if (first == GRADUAL_NO_SAMPLE) {
                continue;
            }
            for (size_t col = first + (size_t)lane * (size_t)job.jump; col < width;
                 col += 64u * (size_t)job.jump) {
                /* a sample on the row's last pixel is taken one to the left, :137-139 */
                const size_t x = col + 1 == width ? col - 1 : col;
                const size_t at = (size_t)row * width + x;
                const GradPixel p = grad_load(data, job.format, at);
                const GradPixel right = grad_load(data, job.format, at + 1);
                const GradPixel below = grad_load(data, job.format, at + width);
                const GradPixel diag = grad_load(data, job.format, at + width + 1);
                bool same = true;
                int s = grad_pair(p, right, th, &same);
                s += grad_pair(p, below, th, &same);
                s += grad_pair(p, diag, th, &same);
                sum += same ? 0 : s; /* squares of identical pixels are ignored, :111-114 */
                count++;
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        sum += __shfl_down(sum, (unsigned)d);
        count += __shfl_down(count, (unsigned)d);
    }
    if (lane == 0 && count) {
        atomicAdd((unsigned long long *)&results[job.result_off], (unsigned long long)sum);
        atomicAdd((unsigned long long *)&results[job.result_off + 1], (unsigned long long)count);
    }
}

void gradual_launch(const GradualDevJob *jobs, int njobs, const GradualDevChunk *chunks,
                    const uint32_t *u32_pool, long long *results, hipStream_t stream)
{
    const int t = spicegpu::timing_begin("gradual_kernel", stream);
    hipLaunchKernelGGL(gradual_kernel, dim3(GRADUAL_BLOCKS_PER_JOB, (unsigned)njobs), dim3(GRADUAL_THREADS),
                       0, stream, jobs, chunks, u32_pool, results);
    spicegpu::timing_end(t, stream);
}
