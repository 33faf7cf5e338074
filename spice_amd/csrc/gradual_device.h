/*
 * Device view of a graduality job (gradual_host.cpp fills it, gradual_kernels.hip
 * reads it).  The sample positions of _get_bitmap_graduality_level form a
 * serial recurrence (server/red_bitmap_utils.h:136-144: a sample that lands on
 * a row's last pixel moves one left, and every later sample with it), but one
 * that depends on the geometry only: the host walks it row by row and hands
 * the kernel the column of each row's first sample.
 */
#ifndef SPICE_GRADUAL_DEVICE_H
#define SPICE_GRADUAL_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define GRADUAL_NO_SAMPLE 0xffffffffu
#define GRADUAL_BLOCKS_PER_JOB 32
#define GRADUAL_THREADS 256

struct GradualDevChunk {
    const uint8_t *data;
    uint32_t rows;
    uint32_t entry_off;   /* into the u32 pool: rows - 1 first-sample columns */
};

struct GradualDevJob {
    int32_t format;       /* SPICE_GPU_BITMAP_FMT_* */
    int32_t width;
    int32_t jump;
    int32_t nchunks;
    uint32_t chunk_off;   /* into the chunk pool */
    uint32_t result_off;  /* into the result pool: {sum of pair scores x 4, samples} */
};

void gradual_launch(const GradualDevJob *jobs, int njobs, const GradualDevChunk *chunks,
                    const uint32_t *u32_pool, long long *results, hipStream_t stream);

#endif
