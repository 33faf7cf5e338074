/*
 * Process-wide device state of libspicegpu: the selected GPU, the library's
 * stream, error text, per-codec counters, and small helpers for pinned
 * staging memory that grows on demand.
 */
#ifndef SPICE_RUNTIME_H
#define SPICE_RUNTIME_H

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>

#include <mutex>

#include "../../include/spice_gpu.h"

namespace spicegpu {

/* Makes sure spice_gpu_init ran (initialises device 0 on first use).
 * Returns false, with the error text set, when no GPU can be used. */
bool ensure_init();
hipStream_t stream();
/* a second stream for work that may run beside stream()'s, and two events to order them */
hipStream_t side_stream();
hipEvent_t fork_event();
hipEvent_t join_event();
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
bool check(hipError_t e, const char *what);

/* Serialises device work issued through the library's single stream. */
std::mutex &device_mutex();

void stat_add(int codec, uint64_t orig, uint64_t comp, uint64_t ns);

/* Brackets one kernel launch with events when timing is enabled (no-ops otherwise). */
int timing_begin(const char *kernel_name, hipStream_t s);
void timing_end(int token, hipStream_t s);
uint64_t now_ns();

/* Host-pinned buffer with a device mirror, grown geometrically. */
struct Staging {
    uint8_t *host = nullptr;
    uint8_t *dev = nullptr;
    size_t cap = 0;
    bool reserve(size_t bytes);
    void release();
};

/* Plain device buffer grown geometrically (contents are not preserved). */
struct DevBuffer {
    uint8_t *ptr = nullptr;
    size_t cap = 0;
    bool reserve(size_t bytes);
    void release();
};

struct PinnedBuffer {
    uint8_t *ptr = nullptr;
    size_t cap = 0;
    bool reserve(size_t bytes);
    void release();
};

} // namespace spicegpu

#endif
