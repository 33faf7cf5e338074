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
/* the device spice_gpu_init selected, or -1 before it ran */
int current_device();
hipStream_t stream();
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
bool check(hipError_t e, const char *what);

/* Serialises the short stretches in which a call enqueues work on the library's own
 * stream() (dictionary reset, graduality, the synchronous JPEG entry points, the entry
 * point that allocates nothing).  It is taken and released inside one API call, never
 * held from one call to the next: batches in flight (spice_gpu_glz_submit /
 * spice_gpu_jpeg_submit) own a flight slot with a stream of its own instead, so a thread
 * that pipelines batches keeps no other thread out of the library.  Lock order: a GLZ
 * dictionary's lock first, then this gate or a flight-slot mutex; never the other way
 * round.  A thread that asks for the gate while it holds it is told so instead of
 * waiting for ever. */
class DeviceGate {
public:
    /* false (with the error text set) when the calling thread already holds the gate */
    bool acquire(const char *who);
    void release();
    /* the calling thread holds the gate right now */
    bool held_by_caller();
};
DeviceGate &device_gate();

/* Holds the gate for a scope; `ok` is false when it could not be taken. */
struct GateHold {
    bool ok;
    explicit GateHold(const char *who) : ok(device_gate().acquire(who)) {}
    ~GateHold()
    {
        if (ok) {
            device_gate().release();
        }
    }
    GateHold(const GateHold &) = delete;
    GateHold &operator=(const GateHold &) = delete;
};

/* A stream (and, on demand, a pair of events) that belongs to one flight slot; made on
 * first use, kept until the process ends. */
struct OwnStream {
    hipStream_t st = nullptr;
    bool ensure();
};

/* kSideBySide streams whose kernels the device runs side by side.  A process has a handful of
 * hardware queues, the runtime binds every stream to one of them when the stream is first
 * given work (the least used one, by a rule of its own), and two streams on one queue run their
 * kernels one after the other however independent they are.  So these are found by trying:
 * candidates are created and bound, and one is kept when a short kernel on it finishes while
 * a longer one still runs on each of those already kept (a few milliseconds, once per
 * process).  Fewer distinct queues than asked for: the rest repeat the ones found. */
const int kSideBySide = 4;
bool side_by_side_streams(hipStream_t out[kSideBySide]);

void stat_add(int codec, uint64_t orig, uint64_t comp, uint64_t ns);

/* Brackets one kernel launch with events when timing is enabled (no-ops otherwise). */
int timing_begin(const char *kernel_name, hipStream_t s);
void timing_end(int token, hipStream_t s);
/* The same for a launch that records its own begin and end (hipExtLaunchKernelGGL's start
 * and stop events): hands out the two events of a new entry and records nothing itself, so
 * no event packet stands between the kernel and its neighbours on the stream.  False when
 * timing is off. */
bool timing_events(const char *kernel_name, hipEvent_t *start, hipEvent_t *stop);
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
