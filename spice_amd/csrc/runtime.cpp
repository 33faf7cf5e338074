/*
 * Process-wide device state of libspicegpu (see runtime.h).
 */
#include "runtime.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <condition_variable>
#include <string>
#include <thread>
#include <vector>

bool jpeg_noalloc_setup(int device); /* jpeg_kernels.hip */

namespace spicegpu {

namespace {
std::mutex g_init_mutex;
std::mutex g_gate_mutex;
std::condition_variable g_gate_cv;
bool g_gate_busy = false;
std::thread::id g_gate_owner;
DeviceGate g_gate;
std::mutex g_stat_mutex;
bool g_ready = false;
int g_device = -1;
hipStream_t g_stream = nullptr;
thread_local char g_error[512] = "";
SpiceGpuCodecStats g_stats[SPICE_GPU_STAT_COUNT];
} // namespace

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

bool check(hipError_t e, const char *what)
{
    if (e == hipSuccess) {
        return true;
    }
    set_error("%s: %s", what, hipGetErrorString(e));
    return false;
}

static bool init_locked(int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_error("no HIP device visible: libspicegpu has no CPU fallback");
        return false;
    }
    if (device < 0 || device >= count) {
        set_error("device %d out of range (%d visible)", device, count);
        return false;
    }
    if (!check(hipSetDevice(device), "hipSetDevice")) {
        return false;
    }
    if (!g_stream && !check(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking),
                            "hipStreamCreate")) {
        return false;
    }
    /* the device-resident scratch of the entry point that allocates nothing */
    if (!jpeg_noalloc_setup(device)) {
        return false;
    }
    g_device = device;
    g_ready = true;
    return true;
}

bool ensure_init()
{
    std::lock_guard<std::mutex> lock(g_init_mutex);
    if (g_ready) {
        /* worker threads other than the initialising one need the device bound */
        return check(hipSetDevice(g_device), "hipSetDevice");
    }
    return init_locked(0);
}

int current_device()
{
    return g_ready ? g_device : -1;
}

hipStream_t stream()
{
    return g_stream;
}

bool OwnStream::ensure()
{
    return st || check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
}

namespace {
/* occupies one wavefront for about `ticks` of the 100 MHz wall clock */
__global__ void queue_probe_spin(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
    }
}

/* does a kernel on b run while one on a is still running? */
bool runs_beside(hipStream_t a, hipStream_t b)
{
    hipLaunchKernelGGL(queue_probe_spin, dim3(1), dim3(64), 0, a, 40000ll); /* 0.4 ms */
    const uint64_t t0 = now_ns();
    hipLaunchKernelGGL(queue_probe_spin, dim3(1), dim3(64), 0, b, 0ll);
    bool ok = hipStreamSynchronize(b) == hipSuccess;
    const uint64_t dt = now_ns() - t0;
    ok = hipStreamSynchronize(a) == hipSuccess && ok;
    if (getenv("SPICE_GPU_DEBUG_QUEUES")) {
        fprintf(stderr, "queue probe: %p beside %p: %.3f ms\n", (void *)b, (void *)a, dt / 1e6);
    }
    return ok && dt < 200000; /* behind the long kernel it would have taken the 0.4 ms */
}

std::mutex g_sbs_mutex;
hipStream_t g_sbs[kSideBySide];
int g_sbs_state = 0; /* 0 not made, 1 made, -1 failed */
} // namespace

bool side_by_side_streams(hipStream_t out[kSideBySide])
{
    std::lock_guard<std::mutex> lk(g_sbs_mutex);
    if (g_sbs_state == 0) {
        g_sbs_state = -1;
        const int kCandidates = 12;
        std::vector<hipStream_t> kept, rejected;
        hipEvent_t ev = nullptr;
        bool ok = check(hipEventCreate(&ev), "hipEventCreate");
        for (int c = 0; ok && c < kCandidates && (int)kept.size() < kSideBySide; c++) {
            hipStream_t st = nullptr;
            /* an event binds the stream to its queue now, not when a caller first uses it */
            ok = check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate") &&
                 check(hipEventRecord(ev, st), "hipEventRecord") &&
                 check(hipStreamSynchronize(st), "hipStreamSynchronize");
            bool keep = ok;
            for (size_t k = 0; keep && k < kept.size(); k++) {
                keep = runs_beside(kept[k], st);
            }
            if (keep) {
                kept.push_back(st);
            } else if (st) {
                rejected.push_back(st); /* stays until the end: destroyed now, its queue would be
                                         * the least used one again and the next candidate's */
            }
        }
        for (hipStream_t st : rejected) {
            hipStreamDestroy(st);
        }
        if (ev) {
            hipEventDestroy(ev);
        }
        if (ok && !kept.empty() && check(hipGetLastError(), "queue probe")) {
            for (int k = 0; k < kSideBySide; k++) {
                g_sbs[k] = kept[(size_t)k % kept.size()];
            }
            g_sbs_state = 1;
        }
    }
    if (g_sbs_state != 1) {
        return false;
    }
    for (int k = 0; k < kSideBySide; k++) {
        out[k] = g_sbs[k];
    }
    return true;
}

bool DeviceGate::acquire(const char *who)
{
    std::unique_lock<std::mutex> lk(g_gate_mutex);
    if (g_gate_busy && g_gate_owner == std::this_thread::get_id()) {
        set_error("%s: this thread has a batch in flight (submit without finish)", who);
        return false;
    }
    g_gate_cv.wait(lk, [] { return !g_gate_busy; });
    g_gate_busy = true;
    g_gate_owner = std::this_thread::get_id();
    return true;
}

void DeviceGate::release()
{
    {
        std::lock_guard<std::mutex> lk(g_gate_mutex);
        g_gate_busy = false;
        g_gate_owner = std::thread::id();
    }
    g_gate_cv.notify_one();
}

bool DeviceGate::held_by_caller()
{
    std::lock_guard<std::mutex> lk(g_gate_mutex);
    return g_gate_busy && g_gate_owner == std::this_thread::get_id();
}

DeviceGate &device_gate()
{
    return g_gate;
}

uint64_t now_ns()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

void stat_add(int codec, uint64_t orig, uint64_t comp, uint64_t ns)
{
    std::lock_guard<std::mutex> lock(g_stat_mutex);
    SpiceGpuCodecStats &s = g_stats[codec];
    if (s.count == 0 || ns < s.min_ns) {
        s.min_ns = ns;
    }
    if (ns > s.max_ns) {
        s.max_ns = ns;
    }
    s.count++;
    s.orig_bytes += orig;
    s.comp_bytes += comp;
    s.total_ns += ns;
}

namespace {
struct TimedLaunch {
    int name_id;
    int id;      /* what timing_begin returned */
    bool closed; /* timing_end has recorded b */
    hipEvent_t a, b;
};
int g_time_next_id = 0;
std::mutex g_time_mutex;
bool g_timing = false;
std::vector<std::string> g_time_names;
std::vector<SpiceGpuKernelTime> g_time_acc;
std::vector<TimedLaunch> g_time_pending;
std::vector<hipEvent_t> g_event_pool;
/* Launches of one kernel on different streams may overlap (two batches in flight); what the
 * device spent on the kernel is then the union of the launches' intervals, not their sum.
 * Every collected launch leaves its [begin, end] in milliseconds after g_time_base, an event
 * recorded when the counters were last reset. */
hipEvent_t g_time_base = nullptr;
std::vector<std::vector<std::pair<float, float>>> g_time_spans;

hipEvent_t take_event()
{
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

/* folds finished launches into the per-name totals; the stream must be idle.
 * A launch whose timing_end has not come yet stays pending. */
void timing_collect_locked()
{
    std::vector<TimedLaunch> open;
    for (TimedLaunch &t : g_time_pending) {
        if (!t.closed) {
            open.push_back(t);
            continue;
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) != hipSuccess) {
            (void)hipGetLastError();
            open.push_back(t); /* on another thread's stream and not finished yet */
            continue;
        }
        g_time_acc[(size_t)t.name_id].launches++;
        g_time_acc[(size_t)t.name_id].total_ms += (double)ms;
        float a_ms = 0.f;
        if (g_time_base && hipEventElapsedTime(&a_ms, g_time_base, t.a) == hipSuccess) {
            g_time_spans[(size_t)t.name_id].push_back(std::make_pair(a_ms, a_ms + ms));
        } else {
            (void)hipGetLastError();
        }
        g_event_pool.push_back(t.a);
        g_event_pool.push_back(t.b);
    }
    g_time_pending.swap(open);
}
} // namespace

namespace {
/* a pending entry for one launch of kernel_name, with its two events; g_time_mutex held */
TimedLaunch &timing_entry_locked(const char *kernel_name, bool closed)
{
    int id = -1;
    for (size_t i = 0; i < g_time_names.size(); i++) {
        if (g_time_names[i] == kernel_name) {
            id = (int)i;
        }
    }
    if (id < 0) {
        id = (int)g_time_names.size();
        g_time_names.push_back(kernel_name);
        SpiceGpuKernelTime t;
        memset(&t, 0, sizeof(t));
        snprintf(t.name, sizeof(t.name), "%s", kernel_name);
        g_time_acc.push_back(t);
        g_time_spans.emplace_back();
    }
    TimedLaunch t{id, g_time_next_id, closed, take_event(), take_event()};
    g_time_next_id = (g_time_next_id + 1) & 0x3fffffff;
    g_time_pending.push_back(t);
    return g_time_pending.back();
}
} // namespace

int timing_begin(const char *kernel_name, hipStream_t s)
{
    std::lock_guard<std::mutex> lock(g_time_mutex);
    if (!g_timing) {
        return -1;
    }
    TimedLaunch &t = timing_entry_locked(kernel_name, false);
    hipEventRecord(t.a, s);
    return t.id;
}

bool timing_events(const char *kernel_name, hipEvent_t *start, hipEvent_t *stop)
{
    std::lock_guard<std::mutex> lock(g_time_mutex);
    if (!g_timing) {
        return false;
    }
    TimedLaunch &t = timing_entry_locked(kernel_name, true);
    *start = t.a;
    *stop = t.b;
    return true;
}

void timing_end(int token, hipStream_t s)
{
    if (token < 0) {
        return;
    }
    std::lock_guard<std::mutex> lock(g_time_mutex);
    for (size_t k = g_time_pending.size(); k-- > 0;) {
        if (g_time_pending[k].id == token && !g_time_pending[k].closed) {
            hipEventRecord(g_time_pending[k].b, s);
            g_time_pending[k].closed = true;
            break;
        }
    }
}

bool Staging::reserve(size_t bytes)
{
    if (bytes <= cap) {
        return true;
    }
    size_t ncap = cap ? cap : 4096;
    while (ncap < bytes) {
        ncap *= 2;
    }
    release();
    if (!check(hipHostMalloc((void **)&host, ncap, hipHostMallocDefault), "hipHostMalloc") ||
        !check(hipMalloc((void **)&dev, ncap), "hipMalloc")) {
        release();
        return false;
    }
    cap = ncap;
    return true;
}

void Staging::release()
{
    if (host) {
        hipHostFree(host);
    }
    if (dev) {
        hipFree(dev);
    }
    host = dev = nullptr;
    cap = 0;
}

bool DevBuffer::reserve(size_t bytes)
{
    if (bytes <= cap) {
        return true;
    }
    size_t ncap = cap ? cap : 4096;
    while (ncap < bytes) {
        ncap *= 2;
    }
    release();
    if (!check(hipMalloc((void **)&ptr, ncap), "hipMalloc")) {
        ptr = nullptr;
        return false;
    }
    cap = ncap;
    return true;
}

void DevBuffer::release()
{
    if (ptr) {
        hipFree(ptr);
    }
    ptr = nullptr;
    cap = 0;
}

bool PinnedBuffer::reserve(size_t bytes)
{
    if (bytes <= cap) {
        return true;
    }
    size_t ncap = cap ? cap : 4096;
    while (ncap < bytes) {
        ncap *= 2;
    }
    release();
    if (!check(hipHostMalloc((void **)&ptr, ncap, hipHostMallocDefault), "hipHostMalloc")) {
        ptr = nullptr;
        return false;
    }
    cap = ncap;
    return true;
}

void PinnedBuffer::release()
{
    if (ptr) {
        hipHostFree(ptr);
    }
    ptr = nullptr;
    cap = 0;
}

} // namespace spicegpu

using namespace spicegpu;

extern "C" {

int spice_gpu_init(int device)
{
    std::lock_guard<std::mutex> lock(g_init_mutex);
    return init_locked(device) ? 0 : -1;
}

void spice_gpu_shutdown(void)
{
    std::lock_guard<std::mutex> lock(g_init_mutex);
    if (g_stream) {
        hipStreamSynchronize(g_stream);
        hipStreamDestroy(g_stream);
        g_stream = nullptr;
    }
    g_ready = false;
}

const char *spice_gpu_last_error(void)
{
    return g_error;
}

void *spice_gpu_stream(void)
{
    return ensure_init() ? (void *)g_stream : nullptr;
}

void *spice_gpu_malloc(size_t bytes)
{
    void *p = nullptr;
    if (!ensure_init() || !check(hipMalloc(&p, bytes ? bytes : 1), "hipMalloc")) {
        return nullptr;
    }
    return p;
}

void spice_gpu_free(void *dev_ptr)
{
    if (dev_ptr) {
        hipFree(dev_ptr);
    }
}

int spice_gpu_upload(void *dev_dst, const void *host_src, size_t bytes)
{
    if (!ensure_init()) {
        return -1;
    }
    return check(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice), "hipMemcpy H2D") ? 0 : -1;
}

int spice_gpu_download(void *host_dst, const void *dev_src, size_t bytes)
{
    if (!ensure_init()) {
        return -1;
    }
    return check(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H") ? 0 : -1;
}

void spice_gpu_get_stats(int codec, SpiceGpuCodecStats *out)
{
    std::lock_guard<std::mutex> lock(g_stat_mutex);
    if (codec >= 0 && codec < SPICE_GPU_STAT_COUNT) {
        *out = g_stats[codec];
    } else {
        memset(out, 0, sizeof(*out));
    }
}

void spice_gpu_reset_stats(void)
{
    std::lock_guard<std::mutex> lock(g_stat_mutex);
    memset(g_stats, 0, sizeof(g_stats));
}

void spice_gpu_timing_enable(int on)
{
    std::lock_guard<std::mutex> lock(g_time_mutex);
    g_timing = on != 0;
}

void spice_gpu_timing_reset(void)
{
    std::lock_guard<std::mutex> lock(g_time_mutex);
    if (g_stream) {
        hipStreamSynchronize(g_stream);
    }
    timing_collect_locked();
    for (SpiceGpuKernelTime &t : g_time_acc) {
        t.launches = 0;
        t.total_ms = 0.0;
        t.union_ms = 0.0;
    }
    for (auto &v : g_time_spans) {
        v.clear();
    }
    if (g_stream) {
        if (!g_time_base) {
            hipEventCreate(&g_time_base);
        }
        hipEventRecord(g_time_base, g_stream);
        hipStreamSynchronize(g_stream);
    }
}

int spice_gpu_timing_read(SpiceGpuKernelTime *out, int max_entries)
{
    std::lock_guard<std::mutex> lock(g_time_mutex);
    if (g_stream) {
        hipStreamSynchronize(g_stream);
    }
    timing_collect_locked();
    int n = 0;
    for (size_t k = 0; k < g_time_acc.size(); k++) {
        /* length of the union of this kernel's launch intervals */
        std::vector<std::pair<float, float>> v(g_time_spans[k]);
        std::sort(v.begin(), v.end());
        double u = 0.0;
        float lo = 0.f, hi = -1.f;
        for (const auto &iv : v) {
            if (hi < lo || iv.first > hi) {
                u += hi > lo ? (double)(hi - lo) : 0.0;
                lo = iv.first;
                hi = iv.second;
            } else if (iv.second > hi) {
                hi = iv.second;
            }
        }
        u += hi > lo ? (double)(hi - lo) : 0.0;
        g_time_acc[k].union_ms = v.empty() ? g_time_acc[k].total_ms : u;
        if (n < max_entries) {
            out[n++] = g_time_acc[k];
        }
    }
    return n;
}

} // extern "C"
