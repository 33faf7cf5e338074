/*
 * TEST INFRASTRUCTURE ONLY -- CPU execution engine behind tests/sim/hip/hip_runtime.h.
 * Work-items of one workgroup are fibers on a single OS thread, switched
 * round-robin at collectives; workgroups run one after another.
 */
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <map>
#include <mutex>
#include <vector>

sim_uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

/* ------------------------------------------------------------ context switch */
extern "C" void sim_switch(void **save_sp, void *load_sp);
asm(".text\n"
    ".globl sim_switch\n"
    ".type sim_switch,@function\n"
    "sim_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size sim_switch, .-sim_switch\n");

namespace {

const size_t kStack = 512 * 1024;
const int kMaxThreads = 1024;

struct Fiber {
    void *sp;
    char *stack;
    bool done;
};

Fiber fibers[kMaxThreads];
int n_fibers, cur_fiber;
void *sched_sp;
const std::function<void()> *cur_body;
uint64_t progress; /* bumped whenever any fiber gets past a rendezvous or ends */
void *wait_site[4096]; /* where each fiber last entered a collective */

/* per wave rendezvous */
uint64_t slot[kMaxThreads];
int arrived_wave[kMaxThreads / 64];
uint64_t gen_wave[kMaxThreads / 64];
int arrived_block;
uint64_t gen_block;

void yield_to_sched()
{
    sim_switch(&fibers[cur_fiber].sp, sched_sp);
}

void fiber_main()
{
    (*cur_body)();
    fibers[cur_fiber].done = true;
    progress++;
    yield_to_sched();
    abort(); /* a finished fiber is never resumed */
}

void prepare_fiber(int i)
{
    Fiber &f = fibers[i];
    if (!f.stack) {
        f.stack = (char *)mmap(NULL, kStack, PROT_READ | PROT_WRITE,
                               MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (f.stack == MAP_FAILED) {
            perror("sim: fiber stack");
            abort();
        }
    }
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    uint64_t *p = (uint64_t *)top;
    p[-1] = 0;                       /* fake return address of fiber_main */
    p[-2] = (uint64_t)&fiber_main;   /* popped by sim_switch's ret */
    for (int k = 3; k <= 8; k++) {
        p[-k] = 0;                   /* rbp rbx r12..r15 */
    }
    f.sp = (void *)(p - 8);
    f.done = false;
}

int wave_width(int wave)
{
    int w = n_fibers - wave * 64;
    return w > 64 ? 64 : w;
}

void rendezvous_wave(int wave)
{
    int width = wave_width(wave);
    if (++arrived_wave[wave] == width) {
        arrived_wave[wave] = 0;
        gen_wave[wave]++;
        progress++;
    } else {
        uint64_t g = gen_wave[wave];
        while (gen_wave[wave] == g) {
            yield_to_sched();
        }
    }
}

void rendezvous_block()
{
    if (++arrived_block == n_fibers) {
        arrived_block = 0;
        gen_block++;
        progress++;
    } else {
        uint64_t g = gen_block;
        while (gen_block == g) {
            yield_to_sched();
        }
    }
}

} // namespace


void sim_barrier(int)
{
    wait_site[cur_fiber] = __builtin_return_address(0);
    rendezvous_block();
}

/* Barrier of a subset of the workgroup's work-items (the wavefronts that share one phase of a
 * kernel in which phases run side by side): nthreads arrivals at `key` release them all. */
namespace {
struct GroupBarrier {
    int arrived = 0;
    uint64_t gen = 0;
};
std::map<void *, GroupBarrier> group_barriers;
} // namespace

void sim_group_barrier(void *key, int nthreads)
{
    wait_site[cur_fiber] = __builtin_return_address(0);
    GroupBarrier &b = group_barriers[key];
    if (++b.arrived == nthreads) {
        b.arrived = 0;
        b.gen++;
        progress++;
    } else {
        uint64_t g = b.gen;
        while (b.gen == g) {
            yield_to_sched();
        }
    }
}

void sim_yield(void)
{
    wait_site[cur_fiber] = __builtin_return_address(0);
    yield_to_sched();
}

void sim_note_progress(void)
{
    progress++;
}

uint64_t sim_exchange64(uint64_t v, int, int mode, int arg)
{
    int tid = cur_fiber, wave = tid / 64, lane = tid % 64, width = wave_width(wave);
    uint64_t r = 0;
    wait_site[tid] = __builtin_return_address(0);
    slot[tid] = v;
    rendezvous_wave(wave);
    const uint64_t *s = &slot[wave * 64];
    switch (mode) {
    case 0:
        for (int l = 0; l < width; l++) {
            if (s[l]) {
                r |= 1ull << l;
            }
        }
        break;
    case 1: {
        int src = arg & 63;
        r = src < width ? s[src] : v;
        break;
    }
    case 2:
        r = lane - arg >= 0 ? s[lane - arg] : v;
        break;
    case 3:
        r = lane + arg < width ? s[lane + arg] : v;
        break;
    case 4: {
        int src = lane ^ arg;
        r = src < width ? s[src] : v;
        break;
    }
    default:
        abort();
    }
    rendezvous_wave(wave);
    return r;
}

/* A fault inside a kernel (typically a read past a guarded buffer) names the
 * work-item and the call chain before the process dies. */
static void sim_on_segv(int, siginfo_t *si, void *)
{
    fprintf(stderr, "sim: fault at address %p in work-item %d of block (%u,%u,%u)\n", si->si_addr,
            cur_fiber, blockIdx.x, blockIdx.y, blockIdx.z);
    void *frames[24];
    int n = backtrace(frames, 24);
    for (int i = 0; i < n; i++) {
        Dl_info di;
        memset(&di, 0, sizeof(di));
        dladdr(frames[i], &di);
        fprintf(stderr, "sim:   +0x%lx %s (%s)\n", (unsigned long)((char *)frames[i] - (char *)di.dli_fbase),
                di.dli_sname ? di.dli_sname : "?", di.dli_fname ? di.dli_fname : "?");
    }
    signal(SIGSEGV, SIG_DFL);
    raise(SIGSEGV);
}

void sim_run_grid(dim3 grid, dim3 block, const std::function<void()> &body)
{
    /* one grid at a time: the fibers and their scheduler are process-wide */
    static std::mutex grid_mutex;
    std::lock_guard<std::mutex> one_grid(grid_mutex);
    static bool handler_installed = false;
    if (!handler_installed) {
        static char alt_stack[64 * 1024];
        stack_t ss;
        ss.ss_sp = alt_stack;
        ss.ss_size = sizeof(alt_stack);
        ss.ss_flags = 0;
        sigaltstack(&ss, NULL);
        struct sigaction sa;
        memset(&sa, 0, sizeof(sa));
        sa.sa_sigaction = sim_on_segv;
        sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
        sigaction(SIGSEGV, &sa, NULL);
        handler_installed = true;
    }
    int threads = (int)(block.x * block.y * block.z);
    if (threads <= 0 || threads > kMaxThreads) {
        fprintf(stderr, "sim: bad block size %d\n", threads);
        abort();
    }
    gridDim = grid;
    blockDim = block;
    cur_body = &body;
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        n_fibers = threads;
        arrived_block = 0;
        memset(arrived_wave, 0, sizeof(arrived_wave));
        for (int i = 0; i < threads; i++) {
            prepare_fiber(i);
        }
        int live = threads;
        while (live > 0) {
            uint64_t before = progress;
            live = 0;
            for (int i = 0; i < threads; i++) {
                if (fibers[i].done) {
                    continue;
                }
                cur_fiber = i;
                blockIdx.x = bx;
                blockIdx.y = by;
                blockIdx.z = bz;
                threadIdx.x = (unsigned)i % block.x;
                threadIdx.y = ((unsigned)i / block.x) % block.y;
                threadIdx.z = (unsigned)i / (block.x * block.y);
                sim_switch(&sched_sp, fibers[i].sp);
                if (!fibers[i].done) {
                    live++;
                }
            }
            if (live > 0 && progress == before) {
                fprintf(stderr, "sim: deadlock -- a collective was not reached by every "
                                "work-item of block (%u,%u,%u)\n", bx, by, bz);
                for (int i = 0; i < threads; i++) {
                    if (!fibers[i].done && (i == 0 || wait_site[i] != wait_site[i - 1])) {
                        Dl_info di;
                        memset(&di, 0, sizeof(di));
                        dladdr(wait_site[i], &di);
                        fprintf(stderr, "sim:   work-item %d waits at +0x%lx (addr2line -e %s)\n", i,
                                (unsigned long)((char *)wait_site[i] - (char *)di.dli_fbase),
                                di.dli_fname ? di.dli_fname : "?");
                    }
                }
                abort();
            }
        }
    }
}

/* ------------------------------------------------------------- host runtime */
namespace {
struct Alloc {
    void *map;
    size_t map_len;
};
std::map<void *, Alloc> allocs;
size_t page_size()
{
    static size_t p = (size_t)sysconf(_SC_PAGESIZE);
    return p;
}
} // namespace

static std::mutex allocs_mutex; /* worker threads allocate side by side */
static unsigned long alloc_calls; /* hipMalloc + hipHostMalloc, for tests of entry points that must not allocate */
extern "C" unsigned long sim_alloc_calls(void)
{
    std::lock_guard<std::mutex> lk(allocs_mutex);
    return alloc_calls;
}

hipError_t hipMalloc(void **out, size_t n)
{
    size_t pg = page_size();
    size_t body = ((n ? n : 1) + 63) & ~(size_t)63;
    size_t pages = (body + pg - 1) / pg * pg;
    size_t len = pages + 2 * pg;
    char *m = (char *)mmap(NULL, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) {
        return hipErrorOutOfMemory;
    }
    mprotect(m, pg, PROT_NONE);
    mprotect(m + pg + pages, pg, PROT_NONE);
    char *p = m + pg + pages - body; /* the end of the buffer touches the guard page */
    memset(p, 0xcd, body);
    std::lock_guard<std::mutex> lk(allocs_mutex);
    alloc_calls++;
    allocs[p] = Alloc{m, len};
    *out = p;
    return hipSuccess;
}

hipError_t hipFree(void *p)
{
    if (!p) {
        return hipSuccess;
    }
    std::lock_guard<std::mutex> lk(allocs_mutex);
    auto it = allocs.find(p);
    if (it == allocs.end()) {
        return hipErrorInvalidValue;
    }
    munmap(it->second.map, it->second.map_len);
    allocs.erase(it);
    return hipSuccess;
}

hipError_t hipHostMalloc(void **p, size_t n, unsigned)
{
    {
        std::lock_guard<std::mutex> lk(allocs_mutex);
        alloc_calls++;
    }
    *p = malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p)
{
    free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind)
{
    memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t)
{
    memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemset(void *d, int v, size_t n)
{
    memset(d, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t)
{
    memset(d, v, n);
    return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t *s)
{
    *s = (hipStream_t)malloc(1);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned)
{
    return hipStreamCreate(s);
}
hipError_t hipStreamDestroy(hipStream_t s)
{
    free(s);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetDeviceCount(int *n)
{
    *n = 1;
    return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int *d)
{
    *d = 0;
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "cpu wave simulator");
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "sim");
    p->multiProcessorCount = 1;
    return hipSuccess;
}
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "sim error"; }

struct sim_event {
    double t;
};
hipError_t hipEventCreate(hipEvent_t *e)
{
    *e = new sim_event{0};
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e)
{
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    e->t = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; } /* launches are synchronous here */
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = (float)(b->t - a->t);
    return hipSuccess;
}
