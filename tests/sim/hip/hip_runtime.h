/*
 * TEST INFRASTRUCTURE ONLY -- a stand-in for <hip/hip_runtime.h> that lets the
 * product's HIP sources be compiled by g++ and executed on the CPU, one
 * workgroup at a time, with every work-item running as a fiber.  Wave-level
 * collectives (__ballot, __shfl*, readfirstlane) and __syncthreads() are
 * rendezvous points between the fibers, so kernels keep their lockstep
 * meaning.  "Device" allocations carry guard pages so that an out-of-bounds
 * access faults here, on the CPU, instead of on a GPU.
 *
 * Only tests/sim/Makefile puts this directory on the include path; the product
 * library is always built by hipcc against the real runtime.
 */
#ifndef SPICE_SIM_HIP_RUNTIME_H
#define SPICE_SIM_HIP_RUNTIME_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <functional>
#include <type_traits>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __restrict__ __restrict
#define __launch_bounds__(...)
#define __shared__ static
#define __constant__ static
#define __align__(n) __attribute__((aligned(n)))

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorUnknown = 999 };
typedef struct sim_stream *hipStream_t;
typedef struct sim_event *hipEvent_t;
enum hipMemcpyKind {
    hipMemcpyHostToHost,
    hipMemcpyHostToDevice,
    hipMemcpyDeviceToHost,
    hipMemcpyDeviceToDevice,
    hipMemcpyDefault
};
enum { hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipStreamNonBlocking = 1 };

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct sim_uint3 {
    unsigned x, y, z;
};
extern sim_uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

struct uint2 {
    unsigned x, y;
};
struct uint4 {
    unsigned x, y, z, w;
};
struct int2 {
    int x, y;
};
struct int4 {
    int x, y, z, w;
};
struct ushort2 {
    unsigned short x, y;
};
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

struct hipDeviceProp_t {
    char name[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
    char gcnArchName[256];
};

/* ---- host runtime ---- */
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags = 0);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st = 0);
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st = 0);
hipError_t hipStreamCreate(hipStream_t *s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize(void);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipGetLastError(void);
/* a __device__ array of the simulator is an ordinary host array */
#define HIP_SYMBOL(x) x
template <typename T> inline hipError_t hipGetSymbolAddress(void **p, const T &symbol)
{
    *p = (void *)&symbol;
    return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
const char *hipGetErrorString(hipError_t e);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = 0);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
template <class T>
static inline hipError_t hipGetSymbolAddress(void **p, T *sym)
{
    *p = (void *)sym;
    return hipSuccess;
}

/* ---- launch: run every workgroup, each work-item a fiber ---- */
void sim_run_grid(dim3 grid, dim3 block, const std::function<void()> &body);

template <class... KArgs, class... Args>
static inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t,
                                      hipStream_t, Args... args)
{
    sim_run_grid(grid, block, [=]() { kernel(static_cast<KArgs>(args)...); });
}

/* ---- device-side collectives ---- */
void sim_barrier(int scope_wave);
uint64_t sim_exchange64(uint64_t v, int src_lane_or_neg, int mode, int arg);
/* mode 0: ballot (v = predicate); 1: shfl abs lane=arg; 2: up by arg; 3: down by arg; 4: xor arg */

/* barrier of nthreads work-items of the workgroup that meet at `key` (wavefronts sharing a phase) */
void sim_group_barrier(void *key, int nthreads);
/* a work-item that polls memory another work-item will write lets the others run meanwhile;
 * the writer says that something has moved on (the deadlock detector counts it) */
void sim_yield(void);
void sim_note_progress(void);

static inline void __syncthreads() { sim_barrier(0); }
/* the device's constant-rate clock (100 MHz) */
static inline long long wall_clock64()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 100000000ll + ts.tv_nsec / 10;
}
static inline uint64_t __ballot(int pred) { return sim_exchange64(pred ? 1 : 0, -1, 0, 0); }
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred);

template <class T>
static inline T sim_shfl(T v, int mode, int arg)
{
    static_assert(sizeof(T) <= 8, "shfl payload");
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    bits = sim_exchange64(bits, -1, mode, arg);
    T r;
    memcpy(&r, &bits, sizeof(T));
    return r;
}
template <class T>
static inline T __shfl(T v, int lane, int width = 64)
{
    (void)width;
    return sim_shfl(v, 1, lane);
}
template <class T>
static inline T __shfl_up(T v, unsigned d, int width = 64)
{
    (void)width;
    return sim_shfl(v, 2, (int)d);
}
template <class T>
static inline T __shfl_down(T v, unsigned d, int width = 64)
{
    (void)width;
    return sim_shfl(v, 3, (int)d);
}
template <class T>
static inline T __shfl_xor(T v, int m, int width = 64)
{
    (void)width;
    return sim_shfl(v, 4, m);
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return sim_shfl(v, 1, 0); }
static inline int __builtin_amdgcn_readlane(int v, int lane) { return sim_shfl(v, 1, lane); }
static inline int __all(int pred) { return __ballot(!pred) == 0; }
/* On the GPU the lanes of a wavefront run in lockstep, so this is only a
 * scheduling fence; here it is what orders one lane's LDS write before the
 * other lanes' reads. */
static inline void __builtin_amdgcn_wave_barrier() { (void)sim_exchange64(0, -1, 0, 0); }

static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline unsigned __brev(unsigned v)
{
    unsigned r = 0;
    for (int i = 0; i < 32; i++) {
        r = (r << 1) | ((v >> i) & 1);
    }
    return r;
}
static inline unsigned __byte_perm(unsigned a, unsigned b, unsigned sel)
{
    uint64_t ab = ((uint64_t)b << 32) | a;
    unsigned r = 0;
    for (int i = 0; i < 4; i++) {
        unsigned s = (sel >> (4 * i)) & 7;
        r |= (unsigned)((ab >> (8 * s)) & 0xff) << (8 * i);
    }
    return r;
}
static inline int __mul24(int a, int b) { return a * b; }
static inline unsigned __umul24(unsigned a, unsigned b) { return a * b; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }

/* fibers are cooperative, so plain read-modify-write is atomic here */
template <class T>
static inline T atomicAdd(T *p, T v)
{
    T o = *p;
    *p = o + v;
    return o;
}
template <class T>
static inline T atomicOr(T *p, T v)
{
    T o = *p;
    *p = o | v;
    return o;
}
template <class T>
static inline T atomicMax(T *p, T v)
{
    T o = *p;
    if (v > o) *p = v;
    return o;
}
template <class T>
static inline T atomicMin(T *p, T v)
{
    T o = *p;
    if (v < o) *p = v;
    return o;
}
template <class T>
static inline T atomicExch(T *p, T v)
{
    T o = *p;
    *p = v;
    return o;
}
static inline void __threadfence() {}
static inline void __threadfence_block() {}

template <class T>
static inline T min(T a, T b) { return a < b ? a : b; }
template <class T>
static inline T max(T a, T b) { return a > b ? a : b; }

#endif
