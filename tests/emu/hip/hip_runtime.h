// TEST INFRASTRUCTURE ONLY -- a tiny HIP execution-model simulator for g++.
//
// The product is libministark_hip.so, built by hipcc for gfx950; it has no CPU
// path.  This header lets the SAME kernel and host sources be compiled by g++
// into tests/emu/_build/libministark_emu.so so that kernel index math, LDS
// exchanges and barrier placement can be checked against the oracle in this
// GPU-less container before a run is spent on a real MI355X.  It is never
// loaded by the ministark_amd package (ministark_amd/_lib.py loads only the
// hipcc-built library and raises if it or the GPU is missing).
//
// Model: blocks run one after another; the threads of a block are ucontext
// fibers; __syncthreads() yields to a round-robin scheduler, so a sweep over
// all fibers advances the block from one barrier to the next.  __shared__
// becomes `static` (one block at a time, so one copy is enough).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__ __restrict
#define MS_EMU 1

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
extern uint3_emu threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

typedef int hipError_t;
typedef void* hipStream_t;
typedef struct { double t; }* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
// code-object API: the simulator build has no runtime compiler (MS_NO_JIT).  A hipFunction_t is the entry of a kernel that
// tests/emu/emu_jit.h compiled with g++ from the generated source: void entry(void** kernelParams); hipModuleLaunchKernel is defined
// below, after the launcher.
typedef void* hipModule_t;
typedef void* hipFunction_t;
static inline hipError_t hipModuleUnload(hipModule_t) { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
#define hipHostMallocPortable 0x1
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
#define hipStreamNonBlocking 1
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
#define hipEventDisableTiming 2
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }

namespace emu {
void run_block_threads(unsigned nthreads, const std::function<void()>& body);
void yield_barrier();
}  // namespace emu

static inline void __syncthreads() { emu::yield_barrier(); }

template <class K, class... Args>
static inline void emu_launch(K kernel, dim3 grid, dim3 block, Args... args) {
    gridDim = grid; blockDim = block;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                blockIdx = {bx, by, bz};
                emu::run_block_threads(block.x * block.y * block.z, [&]() { kernel(args...); });
            }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(kernel, dim3(grid), dim3(block), __VA_ARGS__)
static inline hipError_t hipModuleLaunchKernel(hipFunction_t fn, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, unsigned, hipStream_t,
                                               void** params, void**) {
    if (!fn) return hipErrorInvalidValue;
    emu_launch((void (*)(void**))fn, dim3(gx, gy, gz), dim3(bx, by, bz), params);
    return hipSuccess;
}

struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
// device intrinsics used by the kernels
static inline unsigned __builtin_amdgcn_readfirstlane(unsigned x) { return x; }   // only applied to wave-uniform values
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline unsigned __brev(unsigned x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
}
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; if (v < o) *p = v; return o; }
static inline unsigned long long __brevll(unsigned long long x) {
    return ((unsigned long long)__brev((unsigned)x) << 32) | __brev((unsigned)(x >> 32));
}
