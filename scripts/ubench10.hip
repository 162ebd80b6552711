// Round-4 micro-benchmark: what would the strided pass of a TWO-pass 2^24-point transform cost in bare memory traffic?
// Three radix-256 passes cannot go below ~143 us per column (docs/DESIGN_HISTORY.md 8.1).  A two-pass split 2^24 = R x (2^24 / R) needs a strided
// pass whose tile holds R points of W adjacent sub-transforms: R x W x 8 bytes of LDS, and only W x 8 contiguous bytes per row.
//   R = 4096, W = 4   128 KiB, 32-byte runs      R = 2048, W = 8   128 KiB, 64-byte runs      R = 1024, W = 16  128 KiB, 128-byte runs
//   R = 2048, W = 4    64 KiB (two workgroups per CU), 32-byte runs
// The kernel loads the tile, does SPIN units of stand-in arithmetic per element, passes it through LDS (transposed read), does SPIN
// more, and stores to the same positions of the destination.  MAP 1 places the W-blocks that share a 128-byte line on one XCD,
// next to each other in launch order, so that the line is fetched from HBM once.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench10.hip -o scripts/ubench10
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

static constexpr int NCOL = 8;
static constexpr unsigned LOGN = 24;
struct Cols { const uint64_t* src[NCOL]; uint64_t* dst[NCOL]; };

template <int SPIN>
__device__ __forceinline__ uint64_t work(uint64_t v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    #pragma unroll
    for (int s = 0; s < SPIN; s++) {
        lo += hi; hi ^= lo; lo += 0x9E3779B9u; hi += lo;
        const uint64_t m = (uint64_t)lo * 0x85EBCA6Bu + hi;
        lo = (uint32_t)m; hi = (uint32_t)(m >> 32);
    }
    if (SPIN == 0) lo += 1;
    return ((uint64_t)hi << 32) | lo;
}

template <int R, int W, int NTH, int SPIN, bool NT, int MAP>
__global__ void __launch_bounds__(NTH) k_strided(Cols C) {
    extern __shared__ uint64_t lds[];
    const uint64_t* __restrict__ src = C.src[blockIdx.y];
    uint64_t* __restrict__ dst = C.dst[blockIdx.y];
    constexpr int PER = R * W / NTH;
    const size_t stride = ((size_t)1 << LOGN) / R;
    unsigned b = blockIdx.x;
    if (MAP == 1) { const unsigned x = b & 7, i = b >> 3; b = x * (gridDim.x >> 3) + i; }
    const unsigned t = threadIdx.x;
    uint64_t v[PER];
    #pragma unroll
    for (int i = 0; i < PER; i++) {
        const unsigned e = t + NTH * i, jj = e % W, k = e / W;
        const uint64_t* p = src + (size_t)k * stride + (size_t)b * W + jj;
        v[i] = NT ? __builtin_nontemporal_load(p) : *p;
    }
    #pragma unroll
    for (int i = 0; i < PER; i++) lds[t + NTH * i] = work<SPIN>(v[i]);
    __syncthreads();
    #pragma unroll
    for (int i = 0; i < PER; i++) v[i] = work<SPIN>(lds[((t + NTH * i) * 17u) % (R * W)]);     // a permuted read: 17 is odd, the map is a bijection
    #pragma unroll
    for (int i = 0; i < PER; i++) {
        const unsigned e = t + NTH * i, jj = e % W, k = e / W;
        uint64_t* p = dst + (size_t)k * stride + (size_t)b * W + jj;
        if (NT) __builtin_nontemporal_store(v[i], p); else *p = v[i];
    }
}

// the row pass: one workgroup owns ROWLEN contiguous points
template <int ROWLEN, int NTH, int SPIN, bool NT>
__global__ void __launch_bounds__(NTH) k_rows(Cols C) {
    extern __shared__ uint64_t lds[];
    const uint64_t* __restrict__ src = C.src[blockIdx.y] + (size_t)blockIdx.x * ROWLEN;
    uint64_t* __restrict__ dst = C.dst[blockIdx.y] + (size_t)blockIdx.x * ROWLEN;
    constexpr int PER = ROWLEN / NTH;
    const unsigned t = threadIdx.x;
    uint64_t v[PER];
    #pragma unroll
    for (int i = 0; i < PER; i++) v[i] = NT ? __builtin_nontemporal_load(src + t + NTH * i) : src[t + NTH * i];
    #pragma unroll
    for (int i = 0; i < PER; i++) lds[t + NTH * i] = work<SPIN>(v[i]);
    __syncthreads();
    #pragma unroll
    for (int i = 0; i < PER; i++) v[i] = work<SPIN>(lds[((t + NTH * i) * 17u) % ROWLEN]);
    #pragma unroll
    for (int i = 0; i < PER; i++) { if (NT) __builtin_nontemporal_store(v[i], dst + t + NTH * i); else dst[t + NTH * i] = v[i]; }
}

static uint64_t *IN[NCOL], *SCR[NCOL];
template <class F>
static double timeit(const char* name, F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int rep = 0; rep < 7; rep++) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; i++) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float m; CK(hipEventElapsedTime(&m, e0, e1)); ms.push_back(m / 10);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    const double us = ms[ms.size() / 2] * 1000.0 / NCOL;
    printf("%-96s %8.2f us/column  %6.2f TB/s\n", name, us, 2.0 * 8 * (1 << LOGN) / us * 1e-6);
    return us;
}

template <int R, int W, int NTH, int SPIN, bool NT, int MAP>
static void strided(const char* what) {
    auto k = k_strided<R, W, NTH, SPIN, NT, MAP>;
    const int lds = R * W * 8;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    Cols C; for (int c = 0; c < NCOL; c++) { C.src[c] = IN[c]; C.dst[c] = SCR[c]; }
    char nm[160];
    snprintf(nm, sizeof nm, "strided R=%d W=%d (%d-byte runs, %d KiB LDS, %d threads) %s %s, %d units: %s", R, W, W * 8, lds >> 10, NTH, NT ? "nt" : "default", MAP ? "XCD map" : "launch order", SPIN, what);
    timeit(nm, [&] { hipLaunchKernelGGL(k, dim3((1u << LOGN) / (R * W), NCOL), dim3(NTH), lds, 0, C); });
}
template <int ROWLEN, int NTH, int SPIN, bool NT>
static void rows(const char* what) {
    auto k = k_rows<ROWLEN, NTH, SPIN, NT>;
    const int lds = ROWLEN * 8;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    Cols C; for (int c = 0; c < NCOL; c++) { C.src[c] = SCR[c]; C.dst[c] = SCR[c]; }
    char nm[160];
    snprintf(nm, sizeof nm, "rows of %d points (%d KiB LDS, %d threads) %s, %d units, in place: %s", ROWLEN, lds >> 10, NTH, NT ? "nt" : "default", SPIN, what);
    timeit(nm, [&] { hipLaunchKernelGGL(k, dim3((1u << LOGN) / ROWLEN, NCOL), dim3(NTH), lds, 0, C); });
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d\n", prop.name, prop.multiProcessorCount);
    const size_t bytes = (size_t)8 << LOGN;
    for (int c = 0; c < NCOL; c++) { CK(hipMalloc(&IN[c], bytes)); CK(hipMalloc(&SCR[c], bytes)); CK(hipMemset(IN[c], c + 1, bytes)); CK(hipMemset(SCR[c], 3, bytes)); }
    for (int i = 0; i < 100; i++) strided<1024, 16, 1024, 17, true, 0>("warm-up");
    for (int round = 0; round < 2; round++) {
        strided<1024, 16, 1024, 0, true, 0>("");
        strided<1024, 16, 1024, 0, true, 1>("");
        strided<2048, 8, 1024, 0, true, 0>("");
        strided<2048, 8, 1024, 0, true, 1>("");
        strided<2048, 8, 1024, 0, false, 1>("");
        strided<4096, 4, 1024, 0, true, 0>("");
        strided<4096, 4, 1024, 0, true, 1>("");
        strided<4096, 4, 1024, 0, false, 1>("");
        strided<2048, 4, 512, 0, true, 1>("");
        strided<2048, 4, 512, 0, false, 1>("");
        strided<1024, 8, 512, 0, true, 1>("");
        strided<2048, 8, 1024, 12, true, 1>("~125 instructions per element per pass");
        strided<4096, 4, 1024, 12, true, 1>("~125 instructions per element per pass");
        strided<2048, 4, 512, 12, true, 1>("~125 instructions per element per pass");
        rows<4096, 512, 0, true>("");
        rows<8192, 1024, 0, true>("");
        rows<8192, 512, 0, true>("");
        rows<8192, 1024, 12, true>("~125 instructions per element per pass");
        rows<4096, 512, 12, true>("~125 instructions per element per pass");
    }
    return 0;
}
