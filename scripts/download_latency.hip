// Latency of a small device-to-host copy on an idle stream: pageable destination (what ms_download did) against a pinned staging slot + memcpy,
// and against a kernel that writes into mapped pinned memory.  Build: hipcc --offload-arch=gfx950 -O3 scripts/download_latency.hip -o scripts/download_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void touch(unsigned* p) { p[threadIdx.x] += 1; }
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    char* pinned; CK(hipHostMalloc((void**)&pinned, 1 << 20, 0));
    std::vector<char> pageable(1 << 20);
    for (size_t bytes : {32, 4096, 65536, 262144}) {
        for (int mode = 0; mode < 2; mode++) {
            double best = 1e9, sum = 0;
            const int reps = 300;
            for (int r = 0; r < reps; r++) {
                hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, st, d);       // some work in front of the copy, as in the prover
                auto t0 = std::chrono::steady_clock::now();
                if (mode == 0) { CK(hipMemcpyAsync(pageable.data(), d, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }
                else { CK(hipMemcpyAsync(pinned, d, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); memcpy(pageable.data(), pinned, bytes); }
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                best = us < best ? us : best; sum += us;
            }
            printf("%7zu bytes  %-28s mean %7.1f us  best %7.1f us\n", bytes, mode == 0 ? "pageable destination" : "pinned slot + memcpy", sum / reps, best);
        }
    }
    return 0;
}
