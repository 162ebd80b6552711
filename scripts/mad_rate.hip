// v_mad_u64_u32 on gfx950: issue rate and dependent latency.  One workgroup of 64 x W threads per CU (W waves per SIMD x 4), each lane runs N multiply-adds:
// CHAINS independent accumulators (1 = every multiply-add waits for the previous one).  Prints cycles per multiply-add per wave and the
// aggregate rate.  Build: hipcc --offload-arch=gfx950 -O3 scripts/mad_rate.hip -o scripts/mad_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int CHAINS, bool ADD32>
__global__ void k(uint64_t* out, uint32_t a, uint32_t b, int n) {
    uint64_t acc[CHAINS];
    uint32_t x = a + threadIdx.x;
    #pragma unroll
    for (int c = 0; c < CHAINS; c++) acc[c] = threadIdx.x + c;
    for (int i = 0; i < n; i++) {
        #pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (ADD32) { uint32_t lo = (uint32_t)acc[c], hi = (uint32_t)(acc[c] >> 32); lo = lo * 3u + x; hi ^= lo; acc[c] = ((uint64_t)hi << 32) | lo; }   // two plain 32-bit ops as a reference chain... (v_mad_u32_u24-free)
            else acc[c] = (uint64_t)(uint32_t)acc[c] * b + acc[c];          // v_mad_u64_u32 acc, lo(acc), b, acc
        }
    }
    uint64_t s = 0;
    #pragma unroll
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS, bool ADD32>
int run(int waves_per_simd, uint64_t* d) {
    const int n = 20000, blocks = 256, threads = 64 * 4 * waves_per_simd;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<CHAINS, ADD32>), dim3(blocks), dim3(threads), 0, 0, d, 12345u, 0x9E3779B9u, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<CHAINS, ADD32>), dim3(blocks), dim3(threads), 0, 0, d, 12345u, 0x9E3779B9u, n);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops_per_wave = (double)n * CHAINS, cycles = ms * 1e-3 * 2.4e9;
    printf("%s chains=%d waves/SIMD=%d: %.3f ms, %.2f cycles per op per wave, %.2f cycles per op per SIMD\n", ADD32 ? "32-bit pair " : "v_mad_u64_u32", CHAINS, waves_per_simd, ms,
           cycles / ops_per_wave, cycles / (ops_per_wave * waves_per_simd));
    return 0;
}
int main() {
    uint64_t* d; CK(hipMalloc(&d, 256 * 1024 * 8));
    for (int w : {1, 2, 4}) { run<1, false>(w, d); run<2, false>(w, d); run<4, false>(w, d); run<8, false>(w, d); }
    for (int w : {1, 4}) { run<1, true>(w, d); run<4, true>(w, d); }
    return 0;
}
