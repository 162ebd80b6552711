// v_mad_u64_u32 on gfx950: issue rate by operand kind, and dependent latency.  256 workgroups of 64 x 4 x W threads (W waves per SIMD), each lane runs
// N rounds of C multiply-adds on C independent 64-bit accumulators (inline asm, so the operand kinds are what the label says):
//   VV  both factors in vector registers       SV  one factor in a scalar register (a wave-uniform twiddle)
// Prints cycles per multiply-add per SIMD (4.0 = one issue slot).  Build: hipcc --offload-arch=gfx950 -O3 scripts/mad_rate.hip -o scripts/mad_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int C, bool SV>
__global__ void k(uint64_t* out, uint32_t b, int n) {
    uint64_t acc[C];
    uint32_t x[C];
    #pragma unroll
    for (int c = 0; c < C; c++) { acc[c] = threadIdx.x + c; x[c] = threadIdx.x * 2654435761u + c; }
    for (int i = 0; i < n; i++) {
        #pragma unroll
        for (int c = 0; c < C; c++) {
            if (SV) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "s"(b), "v"(x[c]) : "vcc");
            else asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(x[(c + 1) % C]), "v"(x[c]) : "vcc");
        }
    }
    uint64_t s = 0;
    #pragma unroll
    for (int c = 0; c < C; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int C, bool SV>
int run(int w, uint64_t* d) {
    const int n = 4000, blocks = 256, threads = 256 * w;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<C, SV>), dim3(blocks), dim3(threads), 0, 0, d, 0x9E3779B9u, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<C, SV>), dim3(blocks), dim3(threads), 0, 0, d, 0x9E3779B9u, n);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_simd = (double)n * C * w;
    printf("%s  %2d independent accumulators, %d waves per SIMD: %6.2f cycles per multiply-add per SIMD at 2.4 GHz (%.3f ms)\n", SV ? "SV" : "VV", C, w, ms * 1e-3 * 2.4e9 / per_simd, ms);
    return 0;
}
int main() {
    uint64_t* d; CK(hipMalloc(&d, 256 * 1024 * 8));
    for (int w : {1, 2, 4}) { run<1, false>(w, d); run<4, false>(w, d); run<16, false>(w, d); run<1, true>(w, d); run<4, true>(w, d); run<16, true>(w, d); }
    return 0;
}
