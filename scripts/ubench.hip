// Micro-benchmarks that size the NTT design on MI355X (gfx950):
//   (1) issue rate of the integer instructions Goldilocks arithmetic is made of
//   (2) HBM / Infinity-Cache behaviour of the access patterns the NTT passes use
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench.hip -o scripts/ubench
// Run on the GPU box; prints one line per measurement.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// ---------------------------------------------------------------- (1) issue rates
#define ITERS 4096
#define REP8(x) x x x x x x x x

#define ASM8(S, IN, ...) asm volatile(S("%0") S("%1") S("%2") S("%3") S("%4") S("%5") S("%6") S("%7") \
    : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(IN) : "vcc");
#define RATE_KERNEL(name, TYPE, TTYPE, TINIT, S)                                        \
__global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) {             \
    uint32_t t = threadIdx.x + seed;                                                      \
    TYPE a0=(TYPE)t,a1=(TYPE)(t+1),a2=(TYPE)(t+2),a3=(TYPE)(t+3),a4=(TYPE)(t+4),a5=(TYPE)(t+5),a6=(TYPE)(t+6),a7=(TYPE)(t+7); \
    TTYPE tt = TINIT;                                                                     \
    for (int i = 0; i < ITERS; i++) { REP8(ASM8(S, tt)) }                                 \
    TYPE x = a0; x = x + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                \
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)x;                                    \
}
#define S_ADD(x)      "v_add_u32 " x ", " x ", %8\n"
#define S_MULLO(x)    "v_mul_lo_u32 " x ", " x ", %8\n"
#define S_MULHI(x)    "v_mul_hi_u32 " x ", " x ", %8\n"
#define S_MUL24(x)    "v_mul_u32_u24 " x ", " x ", %8\n"
#define S_MAD24(x)    "v_mad_u32_u24 " x ", " x ", %8, %8\n"
#define S_ALIGN(x)    "v_alignbit_b32 " x ", " x ", %8, 7\n"
#define S_ADD3(x)     "v_add3_u32 " x ", " x ", %8, %8\n"
#define S_CNDMASK(x)  "v_cndmask_b32 " x ", " x ", %8, vcc\n"
#define S_MAD64(x)    "v_mad_u64_u32 " x ", vcc, %8, %8, " x "\n"
#define S_LSHLADD64(x) "v_lshl_add_u64 " x ", " x ", 0, %8\n"
#define S_LSHL64(x)   "v_lshlrev_b64 " x ", 1, " x "\n"
#define S_CMP64(x)    "v_cmp_lt_u64 vcc, " x ", %8\n"
#define S_FMA64(x)    "v_fma_f64 " x ", " x ", %8, %8\n"
#define S_MULF64(x)   "v_mul_f64 " x ", " x ", %8\n"
#define S_ADDCO(x)    "v_add_co_u32 " x ", vcc, " x ", %8\n"
#define S_ADDC(x)     "v_addc_co_u32 " x ", vcc, " x ", %8, vcc\n"
#define S_SUBB(x)     "v_subb_co_u32 " x ", vcc, " x ", %8, vcc\n"
#define S_XOR(x)      "v_xor_b32 " x ", " x ", %8\n"
#define S_BFE(x)      "v_bfe_u32 " x ", " x ", 3, 7\n"

RATE_KERNEL(k_add_u32, uint32_t, uint32_t, t, S_ADD)
RATE_KERNEL(k_mul_lo_u32, uint32_t, uint32_t, t, S_MULLO)
RATE_KERNEL(k_mul_hi_u32, uint32_t, uint32_t, t, S_MULHI)
RATE_KERNEL(k_mul_u32_u24, uint32_t, uint32_t, t, S_MUL24)
RATE_KERNEL(k_mad_u32_u24, uint32_t, uint32_t, t, S_MAD24)
RATE_KERNEL(k_alignbit, uint32_t, uint32_t, t, S_ALIGN)
RATE_KERNEL(k_add3_u32, uint32_t, uint32_t, t, S_ADD3)
RATE_KERNEL(k_cndmask, uint32_t, uint32_t, t, S_CNDMASK)
RATE_KERNEL(k_addco, uint32_t, uint32_t, t, S_ADDCO)
RATE_KERNEL(k_addc, uint32_t, uint32_t, t, S_ADDC)
RATE_KERNEL(k_subb, uint32_t, uint32_t, t, S_SUBB)
RATE_KERNEL(k_mad_u64_u32, uint64_t, uint32_t, t, S_MAD64)
RATE_KERNEL(k_lshl_add_u64, uint64_t, uint64_t, (((uint64_t)t << 32) | t), S_LSHLADD64)
RATE_KERNEL(k_lshlrev_b64, uint64_t, uint32_t, t, S_LSHL64)
RATE_KERNEL(k_cmp_lt_u64, uint64_t, uint64_t, (((uint64_t)t << 32) | t), S_CMP64)
RATE_KERNEL(k_fma_f64, double, double, 1.0000001, S_FMA64)
RATE_KERNEL(k_mul_f64, double, double, 1.0000001, S_MULF64)

// a full Goldilocks multiply (compiler-scheduled), to see what hipcc achieves end to end
__device__ __forceinline__ uint64_t gl_mul_c(uint64_t a, uint64_t b) {
    unsigned __int128 x = (unsigned __int128)a * b;
    uint64_t xl = (uint64_t)x, xh = (uint64_t)(x >> 64);
    uint64_t tmp = xl << 32; uint64_t s = xl + tmp; uint64_t ov = s < xl;
    uint64_t bb = s - (s >> 32) - ov; uint64_t r = xh - bb;
    return (xh < bb) ? r - 0xFFFFFFFFull : r;
}
__global__ void __launch_bounds__(256) k_glmul(uint32_t* out, uint32_t seed) {
    uint64_t t = threadIdx.x + seed;
    uint64_t a0=t*0x9E3779B97F4A7C15ull,a1=a0+1,a2=a0+2,a3=a0+3, w = t * 0xD1B54A32D192ED03ull | 1;
    for (int i = 0; i < ITERS; i++) {
        #pragma unroll
        for (int k = 0; k < 4; k++) { a0 = gl_mul_c(a0, w); a1 = gl_mul_c(a1, w); a2 = gl_mul_c(a2, w); a3 = gl_mul_c(a3, w); }
    }
    uint64_t x = a0^a1^a2^a3;
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32);
}

template <typename K>
static void run_rate(const char* name, K kern, double insts_per_thread, uint32_t* d_out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves/SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, (uint32_t)r);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    double waves = blocks * 4.0;
    double wave_insts = waves * insts_per_thread;
    double per_simd = wave_insts / 1024.0;            // 1024 SIMDs
    double cyc = best * 1e-3 * 2.4e9;                 // at nominal 2.4 GHz
    printf("RATE %-16s %8.3f ms  => %6.2f cycles/wave-inst/SIMD @2.4GHz  (%.1f G lane-ops/s)\n",
           name, best, cyc / per_simd, wave_insts * 64 / (best * 1e-3) / 1e9);
    fflush(stdout);
}

// ---------------------------------------------------------------- (2) memory
__global__ void __launch_bounds__(256) k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (; i < n16; i += stride) { uint4 v = src[i]; v.x ^= 1; dst[i] = v; }
}
__global__ void __launch_bounds__(256) k_copy8(const uint2* __restrict__ src, uint2* __restrict__ dst, size_t n8) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (; i < n8; i += stride) { uint2 v = src[i]; v.x ^= 1; dst[i] = v; }
}
// Tile-strided read+write: each block handles tiles of R rows x SEG u64, row stride = n/R.
// Reads the tile (SEG contiguous words per row), writes it back to dst at the same place.
template <int SEG>
__global__ void __launch_bounds__(256) k_tile(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst,
                                              size_t n, int R, int xcd_swz) {
    size_t row_stride = n / R;
    size_t ntiles = row_stride / SEG;
    int lane_seg = threadIdx.x % SEG, row0 = threadIdx.x / SEG;
    constexpr int ROWS_PER_IT = 256 / SEG;
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        size_t tl = tile;
        if (xcd_swz) { // consecutive tiles -> same XCD (block b runs on XCD b%8)
            size_t per = ntiles / 8; tl = (tile % 8) * per + tile / 8;
        }
        size_t base = tl * SEG + lane_seg;
        uint64_t acc[16];
        for (int r = row0, k = 0; r < R; r += ROWS_PER_IT * 16) {
            #pragma unroll
            for (int u = 0; u < 16; u++) { int rr = r + u * ROWS_PER_IT; acc[u] = rr < R ? src[base + (size_t)rr * row_stride] : 0; }
            #pragma unroll
            for (int u = 0; u < 16; u++) { int rr = r + u * ROWS_PER_IT; if (rr < R) dst[base + (size_t)rr * row_stride] = acc[u] + 1; }
            (void)k;
        }
    }
}

static float time_ms(void (*fn)(void*), void* ctx, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fn(ctx); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0)); fn(ctx); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

struct CopyCtx { const void* src; void* dst; size_t bytes; int blocks; int width; };
static void do_copy(void* p) {
    CopyCtx* c = (CopyCtx*)p;
    if (c->width == 16) hipLaunchKernelGGL(k_copy16, dim3(c->blocks), dim3(256), 0, 0, (const uint4*)c->src, (uint4*)c->dst, c->bytes / 16);
    else hipLaunchKernelGGL(k_copy8, dim3(c->blocks), dim3(256), 0, 0, (const uint2*)c->src, (uint2*)c->dst, c->bytes / 8);
}
struct TileCtx { const uint64_t* src; uint64_t* dst; size_t n; int R; int seg; int blocks; int swz; };
static void do_tile(void* p) {
    TileCtx* c = (TileCtx*)p;
    switch (c->seg) {
    case 4:  hipLaunchKernelGGL(k_tile<4>,  dim3(c->blocks), dim3(256), 0, 0, c->src, c->dst, c->n, c->R, c->swz); break;
    case 8:  hipLaunchKernelGGL(k_tile<8>,  dim3(c->blocks), dim3(256), 0, 0, c->src, c->dst, c->n, c->R, c->swz); break;
    case 16: hipLaunchKernelGGL(k_tile<16>, dim3(c->blocks), dim3(256), 0, 0, c->src, c->dst, c->n, c->R, c->swz); break;
    case 32: hipLaunchKernelGGL(k_tile<32>, dim3(c->blocks), dim3(256), 0, 0, c->src, c->dst, c->n, c->R, c->swz); break;
    case 64: hipLaunchKernelGGL(k_tile<64>, dim3(c->blocks), dim3(256), 0, 0, c->src, c->dst, c->n, c->R, c->swz); break;
    }
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d clock=%d kHz memclk=%d kHz L2=%d\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.memoryClockRate, prop.l2CacheSize);

    uint32_t* d_out; CK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
    const double per = (double)ITERS * 64;
    run_rate("v_add_u32", k_add_u32, per, d_out);
    run_rate("v_mul_lo_u32", k_mul_lo_u32, per, d_out);
    run_rate("v_mul_hi_u32", k_mul_hi_u32, per, d_out);
    run_rate("v_mul_u32_u24", k_mul_u32_u24, per, d_out);
    run_rate("v_mad_u32_u24", k_mad_u32_u24, per, d_out);
    run_rate("v_alignbit_b32", k_alignbit, per, d_out);
    run_rate("v_add3_u32", k_add3_u32, per, d_out);
    run_rate("v_cndmask_b32", k_cndmask, per, d_out);
    run_rate("v_add_co_u32", k_addco, per, d_out);
    run_rate("v_addc_co_u32", k_addc, per, d_out);
    run_rate("v_subb_co_u32", k_subb, per, d_out);
    run_rate("v_mad_u64_u32", k_mad_u64_u32, per, d_out);
    run_rate("v_lshl_add_u64", k_lshl_add_u64, per, d_out);
    run_rate("v_lshlrev_b64", k_lshlrev_b64, per, d_out);
    run_rate("v_cmp_lt_u64", k_cmp_lt_u64, per, d_out);
    run_rate("v_fma_f64", k_fma_f64, per, d_out);
    run_rate("v_mul_f64", k_mul_f64, per, d_out);
    run_rate("gl_mul(hipcc) x16", k_glmul, (double)ITERS * 16, d_out);   // "insts" = field muls here

    // memory
    const size_t GiB = 1ull << 30;
    uint8_t *a, *b; CK(hipMalloc(&a, 2 * GiB)); CK(hipMalloc(&b, 2 * GiB));
    CK(hipMemset(a, 1, 2 * GiB)); CK(hipMemset(b, 2, 2 * GiB));
    for (size_t bytes : {(size_t)128 << 20, (size_t)64 << 20, GiB, 2 * GiB}) {
        for (int width : {16, 8}) for (int blocks : {2048, 8192}) {
            CopyCtx c{a, b, bytes, blocks, width};
            float ms = time_ms(do_copy, &c, 9);
            printf("COPY out-of-place %5zu MiB width=%2d blocks=%5d: %8.3f ms  %7.1f GB/s (r+w)\n", bytes >> 20, width, blocks, ms, 2.0 * bytes / ms / 1e6);
        }
        CopyCtx c{a, a, bytes, 2048, 16};
        float ms = time_ms(do_copy, &c, 9);
        printf("COPY in-place     %5zu MiB width=16 blocks= 2048: %8.3f ms  %7.1f GB/s (r+w)\n", bytes >> 20, ms, 2.0 * bytes / ms / 1e6);
        fflush(stdout);
    }
    // strided tiles: n u64 words
    for (size_t logn : {24, 27}) {
        size_t n = 1ull << logn;
        for (int R : {256, 4096}) for (int seg : {4, 8, 16, 32, 64}) for (int swz : {0, 1}) for (int inplace : {0, 1}) {
            if (R == 4096 && seg > 16) continue;
            TileCtx c{(const uint64_t*)a, inplace ? (uint64_t*)a : (uint64_t*)b, n, R, seg, 2048, swz};
            float ms = time_ms(do_tile, &c, 7);
            printf("TILE n=2^%zu R=%4d seg=%3dB swz=%d %s: %8.3f ms  %7.1f GB/s (r+w)\n", logn, R, seg * 8, swz,
                   inplace ? "in-place " : "out-place", ms, 2.0 * n * 8 / ms / 1e6);
            fflush(stdout);
        }
    }
    return 0;
}
