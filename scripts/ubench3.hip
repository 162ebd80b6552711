// Field-primitive throughput probe: candidate implementations of Goldilocks mul/add/sub on gfx950,
// each checked against the host reference on random + edge inputs, then timed (8 waves/SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ministark_amd/csrc scripts/ubench3.hip -o scripts/ubench3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include "gl.h"
#include "gl_dev.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

static constexpr uint64_t P = gl::P;
#define ITERS 512

// ---- op wrappers: (a, b) -> value.  b is canonical (twiddle / product), a may be weak.
struct OpMulPlainC   { static __device__ __forceinline__ uint64_t f(uint64_t a, uint64_t b) { return gl::mul(a, b); } };
struct OpMontC       { static __device__ __forceinline__ uint64_t f(uint64_t a, uint64_t b) { return gl::mont_mul(a, b); } };
struct OpMontDev     { static __device__ __forceinline__ uint64_t f(uint64_t a, uint64_t b) { return gld::mmul(a, b); } };
struct OpAddC        { static __device__ __forceinline__ uint64_t f(uint64_t a, uint64_t b) { return gl::add(a, b); } };
struct OpSubC        { static __device__ __forceinline__ uint64_t f(uint64_t a, uint64_t b) { return gl::sub(a, b); } };
struct OpAddLazyC    { static __device__ __forceinline__ uint64_t f(uint64_t a, uint64_t b) { return gl::add_lazy(a, b); } };
struct OpSubLazyC    { static __device__ __forceinline__ uint64_t f(uint64_t a, uint64_t b) { return gl::sub_lazy(a, b); } };
struct OpAddLazyDev  { static __device__ __forceinline__ uint64_t f(uint64_t a, uint64_t b) { return gld::add_lazy(a, b); } };
struct OpSubLazyDev  { static __device__ __forceinline__ uint64_t f(uint64_t a, uint64_t b) { return gld::sub_lazy(a, b); } };
struct OpCanonDev    { static __device__ __forceinline__ uint64_t f(uint64_t a, uint64_t b) { return gld::canon(a) ^ (b & 0); } };
template <int S> struct OpShift { static __device__ __forceinline__ uint64_t f(uint64_t a, uint64_t b) { return gld::mul_pow2<S>(a) ^ (b & 0); } };

// correctness: out[i] = f(a[i], b[i])
template <class Op> __global__ void k_apply(const uint64_t* a, const uint64_t* b, uint64_t* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = Op::f(a[i], b[i]);
}
// throughput: 8 independent chains x_k = f(x_k, w) ; w canonical
template <class Op> __global__ void __launch_bounds__(256) k_rate(uint64_t* out, uint64_t seed) {
    uint64_t t = (threadIdx.x + 1) * 0x9E3779B97F4A7C15ull + seed;
    uint64_t x[8];
    #pragma unroll
    for (int k = 0; k < 8; k++) x[k] = t * (k + 3);
    uint64_t w = (t * 0xD1B54A32D192ED03ull) % P;
    for (int i = 0; i < ITERS; i++) {
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            #pragma unroll
            for (int k = 0; k < 8; k++) x[k] = Op::f(x[k], w);
        }
    }
    uint64_t acc = 0;
    #pragma unroll
    for (int k = 0; k < 8; k++) acc ^= x[k];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// a full radix-16 network + 15 twiddle products per 16 elements, as used in the pass kernels
template <int VARIANT> __global__ void __launch_bounds__(256) k_net(uint64_t* out, const uint64_t* tw, uint64_t seed) {
    uint64_t t = (threadIdx.x + 1) * 0x9E3779B97F4A7C15ull + seed;
    uint64_t x[16];
    #pragma unroll
    for (int k = 0; k < 16; k++) x[k] = (t * (k + 3)) % P;
    uint64_t w[16];
    #pragma unroll
    for (int k = 0; k < 16; k++) w[k] = tw[(threadIdx.x * 16 + k) & 4095];
    for (int i = 0; i < ITERS / 16; i++) {
        if constexpr (VARIANT == 0) {
            gld::dft16_ref<false>(x);
            #pragma unroll
            for (int k = 1; k < 16; k++) x[k] = gl::mul(x[k], w[k]);
        } else {
            gld::dft16<false>(x);
            #pragma unroll
            for (int k = 1; k < 16; k++) x[k] = gld::mmul(x[k], w[k]);
            x[0] = gld::canon(x[0]);
        }
    }
    uint64_t acc = 0;
    #pragma unroll
    for (int k = 0; k < 16; k++) acc ^= x[k];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int VARIANT> __global__ void k_net_check(const uint64_t* in, uint64_t* out, int ngroups) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    uint64_t x[16];
    for (int k = 0; k < 16; k++) x[k] = in[g * 16 + k];
    if constexpr (VARIANT == 0) gld::dft16_ref<false>(x);
    else if constexpr (VARIANT == 1) { gld::dft16<false>(x); for (int k = 0; k < 16; k++) x[k] = gld::canon(x[k]); }
    else if constexpr (VARIANT == 2) gld::dft16_ref<true>(x);
    else { gld::dft16<true>(x); for (int k = 0; k < 16; k++) x[k] = gld::canon(x[k]); }
    for (int k = 0; k < 16; k++) out[g * 16 + k] = x[k];
}

static std::vector<uint64_t> test_values(int n, uint64_t seed, bool canonical) {
    std::vector<uint64_t> v;
    const uint64_t edge[] = {0, 1, 2, P - 1, P - 2, P, P + 1, ~0ull, ~0ull - 1, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFF00000000ull,
                             0x8000000000000000ull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFEFFFFFFFFull, 0x00000001FFFFFFFFull};
    for (uint64_t e : edge) v.push_back(canonical ? e % P : e);
    uint64_t s = seed;
    while ((int)v.size() < n) {
        s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        v.push_back(canonical ? z % P : z);
    }
    return v;
}

template <class Op, class Ref>
static void run(const char* name, Ref ref, bool a_canonical, bool out_canonical) {
    const int n = 1 << 14;
    // all pairs of edge values + random
    std::vector<uint64_t> a0 = test_values(128, 1, a_canonical), b0 = test_values(128, 2, true), a, b;
    for (uint64_t x : a0) for (uint64_t y : b0) { a.push_back(x); b.push_back(y); }
    a.resize(n); b.resize(n);
    uint64_t *da, *db, *dout;
    CK(hipMalloc(&da, n * 8)); CK(hipMalloc(&db, n * 8)); CK(hipMalloc(&dout, std::max(n, 256 * 8 * 256) * 8));
    CK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_apply<Op>, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
    std::vector<uint64_t> out(n);
    CK(hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; i++) {
        uint64_t want = ref(a[i], b[i]);
        bool ok = out_canonical ? (out[i] == want) : (out[i] % P == want % P);
        if (!ok && bad++ < 3) printf("   MISMATCH %s a=%016llx b=%016llx got=%016llx want=%016llx\n", name, (unsigned long long)a[i], (unsigned long long)b[i], (unsigned long long)out[i], (unsigned long long)want);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(k_rate<Op>, dim3(blocks), dim3(256), 0, 0, dout, 1ull);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rate<Op>, dim3(blocks), dim3(256), 0, 0, dout, (uint64_t)r);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    double wave_ops = blocks * 4.0 * ITERS * 32.0;
    printf("OP %-18s %s  %8.3f ms => %6.1f cycles/wave-op/SIMD @2.4GHz\n", name, bad ? "WRONG" : "ok   ", best, best * 1e-3 * 2.4e9 / (wave_ops / 1024.0));
    fflush(stdout);
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dout));
}

template <int VARIANT> static void run_net(const char* name) {
    const int ng = 4096;
    std::vector<uint64_t> in = test_values(ng * 16, 77, true), want(ng * 16), got(ng * 16);
    // host reference: naive DFT with w16 (or its inverse)
    const uint64_t w16 = gl::root_of_unity(4);
    const uint64_t w = (VARIANT >= 2) ? gl::inv(w16) : w16;
    for (int g = 0; g < ng; g++) for (int c = 0; c < 16; c++) {
        uint64_t acc = 0;
        for (int a = 0; a < 16; a++) acc = gl::add(acc, gl::mul(in[g * 16 + a], gl::pow(w, (uint64_t)(a * c) % 16)));
        want[g * 16 + c] = acc;
    }
    uint64_t *din, *dout, *dtw;
    CK(hipMalloc(&din, ng * 16 * 8)); CK(hipMalloc(&dout, std::max(ng * 16, 256 * 8 * 256) * 8)); CK(hipMalloc(&dtw, 4096 * 8));
    CK(hipMemcpy(din, in.data(), ng * 16 * 8, hipMemcpyHostToDevice));
    std::vector<uint64_t> tw = test_values(4096, 5, true);
    CK(hipMemcpy(dtw, tw.data(), 4096 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_net_check<VARIANT>, dim3(ng / 256), dim3(256), 0, 0, din, dout, ng);
    CK(hipMemcpy(got.data(), dout, ng * 16 * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < ng * 16; i++) if (got[i] != want[i] && bad++ < 3) printf("   NET MISMATCH %s idx %d got=%016llx want=%016llx\n", name, i, (unsigned long long)got[i], (unsigned long long)want[i]);
    if (VARIANT >= 2) { printf("NET %-22s %s\n", name, bad ? "WRONG" : "ok"); return; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 4;
    hipLaunchKernelGGL(k_net<VARIANT>, dim3(blocks), dim3(256), 0, 0, dout, dtw, 1ull);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_net<VARIANT>, dim3(blocks), dim3(256), 0, 0, dout, dtw, (uint64_t)r);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    double wave_elems = blocks * 4.0 * (ITERS / 16) * 16.0;     // element-visits per wave
    printf("NET %-22s %s  %8.3f ms => %6.1f cycles per element (dft16 + twiddle) per wave/SIMD @2.4GHz\n", name, bad ? "WRONG" : "ok   ",
           best, best * 1e-3 * 2.4e9 / (wave_elems / 1024.0));
    fflush(stdout);
}

int main() {
    run<OpMulPlainC>("mul plain (C)", [](uint64_t a, uint64_t b) { return gl::mul(a, b); }, false, true);
    run<OpMontC>("mont_mul (C)", [](uint64_t a, uint64_t b) { return gl::mont_mul(a % P, b); }, true, true);
    run<OpMontDev>("gld::mmul", [](uint64_t a, uint64_t b) { return gl::mont_mul(a % P, b); }, false, true);
    run<OpAddC>("add canonical (C)", [](uint64_t a, uint64_t b) { return gl::add(a, b); }, true, true);
    run<OpSubC>("sub canonical (C)", [](uint64_t a, uint64_t b) { return gl::sub(a, b); }, true, true);
    run<OpAddLazyC>("add_lazy (C)", [](uint64_t a, uint64_t b) { return gl::add(a % P, b); }, false, false);
    run<OpSubLazyC>("sub_lazy (C)", [](uint64_t a, uint64_t b) { return gl::sub(a % P, b); }, false, false);
    run<OpAddLazyDev>("gld::add_lazy", [](uint64_t a, uint64_t b) { return gl::add(a % P, b); }, false, false);
    run<OpSubLazyDev>("gld::sub_lazy", [](uint64_t a, uint64_t b) { return gl::sub(a % P, b); }, false, false);
    run<OpCanonDev>("gld::canon", [](uint64_t a, uint64_t) { return a % P; }, false, true);
    run<OpShift<12>>("mul_pow2<12>", [](uint64_t a, uint64_t) { return gl::mul(a, 1ull << 12); }, false, true);
    run<OpShift<24>>("mul_pow2<24>", [](uint64_t a, uint64_t) { return gl::mul(a, 1ull << 24); }, false, true);
    run<OpShift<36>>("mul_pow2<36>", [](uint64_t a, uint64_t) { return gl::mul(a, 1ull << 36); }, false, true);
    run<OpShift<48>>("mul_pow2<48>", [](uint64_t a, uint64_t) { return gl::mul(a, 1ull << 48); }, false, true);
    run<OpShift<60>>("mul_pow2<60>", [](uint64_t a, uint64_t) { return gl::mul(a, 1ull << 60); }, false, true);
    run<OpShift<72>>("mul_pow2<72>", [](uint64_t a, uint64_t) { return gl::mul(gl::mul(a, 1ull << 36), 1ull << 36); }, false, true);
    run<OpShift<84>>("mul_pow2<84>", [](uint64_t a, uint64_t) { return gl::mul(gl::mul(a, 1ull << 42), 1ull << 42); }, false, true);
    run_net<0>("dft16_ref + gl::mul");
    run_net<1>("dft16 (lazy/shift) + mmul");
    run_net<2>("dft16_ref inverse");
    run_net<3>("dft16 inverse");
    return 0;
}
