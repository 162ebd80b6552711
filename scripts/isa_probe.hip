// Compile-only probe: instruction counts of the limb-form pieces (no GPU needed).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S -o /tmp/isa_probe.s scripts/isa_probe.hip --cuda-device-only -I ministark_amd/csrc
#include <hip/hip_runtime.h>
#include "gl.h"
#include "gl_dev.h"
#include "gl_limb.h"

using namespace glimb;

typedef const __attribute__((address_space(4))) uint64_t* cptr_t;
__device__ __forceinline__ W4 w4_at(const uint64_t* t, unsigned slot) {
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8), aligned(8)));
    const u32x8 v = *(const __attribute__((address_space(4))) u32x8*)(t + 4 * (size_t)slot);
    W4 r;
    r.lo[0] = v[0]; r.hi[0] = v[1]; r.lo[1] = v[2]; r.hi[1] = v[3];
    r.lo[2] = v[4]; r.hi[2] = v[5]; r.lo[3] = v[6]; r.hi[3] = v[7];
    return r;
}

// ---- candidates -------------------------------------------------------------------------------------
// fold through the carry-out of v_mad_u64_u32
__device__ __forceinline__ uint64_t fold_co(uint32_t a0, uint64_t H) {
    const uint64_t base = ((uint64_t)(uint32_t)H << 32) | a0;
    const uint32_t h1 = (uint32_t)(H >> 32);
    uint64_t z, cm;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(cm) : "v"(h1), "v"(base));
    uint32_t c01;
    asm("v_cndmask_b32 %0, 0, 1, %1" : "=v"(c01) : "s"(cm));
    return (uint64_t)c01 * 0xFFFFFFFFull + z;
}
__device__ __forceinline__ uint64_t mul_fold_co_probe(const L4& x, const W4& w) {
    uint64_t alo = (uint64_t)x.l[0] * w.lo[0];
    #pragma unroll
    for (int i = 1; i < 4; i++) alo += (uint64_t)x.l[i] * w.lo[i];
    uint64_t H = (uint64_t)x.l[0] * w.hi[0] + (alo >> 32);
    #pragma unroll
    for (int i = 1; i < 4; i++) H += (uint64_t)x.l[i] * w.hi[i];
    return fold_co((uint32_t)alo, H);
}
// the same in plain C (does the compiler find the carry-out?)
__device__ __forceinline__ uint64_t mul_fold_c128(const L4& x, const W4& w) {
    uint64_t alo = (uint64_t)x.l[0] * w.lo[0];
    #pragma unroll
    for (int i = 1; i < 4; i++) alo += (uint64_t)x.l[i] * w.lo[i];
    uint64_t H = (uint64_t)x.l[0] * w.hi[0] + (alo >> 32);
    #pragma unroll
    for (int i = 1; i < 4; i++) H += (uint64_t)x.l[i] * w.hi[i];
    const uint64_t base = ((uint64_t)(uint32_t)H << 32) | (uint32_t)alo;
    const u128 t = (u128)(uint32_t)(H >> 32) * 0xFFFFFFFFull + base;
    uint64_t z = (uint64_t)t;
    const uint32_t c = (uint32_t)(t >> 64);
    return (uint64_t)c * 0xFFFFFFFFull + z;
}

// (the three-copy product on loads is gl_limb.h's mul3_to_limbs / Q3 since round 3)

// ---- kernels ------------------------------------------------------------------------------------------
struct Args { const uint64_t* src; uint64_t* dst; const uint64_t* tab; const uint64_t* qtab; unsigned n; };

template <int MODE>
__global__ void __launch_bounds__(256, 4) k_level(Args A) {
    const size_t i = blockIdx.x * 256 + threadIdx.x;
    const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint64_t x[16];
    #pragma unroll
    for (int a = 0; a < 16; a++) x[a] = A.src[i + (size_t)a * A.n];
    L4 v[16];
    if (MODE == 3) {
        Q3 q;
        #pragma unroll
        for (int j = 0; j < 3; j++) { const uint64_t t = A.qtab[3 * threadIdx.x + j]; q.lo[j] = (uint32_t)t; q.hi[j] = (uint32_t)(t >> 32); }
        #pragma unroll
        for (int a = 0; a < 16; a++) v[a] = mul3_to_limbs(x[a], q);
    } else if (MODE == 4) {
        const uint64_t q = A.qtab[threadIdx.x];
        #pragma unroll
        for (int a = 0; a < 16; a++) v[a] = mul_to_limbs(x[a], q);
    } else {
        #pragma unroll
        for (int a = 0; a < 16; a++) v[a] = from_u64(x[a]);
    }
    dft<16, false>(v);
    #pragma unroll
    for (int a = 0; a < 16; a++) {
        const W4 w = w4_at(A.tab, wv * 16 + a);
        uint64_t r;
        if (MODE == 0) r = mul_fold<false>(v[a], w);
        else if (MODE == 2) r = mul_fold_c128(v[a], w);
        else r = mul_fold_co_probe(v[a], w);
        A.dst[i + (size_t)a * A.n] = r;
    }
}
template __global__ void k_level<0>(Args);
template __global__ void k_level<1>(Args);
template __global__ void k_level<2>(Args);
template __global__ void k_level<3>(Args);
template __global__ void k_level<4>(Args);

// the pieces on their own
__global__ void k_from(Args A) {
    const size_t i = blockIdx.x * 256 + threadIdx.x;
    L4 v = from_u64(A.src[i]);
    A.dst[i] = (uint64_t)v.l[0] + v.l[1] + v.l[2] + v.l[3];
}
__global__ void k_net(Args A) {
    const size_t i = blockIdx.x * 256 + threadIdx.x;
    L4 v[16];
    #pragma unroll
    for (int a = 0; a < 16; a++) { const uint64_t x = A.src[i + (size_t)a * A.n]; v[a].l[0] = (uint32_t)x; v[a].l[1] = (uint32_t)(x >> 32); v[a].l[2] = (uint32_t)x >> 3; v[a].l[3] = (uint32_t)(x >> 35); }
    dft<16, false, false>(v);
    #pragma unroll
    for (int a = 0; a < 16; a++) A.dst[i + (size_t)a * A.n] = ((uint64_t)(v[a].l[0] ^ v[a].l[1]) << 32) | (v[a].l[2] ^ v[a].l[3]);
}
