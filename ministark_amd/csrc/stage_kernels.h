// Element-wise "stage" kernels for gfx950: the reference's 17 *Stage wrappers
// (gpu/src/stage.rs:115-1155) over the kernels of gpu/src/metal/evaluation_shaders.h.metal:11-168,
// instantiated there per field pair by host_name strings (:172-513).  Here one template per
// operation, dispatched on (lhs field, rhs field); all values are Montgomery residues.
//   {Add,Mul}{Assign,Into}       dst[i] = lhs[i] (+|*) rhs[(i + shift) mod n]     (:58-99)
//   {Add,Mul}{Assign,Into}Const  dst[i] = lhs[i] (+|*) c                          (:101-145)
//   MulPow                       dst[i] = lhs[i] * rhs[(i + shift) mod n]^e        (:149-161)
//   {Neg,Inverse,Exp}{InPlace,Into}                                                (:11-56)
//   ConvertInto (Fp -> Fq3 embedding), FillBuff                                    (:121-127,163-168)
//   sum of columns (Matrix::sum_columns, src/matrix.rs:357-394, an AddAssign tree there)
// dst may alias lhs ("Assign"/"InPlace" forms).  The Fq3 inverse the reference leaves as todo!()
// (evaluation_shaders.h.metal:393,401; src/eval_gpu.rs:338) is provided.
// Pure streaming kernels: HBM-bound for add/neg/convert/fill, ALU-bound for inverse/exp.
#pragma once
#if !defined(__HIPCC_RTC__)      // hiprtc pre-includes the runtime declarations
#include <hip/hip_runtime.h>
#endif
#include "gl.h"
#include "gl_dev.h"
#include "fp252.h"

namespace msstage {

static constexpr int NT = 256;
static constexpr int MAXCOLS = 128;

struct FpT {
    using T = uint64_t;
    static constexpr int V = 1;
    static __device__ __forceinline__ T load(const uint64_t* p, size_t i) { return p[i]; }
    static __device__ __forceinline__ void store(uint64_t* p, size_t i, T x) { p[i] = x; }
    static __device__ __forceinline__ T add(T a, T b) { return gl::add(a, b); }
    static __device__ __forceinline__ T neg(T a) { return gl::neg(a); }
    static __device__ __forceinline__ T mul(T a, T b) { return gld::mmul(a, b); }
    static __device__ __forceinline__ T one() { return gl::ONE_MONT; }
    static __device__ __forceinline__ T inv(T a) {
        // x^(p-2), the 72-multiplication chain of felt_u64.h.metal:97-109
        auto sqn = [](T x, int n) { for (int i = 0; i < n; i++) x = gld::mmul(x, x); return x; };
        T t2 = gld::mmul(sqn(a, 1), a), t3 = gld::mmul(sqn(t2, 1), a), t6 = gld::mmul(sqn(t3, 3), t3);
        T t12 = gld::mmul(sqn(t6, 6), t6), t24 = gld::mmul(sqn(t12, 12), t12), t30 = gld::mmul(sqn(t24, 6), t6);
        T t31 = gld::mmul(sqn(t30, 1), a), t63 = gld::mmul(sqn(t31, 32), t31);
        return gld::mmul(sqn(t63, 1), a);
    }
};
struct Fq3T {
    using T = gl::Fq3;
    static constexpr int V = 3;
    static __device__ __forceinline__ T load(const uint64_t* p, size_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
    static __device__ __forceinline__ void store(uint64_t* p, size_t i, T x) { p[3 * i] = x.c0; p[3 * i + 1] = x.c1; p[3 * i + 2] = x.c2; }
    static __device__ __forceinline__ T add(T a, T b) { return gl::add(a, b); }
    static __device__ __forceinline__ T neg(T a) { return gl::neg(a); }
    static __device__ __forceinline__ T mul(T a, T b) {
        // 6 base multiplications (felt_u64.h.metal:205-231); the non-residue 2 is a doubling
        const uint64_t ad = gld::mmul(a.c0, b.c0), be = gld::mmul(a.c1, b.c1), cf = gld::mmul(a.c2, b.c2);
        const uint64_t x = gl::sub(gl::sub(gld::mmul(gl::add(a.c1, a.c2), gl::add(b.c1, b.c2)), be), cf);
        const uint64_t y = gl::sub(gl::sub(gld::mmul(gl::add(a.c0, a.c1), gl::add(b.c0, b.c1)), ad), be);
        const uint64_t z = gl::add(gl::sub(gl::sub(gld::mmul(gl::add(a.c0, a.c2), gl::add(b.c0, b.c2)), ad), cf), be);
        return {gl::add(ad, gl::dbl(x)), gl::add(y, gl::dbl(cf)), z};
    }
    static __device__ __forceinline__ T one() { return {gl::ONE_MONT, 0, 0}; }
    static __device__ __forceinline__ T inv(T a) {
        const uint64_t s0 = gl::sub(gld::mmul(a.c0, a.c0), gl::dbl(gld::mmul(a.c1, a.c2)));
        const uint64_t s1 = gl::sub(gl::dbl(gld::mmul(a.c2, a.c2)), gld::mmul(a.c0, a.c1));
        const uint64_t s2 = gl::sub(gld::mmul(a.c1, a.c1), gld::mmul(a.c0, a.c2));
        const uint64_t n = gl::add(gld::mmul(a.c0, s0), gl::dbl(gl::add(gld::mmul(a.c2, s1), gld::mmul(a.c1, s2))));
        const uint64_t ni = FpT::inv(n);
        return {gld::mmul(s0, ni), gld::mmul(s1, ni), gld::mmul(s2, ni)};
    }
};

struct Fp252T {
    using T = f252::E;
    static constexpr int V = 4;
    static __device__ __forceinline__ T load(const uint64_t* p, size_t i) { return {{p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]}}; }
    static __device__ __forceinline__ void store(uint64_t* p, size_t i, const T& x) { p[4 * i] = x.l[0]; p[4 * i + 1] = x.l[1]; p[4 * i + 2] = x.l[2]; p[4 * i + 3] = x.l[3]; }
    // The same words when the 64 lanes of a wave hold 64 CONSECUTIVE elements in the order of gld::wave_elem (lane l: element
    // i = base + wave_elem(l), all lanes active).  A store instruction carries 16 bytes per lane: written straight out, each of an
    // element's two instructions covers half of every 32 bytes, and partially written 64-byte blocks reach HBM once per instruction --
    // twice the bytes of the column (round 6, WRITE_SIZE of every 252-bit kernel; a pair of lanes writing one whole element per
    // instruction, 32 of every 64 bytes, changed nothing).  The lanes of a pair swap halves (one DPP move per register) and every
    // instruction writes one contiguous KiB.
    static __device__ __forceinline__ void store_wave(uint64_t* p, size_t i, const T& x) {
        const unsigned lane = threadIdx.x & 63u;
        const bool odd = lane & 1u;
        const uint64_t r0 = gld::lane_xor1(odd ? x.l[0] : x.l[2]), r1 = gld::lane_xor1(odd ? x.l[1] : x.l[3]);   // even lanes give their high half, odd lanes their low half
        uint64_t* q = p + 4 * (i - gld::wave_elem(lane)) + 2 * lane;
        q[0] = odd ? r0 : x.l[0]; q[1] = odd ? r1 : x.l[1];           // elements base .. base + 31: words 0, 1 from the even lane, 2, 3 from the odd lane
        q[128] = odd ? x.l[2] : r0; q[129] = odd ? x.l[3] : r1;       // elements base + 32 .. base + 63
    }
    static __device__ __forceinline__ T add(const T& a, const T& b) { return f252::add(a, b); }
    static __device__ __forceinline__ T neg(const T& a) { return f252::neg(a); }
    static __device__ __forceinline__ T mul(const T& a, const T& b) { return f252::mul(a, b); }
    static __device__ __forceinline__ T one() { return f252::one(); }
    static __device__ __forceinline__ T inv(const T& a) { return f252::inv(a); }
};

// mixed-field operations: rhs in Fp acts on an Fq3 lhs as in gpu/src/fields.rs:99-188
template <class L, class R> struct Mix;
template <> struct Mix<FpT, FpT> {
    static __device__ __forceinline__ uint64_t add(uint64_t a, uint64_t b) { return gl::add(a, b); }
    static __device__ __forceinline__ uint64_t mul(uint64_t a, uint64_t b) { return gld::mmul(a, b); }
};
template <> struct Mix<Fq3T, Fq3T> {
    static __device__ __forceinline__ gl::Fq3 add(gl::Fq3 a, gl::Fq3 b) { return gl::add(a, b); }
    static __device__ __forceinline__ gl::Fq3 mul(gl::Fq3 a, gl::Fq3 b) { return Fq3T::mul(a, b); }
};
template <> struct Mix<Fp252T, Fp252T> {
    static __device__ __forceinline__ f252::E add(const f252::E& a, const f252::E& b) { return f252::add(a, b); }
    static __device__ __forceinline__ f252::E mul(const f252::E& a, const f252::E& b) { return f252::mul(a, b); }
};
template <> struct Mix<Fq3T, FpT> {
    static __device__ __forceinline__ gl::Fq3 add(gl::Fq3 a, uint64_t b) { return {gl::add(a.c0, b), a.c1, a.c2}; }
    static __device__ __forceinline__ gl::Fq3 mul(gl::Fq3 a, uint64_t b) { return {gld::mmul(a.c0, b), gld::mmul(a.c1, b), gld::mmul(a.c2, b)}; }
};

template <class F>
__device__ __forceinline__ typename F::T powu(typename F::T a, unsigned e) {
    typename F::T r = F::one();
    while (e) { if (e & 1) r = F::mul(r, a); e >>= 1; if (e) a = F::mul(a, a); }
    return r;
}

// shift already normalised to [0, n)
template <class L, class R, int OP>
__global__ void __launch_bounds__(NT) k_binary(uint64_t* dst, const uint64_t* lhs, const uint64_t* rhs, size_t n, size_t shift) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        size_t j = i + shift; if (j >= n) j -= n;
        const auto a = L::load(lhs, i);
        const auto b = R::load(rhs, j);
        L::store(dst, i, OP == 0 ? Mix<L, R>::add(a, b) : Mix<L, R>::mul(a, b));
    }
}
struct Const3 { uint64_t w[4]; };   // one element of any field (<= 4 words)
template <class L, class R, int OP>
__global__ void __launch_bounds__(NT) k_binary_const(uint64_t* dst, const uint64_t* lhs, Const3 c, size_t n) {
    const auto b = R::load(c.w, 0);
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const auto a = L::load(lhs, i);
        L::store(dst, i, OP == 0 ? Mix<L, R>::add(a, b) : Mix<L, R>::mul(a, b));
    }
}
template <class L, class R>
__global__ void __launch_bounds__(NT) k_mul_pow(uint64_t* dst, const uint64_t* lhs, const uint64_t* rhs, size_t n, size_t shift, unsigned e) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        size_t j = i + shift; if (j >= n) j -= n;
        L::store(dst, i, Mix<L, R>::mul(L::load(lhs, i), powu<R>(R::load(rhs, j), e)));
    }
}
// OP: 0 neg, 1 inverse, 2 exp
template <class F, int OP>
__global__ void __launch_bounds__(NT) k_unary(uint64_t* dst, const uint64_t* src, size_t n, unsigned e) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const auto a = F::load(src, i);
        F::store(dst, i, OP == 0 ? F::neg(a) : OP == 1 ? F::inv(a) : powu<F>(a, e));
    }
}
// InverseInto / InverseInPlace stages (gpu/src/stage.rs:808-853, 949-997) on long columns: Montgomery's trick over the K elements of a
// lane (elements tid + j * stride: coalesced) -- K - 1 products forward, ONE Fermat inverse, 2 (K - 1) products backward: three
// multiplications and 1/K of an inversion per element instead of 72 (Fp) / ~370 (Fp252).  Inverses are unique and every product is
// canonical, so the result is what the per-element stage writes, 0^-1 = 0 included.  dst may be src.
template <class T> __device__ __forceinline__ bool is_zero(const T& v);
template <> __device__ __forceinline__ bool is_zero<uint64_t>(const uint64_t& v) { return v == 0; }
template <> __device__ __forceinline__ bool is_zero<gl::Fq3>(const gl::Fq3& v) { return (v.c0 | v.c1 | v.c2) == 0; }
template <> __device__ __forceinline__ bool is_zero<f252::E>(const f252::E& v) { return (v.l[0] | v.l[1] | v.l[2] | v.l[3]) == 0; }
template <class F, int K>
__global__ void __launch_bounds__(NT) k_batch_inverse(uint64_t* dst, const uint64_t* src, size_t n) {
    using T = typename F::T;
    const size_t tid = (size_t)blockIdx.x * NT + threadIdx.x, stride = (size_t)gridDim.x * NT;
    T val[K], pre[K];
    T acc = F::one();
    #pragma unroll
    for (int j = 0; j < K; j++) {
        const size_t i = tid + (size_t)j * stride;
        pre[j] = acc;
        if (i < n) { val[j] = F::load(src, i); if (!is_zero<T>(val[j])) acc = F::mul(acc, val[j]); }
    }
    T inv = F::inv(acc);
    #pragma unroll
    for (int j = K - 1; j >= 0; j--) {
        const size_t i = tid + (size_t)j * stride;
        if (i < n) {
            if (is_zero<T>(val[j])) F::store(dst, i, val[j]);
            else { F::store(dst, i, F::mul(inv, pre[j])); inv = F::mul(inv, val[j]); }
        }
    }
}
static __global__ void __launch_bounds__(NT) k_convert_fp_fq3(uint64_t* dst, const uint64_t* src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const uint64_t x = src[i];
        dst[3 * i] = x; dst[3 * i + 1] = 0; dst[3 * i + 2] = 0;
    }
}
// word-granular fill: element pattern c.w[0..V)
static __global__ void __launch_bounds__(NT) k_fill(uint64_t* dst, Const3 c, size_t nwords, unsigned V) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < nwords; i += (size_t)gridDim.x * NT) dst[i] = c.w[i % V];
}
struct SumParams { const uint64_t* cols[MAXCOLS]; uint64_t* dst; size_t nwords; unsigned ncols; };
static __global__ void __launch_bounds__(NT) k_sum_columns(SumParams P) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < P.nwords; i += (size_t)gridDim.x * NT) {
        uint64_t acc = P.cols[0][i];                         // canonical
        for (unsigned c = 1; c < P.ncols; c++) acc = gl::add(acc, P.cols[c][i]);
        P.dst[i] = acc;
    }
}

static __global__ void __launch_bounds__(NT) k_sum_columns252(SumParams P) {      // nwords = number of elements here
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < P.nwords; i += (size_t)gridDim.x * NT) {
        f252::E acc = Fp252T::load(P.cols[0], i);
        for (unsigned c = 1; c < P.ncols; c++) acc = f252::add(acc, Fp252T::load(P.cols[c], i));
        Fp252T::store(P.dst, i, acc);
    }
}

}  // namespace msstage
