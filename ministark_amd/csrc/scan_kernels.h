// Running products / running evaluations along a column, and row / digest gathers
// (SURVEY.md 8(f) rank 4): the reference builds its extension columns with sequential host loops
//     state = init;  for row: ext[row] = state;  state = a_row * state + b_row
// (examples/brainfuck/trace.rs:108-289: permutation running products are the b = 0 case, the
// input / output running evaluations state*gamma + value the general one, masked rows are the
// identity map a = 1, b = 0) and reads query rows / authentication nodes back one by one
// (src/trace.rs:115-157, src/merkle.rs:149-206).
//
// scan: the maps x -> a*x + b compose associatively, so the column is scanned in three launches:
//   scan_reduce  one workgroup per 4096 (or 1024) rows: each lane composes its 16 (4) consecutive maps, the
//                workgroup composes the 256 lane maps in order -> one aggregate map per block
//   scan_blocks  one workgroup walks the block aggregates -> the state at the start of every block
//   scan_apply   per block: lane maps again, an in-order prefix over the 256 lanes gives every
//                lane its incoming state, then the 16 rows are written
// All arithmetic is exact field arithmetic: results equal the sequential loop bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include "gl.h"
#include "gl_dev.h"
#include "stage_kernels.h"

namespace msscan {

static constexpr int NT = 256;
// rows per lane: 16 (Fp) / 8 (wider elements) for long columns, 4 below 2^20 rows (four times the workgroups, a quarter of the
// serial chain per lane: a 2^16-row column is latency-bound, not throughput-bound)

struct ScanParams {
    const uint64_t* a;       // multipliers (nullptr: all one)
    const uint64_t* b;       // addends     (nullptr: all zero)
    uint64_t* out;
    uint64_t* agg;           // [nblocks][2] elements: aggregate map of every block
    uint64_t* block_state;   // [nblocks] elements: state at the start of every block
    uint64_t init[4];
    size_t n;
    unsigned nblocks;
    int inclusive;
};

template <class F> struct Map { typename F::T a, b; };

template <class F> __device__ __forceinline__ typename F::T f_zero();
template <> __device__ __forceinline__ uint64_t f_zero<msstage::FpT>() { return 0; }
template <> __device__ __forceinline__ gl::Fq3 f_zero<msstage::Fq3T>() { return {0, 0, 0}; }
template <> __device__ __forceinline__ f252::E f_zero<msstage::Fp252T>() { return f252::zero(); }

// `l` first, then `r`:  x -> r.a*(l.a*x + l.b) + r.b
template <class F, bool HAS_A, bool HAS_B>
__device__ __forceinline__ Map<F> compose(const Map<F>& l, const Map<F>& r) {
    Map<F> m;
    m.a = HAS_A ? F::mul(r.a, l.a) : l.a;
    if constexpr (HAS_B) m.b = HAS_A ? F::add(F::mul(r.a, l.b), r.b) : F::add(l.b, r.b);
    else m.b = l.b;
    return m;
}
template <class F, bool HAS_A, bool HAS_B>
__device__ __forceinline__ typename F::T apply(const Map<F>& m, const typename F::T& x) {
    typename F::T y = HAS_A ? F::mul(m.a, x) : x;
    if constexpr (HAS_B) y = F::add(y, m.b);
    return y;
}
template <class F> __device__ __forceinline__ Map<F> identity() { return {F::one(), f_zero<F>()}; }

template <class F, bool HAS_A, bool HAS_B>
__device__ __forceinline__ Map<F> load_map(const ScanParams& P, size_t i) {
    Map<F> m = identity<F>();
    if (i < P.n) {
        if constexpr (HAS_A) m.a = F::load(P.a, i);
        if constexpr (HAS_B) m.b = F::load(P.b, i);
    }
    return m;
}

// in-order inclusive scan of one map per lane across the workgroup (Hillis-Steele through LDS);
// returns the inclusive result of this lane, *excl = composition of all lanes before it
template <class F, bool HAS_A, bool HAS_B>
__device__ __forceinline__ Map<F> wg_scan(Map<F> m, Map<F>* sh, Map<F>* excl) {
    const unsigned t = threadIdx.x;
    for (unsigned off = 1; off < (unsigned)NT; off <<= 1) {
        sh[t] = m;
        __syncthreads();
        if (t >= off) m = compose<F, HAS_A, HAS_B>(sh[t - off], m);
        __syncthreads();
    }
    sh[t] = m;
    __syncthreads();
    *excl = t ? sh[t - 1] : identity<F>();
    __syncthreads();
    return m;
}

// ---- the block's rows, read and written coalesced ----------------------------------------------------------------------
// A lane composes PER CONSECUTIVE rows, so reading them where they lie would make a wave touch 64 different lines per load
// (measured: 0.5 TB/s on a 2^22-row column).  Instead the NT * PER elements of a block cross LDS once: lanes run along the WORDS
// of the block (coalesced, whatever the element width), then every lane picks up its run.  One pad word per Q = lowbit(PER V)
// words makes the run stride odd, i.e. conflict-free.
template <int S> __device__ __forceinline__ unsigned tile_slot(unsigned p) { constexpr unsigned Q = S & (0u - S); return p + p / Q; }
template <class F, int PER> struct Tile { static constexpr int S = PER * F::V; static constexpr unsigned WORDS = NT * S + NT * S / (S & (0u - S)) + 1; };

template <class F, int PER>
__device__ __forceinline__ void load_runs(const uint64_t* __restrict__ src, size_t e0, size_t n, uint64_t* tile, typename F::T* out, const typename F::T& fill) {
    constexpr int V = F::V, S = PER * V;
    const unsigned t = threadIdx.x;
    const size_t w0 = e0 * V, wn = n * V;
    #pragma unroll
    for (int j = 0; j < S; j++) {
        const size_t w = w0 + (size_t)j * NT + t;
        tile[tile_slot<S>(j * NT + t)] = w < wn ? src[w] : 0;
    }
    __syncthreads();
    #pragma unroll
    for (int j = 0; j < PER; j++) {
        uint64_t words[V];
        #pragma unroll
        for (int v = 0; v < V; v++) words[v] = tile[tile_slot<S>((t * PER + j) * V + v)];
        out[j] = e0 + (size_t)t * PER + j < n ? F::load(words, 0) : fill;
    }
    __syncthreads();
}
template <class F, int PER>
__device__ __forceinline__ void store_runs(uint64_t* __restrict__ dst, size_t e0, size_t n, uint64_t* tile, const typename F::T* vals) {
    constexpr int V = F::V, S = PER * V;
    const unsigned t = threadIdx.x;
    #pragma unroll
    for (int j = 0; j < PER; j++) {
        uint64_t words[V];
        F::store(words, 0, vals[j]);
        #pragma unroll
        for (int v = 0; v < V; v++) tile[tile_slot<S>((t * PER + j) * V + v)] = words[v];
    }
    __syncthreads();
    const size_t w0 = e0 * V, wn = n * V;
    #pragma unroll
    for (int j = 0; j < S; j++) {
        const size_t w = w0 + (size_t)j * NT + t;
        if (w < wn) dst[w] = tile[tile_slot<S>(j * NT + t)];
    }
}
// the PER maps of this lane -> (a[], b[]) in registers and their in-order composition
template <class F, bool HAS_A, bool HAS_B, int PER>
__device__ __forceinline__ Map<F> lane_maps(const ScanParams& P, size_t e0, uint64_t* tile, typename F::T* a, typename F::T* b) {
    if constexpr (HAS_A) load_runs<F, PER>(P.a, e0, P.n, tile, a, F::one());
    if constexpr (HAS_B) load_runs<F, PER>(P.b, e0, P.n, tile, b, f_zero<F>());
    Map<F> m = identity<F>();
    #pragma unroll
    for (int j = 0; j < PER; j++) {
        Map<F> mj = identity<F>();
        if constexpr (HAS_A) mj.a = a[j];
        if constexpr (HAS_B) mj.b = b[j];
        m = j ? compose<F, HAS_A, HAS_B>(m, mj) : mj;
    }
    return m;
}

template <class F, bool HAS_A, bool HAS_B, int PER>
__global__ void __launch_bounds__(NT) scan_reduce(ScanParams P) {
    __shared__ Map<F> sh[NT];
    __shared__ uint64_t tile[Tile<F, PER>::WORDS];
    typename F::T a[HAS_A ? PER : 1], b[HAS_B ? PER : 1];
    Map<F> m = lane_maps<F, HAS_A, HAS_B, PER>(P, (size_t)blockIdx.x * (NT * PER), tile, a, b);
    Map<F> excl;
    m = wg_scan<F, HAS_A, HAS_B>(m, sh, &excl);
    if (threadIdx.x == NT - 1) {
        F::store(P.agg, 2 * (size_t)blockIdx.x, m.a);
        F::store(P.agg, 2 * (size_t)blockIdx.x + 1, m.b);
    }
}

// one workgroup: lane t walks blocks [t*chunk, (t+1)*chunk)
template <class F, bool HAS_A, bool HAS_B>
__global__ void __launch_bounds__(NT) scan_blocks(ScanParams P) {
    __shared__ Map<F> sh[NT];
    const unsigned chunk = (P.nblocks + NT - 1) / NT;
    const unsigned b0 = threadIdx.x * chunk;
    Map<F> m = identity<F>();
    for (unsigned k = 0; k < chunk; k++) {
        const unsigned blk = b0 + k;
        if (blk < P.nblocks) m = compose<F, HAS_A, HAS_B>(m, Map<F>{F::load(P.agg, 2 * (size_t)blk), F::load(P.agg, 2 * (size_t)blk + 1)});
    }
    Map<F> excl;
    wg_scan<F, HAS_A, HAS_B>(m, sh, &excl);
    typename F::T s = apply<F, HAS_A, HAS_B>(excl, F::load(P.init, 0));
    for (unsigned k = 0; k < chunk; k++) {
        const unsigned blk = b0 + k;
        if (blk >= P.nblocks) break;
        F::store(P.block_state, blk, s);
        s = apply<F, HAS_A, HAS_B>(Map<F>{F::load(P.agg, 2 * (size_t)blk), F::load(P.agg, 2 * (size_t)blk + 1)}, s);
    }
}

template <class F, bool HAS_A, bool HAS_B, int PER>
__global__ void __launch_bounds__(NT) scan_apply(ScanParams P) {
    __shared__ Map<F> sh[NT];
    __shared__ uint64_t tile[Tile<F, PER>::WORDS];
    const size_t e0 = (size_t)blockIdx.x * (NT * PER);
    typename F::T a[HAS_A ? PER : 1], b[HAS_B ? PER : 1];
    Map<F> m = lane_maps<F, HAS_A, HAS_B, PER>(P, e0, tile, a, b);   // the maps stay in registers: no second read of the column
    Map<F> excl;
    wg_scan<F, HAS_A, HAS_B>(m, sh, &excl);
    typename F::T s = apply<F, HAS_A, HAS_B>(excl, F::load(P.block_state, blockIdx.x));
    typename F::T out[PER];
    #pragma unroll
    for (int j = 0; j < PER; j++) {
        Map<F> mj = identity<F>();
        if constexpr (HAS_A) mj.a = a[j];
        if constexpr (HAS_B) mj.b = b[j];
        if (!P.inclusive) out[j] = s;
        s = apply<F, HAS_A, HAS_B>(mj, s);
        if (P.inclusive) out[j] = s;
    }
    store_runs<F, PER>(P.out, e0, P.n, tile, out);                  // in place is fine: the block's inputs were read above
}

// ---- gathers ---------------------------------------------------------------------------------
// out[p][c] = cols[c][pos[p]]  (row-major rows of the queried positions: Matrix::get_row, src/matrix.rs)
struct GatherRowsParams {
    const uint64_t* cols[msstage::MAXCOLS];
    const uint64_t* pos;     // device
    uint64_t* out;
    size_t npos;
    unsigned ncols, V;
};
static __global__ void __launch_bounds__(NT) gather_rows(GatherRowsParams P) {
    const size_t words_per_row = (size_t)P.ncols * P.V;
    const size_t total = P.npos * words_per_row;
    for (size_t idx = (size_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (size_t)gridDim.x * NT) {
        const size_t p = idx / words_per_row, w = idx % words_per_row;
        const unsigned c = (unsigned)(w / P.V), v = (unsigned)(w % P.V);
        P.out[idx] = P.cols[c][(size_t)P.pos[p] * P.V + v];
    }
}
// out[k] = src[idx[k]] for records of `words` u64 words (Merkle digests: 4 words)
static __global__ void __launch_bounds__(NT) gather_records(const uint64_t* __restrict__ src, const uint64_t* __restrict__ idx, uint64_t* __restrict__ out, size_t count, unsigned words) {
    const size_t total = count * words;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += (size_t)gridDim.x * NT)
        out[i] = src[(size_t)idx[i / words] * words + i % words];
}

// 32-byte records from anywhere to anywhere: pairs[2 r] = address of record r, pairs[2 r + 1] = where it goes (ms_gather_digests_multi builds
// the list on the host from validated indices: every digest gather of a proof's openings -- two dozen small launches -- in ONE)
static __global__ void __launch_bounds__(NT) copy_records32(const uint64_t* __restrict__ pairs, size_t count) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < count * 4; i += (size_t)gridDim.x * NT) {
        const size_t r = i >> 2;
        const uint64_t* s = (const uint64_t*)pairs[2 * r];
        uint64_t* d = (uint64_t*)pairs[2 * r + 1];
        d[i & 3] = s[i & 3];
    }
}

// composition_poly.chunks(k) into k columns (src/prover.rs:113-121): out[c][j] = in[j*k + c].  Lanes run along
// the interleaved input (coalesced reads; the k output streams are each contiguous per lane group).
struct DeinterleaveParams {
    uint64_t* out[msstage::MAXCOLS];
    const uint64_t* in;
    size_t n_out;
    unsigned k, V;
};
static __global__ void __launch_bounds__(NT) deinterleave(DeinterleaveParams P) {
    const size_t total = P.n_out * P.k * P.V;
    for (size_t idx = (size_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (size_t)gridDim.x * NT) {
        const size_t e = idx / P.V;                    // input element
        const unsigned v = (unsigned)(idx % P.V);
        P.out[e % P.k][(e / P.k) * P.V + v] = P.in[idx];
    }
}

}  // namespace msscan
