// Element-wise stages, FRI fold, running scans and query gathers (gpu/src/stage.rs:115-1155, src/fri.rs:526-567,
// examples/brainfuck/trace.rs:108-289, src/trace.rs:113-157).
#include "ms_internal.h"
#include "stage_kernels.h"
#include "fri_kernels.h"
#include "scan_kernels.h"
#include "fp252_kernels.h"

// ---------------------------------------------------------------------------------------
// element-wise stages
// ---------------------------------------------------------------------------------------
unsigned stream_grid(size_t n) { return (unsigned)std::max<size_t>(1, std::min<size_t>((n + msstage::NT - 1) / msstage::NT, 256 * 16)); }
static int field_pair(int lf, int rf, unsigned* VL, unsigned* VR) {
    MSCHK(field_words(lf, VL));
    MSCHK(field_words(rf, VR));
    if (*VR > *VL || ((*VL == 4) != (*VR == 4)))
        return fail(MS_ERR_UNSUPPORTED, "rhs field must embed into the lhs field (Fp,Fp / Fq3,Fq3 / Fq3,Fp / Fp252,Fp252)");
    return MS_OK;
}
static size_t norm_shift(long shift, size_t n) {
    if (n == 0) return 0;
    long long m = (long long)shift % (long long)n;
    if (m < 0) m += (long long)n;
    return (size_t)m;
}
extern "C" int ms_binary(ms_ctx* ctx, int op, int lf, int rf, size_t n, void* d_dst, const void* d_lhs, const void* d_rhs, long shift) {
    if (!ctx || !d_dst || !d_lhs || !d_rhs) return fail(MS_ERR_INVALID, "ms_binary: null argument");
    if (op != MS_ADD && op != MS_MUL) return fail(MS_ERR_INVALID, "unknown binary op %d", op);
    unsigned VL = 0, VR = 0;
    MSCHK(field_pair(lf, rf, &VL, &VR));
    if (n == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const size_t sh = norm_shift(shift, n);
    uint64_t* dst = (uint64_t*)d_dst; const uint64_t* l = (const uint64_t*)d_lhs; const uint64_t* r = (const uint64_t*)d_rhs;
    dim3 g(stream_grid(n)), b(msstage::NT);
    ProfScope ps(ctx, op == MS_ADD ? "stage_add" : "stage_mul", 8.0 * n * (2 * VL + VR));
    using namespace msstage;
    if (VL == 3 && VR == 3 && op == MS_ADD) {
        // component-wise: an Fq3 + Fq3 column add is an Fp add over 3n consecutive words (fully coalesced)
        hipLaunchKernelGGL((k_binary<FpT, FpT, 0>), dim3(stream_grid(3 * n)), b, 0, ctx->stream, dst, l, r, 3 * n, 3 * sh);
    }
    else if (VL == 4) { if (op == MS_ADD) hipLaunchKernelGGL((k_binary<Fp252T, Fp252T, 0>), g, b, 0, ctx->stream, dst, l, r, n, sh); else hipLaunchKernelGGL((k_binary<Fp252T, Fp252T, 1>), g, b, 0, ctx->stream, dst, l, r, n, sh); }
    else if (VL == 1) { if (op == MS_ADD) hipLaunchKernelGGL((k_binary<FpT, FpT, 0>), g, b, 0, ctx->stream, dst, l, r, n, sh); else hipLaunchKernelGGL((k_binary<FpT, FpT, 1>), g, b, 0, ctx->stream, dst, l, r, n, sh); }
    else if (VR == 3) { if (op == MS_ADD) hipLaunchKernelGGL((k_binary<Fq3T, Fq3T, 0>), g, b, 0, ctx->stream, dst, l, r, n, sh); else hipLaunchKernelGGL((k_binary<Fq3T, Fq3T, 1>), g, b, 0, ctx->stream, dst, l, r, n, sh); }
    else { if (op == MS_ADD) hipLaunchKernelGGL((k_binary<Fq3T, FpT, 0>), g, b, 0, ctx->stream, dst, l, r, n, sh); else hipLaunchKernelGGL((k_binary<Fq3T, FpT, 1>), g, b, 0, ctx->stream, dst, l, r, n, sh); }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_binary_const(ms_ctx* ctx, int op, int lf, int rf, size_t n, void* d_dst, const void* d_lhs, const void* h_const) {
    if (!ctx || !d_dst || !d_lhs || !h_const) return fail(MS_ERR_INVALID, "ms_binary_const: null argument");
    if (op != MS_ADD && op != MS_MUL) return fail(MS_ERR_INVALID, "unknown binary op %d", op);
    unsigned VL = 0, VR = 0;
    MSCHK(field_pair(lf, rf, &VL, &VR));
    if (n == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    msstage::Const3 c = {{0, 0, 0, 0}};
    memcpy(c.w, h_const, VR * 8);
    uint64_t* dst = (uint64_t*)d_dst; const uint64_t* l = (const uint64_t*)d_lhs;
    dim3 g(stream_grid(n)), b(msstage::NT);
    ProfScope ps(ctx, op == MS_ADD ? "stage_add_const" : "stage_mul_const", 16.0 * n * VL);
    using namespace msstage;
    if (VL == 3 && VR == 1 && op == MS_MUL) {
        // scaling an Fq3 column by an Fp constant acts on every word alike
        hipLaunchKernelGGL((k_binary_const<FpT, FpT, 1>), dim3(stream_grid(3 * n)), b, 0, ctx->stream, dst, l, c, 3 * n);
    }
    else if (VL == 4) { if (op == MS_ADD) hipLaunchKernelGGL((k_binary_const<Fp252T, Fp252T, 0>), g, b, 0, ctx->stream, dst, l, c, n); else hipLaunchKernelGGL((k_binary_const<Fp252T, Fp252T, 1>), g, b, 0, ctx->stream, dst, l, c, n); }
    else if (VL == 1) { if (op == MS_ADD) hipLaunchKernelGGL((k_binary_const<FpT, FpT, 0>), g, b, 0, ctx->stream, dst, l, c, n); else hipLaunchKernelGGL((k_binary_const<FpT, FpT, 1>), g, b, 0, ctx->stream, dst, l, c, n); }
    else if (VR == 3) { if (op == MS_ADD) hipLaunchKernelGGL((k_binary_const<Fq3T, Fq3T, 0>), g, b, 0, ctx->stream, dst, l, c, n); else hipLaunchKernelGGL((k_binary_const<Fq3T, Fq3T, 1>), g, b, 0, ctx->stream, dst, l, c, n); }
    else { if (op == MS_ADD) hipLaunchKernelGGL((k_binary_const<Fq3T, FpT, 0>), g, b, 0, ctx->stream, dst, l, c, n); else hipLaunchKernelGGL((k_binary_const<Fq3T, FpT, 1>), g, b, 0, ctx->stream, dst, l, c, n); }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_mul_pow(ms_ctx* ctx, int lf, int rf, size_t n, void* d_dst, const void* d_lhs, const void* d_rhs, unsigned power, long shift) {
    if (!ctx || !d_dst || !d_lhs || !d_rhs) return fail(MS_ERR_INVALID, "ms_mul_pow: null argument");
    unsigned VL = 0, VR = 0;
    MSCHK(field_pair(lf, rf, &VL, &VR));
    if (n == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const size_t sh = norm_shift(shift, n);
    uint64_t* dst = (uint64_t*)d_dst; const uint64_t* l = (const uint64_t*)d_lhs; const uint64_t* r = (const uint64_t*)d_rhs;
    dim3 g(stream_grid(n)), b(msstage::NT);
    ProfScope ps(ctx, "stage_mul_pow", 8.0 * n * (2 * VL + VR));
    using namespace msstage;
    if (VL == 4) hipLaunchKernelGGL((k_mul_pow<Fp252T, Fp252T>), g, b, 0, ctx->stream, dst, l, r, n, sh, power);
    else if (VL == 1) hipLaunchKernelGGL((k_mul_pow<FpT, FpT>), g, b, 0, ctx->stream, dst, l, r, n, sh, power);
    else if (VR == 3) hipLaunchKernelGGL((k_mul_pow<Fq3T, Fq3T>), g, b, 0, ctx->stream, dst, l, r, n, sh, power);
    else hipLaunchKernelGGL((k_mul_pow<Fq3T, FpT>), g, b, 0, ctx->stream, dst, l, r, n, sh, power);
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_unary(ms_ctx* ctx, int op, int field, size_t n, void* d_dst, const void* d_src, unsigned exponent) {
    if (!ctx || !d_dst || !d_src) return fail(MS_ERR_INVALID, "ms_unary: null argument");
    if (op != MS_NEG && op != MS_INV && op != MS_EXP) return fail(MS_ERR_INVALID, "unknown unary op %d", op);
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (n == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    uint64_t* dst = (uint64_t*)d_dst; const uint64_t* src = (const uint64_t*)d_src;
    dim3 g(stream_grid(n)), b(msstage::NT);
    ProfScope ps(ctx, op == MS_NEG ? "stage_neg" : op == MS_INV ? "stage_inverse" : "stage_exp", 16.0 * n * V);
    using namespace msstage;
    if (op == MS_INV && n >= 4096) {          // long columns: batch inversion (k_batch_inverse), K elements per lane
        auto blocks = [&](unsigned K) { return dim3((unsigned)std::min<size_t>((n + (size_t)NT * K - 1) / ((size_t)NT * K), 0x7FFFFFFFu)); };
        if (V == 1) hipLaunchKernelGGL((k_batch_inverse<FpT, 16>), blocks(16), b, 0, ctx->stream, dst, src, n);
        else if (V == 3) hipLaunchKernelGGL((k_batch_inverse<Fq3T, 8>), blocks(8), b, 0, ctx->stream, dst, src, n);
        else hipLaunchKernelGGL((k_batch_inverse<Fp252T, 8>), blocks(8), b, 0, ctx->stream, dst, src, n);
        HIPCHK(hipGetLastError());
        return MS_OK;
    }
    if (V == 4) {
        if (op == MS_NEG) hipLaunchKernelGGL((k_unary<Fp252T, 0>), g, b, 0, ctx->stream, dst, src, n, exponent);
        else if (op == MS_INV) hipLaunchKernelGGL((k_unary<Fp252T, 1>), g, b, 0, ctx->stream, dst, src, n, exponent);
        else hipLaunchKernelGGL((k_unary<Fp252T, 2>), g, b, 0, ctx->stream, dst, src, n, exponent);
    } else if (V == 1) {
        if (op == MS_NEG) hipLaunchKernelGGL((k_unary<FpT, 0>), g, b, 0, ctx->stream, dst, src, n, exponent);
        else if (op == MS_INV) hipLaunchKernelGGL((k_unary<FpT, 1>), g, b, 0, ctx->stream, dst, src, n, exponent);
        else hipLaunchKernelGGL((k_unary<FpT, 2>), g, b, 0, ctx->stream, dst, src, n, exponent);
    } else {
        if (op == MS_NEG) hipLaunchKernelGGL((k_unary<FpT, 0>), dim3(stream_grid(3 * n)), b, 0, ctx->stream, dst, src, 3 * n, exponent);   // component-wise
        else if (op == MS_INV) hipLaunchKernelGGL((k_unary<Fq3T, 1>), g, b, 0, ctx->stream, dst, src, n, exponent);
        else hipLaunchKernelGGL((k_unary<Fq3T, 2>), g, b, 0, ctx->stream, dst, src, n, exponent);
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_convert(ms_ctx* ctx, int dst_field, int src_field, size_t n, void* d_dst, const void* d_src) {
    if (!ctx || !d_dst || !d_src) return fail(MS_ERR_INVALID, "ms_convert: null argument");
    unsigned VD = 0, VS = 0;
    MSCHK(field_pair(dst_field, src_field, &VD, &VS));
    if (n == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    if (VD == VS) {
        if (d_dst != d_src) HIPCHK(hipMemcpyAsync(d_dst, d_src, n * VD * 8, hipMemcpyDeviceToDevice, ctx->stream));
        return MS_OK;
    }
    if (VD != 3 || VS != 1) return fail(MS_ERR_UNSUPPORTED, "only the Fp -> Fq3 embedding exists");
    ProfScope ps(ctx, "stage_convert", 8.0 * n * (VD + VS));
    hipLaunchKernelGGL(msstage::k_convert_fp_fq3, dim3(stream_grid(n)), dim3(msstage::NT), 0, ctx->stream, (uint64_t*)d_dst, (const uint64_t*)d_src, n);
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_fill(ms_ctx* ctx, int field, size_t n, void* d_dst, const void* h_value) {
    if (!ctx || !d_dst || !h_value) return fail(MS_ERR_INVALID, "ms_fill: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (n == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    msstage::Const3 c = {{0, 0, 0, 0}};
    memcpy(c.w, h_value, V * 8);
    ProfScope ps(ctx, "stage_fill", 8.0 * n * V);
    hipLaunchKernelGGL(msstage::k_fill, dim3(stream_grid(n * V)), dim3(msstage::NT), 0, ctx->stream, (uint64_t*)d_dst, c, n * V, V);
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_sum_columns(ms_ctx* ctx, int field, size_t n, const void* const* d_cols, unsigned ncols, void* d_dst) {
    if (!ctx || !d_cols || !d_dst) return fail(MS_ERR_INVALID, "ms_sum_columns: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (ncols == 0) return fail(MS_ERR_INVALID, "sum of zero columns");
    if (ncols > (unsigned)msstage::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d columns per call", msstage::MAXCOLS);
    if (n == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    msstage::SumParams P;
    memset(&P, 0, sizeof P);
    for (unsigned c = 0; c < ncols; c++) P.cols[c] = (const uint64_t*)d_cols[c];
    P.dst = (uint64_t*)d_dst; P.nwords = n * V; P.ncols = ncols;
    ProfScope ps(ctx, "sum_columns", 8.0 * n * V * (ncols + 1));
    if (V == 4) { P.nwords = n; hipLaunchKernelGGL(msstage::k_sum_columns252, dim3(stream_grid(n)), dim3(msstage::NT), 0, ctx->stream, P); }
    else hipLaunchKernelGGL(msstage::k_sum_columns, dim3(stream_grid(n * V)), dim3(msstage::NT), 0, ctx->stream, P);
    HIPCHK(hipGetLastError());
    return MS_OK;
}

// ---------------------------------------------------------------------------------------
// FRI fold
// ---------------------------------------------------------------------------------------
template <int V>
static void launch_fold(unsigned ff, dim3 g, hipStream_t st, const msfri::FoldParams& P) {
    switch (ff) {
    case 2: hipLaunchKernelGGL((msfri::fri_fold<2, V>), g, dim3(msfri::NT), 0, st, P); break;
    case 4: hipLaunchKernelGGL((msfri::fri_fold<4, V>), g, dim3(msfri::NT), 0, st, P); break;
    case 8: hipLaunchKernelGGL((msfri::fri_fold<8, V>), g, dim3(msfri::NT), 0, st, P); break;
    default: hipLaunchKernelGGL((msfri::fri_fold<16, V>), g, dim3(msfri::NT), 0, st, P); break;
    }
}
static int fri_fold_impl(ms_ctx* ctx, int field, unsigned log_n, unsigned folding_factor, const void* h_alpha,
                         const void* h_offset, size_t first_chunk, size_t nchunks, bool whole, const void* d_evals, void* d_out);
extern "C" int ms_fri_fold(ms_ctx* ctx, int field, unsigned log_n, unsigned folding_factor, const void* h_alpha,
                           const void* h_offset, const void* d_evals, void* d_out) {
    return fri_fold_impl(ctx, field, log_n, folding_factor, h_alpha, h_offset, 0, 0, true, d_evals, d_out);
}
// A row shard of a layer: chunks [first_chunk, first_chunk + nchunks) of the bit-reversed layer of 2^log_n evaluations (d_evals holds
// those nchunks * folding_factor evaluations, d_out receives nchunks).  The fold of a chunk needs nothing but the chunk and its
// position, so a rank that holds rows [r n / G, (r + 1) n / G) of a layer produces its rows of the next one without communication.
extern "C" int ms_fri_fold_rows(ms_ctx* ctx, int field, unsigned log_n, unsigned folding_factor, const void* h_alpha,
                                const void* h_offset, size_t first_chunk, size_t nchunks, const void* d_evals, void* d_out) {
    return fri_fold_impl(ctx, field, log_n, folding_factor, h_alpha, h_offset, first_chunk, nchunks, false, d_evals, d_out);
}
static int fri_fold_impl(ms_ctx* ctx, int field, unsigned log_n, unsigned folding_factor, const void* h_alpha,
                         const void* h_offset, size_t first_chunk, size_t nchunks, bool whole, const void* d_evals, void* d_out) {
    if (!ctx || !h_alpha || !d_evals || !d_out) return fail(MS_ERR_INVALID, "ms_fri_fold: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (folding_factor != 2 && folding_factor != 4 && folding_factor != 8 && folding_factor != 16)
        return fail(MS_ERR_UNSUPPORTED, "folding factor %u not supported (2, 4, 8, 16)", folding_factor);   // src/fri.rs:186-192
    unsigned log_ff = 0;
    while ((1u << log_ff) < folding_factor) log_ff++;
    if (log_n < log_ff || log_n > 32) return fail(MS_ERR_INVALID, "bad layer size 2^%u for folding factor %u", log_n, folding_factor);
    const size_t all_chunks = (size_t)1 << (log_n - log_ff);
    if (whole) { first_chunk = 0; nchunks = all_chunks; }
    else if (V == 4) return fail(MS_ERR_UNSUPPORTED, "ms_fri_fold_rows: Goldilocks fields");
    if (first_chunk > all_chunks || nchunks > all_chunks - first_chunk) return fail(MS_ERR_INVALID, "ms_fri_fold_rows: chunks [%zu, %zu) outside the layer", first_chunk, first_chunk + nchunks);
    if (nchunks == 0) return MS_OK;
    {   // lane c reads d_evals[c*ff .. c*ff + ff) and writes d_out[c]: overlapping buffers would corrupt the next layer
        const size_t in_bytes = nchunks * folding_factor * V * 8, out_bytes = in_bytes / folding_factor;
        const char *a = (const char*)d_evals, *b = (const char*)d_out;
        if (a < b + out_bytes && b < a + in_bytes) return fail(MS_ERR_INVALID, "ms_fri_fold: d_out overlaps d_evals (the fold is not an in-place operation)");
    }
    if (V == 4) {
        f252::E h252 = f252::one();
        if (h_offset) memcpy(h252.l, h_offset, 32);
        if (f252::is_zero(h252) || f252::geq_p(h252)) return fail(MS_ERR_INVALID, "domain offset must be a non-zero canonical element");
        std::lock_guard<std::mutex> lk(ctx->mu);
        HIPCHK(hipSetDevice(ctx->device));
        ms_ntt_plan* plan = nullptr;
        MSCHK(plan252_cached(ctx, log_n, true, f252::one(), &plan));
        ms252::Fold252Params P;
        memset(&P, 0, sizeof P);
        P.src = (const uint64_t*)d_evals; P.dst = (uint64_t*)d_out;
        P.tw_lo = plan->d252_tw_lo; P.tw_hi = plan->d252_tw_hi; P.lo_bits = plan->lo_bits; P.log_m = log_n - log_ff;
        const f252::E hinv = f252::inv(h252);
        memcpy(P.hinv, hinv.l, 32);
        memcpy(P.alpha, h_alpha, 32);
        const f252::E zinv = f252::pow_u64(f252::inv(f252::root_of_unity(log_n)), (uint64_t)1 << (log_n - log_ff));
        f252::E zp = f252::one();
        for (unsigned k = 0; k < folding_factor / 2; k++) { memcpy(P.zinv[k], zp.l, 32); zp = f252::mul(zp, zinv); }
        const size_t m = (size_t)1 << (log_n - log_ff);
        dim3 g((unsigned)((m + ms252::NT - 1) / ms252::NT));
        ProfScope ps(ctx, "fri_fold252", 32.0 * (((size_t)1 << log_n) + m));
        switch (folding_factor) {
        case 2: hipLaunchKernelGGL(ms252::fri_fold252<2>, g, dim3(ms252::NT), 0, ctx->stream, P); break;
        case 4: hipLaunchKernelGGL(ms252::fri_fold252<4>, g, dim3(ms252::NT), 0, ctx->stream, P); break;
        case 8: hipLaunchKernelGGL(ms252::fri_fold252<8>, g, dim3(ms252::NT), 0, ctx->stream, P); break;
        default: hipLaunchKernelGGL(ms252::fri_fold252<16>, g, dim3(ms252::NT), 0, ctx->stream, P); break;
        }
        HIPCHK(hipGetLastError());
        return MS_OK;
    }
    uint64_t h = 1;
    if (h_offset) { uint64_t h_m; memcpy(&h_m, h_offset, 8); h = gl::from_mont(h_m); }
    if (h == 0) return fail(MS_ERR_INVALID, "domain offset must be non-zero");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    // powers of w_n^-1: the tables of the size-n inverse plan (multi-pass layout needs log_n >= 12;
    // smaller layers get a dedicated two-level table through a plan of size max(n, 4096))
    ms_ntt_plan* plan = nullptr;
    const unsigned tl = std::max(log_n, 12u);
    MSCHK(ctx_plan(ctx, 1, tl, true, 1, &plan));
    msfri::FoldParams P;
    memset(&P, 0, sizeof P);
    P.src = (const uint64_t*)d_evals; P.dst = (uint64_t*)d_out;
    P.tw_lo = plan->d_tw_lo; P.tw_hi = plan->d_tw_hi; P.lo_bits = plan->lo_bits;
    P.log_m = log_n - log_ff;
    P.hinv = gl::to_mont(gl::inv(h));
    memcpy(P.alpha, h_alpha, V * 8);
    // table exponent scale: w_n = w_(2^tl)^(2^(tl-log_n)); fold it into the index below
    P.log_m |= (tl - log_n) << 8;
    P.c0 = first_chunk; P.count = nchunks;
    const size_t m = nchunks;
    dim3 g((unsigned)((m + msfri::NT - 1) / msfri::NT));
    ProfScope ps(ctx, "fri_fold", 8.0 * V * (m * folding_factor + m));
    if (V == 1) launch_fold<1>(folding_factor, g, ctx->stream, P); else launch_fold<3>(folding_factor, g, ctx->stream, P);
    HIPCHK(hipGetLastError());
    return MS_OK;
}

// ---------------------------------------------------------------------------------------
// running products / evaluations, query gathers
// ---------------------------------------------------------------------------------------
template <class F, bool HAS_A, bool HAS_B, int PER>
static void scan_launch_per(ms_ctx* ctx, const msscan::ScanParams& P) {
    using namespace msscan;
    { ProfScope ps(ctx, "scan_reduce", 8.0 * P.n * F::V * ((HAS_A ? 1 : 0) + (HAS_B ? 1 : 0)));
      hipLaunchKernelGGL((scan_reduce<F, HAS_A, HAS_B, PER>), dim3(P.nblocks), dim3(NT), 0, ctx->stream, P); }
    { ProfScope ps(ctx, "scan_blocks", 0.0);
      hipLaunchKernelGGL((scan_blocks<F, HAS_A, HAS_B>), dim3(1), dim3(NT), 0, ctx->stream, P); }
    { ProfScope ps(ctx, "scan_apply", 8.0 * P.n * F::V * (1 + (HAS_A ? 1 : 0) + (HAS_B ? 1 : 0)));
      hipLaunchKernelGGL((scan_apply<F, HAS_A, HAS_B, PER>), dim3(P.nblocks), dim3(NT), 0, ctx->stream, P); }
}
// rows per lane: the block's rows cross LDS once (NT * PER elements), so wide elements take fewer
static unsigned scan_rows_per_lane(size_t n, unsigned V) { return (n < ((size_t)1 << 20) || V == 4) ? 4 : V == 1 ? 16 : 8; }
template <class F, bool HAS_A, bool HAS_B>
static void scan_launch(ms_ctx* ctx, const msscan::ScanParams& P) {
    if constexpr (F::V == 4) scan_launch_per<F, HAS_A, HAS_B, 4>(ctx, P);        // 32-byte elements: 4 rows per lane at every length
    else if (scan_rows_per_lane(P.n, F::V) == 4) scan_launch_per<F, HAS_A, HAS_B, 4>(ctx, P);
    else scan_launch_per<F, HAS_A, HAS_B, F::V == 1 ? 16 : 8>(ctx, P);
}
extern "C" int ms_scan_affine(ms_ctx* ctx, int field, size_t n, const void* d_a, const void* d_b, const void* h_init, int inclusive, void* d_out) {
    if (!ctx || !d_out || !h_init) return fail(MS_ERR_INVALID, "ms_scan_affine: null argument");
    if (!d_a && !d_b) return fail(MS_ERR_INVALID, "ms_scan_affine: neither multipliers nor addends given");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (n == 0) return MS_OK;
    const size_t tile = (size_t)msscan::NT * scan_rows_per_lane(n, V);
    if ((n + tile - 1) / tile > 0xFFFFFFFFull) return fail(MS_ERR_UNSUPPORTED, "column too long");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    msscan::ScanParams P;
    memset(&P, 0, sizeof P);
    P.a = (const uint64_t*)d_a; P.b = (const uint64_t*)d_b; P.out = (uint64_t*)d_out;
    memcpy(P.init, h_init, V * 8);
    P.n = n; P.nblocks = (unsigned)((n + tile - 1) / tile); P.inclusive = inclusive != 0;
    void* tmp = nullptr;
    LockedPoolGuard pooled(ctx);
    MSCHK(pooled.alloc((size_t)P.nblocks * 3 * V * 8, &tmp));
    P.agg = (uint64_t*)tmp; P.block_state = (uint64_t*)tmp + (size_t)P.nblocks * 2 * V;
    using msstage::FpT; using msstage::Fq3T; using msstage::Fp252T;
    if (V == 1) {
        if (d_a && d_b) scan_launch<FpT, true, true>(ctx, P); else if (d_a) scan_launch<FpT, true, false>(ctx, P); else scan_launch<FpT, false, true>(ctx, P);
    } else if (V == 4) {
        if (d_a && d_b) scan_launch<Fp252T, true, true>(ctx, P); else if (d_a) scan_launch<Fp252T, true, false>(ctx, P); else scan_launch<Fp252T, false, true>(ctx, P);
    } else {
        if (d_a && d_b) scan_launch<Fq3T, true, true>(ctx, P); else if (d_a) scan_launch<Fq3T, true, false>(ctx, P); else scan_launch<Fq3T, false, true>(ctx, P);
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_gather_rows(ms_ctx* ctx, int field, size_t nrows, const void* const* d_cols, unsigned ncols,
                              const uint64_t* h_positions, size_t npos, void* d_out) {
    if (!ctx || !d_cols || !d_out || (npos && !h_positions)) return fail(MS_ERR_INVALID, "ms_gather_rows: null argument");
    const size_t fb = ms_field_bytes(field);
    if (!fb) return fail(MS_ERR_UNSUPPORTED, "unknown field %d", field);
    if (ncols == 0 || ncols > (unsigned)msstage::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "1..%d columns", msstage::MAXCOLS);
    for (size_t p = 0; p < npos; p++) if (h_positions[p] >= nrows) return fail(MS_ERR_INVALID, "row %llu out of range", (unsigned long long)h_positions[p]);
    if (npos == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const void* d_pos = nullptr;
    LockedPoolGuard pooled(ctx);
    MSCHK(stage_view(ctx, h_positions, npos * 8, &d_pos, pooled));
    msscan::GatherRowsParams P;
    memset(&P, 0, sizeof P);
    for (unsigned c = 0; c < ncols; c++) { if (!d_cols[c]) return fail(MS_ERR_INVALID, "null column %u", c); P.cols[c] = (const uint64_t*)d_cols[c]; }
    P.pos = (const uint64_t*)d_pos; P.out = (uint64_t*)d_out; P.npos = npos; P.ncols = ncols; P.V = (unsigned)(fb / 8);
    const size_t total = npos * ncols * P.V;
    { ProfScope ps(ctx, "gather_rows", 16.0 * total);
      hipLaunchKernelGGL(msscan::gather_rows, dim3(stream_grid(total)), dim3(msscan::NT), 0, ctx->stream, P); }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
// MerkleTreeImpl::prove's index walk (src/merkle.rs:149-206): two queues, leaves first, then the internal nodes level by level
extern "C" int ms_merkle_view_ids(size_t nleaves, const uint64_t* h_indices, size_t nidx, uint64_t* h_leaf_ids, unsigned char* h_leaf_is_sibling,
                                  size_t* n_leaf_ids, uint64_t* h_node_ids, size_t* n_node_ids) {
    if ((nidx && !h_indices) || !h_leaf_ids || !h_leaf_is_sibling || !n_leaf_ids || !h_node_ids || !n_node_ids) return fail(MS_ERR_INVALID, "ms_merkle_view_ids: null argument");
    if (nleaves < 2 || (nleaves & (nleaves - 1))) return fail(MS_ERR_INVALID, "number of leaves must be a power of two >= 2");
    std::vector<uint64_t> idx(h_indices, h_indices + nidx);
    for (uint64_t i : idx) if (i >= nleaves) return fail(MS_ERR_INVALID, "leaf index %llu out of bounds (%zu)", (unsigned long long)i, nleaves);   // Error::LeafIndexOutOfBounds
    std::sort(idx.begin(), idx.end());
    idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
    std::vector<uint64_t> queue;
    size_t nl = 0, nn = 0;
    for (size_t k = 0; k < idx.size(); k++) {
        const uint64_t index = idx[k];
        h_leaf_ids[nl] = index; h_leaf_is_sibling[nl++] = 0;
        queue.push_back((nleaves + index) >> 1);
        if (k + 1 < idx.size() && (index ^ 1) == idx[k + 1]) { h_leaf_ids[nl] = idx[++k]; h_leaf_is_sibling[nl++] = 0; continue; }
        h_leaf_ids[nl] = index ^ 1; h_leaf_is_sibling[nl++] = 1;
    }
    for (size_t head = 0; head < queue.size(); head++) {
        const uint64_t index = queue[head];
        if (index > 2) queue.push_back(index >> 1);
        if (head + 1 < queue.size() && (index ^ 1) == queue[head + 1]) { head++; continue; }
        h_node_ids[nn++] = index ^ 1;
    }
    *n_leaf_ids = nl; *n_node_ids = nn;
    return MS_OK;
}

extern "C" int ms_gather_digests_multi(ms_ctx* ctx, unsigned nseg, const void* const* d_digests, const size_t* ndigests, const uint64_t* h_indices,
                                       const size_t* counts, void* const* d_out) {
    if (!ctx || (nseg && (!d_digests || !ndigests || !counts || !d_out))) return fail(MS_ERR_INVALID, "ms_gather_digests_multi: null argument");
    size_t total = 0;
    for (unsigned s = 0; s < nseg; s++) {
        if (counts[s] && (!d_digests[s] || !d_out[s])) return fail(MS_ERR_INVALID, "ms_gather_digests_multi: null pointer in segment %u", s);
        total += counts[s];
    }
    if (total && !h_indices) return fail(MS_ERR_INVALID, "ms_gather_digests_multi: null indices");
    if (total == 0) return MS_OK;
    std::vector<uint64_t> pairs(2 * total);
    size_t r = 0;
    for (unsigned s = 0; s < nseg; s++)
        for (size_t k = 0; k < counts[s]; k++, r++) {
            if (h_indices[r] >= ndigests[s]) return fail(MS_ERR_INVALID, "digest %llu out of range (segment %u has %zu)", (unsigned long long)h_indices[r], s, ndigests[s]);
            pairs[2 * r] = (uint64_t)(uintptr_t)d_digests[s] + 32 * h_indices[r];
            pairs[2 * r + 1] = (uint64_t)(uintptr_t)d_out[s] + 32 * k;
        }
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    // the list goes to device memory by one stream-ordered copy: read in place from the pinned ring (as the short index lists of the single
    // gathers are) its ~3 000 records cost a PCIe round trip each -- 45 us for the openings of a proof
    void* d_pairs = nullptr;
    LockedPoolGuard pooled(ctx);
    MSCHK(pooled.alloc(pairs.size() * 8, &d_pairs));
    MSCHK(stage_upload(ctx, d_pairs, pairs.data(), pairs.size() * 8));
    { ProfScope ps(ctx, "gather_digests", 64.0 * total);
      hipLaunchKernelGGL(msscan::copy_records32, dim3(stream_grid(total * 4)), dim3(msscan::NT), 0, ctx->stream, (const uint64_t*)d_pairs, total); }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_gather_digests(ms_ctx* ctx, size_t ndigests, const void* d_digests, const uint64_t* h_indices, size_t count, void* d_out) {
    if (!ctx || !d_digests || !d_out || (count && !h_indices)) return fail(MS_ERR_INVALID, "ms_gather_digests: null argument");
    for (size_t k = 0; k < count; k++) if (h_indices[k] >= ndigests) return fail(MS_ERR_INVALID, "digest %llu out of range", (unsigned long long)h_indices[k]);
    if (count == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const void* d_idx = nullptr;
    LockedPoolGuard pooled(ctx);
    MSCHK(stage_view(ctx, h_indices, count * 8, &d_idx, pooled));
    { ProfScope ps(ctx, "gather_digests", 64.0 * count);
      hipLaunchKernelGGL(msscan::gather_records, dim3(stream_grid(count * 4)), dim3(msscan::NT), 0, ctx->stream,
                         (const uint64_t*)d_digests, (const uint64_t*)d_idx, (uint64_t*)d_out, count, 4u); }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
