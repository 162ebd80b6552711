// SHA-256 and RPO-256 commitments, proof-of-work grinding (src/merkle.rs:412-508, src/hash.rs:58-100, gpu/src/plan.rs:32-174,
// src/channel.rs:76-93).
#include "ms_internal.h"
#include "sha256_kernels.h"
#include "rpo_kernels.h"

// ---------------------------------------------------------------------------------------
// SHA-256 commitments
// ---------------------------------------------------------------------------------------
extern "C" int ms_sha256_rows(ms_ctx* ctx, int field, size_t nrows, const void* const* d_cols, unsigned ncols, void* d_leaves) {
    if (!ctx || (!d_cols && ncols) || !d_leaves) return fail(MS_ERR_INVALID, "ms_sha256_rows: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (ncols > (unsigned)mssha::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d columns per commitment", mssha::MAXCOLS);
    if (nrows == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    mssha::RowsParams P;
    memset(&P, 0, sizeof P);
    for (unsigned c = 0; c < ncols; c++) P.cols[c] = (const uint64_t*)d_cols[c];
    P.leaves = (uint8_t*)d_leaves; P.nrows = nrows; P.ncols = ncols; P.V = V; P.row_stride = V;
    if (ncols && (ncols * V) % 8 == 0) { P.fold_last = 1; mssha::sha256_fold_pad_block((uint64_t)ncols * V * 64, P.kw_last); }
    {
        ProfScope ps(ctx, "sha256_rows", (double)nrows * ncols * V * 8 + 32.0 * nrows);
        hipLaunchKernelGGL(mssha::sha256_rows, dim3((unsigned)((nrows + mssha::NT - 1) / mssha::NT)), dim3(mssha::NT), 0, ctx->stream, P);
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_sha256_rows_row_major(ms_ctx* ctx, int field, size_t nrows, unsigned ncols, const void* d_matrix, void* d_leaves) {
    if (!ctx || !d_matrix || !d_leaves) return fail(MS_ERR_INVALID, "ms_sha256_rows_row_major: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (ncols == 0 || ncols > (unsigned)mssha::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "1..%d columns per row", mssha::MAXCOLS);
    if (nrows == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    mssha::RowsParams P;
    memset(&P, 0, sizeof P);
    for (unsigned c = 0; c < ncols; c++) P.cols[c] = (const uint64_t*)d_matrix + (size_t)c * V;
    P.leaves = (uint8_t*)d_leaves; P.nrows = nrows; P.ncols = ncols; P.V = V; P.row_stride = ncols * V;
    if ((ncols * V) % 8 == 0) { P.fold_last = 1; mssha::sha256_fold_pad_block((uint64_t)ncols * V * 64, P.kw_last); }
    {
        ProfScope ps(ctx, "sha256_rows", (double)nrows * ncols * V * 8 + 32.0 * nrows);
        hipLaunchKernelGGL(mssha::sha256_rows, dim3((unsigned)((nrows + mssha::NT - 1) / mssha::NT)), dim3(mssha::NT), 0, ctx->stream, P);
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_sha256_merkle(ms_ctx* ctx, size_t nleaves, const void* d_leaves, void* d_nodes) {
    if (!ctx || !d_leaves || !d_nodes) return fail(MS_ERR_INVALID, "ms_sha256_merkle: null argument");
    if (nleaves < 2 || (nleaves & (nleaves - 1))) return fail(MS_ERR_INVALID, "number of leaves must be a power of two >= 2");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    uint8_t* nodes = (uint8_t*)d_nodes;                       // (nodes[0] is cleared by the launch that writes the root: sha256_merkle_top)
    const uint8_t* src = (const uint8_t*)d_leaves;
    for (size_t count = nleaves / 2; count >= 1;) {
        uint8_t* dst = nodes + count * 32;
        if (count <= (size_t)mssha::NT) {                        // the remaining levels in one launch
            ProfScope ps(ctx, "sha256_merkle_top", 96.0 * (2 * count - 1));
            hipLaunchKernelGGL(mssha::sha256_merkle_top<1>, dim3(1), dim3(mssha::NT), 0, ctx->stream, src, nodes, (unsigned)count);
            break;
        }
        if (count <= ((size_t)1 << 17)) {                        // log2(NT) + 1 levels at once: count / NT subtrees, one workgroup each
            // more subtrees than CUs: two parents per lane, so that every wave keeps a SIMD to itself (sha256_kernels.h)
            // Measured per tree (scripts/merkle_top_probe.py, same box): 2^18 leaves 123 -> 105 us, 2^21 120 -> 108; 2^23 / 2^24 leaves 116 -> 119
            // (after the long level launches of a big tree the 512-workgroup form is the faster one), hence the bound on the tree's size.
            const unsigned per = nleaves <= ((size_t)1 << 21) && count / mssha::NT > 256 && count % (2 * mssha::NT) == 0 ? 2u : 1u;
            ProfScope ps(ctx, "sha256_merkle_top", 96.0 * (2 * count - count / (per * mssha::NT)));
            if (per == 2) hipLaunchKernelGGL(mssha::sha256_merkle_top<2>, dim3((unsigned)(count / (2 * mssha::NT))), dim3(mssha::NT), 0, ctx->stream, src, nodes, (unsigned)count);
            else hipLaunchKernelGGL(mssha::sha256_merkle_top<1>, dim3((unsigned)(count / mssha::NT)), dim3(mssha::NT), 0, ctx->stream, src, nodes, (unsigned)count);
            const size_t last = count / (per * mssha::NT);       // the level the subtrees end in
            src = nodes + last * 32;
            count = last / 2;
            continue;
        }
        ProfScope ps(ctx, "sha256_merkle_level", 96.0 * count);
        hipLaunchKernelGGL(mssha::sha256_merge_level, dim3((unsigned)((count + mssha::NT - 1) / mssha::NT)), dim3(mssha::NT), 0, ctx->stream, src, dst, count);
        src = dst;
        count >>= 1;
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}

// ---------------------------------------------------------------------------------------
// RPO-256 commitments
// ---------------------------------------------------------------------------------------
static int rpo_rows(ms_ctx* ctx, size_t nrows, const uint64_t* const* cols, unsigned ncols, unsigned stride, void* d_digests) {
    if (ncols == 0) return fail(MS_ERR_INVALID, "the zero-length input is not allowed");          // plan.rs:72
    if (ncols > (unsigned)msrpo::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d columns per commitment", msrpo::MAXCOLS);
    if (nrows == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    msrpo::RowsParams P;
    memset(&P, 0, sizeof P);
    for (unsigned c = 0; c < ncols; c++) P.cols[c] = cols[c];
    P.digests = (uint64_t*)d_digests; P.nrows = nrows; P.ncols = ncols; P.row_stride = stride;
    ProfScope ps(ctx, "rpo256_rows", 8.0 * nrows * (ncols + 4));
    hipLaunchKernelGGL(msrpo::rpo256_rows, dim3((unsigned)((nrows + msrpo::NT - 1) / msrpo::NT)), dim3(msrpo::NT), 0, ctx->stream, P);
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_rpo256_rows(ms_ctx* ctx, size_t nrows, const void* const* d_cols, unsigned ncols, void* d_digests) {
    if (!ctx || !d_cols || !d_digests) return fail(MS_ERR_INVALID, "ms_rpo256_rows: null argument");
    std::vector<const uint64_t*> cols(ncols);
    for (unsigned c = 0; c < ncols; c++) cols[c] = (const uint64_t*)d_cols[c];
    return rpo_rows(ctx, nrows, cols.data(), ncols, 1, d_digests);
}
// rows of a column-major matrix of `field`: an Fq3 column contributes its components c0, c1, c2 in the order
// the SHA-256 leaves serialise them (src/hash.rs:93-98) -- the column pointers are simply taken at word stride 3
extern "C" int ms_rpo256_rows_field(ms_ctx* ctx, int field, size_t nrows, const void* const* d_cols, unsigned ncols, void* d_digests) {
    if (!ctx || !d_cols || !d_digests) return fail(MS_ERR_INVALID, "ms_rpo256_rows_field: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (V != 1 && V != 3) return fail(MS_ERR_UNSUPPORTED, "RPO-256 absorbs Goldilocks elements (Fp or Fq3 columns)");
    std::vector<const uint64_t*> cols;
    for (unsigned c = 0; c < ncols; c++) {
        if (!d_cols[c]) return fail(MS_ERR_INVALID, "null column %u", c);
        for (unsigned k = 0; k < V; k++) cols.push_back((const uint64_t*)d_cols[c] + k);
    }
    return rpo_rows(ctx, nrows, cols.data(), (unsigned)cols.size(), V, d_digests);
}
extern "C" int ms_rpo256_rows_row_major(ms_ctx* ctx, size_t nrows, unsigned ncols, const void* d_matrix, void* d_digests) {
    if (!ctx || !d_matrix || !d_digests) return fail(MS_ERR_INVALID, "ms_rpo256_rows_row_major: null argument");
    std::vector<const uint64_t*> cols(ncols);
    for (unsigned c = 0; c < ncols; c++) cols[c] = (const uint64_t*)d_matrix + c;
    return rpo_rows(ctx, nrows, cols.data(), ncols, ncols, d_digests);
}
extern "C" int ms_rpo256_merkle(ms_ctx* ctx, size_t nleaves, const void* d_leaves, void* d_nodes) {
    if (!ctx || !d_leaves || !d_nodes) return fail(MS_ERR_INVALID, "ms_rpo256_merkle: null argument");
    if (nleaves < 2 || (nleaves & (nleaves - 1))) return fail(MS_ERR_INVALID, "number of leaves must be a power of two >= 2");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    uint64_t* nodes = (uint64_t*)d_nodes;
    HIPCHK(hipMemsetAsync(nodes, 0, 32, ctx->stream));
    const uint64_t* src = (const uint64_t*)d_leaves;
    for (size_t count = nleaves / 2; count >= 1; count >>= 1) {
        uint64_t* dst = nodes + count * 4;
        ProfScope ps(ctx, "rpo256_merkle_level", 96.0 * count);
        if (count <= ((size_t)1 << 15))      // few nodes: sixteen lanes per node (latency of one element's chains, not of twelve)
            hipLaunchKernelGGL(msrpo::rpo256_merge_level_wide, dim3((unsigned)((count + 3) / 4)), dim3(msrpo::NTW), 0, ctx->stream, src, dst, count);
        else
            hipLaunchKernelGGL(msrpo::rpo256_merge_level, dim3((unsigned)((count + msrpo::NT - 1) / msrpo::NT)), dim3(msrpo::NT), 0, ctx->stream, src, dst, count);
        src = dst;
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}

// ---------------------------------------------------------------------------------------
// proof-of-work grinding
// ---------------------------------------------------------------------------------------
extern "C" int ms_sha256_pow_grind(ms_ctx* ctx, const void* h_seed32, unsigned bits, uint64_t max_nonce, uint64_t* nonce) {
    if (!ctx || !h_seed32 || !nonce) return fail(MS_ERR_INVALID, "ms_sha256_pow_grind: null argument");
    if (bits > 64) return fail(MS_ERR_INVALID, "proof-of-work bits must be <= 64");
    void* d_found = nullptr;
    PoolGuard pooled(ctx);                                 // temporaries go back to the pool on every exit path
    MSCHK(pooled.alloc(8, &d_found));
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    mssha::PowParams P;
    const uint8_t* sb = (const uint8_t*)h_seed32;
    for (int q = 0; q < 8; q++) P.seed[q] = ((uint32_t)sb[4 * q] << 24) | ((uint32_t)sb[4 * q + 1] << 16) | ((uint32_t)sb[4 * q + 2] << 8) | sb[4 * q + 3];
    P.bits = bits; P.found = (unsigned long long*)d_found;
    unsigned long long window = 1ull << 12;             // grows to 2^24 nonces per launch
    unsigned long long none = ~0ull, found = ~0ull;
    int rc = MS_OK;
    for (unsigned long long base = 1; base <= max_nonce && rc == MS_OK; base += P.count, window = std::min(window * 4, 1ull << 24)) {
        P.base = base; P.count = std::min<unsigned long long>(window, max_nonce - base + 1);
        if (hipMemcpyAsync(d_found, &none, 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = fail(MS_ERR_HIP, "pow: memcpy"); break; }
        {
            ProfScope ps(ctx, "sha256_pow_grind", 0.0);
            hipLaunchKernelGGL(mssha::sha256_pow_grind, dim3((unsigned)((P.count + mssha::NT - 1) / mssha::NT)), dim3(mssha::NT), 0, ctx->stream, P);
        }
        if (hipMemcpyAsync(&found, d_found, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = fail(MS_ERR_HIP, "pow: readback"); break; }
        if (found != none) break;
    }
    if (rc != MS_OK) return rc;
    if (found == none) return fail(MS_ERR_INVALID, "no nonce below %llu has %u leading zero bits", (unsigned long long)max_nonce, bits);
    *nonce = found;
    return MS_OK;
}
