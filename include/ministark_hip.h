/* ministark_hip.h -- C ABI of the MI355X (gfx950) backend for miniSTARK's gpu-poly path.
 *
 * The reference has no FFI seam: its GPU boundary is the Rust API of crate
 * `ministark-gpu` (Apple-Metal only).  Every entry point below names the Rust
 * item it stands in for (path:line under the reference tree); INTEGRATION.md
 * shows the `cfg(feature = "hip")` Rust shim that binds them.
 *
 * Conventions
 *   - All field data is the reference's in-memory format, untouched at the
 *     boundary: Montgomery residues, little-endian u64 limbs
 *     (Fp = 1 limb, R = 2^64; Fq3 = 3 consecutive Fp; Fp252 = 4 limbs, R = 2^256).
 *     Every element handed in -- columns, constants, challenges, offsets -- must be
 *     CANONICAL (< p), as every value of the reference's field types is (ark-ff keeps
 *     residues reduced): the kernels' unreduced accumulators are sized for canonical
 *     operands, a word in [p, 2^64) or a 252-bit value in [p, 2^256) gives a wrong
 *     (not a rejected) result.  Every element handed back is canonical.
 *   - `void* d_*` are DEVICE pointers (hipMalloc / ms_alloc / a torch tensor's
 *     data_ptr); `const void* h_*` are small HOST constants (one field element).
 *   - Every function returns 0 on success, a negative MS_ERR_* otherwise, and
 *     never throws; ms_last_error() describes the last failure of the calling
 *     thread.  The reference panics (`assert!`/`unwrap()`, e.g.
 *     gpu/src/plan.rs:248,255,360); the Rust shim turns non-zero into panic!.
 *   - Work is enqueued on the context's HIP stream; functions documented
 *     "blocks" synchronise it (the reference's execute() = commit +
 *     wait_until_completed, gpu/src/plan.rs:229-232).
 */
#ifndef MINISTARK_HIP_H
#define MINISTARK_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* GpuField::field_name() (gpu/src/lib.rs:20-26, gpu/src/fields.rs:55-60,199-205,250-256) */
typedef enum {
    MS_GOLDILOCKS_FP = 0,  /* "p18446744069414584321_fp"  */
    MS_GOLDILOCKS_FQ3 = 1, /* "p18446744069414584321_fq3" */
    MS_STARK252_FP = 2     /* "p3618502788666131213697322783095070105623107215331596699973092056135872020481_fp" */
} ms_field;

typedef enum { MS_ADD = 0, MS_MUL = 1 } ms_binop;          /* {Add,Mul}{Assign,Into}[Const] */
typedef enum { MS_NEG = 0, MS_INV = 1, MS_EXP = 2 } ms_unop; /* {Neg,Inverse,Exp}{InPlace,Into} */

enum {
    MS_OK = 0,
    MS_ERR_INVALID = -1,     /* bad argument (size not a power of two, null pointer, ...) */
    MS_ERR_UNSUPPORTED = -2, /* field / size / option outside what the backend implements */
    MS_ERR_HIP = -3,         /* a HIP runtime call failed */
    MS_ERR_NOMEM = -4,
    MS_ERR_INTERNAL = -5     /* a self-check of the library failed (MS_EVAL_SELFCHECK): a bug, please report */
};

typedef struct ms_ctx ms_ctx;
typedef struct ms_ntt_plan ms_ntt_plan;

/* ---- runtime: Planner / get_planner (gpu/src/plan.rs:327-351, 464-469) ---------------- */
int ms_ctx_create(int device, ms_ctx** out);
int ms_ctx_destroy(ms_ctx* ctx);
int ms_sync(ms_ctx* ctx);                      /* command_buffer.wait_until_completed() */
void* ms_ctx_stream(ms_ctx* ctx);              /* the hipStream_t work is enqueued on   */
const char* ms_last_error(void);
size_t ms_field_bytes(int field);              /* 8 / 24 / 32 */
/* Per-launch timing (the reference's `Timer` / Instant prints, src/utils.rs:32-51,
 * src/prover.rs:30-159): with profiling on, every kernel launch is bracketed by hipEvents
 * on the context's stream.  ms_profile_read blocks and writes lines
 * "kernel_name calls total_microseconds algorithmic_bytes_per_call". */
int ms_profile_enable(ms_ctx* ctx, int on);
int ms_profile_read(ms_ctx* ctx, char* buf, size_t cap);

/* ---- memory: GpuAllocator + buffer_no_copy (src/utils.rs:438-470, gpu/src/utils.rs:103-134).
 * The reference aliases page-aligned host Vecs (unified memory); on a discrete GPU
 * columns live in HBM and are mirrored explicitly. */
int ms_alloc(ms_ctx* ctx, size_t bytes, void** d_ptr);
int ms_free(ms_ctx* ctx, void* d_ptr);
/* ms_alloc / ms_free recycle device blocks through a per-context pool (a freed block may be handed
 * out again without a hipFree: all work is ordered on the context's stream).
 * ms_copy = GpuVec clone on device (Matrix::clone in interpolate/evaluate, src/matrix.rs:155-163,237-243). */
int ms_copy(ms_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);      /* asynchronous */
int ms_upload(ms_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);   /* blocks */
int ms_download(ms_ctx* ctx, void* h_dst, const void* d_src, size_t bytes); /* blocks */

/* ---- NTT plans: GpuFft / GpuIfft (gpu/src/plan.rs:236-325, 353-462) --------------------
 * ms_ntt_plan_create   = GpuFft::from(domain) / GpuIfft::from(domain)
 *     field       element type of the columns (Fp or Fq3; twiddles are in Fp = F::FftField)
 *     log_n       domain.size() = 2^log_n      (any log_n >= 0; the reference needs >= 2048)
 *     inverse     0: GpuFft (c_i *= offset^i, then DFT)   1: GpuIfft (DFT^-1, then e_i *= offset^-i / n)
 *     h_offset    domain.offset (coset shift h), one Fp element, Montgomery form; NULL = 1
 *     h_group_gen domain.group_gen, Montgomery form, or NULL.  When given it must equal
 *                 arkworks' get_root_of_unity(n) (the only root the kernels' constants are
 *                 built for); otherwise MS_ERR_UNSUPPORTED.
 *     The twiddle tables of a (field, size, direction, offset) are built once per context: the object returned
 *     is a handle (own queue) on the context's cached plan, so constructing a GpuFft per call, as the reference's
 *     callers do (src/matrix.rs:119-131), costs no table build or upload after the first.  Destroy handles before
 *     the context.  (Goldilocks fields; an Fp252 plan owns its tables.)
 * ms_ntt_encode        = fft.encode(&mut column): queue one column (n elements, in place)
 * ms_ntt_execute       = fft.execute(): run every queued column, BLOCKS, clears the queue
 *                        (the plan stays usable; the reference consumes it)
 * ms_ntt_enqueue       = encode + launch without blocking (for pipelines and timing)
 *
 * Threads: calls on one context serialise on its lock.  Destroying a plan, or the context it was created on, must not run
 * concurrently with a call that uses that plan (the liveness check and the use are not one atomic step); a plan that merely
 * OUTLIVES its context is safe -- its entry points return MS_ERR_INVALID and ms_ntt_plan_destroy is a no-op. */
int ms_ntt_plan_create(ms_ctx* ctx, int field, unsigned log_n, int inverse, const void* h_offset,
                       const void* h_group_gen, ms_ntt_plan** out);
int ms_ntt_plan_destroy(ms_ntt_plan* plan);
int ms_ntt_encode(ms_ntt_plan* plan, void* d_column);
int ms_ntt_execute(ms_ntt_plan* plan);
int ms_ntt_enqueue(ms_ntt_plan* plan, void* const* d_columns, unsigned ncols);
/* Out of place, non-blocking: d_dst[c] = transform(d_src[c]); d_src[c] is left untouched (d_dst[c] may be d_src[c]).
 * Matrix::interpolate / evaluate are `self.clone().into_polynomials / into_evaluations` (src/matrix.rs:155-163, 237-243): this is the clone
 * and the transform in one -- the first pass reads the source, the last one writes the destination, no copy of the columns is made
 * (the clones of the trace and of its polynomials were 0.57 GB of device copies per 2^22-row proof). */
int ms_ntt_enqueue_to(ms_ntt_plan* plan, const void* const* d_src, void* const* d_dst, unsigned ncols);

/* ---- bit reversal: BitReverseGpuStage + bit_reverse (gpu/src/stage.rs:280-332,
 * gpu/src/utils.rs:32-78), Matrix::bit_reverse_rows (src/matrix.rs:352-354).
 * Reverses the first 2^log_n elements of each column in place (log_n smaller than the
 * column = prover.rs:185-194 bit_reverse_ce_trace). */
int ms_bit_reverse(ms_ctx* ctx, int field, unsigned log_n, void* const* d_columns, unsigned ncols);

/* ---- fused LDE: Matrix::interpolate + bit_reversed_evaluate (src/prover.rs:50-51,
 * src/matrix.rs:142-163,211-251): per column  iNTT on subgroup(2^log_n)  ->  zero-extend
 * -> NTT on coset(2^(log_n+log_blowup), h_offset) -> optional bit-reversed order.
 * d_in[c] (2^log_n elements) is preserved; d_out[c] has 2^(log_n+log_blowup) elements. */
int ms_lde(ms_ctx* ctx, int field, unsigned log_n, unsigned log_blowup, const void* h_offset,
           const void* const* d_in, void* const* d_out, unsigned ncols, int bit_reversed);

/* ---- evaluation of coefficient columns shorter than the domain: Matrix::into_evaluations /
 * bit_reversed_evaluate with their "resize the column to the domain size" (src/matrix.rs:193-251), e.g. the
 * composition-trace and DEEP polynomials (src/prover.rs:122-124, 156).  d_in[c] holds 2^log_n coefficients
 * (untouched unless d_out[c] aliases it), d_out[c] receives 2^log_domain evaluations on coset(2^log_domain,
 * h_offset), optionally in bit-reversed order.  Blow-ups 4..16 never materialise the zero padding.
 * ms_deinterleave: `composition_poly.chunks(k)` spread over k columns (src/prover.rs:113-121):
 * d_out[c][j] = d_in[j*k + c], j < n_out. */
int ms_evaluate(ms_ctx* ctx, int field, unsigned log_n, unsigned log_domain, const void* h_offset,
                const void* const* d_in, void* const* d_out, unsigned ncols, int bit_reversed);
int ms_deinterleave(ms_ctx* ctx, int field, size_t n_out, unsigned k, const void* d_in, void* const* d_out);

/* ---- element-wise stages (gpu/src/stage.rs:115-1155; kernels evaluation_shaders.h.metal:11-168).
 * lf / rf are the fields of lhs(=dst) and rhs: (Fp,Fp), (Fq3,Fq3) or (Fq3,Fp) -- the GpuMul / GpuAdd
 * impls of gpu/src/fields.rs:55-216.  `shift` rotates the rhs index: rhs[(i + shift) mod n], any sign
 * (the reference normalises with (n + shift) % n, stage.rs:168,227,449,515).  d_dst may alias d_lhs
 * (the *Assign / *InPlace stages) but not d_rhs when shift != 0.  All calls are asynchronous.
 *   ms_binary        MulAssign / MulInto / AddAssign / AddInto            (stage.rs:115-233, 393-521)
 *   ms_binary_const  {Mul,Add}{Into,Assign}Const                          (stage.rs:523-806)
 *   ms_mul_pow       MulPowStage: dst = lhs * rhs[(i+shift)%n]^power      (stage.rs:334-391)
 *   ms_unary         Neg / Inverse / Exp, in place or into                (stage.rs:808-1109);
 *                    inverse of 0 is 0; Fq3 inverse is implemented (a todo!() in the reference)
 *   ms_convert       ConvertIntoStage: Fp -> Fq3 embedding                (stage.rs:581-635)
 *   ms_fill          FillBuffStage                                        (stage.rs:1111-1155)
 *   ms_sum_columns   Matrix::sum_columns (src/matrix.rs:357-394)                              */
int ms_binary(ms_ctx* ctx, int op, int lf, int rf, size_t n, void* d_dst, const void* d_lhs, const void* d_rhs, long shift);
int ms_binary_const(ms_ctx* ctx, int op, int lf, int rf, size_t n, void* d_dst, const void* d_lhs, const void* h_const);
int ms_mul_pow(ms_ctx* ctx, int lf, int rf, size_t n, void* d_dst, const void* d_lhs, const void* d_rhs, unsigned power, long shift);
int ms_unary(ms_ctx* ctx, int op, int field, size_t n, void* d_dst, const void* d_src, unsigned exponent);
int ms_convert(ms_ctx* ctx, int dst_field, int src_field, size_t n, void* d_dst, const void* d_src);
int ms_fill(ms_ctx* ctx, int field, size_t n, void* d_dst, const void* h_value);
int ms_sum_columns(ms_ctx* ctx, int field, size_t n, const void* const* d_cols, unsigned ncols, void* d_dst);

/* ---- commitments: hash_rows + build_merkle_nodes with Sha256HashFn (src/merkle.rs:412-508,
 * src/hash.rs:58-100; CPU-only in the reference).  Digests are 32 raw bytes.
 * ms_sha256_rows   leaves[r] = SHA-256( ||_c canonical little-endian bytes of d_cols[c][r] )
 *                  (Matrix::hash_rows / MatrixMerkleTree::from_matrix's leaf layer)
 * ms_sha256_merkle nodes[] has nleaves slots of 32 bytes: nodes[k] = SHA-256(nodes[2k]||nodes[2k+1]),
 *                  leaf pairs feed nodes[nleaves/2 ..), nodes[1] = root, nodes[0] = zero
 *                  (MerkleTreeImpl::new -> build_merkle_nodes).  nleaves = 2^k >= 2. */
int ms_sha256_rows(ms_ctx* ctx, int field, size_t nrows, const void* const* d_cols, unsigned ncols, void* d_leaves);
int ms_sha256_merkle(ms_ctx* ctx, size_t nleaves, const void* d_leaves, void* d_nodes);
/* Row hashing of a ROW-MAJOR matrix [nrows][ncols] of `field` elements: the FRI layer commitment,
 * Matrix::from_arrays(evaluations.as_chunks::<N>()) + from_matrix (src/fri.rs:213-216) -- row r is the
 * coset of ncols = folding_factor consecutive evaluations; no de-interleaving copy is made. */
int ms_sha256_rows_row_major(ms_ctx* ctx, int field, size_t nrows, unsigned ncols, const void* d_matrix, void* d_leaves);

/* ---- fused constraint evaluation: eval_gpu::eval / eval_cpu::eval
 * (src/eval_gpu.rs:46-131, src/eval_cpu.rs:33-150; called from AirConfig::eval_constraint,
 * src/air.rs:86-128).  The `Expr<AlgebraicItem<..>>` DAG (src/expression.rs:33-40,
 * src/constraints.rs:21-28) is lowered by the host to a typed register program:
 *
 *   constraint program = ninstr x { u32 op, dst, a, b }, executed in order at every point i < 2^log_n.
 *   Two register files per point: P (Fp, 8 B) and Q (Fq3, 24 B); register numbers < 256 / < 128.
 *     0 X_P        P[dst] = x_i = offset * w^i                  (AlgebraicItem::X)
 *     1 CONST_P    P[dst] = consts[a]          2 CONST_Q    Q[dst] = consts[a..a+3)
 *                  (Constant / Challenge / Hint; `a` indexes u64 words of h_consts)
 *     3 TRACE_P    P[dst] = base_col[a][(i + lde_step * (int32)b) mod n]        (Trace(col, offset))
 *     4 TRACE_Q    Q[dst] = ext_col [a][(i + lde_step * (int32)b) mod n]
 *     5 PERIODIC_P P[dst] = periodic[a][i mod periodic_len[a]]     6 PERIODIC_Q likewise
 *     7 NEG_P  8 NEG_Q      dst = -a
 *     9 ADD_PP 10 ADD_QQ 11 ADD_QP (Q[a] + P[b] -> Q)     12 MUL_PP 13 MUL_QQ 14 MUL_QP
 *     15 INV_P 16 INV_Q     dst = a^-1, 0^-1 = 0 (Div(x, y) = MUL(x, INV(y)), eval_cpu.rs:440-442)
 *     17 POW_P 18 POW_Q     dst = a^b  (b: u32 exponent)
 *     19 EMBED              Q[dst] = (P[a], 0, 0)
 *     20 STORE_Q            out[i] = Q[a]  (out_field = MS_GOLDILOCKS_FQ3)
 *     21 STORE_P            out[i] = P[a]  (out_field = MS_GOLDILOCKS_FP, i.e. Fq = Fp AIRs)
 *   The program is validated on the host (MS_ERR_INVALID on any out-of-range operand; `b` of a STORE
 *   must be 0).  Before launch the library rewrites it (csrc/eval_opt.h): sub-expressions of short
 *   period in i (zerofier inverses, x^N) are evaluated once into tables, long x^e chains become
 *   twiddle-table lookups, divisions by values of x alone are batch-inverted into tables (ONE table for all the divisors X - a whose
 *   roots differ by a power of the trace generator: they are rotations of each other, csrc/eval_shift.h), and the result is regrouped into
 *   sums of products that are accumulated unreduced with one reduction per sum (csrc/eval_regroup.h; MS_EVAL_REGROUP=0
 *   switches that step off); on domains of >= 2^16 points the result is compiled (hiprtc, cached per
 *   context) into a specialised straight-line kernel (csrc/eval_jit.h; MS_EVAL_JIT=0 keeps the
 *   interpreter).  Results are bit-identical either way.
 * d_x_lde may be NULL (x generated on the fly from h_domain_offset; a caller-supplied x array
 * disables the rewrites that rely on x_i = offset * w^i); d_out has 2^log_n elements of out_field.
 * At most 16 periodic columns.  Asynchronous. */
/* The opcodes as values: the ONE table the three lowerings of the expression DAG are generated from (scripts/gen_opcodes.py writes the
 * marked blocks of ministark_amd/expr.py, ministark_amd/csrc/host/expr.hpp and rust/src/eval_hip.rs; tests/test_opcode_tables.py
 * re-runs it in check mode and compares the kernels' own enum, csrc/eval_kernels.h, with it). */
enum ms_eval_op {
    MS_OP_X_P = 0, MS_OP_CONST_P = 1, MS_OP_CONST_Q = 2, MS_OP_TRACE_P = 3, MS_OP_TRACE_Q = 4, MS_OP_PERIODIC_P = 5, MS_OP_PERIODIC_Q = 6,
    MS_OP_NEG_P = 7, MS_OP_NEG_Q = 8, MS_OP_ADD_PP = 9, MS_OP_ADD_QQ = 10, MS_OP_ADD_QP = 11, MS_OP_MUL_PP = 12, MS_OP_MUL_QQ = 13,
    MS_OP_MUL_QP = 14, MS_OP_INV_P = 15, MS_OP_INV_Q = 16, MS_OP_POW_P = 17, MS_OP_POW_Q = 18, MS_OP_EMBED = 19, MS_OP_STORE_Q = 20,
    MS_OP_STORE_P = 21, MS_OP_PUBLIC_COUNT = 22
};
int ms_eval_program(ms_ctx* ctx, const uint32_t* h_prog, unsigned ninstr, const void* h_consts, unsigned nconst_words,
                    unsigned log_n, unsigned lde_step, const void* h_domain_offset, const void* d_x_lde,
                    const void* const* d_base_cols, unsigned nbase, const void* const* d_ext_cols, unsigned next,
                    const void* const* d_periodic, const unsigned* periodic_len, unsigned nperiodic,
                    int out_field, void* d_out);

/* The same with flags.  MS_EVAL_BIT_REVERSED: the first 2^log_n entries of every trace column, and d_out, are in
 * bit-reversed order over the evaluation domain (position R holds point bitrev(R)) -- the layout the LDE is committed in.
 * The reference re-orders every column into natural order before evaluating and back afterwards
 * (bit_reverse_ce_trace, src/prover.rs:88-91, 126-129, 185-194); here the evaluator reads the committed layout directly:
 * consecutive positions differ in the high bits of the point index, so rotated rows are again consecutive positions
 * and the loads stay coalesced.  d_out = the natural-order result, bit-reversed. */
#define MS_EVAL_BIT_REVERSED 1u
/* MS_EVAL_PLAIN (diagnostics): evaluate the program exactly as given on the interpreter -- none of the rewriting passes (tables of
 * short-period / x-only sub-expressions, shared inverse tables, sums of products), no specialised kernel.  Same output words, several
 * times slower.  With MS_EVAL_SELFCHECK=1 in the environment every evaluation is followed by this one and compared word by word
 * (MS_ERR_INTERNAL on a difference). */
#define MS_EVAL_PLAIN 2u
int ms_eval_program_ex(ms_ctx* ctx, const uint32_t* h_prog, unsigned ninstr, const void* h_consts, unsigned nconst_words,
                       unsigned log_n, unsigned lde_step, const void* h_domain_offset, const void* d_x_lde,
                       const void* const* d_base_cols, unsigned nbase, const void* const* d_ext_cols, unsigned next,
                       const void* const* d_periodic, const unsigned* periodic_len, unsigned nperiodic,
                       int out_field, void* d_out, unsigned flags);

/* Diagnostics: compile the specialised kernel for a (validated-shape) program without launching it;
 * needs no device.  *code_bytes receives the size of the gfx950 code object; on failure returns
 * MS_ERR_UNSUPPORTED with the compiler log in ms_last_error(). */
int ms_eval_jit_check(const uint32_t* h_prog, unsigned ninstr, int out_field, size_t* code_bytes);

/* Where the specialised kernels came from and what they cost.  The reference ships its shaders
 * precompiled (gpu/src/plan.rs:30 include_bytes!("metal/shaders.metallib")) and pays nothing at run
 * time; here a program's first evaluation in a PROCESS either loads its gfx950 code object from the
 * on-disk cache ($MS_JIT_CACHE, default ~/.cache/ministark_hip; MS_JIT_CACHE=0 switches it off; entries
 * are keyed by SHA-256 of source + options + compiler version + the library's device headers, carry a
 * digest, and a damaged entry is dropped and recompiled) or compiles it with hiprtc and stores it.
 * compile_failures > 0 means some program runs on the interpreter (same words, several times slower;
 * a warning is printed once per process).  ctx NULL: the totals of the process (ms_eval_jit_check
 * counts there). */
typedef struct ms_jit_stats {
    uint64_t kernels_compiled;   /* hiprtc runs that produced a kernel                         */
    uint64_t kernels_from_disk;  /* kernels loaded from the cache without compiling            */
    uint64_t compile_failures;   /* programs left to the interpreter                           */
    uint64_t damaged_entries;    /* cache files rejected (digest / size / loader) and replaced */
    double compile_ms;           /* wall time inside hiprtc                                    */
    double load_ms;              /* reading cache entries + hipModuleLoadData                  */
} ms_jit_stats;
int ms_eval_jit_stats(ms_ctx* ctx, ms_jit_stats* out);

/* ---- extension columns and queries.
 * ms_scan_affine     the sequential host loops that build running-product / running-evaluation
 *                    extension columns (examples/brainfuck/trace.rs:108-289):
 *                        state = init;  for i in 0..n:  out[i] = state;  state = a[i]*state + b[i]
 *                    (inclusive != 0: out[i] = the state AFTER row i).  d_a NULL = all ones (running sum),
 *                    d_b NULL = all zeros (running product); masked rows are a = 1, b = 0.  a, b, init, out
 *                    are elements of `field` (any of the three); out may alias a or b.  Any n >= 0.
 * ms_gather_rows     out[p][c] = cols[c][positions[p]], row-major: Matrix::get_row over the query
 *                    positions (src/trace.rs:139-152, src/matrix.rs get_row)
 * ms_gather_digests  out[k] = digests[indices[k]] (32-byte records): the leaves / sibling leaves / nodes
 *                    a batched Merkle opening lists (MerkleTreeImpl::prove, src/merkle.rs:149-206)
 * ms_merkle_view_ids the index walk of MerkleTreeImpl::prove itself (src/merkle.rs:149-206; host-only, no device work): for a tree of
 *                    nleaves leaves and the queried leaf indices -> h_leaf_ids (the leaves to fetch: each queried leaf followed by its
 *                    pair partner or its sibling; h_leaf_is_sibling[k] = 1 where entry k is a sibling that was NOT queried; at most
 *                    2 nidx entries) and h_node_ids (the internal nodes of the batched opening in the reference's order; at most
 *                    nidx * log2(nleaves) entries); the two counts come back through n_leaf_ids / n_node_ids.  Bindings that
 *                    would walk the two queues in an interpreter call this instead.
 * Positions / indices are host arrays of u64 (they come from the channel); outputs are device buffers. */
int ms_merkle_view_ids(size_t nleaves, const uint64_t* h_indices, size_t nidx, uint64_t* h_leaf_ids, unsigned char* h_leaf_is_sibling,
                       size_t* n_leaf_ids, uint64_t* h_node_ids, size_t* n_node_ids);
int ms_scan_affine(ms_ctx* ctx, int field, size_t n, const void* d_a, const void* d_b, const void* h_init, int inclusive, void* d_out);
int ms_gather_rows(ms_ctx* ctx, int field, size_t nrows, const void* const* d_cols, unsigned ncols,
                   const uint64_t* h_positions, size_t npos, void* d_out);
int ms_gather_digests(ms_ctx* ctx, size_t ndigests, const void* d_digests, const uint64_t* h_indices, size_t count, void* d_out);
/* The same for nseg digest arrays in ONE launch: segment s gathers counts[s] records of d_digests[s] (ndigests[s] records long) into
 * d_out[s]; h_indices holds the segments' index lists one after another.  The openings of a proof list leaves, siblings and nodes of
 * every committed tree (two trace trees + one per FRI layer: src/prover.rs:161-173, src/fri.rs:148-165): two dozen gathers of a few
 * records each, i.e. two dozen launch latencies -- the bindings collect them and call this once. */
int ms_gather_digests_multi(ms_ctx* ctx, unsigned nseg, const void* const* d_digests, const size_t* ndigests, const uint64_t* h_indices,
                            const size_t* counts, void* const* d_out);

/* ---- FRI fold: apply_drp (src/fri.rs:526-567) as called by FriProver::build_layer
 * (src/fri.rs:199-231).  d_evals holds 2^log_n elements in bit-reversed order (the layer that
 * was just committed); d_out receives 2^log_n / folding_factor elements, bit-reversed, the next
 * layer's evaluations.  folding_factor in {2,4,8,16} (src/fri.rs:186-192); h_alpha = the drawn
 * challenge (one element of `field`); h_offset = domain_offset (Fp, NULL = 1 as build_layer
 * passes).  Bit-identical to bit_reverse + ifft + fold + fft + bit_reverse.  Asynchronous.
 * NOT in place: [d_out, d_out + 2^log_n / folding_factor) must not overlap [d_evals, d_evals + 2^log_n)
 * (MS_ERR_INVALID otherwise). */
int ms_fri_fold(ms_ctx* ctx, int field, unsigned log_n, unsigned folding_factor, const void* h_alpha,
                const void* h_offset, const void* d_evals, void* d_out);
/* ms_fri_fold_rows: the same fold on a ROW SHARD of the layer -- chunks [first_chunk, first_chunk + nchunks) of the 2^log_n /
 * folding_factor chunks; d_evals holds those nchunks * folding_factor evaluations, d_out receives nchunks.  (A chunk folds from its own
 * values and its position alone: the multi-GPU FRI prover keeps every layer sharded by rows and folds without communication.) */
int ms_fri_fold_rows(ms_ctx* ctx, int field, unsigned log_n, unsigned folding_factor, const void* h_alpha,
                     const void* h_offset, size_t first_chunk, size_t nchunks, const void* d_evals, void* d_out);

/* ---- DEEP composition (host-side and sequential in the reference):
 * DeepPolyComposer::get_ood_evals / into_deep_poly (src/composer.rs:43-188).  `point_field` is Fq
 * (MS_GOLDILOCKS_FQ3, or MS_GOLDILOCKS_FP / MS_STARK252_FP for Fq = Fp AIRs; with the 252-bit field all
 * polynomials are passed as base columns and h_offset defaults to its generator 3); every host array of
 * points / alphas / values holds elements of point_field (3, 1 or 4 Montgomery words each), packed.
 * ms_horner_eval   out[q] = P_{qcol[q]}(qpoint[q]) for coefficient-form columns of `coeff_field`
 *                  (horner_evaluate, src/utils.rs:124-133).  Blocks; results on the host.
 * ms_deep_compose  coefficients (2^log_n elements of point_field, d_out) of
 *                      (alpha + beta X) * sum_t alpha_t (P_ct(X) - ood_t) / (X - z_pt)
 *                  where column ct < nbase is d_base_polys[ct] (Fp) and otherwise d_ext_polys[ct - nbase]
 *                  (point_field; the caller appends the composition-trace polynomials there), ood_t must be
 *                  P_ct(z_pt) (from ms_horner_eval).  Equals divide_out_point(s)_into + sum_columns + the
 *                  degree adjustment of src/composer.rs:89-188, computed through n coset evaluations.
 *                  h_offset: coset used internally (Fp, NULL = the generator 7).  Asynchronous.
 * ms_deep_rows     the same polynomial EVALUATED at rows [first, first + count) of the bit-reversed LDE domain (2^log_domain
 *                  points, offset h_offset, NULL = 7) from those rows of the committed LDE columns (d_base_rows: Fp,
 *                  d_ext_rows: point_field; `count` elements each): deep_composition_poly.into_bit_reversed_evaluations
 *                  (src/prover.rs:149-152) without forming the coefficients -- the quotient is a polynomial, so its values
 *                  at the LDE points are these, bit for bit.  With first = 0, count = 2^log_domain one GPU's whole first FRI
 *                  layer; with a row shard (rank r of G: first = r 2^log_domain / G) the multi-GPU form, no communication.
 *                  Goldilocks fields.  Asynchronous. */
int ms_horner_eval(ms_ctx* ctx, int coeff_field, int point_field, size_t n, const void* const* d_cols, unsigned ncols,
                   const unsigned* h_qcol, const void* h_qpoints, unsigned nq, void* h_out);
int ms_deep_rows(ms_ctx* ctx, int point_field, unsigned log_domain, const void* h_offset, size_t first, size_t count,
                 const void* const* d_base_rows, unsigned nbase, const void* const* d_ext_rows, unsigned next,
                 const void* h_points, unsigned npoints, const unsigned* h_term_col, const unsigned* h_term_point,
                 const void* h_term_alpha, const void* h_term_ood, unsigned nterms,
                 const void* h_degree_alpha, const void* h_degree_beta, void* d_out);
int ms_deep_compose(ms_ctx* ctx, int point_field, unsigned log_n, const void* h_offset,
                    const void* const* d_base_polys, unsigned nbase, const void* const* d_ext_polys, unsigned next,
                    const void* h_points, unsigned npoints, const unsigned* h_term_col, const unsigned* h_term_point,
                    const void* h_term_alpha, const void* h_term_ood, unsigned nterms,
                    const void* h_degree_alpha, const void* h_degree_beta, void* d_out);

/* ---- proof-of-work grinding: PublicCoin::grind_proof_of_work with Sha256HashFn
 * (src/random.rs:48-58,129-132; src/hash.rs:84-89): *nonce = the smallest n >= 1 such that
 * SHA-256(seed32 || n as 8 big-endian bytes) has at least `bits` leading zero bits (what the reference's
 * sequential search returns; its rayon path returns any valid nonce).  Blocks.  MS_ERR_INVALID if
 * bits > 64 or no nonce exists below max_nonce. */
int ms_sha256_pow_grind(ms_ctx* ctx, const void* h_seed32, unsigned bits, uint64_t max_nonce, uint64_t* nonce);

/* ---- RPO-256 commitments over Goldilocks Fp: GpuRpo256ColumnMajor / GpuRpo256RowMajor /
 * gen_rpo_merkle_tree (gpu/src/plan.rs:32-174; kernels gpu/src/metal/hash_shaders.h.metal:215-380).
 * Digests are 4 Fp elements (Montgomery form, 32 bytes).
 * ms_rpo256_rows            = update(col) x ncols + finish(): digests[r] = RPO(row r of the columns), with
 *                             the reference's padding rule when ncols is not a multiple of 8
 * ms_rpo256_rows_row_major  = the same for a row-major matrix [nrows][ncols] (GpuRpo256RowMajor: ncols = 8)
 * ms_rpo256_merkle          = gen_rpo_merkle_tree: nodes[k] = merge(nodes[2k], nodes[2k+1]), leaf pairs feed
 *                             nodes[n/2 ..), nodes[1] = root, nodes[0] = zero */
int ms_rpo256_rows(ms_ctx* ctx, size_t nrows, const void* const* d_cols, unsigned ncols, void* d_digests);
int ms_rpo256_rows_row_major(ms_ctx* ctx, size_t nrows, unsigned ncols, const void* d_matrix, void* d_digests);
/* MatrixMerkleTree over RPO-256 (README.md:90 "coming soon"): the leaf of row r absorbs the
 * row's elements column by column; an Fq3 element contributes c0, c1, c2, the order its SHA-256 leaf serialises
 * (src/hash.rs:93-98).  field = MS_GOLDILOCKS_FP or MS_GOLDILOCKS_FQ3. */
int ms_rpo256_rows_field(ms_ctx* ctx, int field, size_t nrows, const void* const* d_cols, unsigned ncols, void* d_digests);
int ms_rpo256_merkle(ms_ctx* ctx, size_t nleaves, const void* d_leaves, void* d_nodes);

/* ---- multi-GPU exchange (new work: the reference has one metal::Device,
 * gpu/src/plan.rs:465-468).  One context per process and GPU; RCCL (librccl.so.1, loaded on first use)
 * over xGMI.  Transforms need no communication -- rank g owns the columns {c : c mod nranks == g}, in that
 * order -- and exactly two steps do:
 *   ms_cols_to_rows_alltoall  a Merkle leaf hashes one element of EVERY column of a row (src/merkle.rs:428-431):
 *                             column shards -> row shards.  d_my_cols[j] = my j-th column, nrows elements (the
 *                             bit-reversed LDE); d_shard_cols[c], c < total_cols, receives rows
 *                             [rank * nrows/nranks, (rank+1) * nrows/nranks) of column c.  Row blocks are sent
 *                             straight out of the LDE columns (ncclSend/ncclRecv pairs in one group: every pair
 *                             of GPUs talks directly, all xGMI links carry payload at once); no staging copies.
 *   ms_allgather_digests      the nranks subtree roots (32 bytes each) -> every rank, device memory; the top
 *                             log2(nranks) levels are then ms_sha256_merkle / ms_rpo256_merkle over them, which
 *                             gives the root MerkleTree::from_matrix computes on one device, byte for byte.
 * ms_comm_unique_id is called by ONE rank; the 128 bytes reach the others by any host channel (the mirror uses
 * the torch.distributed store, a Rust host would use its own launcher).  Asynchronous on the context's stream.
 * nranks must be a power of two dividing nrows. */
#define MS_COMM_ID_BYTES 128
int ms_comm_unique_id(void* h_id128);
int ms_comm_init(ms_ctx* ctx, int nranks, int rank, const void* h_id128);
int ms_comm_destroy(ms_ctx* ctx);
int ms_comm_rank(ms_ctx* ctx, int* rank, int* nranks);
int ms_cols_to_rows_alltoall(ms_ctx* ctx, int field, size_t nrows, const void* const* d_my_cols, unsigned my_ncols,
                             unsigned total_cols, void* const* d_shard_cols);
int ms_allgather_digests(ms_ctx* ctx, const void* d_my_digest32, void* d_all_digests);
/* The point-to-point schedule ms_cols_to_rows_alltoall executes on rank `rank` of `nranks`, as data (a pure function of
 * its arguments; no context, no GPU): ops[0 .. *count) in issue order.  MS_XCHG_SEND: blk_bytes bytes of my column
 * `src_col` (index into d_my_cols) from byte `src_offset` to `peer`; MS_XCHG_RECV: blk_bytes bytes from `peer` into the start
 * of shard column `dst_col` (index into d_shard_cols); MS_XCHG_COPY: the device-to-device copy of my own block.  Sends and
 * receives between a pair of ranks match in issue order (RCCL / NCCL point-to-point semantics).  The entry point exists so
 * that the offsets which run over xGMI can be executed and checked anywhere -- tests/test_exchange_schedule.py replays the
 * schedules of all ranks on the host for 2, 4 and 8 ranks, and the gloo stand-in of the CPU tests (tests/gloo_comm.py)
 * issues exactly these operations.  Returns MS_ERR_INVALID when cap is too small (*count = the number needed). */
enum { MS_XCHG_SEND = 0, MS_XCHG_RECV = 1, MS_XCHG_COPY = 2 };
typedef struct { uint32_t kind, peer, src_col, dst_col; uint64_t src_offset, bytes; } ms_xchg_op;
int ms_cols_to_rows_schedule(unsigned nranks, unsigned rank, unsigned my_ncols, unsigned total_cols, size_t blk_bytes,
                             ms_xchg_op* ops, size_t cap, size_t* count);
/* A batch of point-to-point transfers of device memory in one RCCL group (ncclSend / ncclRecv; kind = MS_XCHG_SEND or
 * MS_XCHG_RECV): the exchange of whole row shards between the ranks that hold the constraint-evaluation rows when the
 * number of ranks does not divide the blow-up (ministark_amd/distributed.py eval_constraints_sharded; SURVEY.md 8(e):
 * src/eval_cpu.rs:116-119 reads rotated rows).  Asynchronous on the context's stream. */
typedef struct { uint32_t kind, peer; void* d_ptr; uint64_t bytes; } ms_p2p_op;
int ms_p2p_batch(ms_ctx* ctx, const ms_p2p_op* ops, size_t count);

#ifdef __cplusplus
}
#endif
#endif /* MINISTARK_HIP_H */
