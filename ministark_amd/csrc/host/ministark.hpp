// ministark.hpp -- C++17 host mirror of the reference's gpu-poly interface over the C ABI
// (include/ministark_hip.h).  The reference is Rust; this header stands where its
// `cfg(feature = "hip")` arm would: same item names, argument meaning and
// error behaviour -- the reference panics on GPU-side failures (gpu/src/plan.rs:248,255,360),
// here every non-zero status becomes a std::runtime_error.
//   Planner / get_planner()          gpu/src/plan.rs:327-351
//   GpuVec<F>                        src/utils.rs:438-470
//   Radix2EvaluationDomain           ark-poly, as consumed at gpu/src/plan.rs:386-423
//   GpuFft<F> / GpuIfft<F>           gpu/src/plan.rs:236-325
//   Matrix<F>                        src/matrix.rs:26-394
//   MerkleTree                       src/merkle.rs:296-361 (MatrixMerkleTreeImpl<Sha256HashFn>), prove :149-206
// stages.hpp: the 17 element-wise stages; expr.hpp: constraint DAG -> program -> eval; prover.hpp: FRI fold,
// DEEP composer, extension-column scans, queries, proof-of-work, RPO front-ends.
// Header-only; needs no HIP headers, link with -lministark_hip.
#pragma once
#include <algorithm>
#include <array>
#include <deque>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/ministark_hip.h"

namespace ms {

inline void check(int rc) {
    if (rc != MS_OK) throw std::runtime_error("ministark_hip error " + std::to_string(rc) + ": " + ms_last_error());
}

// field tags = GpuField::field_name() (gpu/src/fields.rs)
struct Fp { static constexpr int id = MS_GOLDILOCKS_FP; static constexpr unsigned words = 1; };
struct Fq3 { static constexpr int id = MS_GOLDILOCKS_FQ3; static constexpr unsigned words = 3; };
struct Fp252 { static constexpr int id = MS_STARK252_FP; static constexpr unsigned words = 4; };

namespace gl {   // just enough Goldilocks host arithmetic to build domain constants
constexpr uint64_t P = 0xFFFFFFFF00000001ull;
inline uint64_t mul(uint64_t a, uint64_t b) { return (uint64_t)((unsigned __int128)a * b % P); }
inline uint64_t pow(uint64_t a, uint64_t e) { uint64_t r = 1; while (e) { if (e & 1) r = mul(r, a); a = mul(a, a); e >>= 1; } return r; }
inline uint64_t to_mont(uint64_t x) { return mul(x % P, 0xFFFFFFFFull); }
inline uint64_t add(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + b) % P); }
inline uint64_t neg(uint64_t a) { return a ? P - a : 0; }
inline uint64_t sub(uint64_t a, uint64_t b) { return add(a, neg(b)); }
inline uint64_t inv(uint64_t a) { return pow(a, P - 2); }
inline uint64_t from_mont(uint64_t m) { return mul(m, 0xFFFFFFFE00000001ull); }        // 2^-64 mod p
}  // namespace gl

class Planner {
public:
    explicit Planner(int device = 0) { check(ms_ctx_create(device, &ctx_)); }
    ~Planner() { ms_ctx_destroy(ctx_); }
    Planner(const Planner&) = delete;
    Planner& operator=(const Planner&) = delete;
    ms_ctx* ctx() const { return ctx_; }
    void sync() const { check(ms_sync(ctx_)); }        // command_buffer.wait_until_completed()
    // specialised constraint kernels of this context: compiled / loaded from the on-disk cache / left to the interpreter (include/ministark_hip.h)
    ms_jit_stats jit_stats() const { ms_jit_stats st{}; check(ms_eval_jit_stats(ctx_, &st)); return st; }
private:
    ms_ctx* ctx_ = nullptr;
};
inline Planner& get_planner() { static Planner p(0); return p; }

// One device buffer shared by the gathers of an opening phase, downloaded once: every gather that is handed the arena writes
// into its own slice, and the first fetch brings the whole used part to the host with a single copy.  Six to forty small
// downloads cost more host time than the gathers cost on the device (the query phase of a 2^21-row proof: 0.75 ms against
// 0.16 ms of kernels).  A gather that does not fit falls back to a buffer of its own.
class GatherArena {
public:
    explicit GatherArena(Planner& pl, size_t capacity = (size_t)4 << 20) : pl_(&pl), cap_(capacity) { check(ms_alloc(pl.ctx(), cap_, &d_)); }
    ~GatherArena() { if (d_) ms_free(pl_->ctx(), d_); }
    GatherArena(const GatherArena&) = delete;
    GatherArena& operator=(const GatherArena&) = delete;
    bool reserve(size_t bytes, size_t* off) {           // 32-byte aligned slices (the digest gather writes whole records)
        const size_t at = (used_ + 31) & ~(size_t)31;
        if (at + bytes > cap_) return false;
        *off = at; used_ = at + bytes;
        return true;
    }
    void* dev(size_t off) const { return (char*)d_ + off; }
    bool owns(const void* p) const { return (const char*)p >= (const char*)d_ && (const char*)p < (const char*)d_ + cap_; }
    // a digest gather into a slice of this arena joins the ONE launch `flush` issues (ms_gather_digests_multi): the openings of a proof
    // are two dozen gathers of a few records each, i.e. two dozen launch latencies
    void defer_digests(const void* digests, size_t ndigests, const std::vector<uint64_t>& ids, void* out) {
        seg_src_.push_back(digests); seg_n_.push_back(ndigests); seg_cnt_.push_back(ids.size()); seg_out_.push_back(out);
        seg_idx_.insert(seg_idx_.end(), ids.begin(), ids.end());
    }
    void flush() {
        if (seg_src_.empty()) return;
        check(ms_gather_digests_multi(pl_->ctx(), (unsigned)seg_src_.size(), seg_src_.data(), seg_n_.data(), seg_idx_.data(), seg_cnt_.data(), seg_out_.data()));
        seg_src_.clear(); seg_n_.clear(); seg_cnt_.clear(); seg_out_.clear(); seg_idx_.clear();
    }
    const uint8_t* host(size_t off) {                   // waits for the stream and copies on the first call after a reserve
        flush();
        if (fetched_ < used_) { host_.resize(used_); check(ms_download(pl_->ctx(), host_.data() + fetched_, (const char*)d_ + fetched_, used_ - fetched_)); fetched_ = used_; }
        return host_.data() + off;
    }
private:
    Planner* pl_; void* d_ = nullptr; size_t cap_, used_ = 0, fetched_ = 0; std::vector<uint8_t> host_;
    std::vector<const void*> seg_src_; std::vector<size_t> seg_n_, seg_cnt_; std::vector<void*> seg_out_; std::vector<uint64_t> seg_idx_;
};

// one digest gather: deferred into the arena's single launch when its output is a slice of the arena, launched at once otherwise
inline void gather_digests_into(Planner& pl, const void* digests, size_t ndigests, const std::vector<uint64_t>& ids, void* out, GatherArena* arena) {
    if (ids.empty()) return;
    if (arena && arena->owns(out)) arena->defer_digests(digests, ndigests, ids, out);
    else check(ms_gather_digests(pl.ctx(), ndigests, digests, ids.data(), ids.size(), out));
}

// The result of a gather that has been launched but not fetched: an opening launches every gather first and fetches afterwards,
// so the host waits for the device once instead of once per call (the device runs them back to back on the context's stream).
// With an arena the result is a slice of the arena's buffer and fetch() reads the arena's single download.
class Pending {
public:
    Pending() = default;
    Pending(Planner& pl, size_t bytes, GatherArena* arena = nullptr) : pl_(&pl), bytes_(bytes) {
        if (!bytes) return;
        if (arena && arena->reserve(bytes, &off_)) { arena_ = arena; d_ = arena->dev(off_); }
        else check(ms_alloc(pl.ctx(), bytes, &d_));
    }
    ~Pending() { release(); }
    Pending(Pending&& o) noexcept : pl_(o.pl_), d_(o.d_), bytes_(o.bytes_), arena_(o.arena_), off_(o.off_) { o.d_ = nullptr; }
    Pending& operator=(Pending&& o) noexcept { if (this != &o) { release(); pl_ = o.pl_; d_ = o.d_; bytes_ = o.bytes_; arena_ = o.arena_; off_ = o.off_; o.d_ = nullptr; } return *this; }
    Pending(const Pending&) = delete;
    Pending& operator=(const Pending&) = delete;
    void* ptr() const { return d_; }
    size_t bytes() const { return bytes_; }
    template <class T> std::vector<T> fetch() const {                   // waits for the stream, then copies
        std::vector<T> out(bytes_ / sizeof(T));
        if (!bytes_) return out;
        if (arena_) memcpy(out.data(), arena_->host(off_), bytes_);
        else check(ms_download(pl_->ctx(), out.data(), d_, bytes_));
        return out;
    }
private:
    void release() { if (d_ && !arena_) ms_free(pl_->ctx(), d_); d_ = nullptr; }
    Planner* pl_ = nullptr; void* d_ = nullptr; size_t bytes_ = 0; GatherArena* arena_ = nullptr; size_t off_ = 0;
};

template <class F>
class GpuVec {
public:
    GpuVec(Planner& pl, size_t n) : pl_(&pl), n_(n) { check(ms_alloc(pl.ctx(), n * F::words * 8 + 8, &ptr_)); }
    GpuVec(Planner& pl, const std::vector<uint64_t>& host) : GpuVec(pl, host.size() / F::words) { upload(host); }
    ~GpuVec() { if (ptr_) ms_free(pl_->ctx(), ptr_); }
    GpuVec(GpuVec&& o) noexcept : pl_(o.pl_), n_(o.n_), ptr_(o.ptr_) { o.ptr_ = nullptr; }
    GpuVec& operator=(GpuVec&& o) noexcept {
        if (this != &o) { if (ptr_) ms_free(pl_->ctx(), ptr_); pl_ = o.pl_; n_ = o.n_; ptr_ = o.ptr_; o.ptr_ = nullptr; }
        return *this;
    }
    GpuVec(const GpuVec&) = delete;
    size_t len() const { return n_; }
    void* ptr() const { return ptr_; }
    Planner& planner() const { return *pl_; }
    void upload(const std::vector<uint64_t>& host) { check(ms_upload(pl_->ctx(), ptr_, host.data(), n_ * F::words * 8)); }
    std::vector<uint64_t> to_host() const {
        std::vector<uint64_t> h(n_ * F::words);
        check(ms_download(pl_->ctx(), h.data(), ptr_, h.size() * 8));
        return h;
    }
    GpuVec clone() const { GpuVec c(*pl_, n_); check(ms_copy(pl_->ctx(), c.ptr_, ptr_, n_ * F::words * 8)); return c; }   // device copy
private:
    Planner* pl_; size_t n_; void* ptr_ = nullptr;
};

// Goldilocks domains only (Fp252 domains carry 4-limb constants; see the Python mirror)
struct Radix2EvaluationDomain {
    size_t size; unsigned log_size; uint64_t group_gen, offset;      // canonical integers
    explicit Radix2EvaluationDomain(size_t n, uint64_t coset_offset = 1) : size(n), log_size(0), offset(coset_offset % gl::P) {
        if (n == 0 || (n & (n - 1))) throw std::invalid_argument("domain size must be a power of two");
        while (((size_t)1 << log_size) < n) log_size++;
        if (log_size > 32) throw std::invalid_argument("domain exceeds the two-adicity");
        group_gen = gl::pow(1753635133440165772ull, (uint64_t)1 << (32 - log_size));
    }
    static Radix2EvaluationDomain new_coset(size_t n, uint64_t off) { return Radix2EvaluationDomain(n, off); }
};

template <class F, int INVERSE>
class FftBase {
public:
    static constexpr size_t MIN_SIZE = 1;      // 2048 in the reference (gpu/src/plan.rs:248,294)
    FftBase(Planner& pl, const Radix2EvaluationDomain& d) : pl_(&pl), n_(d.size) {
        const uint64_t off = gl::to_mont(d.offset), gen = gl::to_mont(d.group_gen);
        check(ms_ntt_plan_create(pl.ctx(), F::id, d.log_size, INVERSE, &off, &gen, &plan_));
    }
    ~FftBase() { if (plan_) ms_ntt_plan_destroy(plan_); }
    FftBase(const FftBase&) = delete;
    void encode(GpuVec<F>& column) {            // plan.rs:254-263 / 300-309
        if (column.len() != n_) throw std::invalid_argument("column length differs from the domain size");   // plan.rs:257 assert_eq!
        check(ms_ntt_encode(plan_, column.ptr()));
    }
    void execute() { check(ms_ntt_execute(plan_)); }   // plan.rs:229-232 (blocks)
    // encode of every column + execute WITHOUT the wait (the Matrix methods): in place, ordered on the planner's stream
    void enqueue(const std::vector<GpuVec<F>>& cols) {
        std::vector<void*> p;
        for (auto& c : cols) { if (c.len() != n_) throw std::invalid_argument("column length differs from the domain size"); p.push_back(c.ptr()); }
        check(ms_ntt_enqueue(plan_, p.data(), (unsigned)p.size()));
    }
    ms_ntt_plan* plan() const { return plan_; }
private:
    Planner* pl_; size_t n_; ms_ntt_plan* plan_ = nullptr;
};
template <class F> using GpuFft = FftBase<F, 0>;
template <class F> using GpuIfft = FftBase<F, 1>;

class MerkleTree;

// The transforms ENQUEUE on the planner's stream and return; whatever consumes their result is ordered behind them on the same stream, and
// whatever brings bytes to the host (to_host, MerkleTree::root, a gather's fetch) waits.  The reference's block (GpuFft::execute waits for its
// command buffer, gpu/src/plan.rs:378-386): a caller that wants that calls planner().sync() -- the prover does not (a wait after every
// transform left the device idle until the host's next launch arrived).
template <class F>
class Matrix {
public:
    std::vector<GpuVec<F>> columns;
    Matrix() = default;
    explicit Matrix(std::vector<GpuVec<F>>&& cols) : columns(std::move(cols)) {}
    // Matrix::new (src/matrix.rs:32-38): every column has the length of the first -- the kernels take ONE row count per matrix
    void assert_rectangular() const { for (auto& c : columns) if (c.len() != columns[0].len()) throw std::invalid_argument("all columns of a matrix must have the same length"); }
    size_t num_rows() const { assert_rectangular(); return columns.empty() ? 0 : columns[0].len(); }
    size_t num_cols() const { return columns.size(); }
    Planner& planner() const { return columns.at(0).planner(); }
    Matrix clone() const { Matrix m; for (auto& c : columns) m.columns.push_back(c.clone()); return m; }
    Matrix& into_polynomials(const Radix2EvaluationDomain& d) {       // src/matrix.rs:102-116
        GpuIfft<F> ifft(planner(), d);
        ifft.enqueue(columns);                                         // encode + execute without the wait
        return *this;
    }
    // `self.clone().into_polynomials(d)` (src/matrix.rs:155-163) without the device copy: the out-of-place transform (ms_ntt_enqueue_to)
    Matrix interpolate(const Radix2EvaluationDomain& d) const {
        Matrix out;
        for (auto& c : columns) out.columns.emplace_back(planner(), c.len());
        GpuIfft<F> ifft(planner(), d);
        std::vector<const void*> in; for (auto& c : columns) in.push_back(c.ptr());
        auto o = out.ptrs();
        check(ms_ntt_enqueue_to(ifft.plan(), in.data(), o.data(), (unsigned)in.size()));
        return out;                                                    // enqueued: consumers are ordered behind it on the stream (see the class comment)
    }
    Matrix& into_evaluations(const Radix2EvaluationDomain& d) {       // src/matrix.rs:193-208 (columns already of domain size)
        GpuFft<F> fft(planner(), d);
        fft.enqueue(columns);
        return *this;
    }
    // evaluate / bit_reversed_evaluate (src/matrix.rs:237-251): columns shorter than the domain are coefficient
    // vectors, zero-extended on the device ("resize", :201); the input is preserved
    Matrix evaluate(const Radix2EvaluationDomain& d, bool bit_reversed = false) const {
        if (num_rows() == d.size && !bit_reversed) { Matrix m = clone(); m.into_evaluations(d); return m; }
        Matrix out;
        for (size_t c = 0; c < columns.size(); c++) out.columns.emplace_back(planner(), d.size);
        std::vector<const void*> in; for (auto& c : columns) in.push_back(c.ptr());
        auto o = out.ptrs();
        unsigned lg = 0; while (((size_t)1 << lg) < num_rows()) lg++;
        const uint64_t off = gl::to_mont(d.offset);
        check(ms_evaluate(planner().ctx(), F::id, lg, d.log_size, &off, in.data(), o.data(), (unsigned)in.size(), bit_reversed ? 1 : 0));
        return out;
    }
    Matrix bit_reversed_evaluate(const Radix2EvaluationDomain& d) const { return evaluate(d, true); }
    // composition_poly.chunks(k) spread over k columns (src/prover.rs:113-121)
    static Matrix from_chunks(const GpuVec<F>& poly, unsigned k) {
        if (k == 0 || poly.len() % k) throw std::invalid_argument("the coefficients do not split into that many columns");
        Matrix out;
        for (unsigned c = 0; c < k; c++) out.columns.emplace_back(poly.planner(), poly.len() / k);
        auto o = out.ptrs();
        check(ms_deinterleave(poly.planner().ctx(), F::id, poly.len() / k, k, poly.ptr(), o.data()));
        return out;
    }
    Matrix& bit_reverse_rows() {                                       // src/matrix.rs:352-354
        auto p = ptrs();
        unsigned lg = 0; while (((size_t)1 << lg) < num_rows()) lg++;
        check(ms_bit_reverse(planner().ctx(), F::id, lg, p.data(), (unsigned)p.size()));
        return *this;
    }
    // interpolate(trace_domain) + bit_reversed_evaluate(lde_domain), src/prover.rs:50-51, fused
    Matrix lde(unsigned log_blowup, uint64_t offset = 7, bool bit_reversed = true) const {
        Matrix out;
        for (size_t c = 0; c < columns.size(); c++) out.columns.emplace_back(planner(), num_rows() << log_blowup);
        std::vector<const void*> in; for (auto& c : columns) in.push_back(c.ptr());
        auto o = out.ptrs();
        unsigned lg = 0; while (((size_t)1 << lg) < num_rows()) lg++;
        const uint64_t off = gl::to_mont(offset);
        check(ms_lde(planner().ctx(), F::id, lg, log_blowup, &off, in.data(), o.data(), (unsigned)in.size(), bit_reversed ? 1 : 0));
        return out;
    }
    GpuVec<F> sum_columns() const {                                    // src/matrix.rs:357-394
        GpuVec<F> dst(planner(), num_rows());
        std::vector<const void*> in; for (auto& c : columns) in.push_back(c.ptr());
        check(ms_sum_columns(planner().ctx(), F::id, num_rows(), in.data(), (unsigned)in.size(), dst.ptr()));
        return dst;
    }
    // Matrix::get_row for every queried position (src/trace.rs:139-152): row-major [positions][num_cols * words]
    std::vector<uint64_t> get_rows(const std::vector<uint64_t>& positions) const { return get_rows_launch(positions).template fetch<uint64_t>(); }
    Pending get_rows_launch(const std::vector<uint64_t>& positions, GatherArena* arena = nullptr) const {
        const size_t words = num_cols() * F::words;
        Pending out(planner(), positions.size() * words * 8, arena);
        if (!out.bytes()) return out;
        std::vector<const void*> in; for (auto& c : columns) in.push_back(c.ptr());
        check(ms_gather_rows(planner().ctx(), F::id, num_rows(), in.data(), (unsigned)in.size(), positions.data(), positions.size(), out.ptr()));
        return out;
    }
    std::vector<void*> ptrs() const { std::vector<void*> p; for (auto& c : columns) p.push_back(c.ptr()); return p; }
};

// H of MatrixMerkleTreeImpl<H> (src/merkle.rs:296-361): Sha256HashFn (src/hash.rs:58-100) or RPO-256 over
// Goldilocks (gpu/src/plan.rs:32-174, README.md:90).  Both digests are 32 bytes, proofs have the same shape.
enum class Hash { Sha256, Rpo256 };

class MerkleTree {
public:
    template <class F>
    static MerkleTree from_matrix(const Matrix<F>& m, Hash h = Hash::Sha256) {                // src/merkle.rs:356-361
        MerkleTree t(m.planner(), m.num_rows());
        std::vector<const void*> in; for (auto& c : m.columns) in.push_back(c.ptr());
        if (h == Hash::Sha256) check(ms_sha256_rows(t.pl_->ctx(), F::id, t.n_, in.data(), (unsigned)in.size(), t.leaves_));
        else check(ms_rpo256_rows_field(t.pl_->ctx(), F::id, t.n_, in.data(), (unsigned)in.size(), t.leaves_));
        t.build(h);
        return t;
    }
    // Matrix::from_arrays(evaluations.as_chunks::<N>()) + from_matrix (src/fri.rs:213-216): commit to a bit-reversed
    // FRI layer whose rows are the cosets of `folding_factor` consecutive evaluations
    template <class F>
    static MerkleTree from_fri_layer(const GpuVec<F>& evaluations, unsigned folding_factor, Hash h = Hash::Sha256) {
        MerkleTree t(evaluations.planner(), evaluations.len() / folding_factor);
        if (h == Hash::Sha256) check(ms_sha256_rows_row_major(t.pl_->ctx(), F::id, t.n_, folding_factor, evaluations.ptr(), t.leaves_));
        else check(ms_rpo256_rows_row_major(t.pl_->ctx(), t.n_, folding_factor * (unsigned)(ms_field_bytes(F::id) / 8), evaluations.ptr(), t.leaves_));
        t.build(h);
        return t;
    }
    using Digest = std::array<uint8_t, 32>;
    struct MerkleView { std::vector<Digest> nodes, initial_leaves, sibling_leaves; unsigned height = 0; };   // src/merkle.rs:72-84
    // MerkleTreeImpl::prove (src/merkle.rs:149-206): the walk over indices is bookkeeping, the digests it lists
    // are gathered on the device and come back in one copy per array
    // `prove` in two halves: the gathers are launched now, PendingView::fetch() downloads and assembles the view
    struct PendingView {
        Pending leaves, nodes; std::vector<size_t> initial, sibling; unsigned height = 0;
        MerkleView fetch() const {
            const auto lv = leaves.fetch<Digest>(), nd = nodes.fetch<Digest>();
            MerkleView v;
            v.nodes = nd;
            for (size_t k : initial) v.initial_leaves.push_back(lv[k]);
            for (size_t k : sibling) v.sibling_leaves.push_back(lv[k]);
            v.height = height;
            return v;
        }
    };
    MerkleView prove(std::vector<size_t> indices) const { return prove_launch(std::move(indices)).fetch(); }
    PendingView prove_launch(std::vector<size_t> indices, GatherArena* arena = nullptr) const {
        for (size_t i : indices) if (i >= n_) throw std::out_of_range("leaf index out of bounds");          // Error::LeafIndexOutOfBounds
        std::sort(indices.begin(), indices.end());
        indices.erase(std::unique(indices.begin(), indices.end()), indices.end());
        std::vector<uint64_t> leaf_ids, node_ids;
        PendingView pv;
        std::deque<size_t> node_queue, leaf_queue(indices.begin(), indices.end());
        while (!leaf_queue.empty()) {
            const size_t index = leaf_queue.front(); leaf_queue.pop_front();
            pv.initial.push_back(leaf_ids.size()); leaf_ids.push_back(index);
            node_queue.push_back((n_ + index) >> 1);
            if (!leaf_queue.empty() && (index ^ 1) == leaf_queue.front()) { pv.initial.push_back(leaf_ids.size()); leaf_ids.push_back(leaf_queue.front()); leaf_queue.pop_front(); continue; }
            pv.sibling.push_back(leaf_ids.size()); leaf_ids.push_back(index ^ 1);
        }
        while (!node_queue.empty()) {
            const size_t index = node_queue.front(); node_queue.pop_front();
            if (index > 2) node_queue.push_back(index >> 1);
            if (!node_queue.empty() && (index ^ 1) == node_queue.front()) { node_queue.pop_front(); continue; }
            node_ids.push_back(index ^ 1);
        }
        pv.leaves = gather_launch(leaves_, leaf_ids, arena);
        pv.nodes = gather_launch(nodes_, node_ids, arena);
        while (((size_t)1 << pv.height) < n_) pv.height++;
        return pv;
    }
    size_t num_leaves() const { return n_; }
    std::array<uint8_t, 32> root() const {                              // nodes[1], src/merkle.rs:145-147
        std::array<uint8_t, 32> r{};
        check(ms_download(pl_->ctx(), r.data(), (const char*)nodes_ + 32, 32));
        return r;
    }
    ~MerkleTree() { if (leaves_) ms_free(pl_->ctx(), leaves_); if (nodes_) ms_free(pl_->ctx(), nodes_); }
    MerkleTree(MerkleTree&& o) noexcept : pl_(o.pl_), n_(o.n_), leaves_(o.leaves_), nodes_(o.nodes_) { o.leaves_ = o.nodes_ = nullptr; }
private:
    MerkleTree(Planner& pl, size_t n) : pl_(&pl), n_(n) { check(ms_alloc(pl.ctx(), n * 32, &leaves_)); check(ms_alloc(pl.ctx(), n * 32, &nodes_)); }
    void build(Hash h) {
        if (h == Hash::Sha256) check(ms_sha256_merkle(pl_->ctx(), n_, leaves_, nodes_));
        else check(ms_rpo256_merkle(pl_->ctx(), n_, leaves_, nodes_));
    }
    Pending gather_launch(const void* digests, const std::vector<uint64_t>& ids, GatherArena* arena) const {
        Pending out(*pl_, ids.size() * 32, arena);
        gather_digests_into(*pl_, digests, n_, ids, out.ptr(), arena);
        return out;
    }
    Planner* pl_; size_t n_; void* leaves_ = nullptr; void* nodes_ = nullptr;
};

}  // namespace ms
