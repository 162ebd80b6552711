// prover.hpp -- the remaining data-parallel pieces of default_prove (src/prover.rs:25-174) over the C ABI:
//   apply_drp                         src/fri.rs:526-567
//   DeepPolyComposer                  src/composer.rs:17-188
//   scan_affine / running_product     examples/brainfuck/trace.rs:108-289 (extension-column loops)
//   Queries                           src/trace.rs:113-157
//   grind_proof_of_work               src/random.rs:48-55
//   GpuRpo256ColumnMajor / RowMajor / gen_rpo_merkle_tree   gpu/src/plan.rs:32-174
// Host values of Fq are canonical integers: FqVal{c0,c1,c2} (c1 = c2 = 0 when Fq = Fp).
#pragma once
#include "ministark.hpp"

namespace ms {

struct FqVal {
    uint64_t c[3] = {0, 0, 0};
    bool operator==(const FqVal& o) const { return c[0] == o.c[0] && c[1] == o.c[1] && c[2] == o.c[2]; }
    bool operator<(const FqVal& o) const { return std::lexicographical_compare(c, c + 3, o.c, o.c + 3); }
};
namespace fq {          // Fq3 = Fp[x]/(x^3 - 2) on canonical values: bookkeeping of evaluation points only
inline uint64_t addp(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + b) % gl::P); }
inline FqVal mul(const FqVal& a, const FqVal& b) {
    auto m = gl::mul;
    return {{addp(m(a.c[0], b.c[0]), m(2, addp(m(a.c[1], b.c[2]), m(a.c[2], b.c[1])))),
             addp(addp(m(a.c[0], b.c[1]), m(a.c[1], b.c[0])), m(2, m(a.c[2], b.c[2]))),
             addp(addp(m(a.c[0], b.c[2]), m(a.c[1], b.c[1])), m(a.c[2], b.c[0]))}};
}
inline FqVal mul_base(const FqVal& a, uint64_t s) { return {{gl::mul(a.c[0], s), gl::mul(a.c[1], s), gl::mul(a.c[2], s)}}; }
inline FqVal pow(FqVal a, uint64_t e) { FqVal r{{1, 0, 0}}; while (e) { if (e & 1) r = mul(r, a); a = mul(a, a); e >>= 1; } return r; }
template <class F> inline void push_words(std::vector<uint64_t>& out, const FqVal& v) { for (unsigned w = 0; w < F::words; w++) out.push_back(gl::to_mont(v.c[w])); }
inline uint64_t from_mont(uint64_t m) { return gl::mul(m, gl::pow(0xFFFFFFFFull, gl::P - 2)); }
template <class F> inline FqVal from_words(const uint64_t* w) { FqVal v; for (unsigned k = 0; k < F::words; k++) v.c[k] = from_mont(w[k]); return v; }
}  // namespace fq

// apply_drp(evals, domain_offset, alpha, folding_factor): `evals` in bit-reversed order; returns the next layer
// (bit-reversed).  alpha: Montgomery words of one element of F.
template <class F>
inline GpuVec<F> apply_drp(const GpuVec<F>& evals, const std::vector<uint64_t>& alpha, unsigned folding_factor, uint64_t domain_offset = 1) {
    if (alpha.size() != F::words) throw std::invalid_argument("alpha has the wrong number of limbs");
    GpuVec<F> out(evals.planner(), evals.len() / folding_factor);
    unsigned log_n = 0; while (((size_t)1 << log_n) < evals.len()) log_n++;
    const uint64_t off = gl::to_mont(domain_offset);
    check(ms_fri_fold(evals.planner().ctx(), F::id, log_n, folding_factor, alpha.data(), &off, evals.ptr(), out.ptr()));
    return out;
}

// state = init; for every row: out[row] = state; state = a[row]*state + b[row]   (a or b may be null)
template <class F>
inline GpuVec<F> scan_affine(const GpuVec<F>* a, const GpuVec<F>* b, const std::vector<uint64_t>& init, bool inclusive = false) {
    const GpuVec<F>* ref = a ? a : b;
    if (!ref) throw std::invalid_argument("scan_affine: neither multipliers nor addends given");
    if (init.size() != F::words) throw std::invalid_argument("init has the wrong number of limbs");
    GpuVec<F> out(ref->planner(), ref->len());
    check(ms_scan_affine(ref->planner().ctx(), F::id, ref->len(), a ? a->ptr() : nullptr, b ? b->ptr() : nullptr, init.data(), inclusive ? 1 : 0, out.ptr()));
    return out;
}
template <class F> inline GpuVec<F> running_product(const GpuVec<F>& factors, const std::vector<uint64_t>& init) { return scan_affine<F>(&factors, nullptr, init); }

// `fold_positions` (src/fri.rs:615-622): strictly increasing positions -> their cosets, deduplicated
inline std::vector<size_t> fold_positions(const std::vector<size_t>& positions, unsigned folding_factor) {
    std::vector<size_t> out;
    for (size_t i = 1; i < positions.size(); i++)          // the reference's precondition (src/fri.rs:613), asserted by the Python mirror too
        if (positions[i] <= positions[i - 1]) throw std::invalid_argument("fold_positions: positions must be strictly increasing");
    for (size_t p : positions) if (out.empty() || out.back() != p / folding_factor) out.push_back(p / folding_factor);
    return out;
}
// rows `positions` of Matrix::from_arrays(evaluations.as_chunks::<N>()) (src/fri.rs:213-215): N consecutive evaluations each,
// gathered on the device as 32-byte records (FriProver::into_proof, src/fri.rs:148-165)
template <class F>
inline Pending fri_layer_rows_launch(const GpuVec<F>& layer, unsigned folding_factor, const std::vector<size_t>& positions, GatherArena* arena = nullptr) {
    const size_t words = (size_t)folding_factor * F::words;
    if (words % 4) throw std::invalid_argument("fri_layer_rows_launch: rows shorter than a 32-byte record");
    Planner& pl = layer.planner();
    Pending out(pl, positions.size() * words * 8, arena);
    if (!out.bytes()) return out;
    const size_t per = words / 4;
    std::vector<uint64_t> ids;
    for (size_t p : positions) for (size_t k = 0; k < per; k++) ids.push_back(p * per + k);
    gather_digests_into(pl, layer.ptr(), layer.len() * F::words / 4, ids, out.ptr(), arena);
    return out;
}
template <class F>
inline std::vector<uint64_t> fri_layer_rows(const GpuVec<F>& layer, unsigned folding_factor, const std::vector<size_t>& positions) {
    const size_t words = (size_t)folding_factor * F::words;
    if (words % 4) {                                       // rows shorter than a record: tiny layers only
        std::vector<uint64_t> out(positions.size() * words);
        if (out.empty()) return out;
        const auto all = layer.to_host();
        for (size_t i = 0; i < positions.size(); i++) memcpy(&out[i * words], &all[positions[i] * words], words * 8);
        return out;
    }
    return fri_layer_rows_launch(layer, folding_factor, positions).template fetch<uint64_t>();
}

// Queries::new: rows of the three LDE matrices at the query positions + batched openings of the three trees.  With an arena
// the constructor only launches the gathers (into the arena) and fetch() fills the fields, so that a prover can put the
// FRI layer openings into the same arena and download everything once.
template <class FqT>
struct Queries {
    std::vector<uint64_t> base_trace_values, extension_trace_values, composition_trace_values;
    MerkleTree::MerkleView base_trace_proof, extension_trace_proof, composition_trace_proof;
    Queries(const Matrix<Fp>& base_lde, const Matrix<FqT>* extension_lde, const Matrix<FqT>& composition_lde,
            const MerkleTree& base_tree, const MerkleTree* extension_tree, const MerkleTree& composition_tree, const std::vector<size_t>& positions,
            GatherArena* arena = nullptr) : has_ext_tree_(extension_tree), has_ext_lde_(extension_lde) {
        std::vector<uint64_t> pos(positions.begin(), positions.end());
        // every gather first, then the downloads: one wait for the device instead of eight
        bp_ = base_tree.prove_launch(positions, arena);
        if (extension_tree) ep_ = extension_tree->prove_launch(positions, arena);
        cp_ = composition_tree.prove_launch(positions, arena);
        bv_ = base_lde.get_rows_launch(pos, arena);
        cv_ = composition_lde.get_rows_launch(pos, arena);
        if (extension_lde) ev_ = extension_lde->get_rows_launch(pos, arena);
        if (!arena) fetch();
    }
    void fetch() {
        if (fetched_) return;
        fetched_ = true;
        base_trace_proof = bp_.fetch();
        if (has_ext_tree_) extension_trace_proof = ep_.fetch();
        composition_trace_proof = cp_.fetch();
        base_trace_values = bv_.template fetch<uint64_t>();
        if (has_ext_lde_) extension_trace_values = ev_.template fetch<uint64_t>();
        composition_trace_values = cv_.template fetch<uint64_t>();
    }
private:
    MerkleTree::PendingView bp_, ep_, cp_;
    Pending bv_, ev_, cv_;
    bool has_ext_tree_, has_ext_lde_, fetched_ = false;
};

// PublicCoin::grind_proof_of_work(bits): the smallest nonce >= 1 with `bits` leading zero bits of SHA-256(seed || nonce_be)
inline uint64_t grind_proof_of_work(Planner& pl, const std::array<uint8_t, 32>& seed, unsigned proof_of_work_bits, uint64_t max_nonce = (uint64_t)1 << 40) {
    uint64_t nonce = 0;
    check(ms_sha256_pow_grind(pl.ctx(), seed.data(), proof_of_work_bits, max_nonce, &nonce));
    return nonce;
}

// DeepPolyComposer::new(air, z, base_trace_polys, extension_trace_polys, composition_trace_polys)
// trace_arguments: the AIR's (column, offset) pairs (air.trace_arguments()).  Polynomials are coefficient-form
// matrices on the device (what interpolate / into_polynomials return).  FqT = Fq3 or Fp (Fq = Fp AIRs).
struct DeepCompositionCoeffs { std::vector<FqVal> execution_trace, composition_trace; FqVal degree[2]; };   // src/composer.rs:191-198
template <class FqT>
class DeepPolyComposer {
public:
    DeepPolyComposer(std::vector<std::pair<unsigned, int>> trace_arguments, size_t trace_len, FqVal z, const Matrix<Fp>& base_polys,
                     const Matrix<FqT>* extension_polys, const Matrix<FqT>& composition_polys)
        : args_(std::move(trace_arguments)), n_(trace_len), z_(z), base_(base_polys), ext_(extension_polys), comp_(composition_polys) {
        Radix2EvaluationDomain d(trace_len);
        g_ = d.group_gen; g_inv_ = gl::pow(g_, gl::P - 2);
        nbase_ = (unsigned)base_.num_cols();
    }
    // -> (execution trace values in trace_arguments order, composition trace values)   src/composer.rs:43-86
    std::pair<std::vector<FqVal>, std::vector<FqVal>> get_ood_evals() {
        std::vector<unsigned> bq_col, eq_col; std::vector<FqVal> bq_pt, eq_pt;
        for (auto& a : args_) { if (a.first < nbase_) { bq_col.push_back(a.first); bq_pt.push_back(point(a.second)); } else { eq_col.push_back(a.first - nbase_); eq_pt.push_back(point(a.second)); } }
        const FqVal z_n = fq::pow(z_, comp_.num_cols());
        std::vector<unsigned> cc; std::vector<FqVal> cp;
        for (unsigned c = 0; c < comp_.num_cols(); c++) { cc.push_back(c); cp.push_back(z_n); }
        // matrices over the same field with the same number of rows share ONE launch and one download (every call is a wait of the host for
        // the device and of the device for the host's next launch): the composition-trace polynomials ride with the trace polynomials of
        // their field -- the base trace's when Fq = Fp, the extension trace's otherwise
        std::vector<FqVal> bv, ev, cv;
        bool comp_done = false;
        if constexpr (FqT::words == 1) {
            if (comp_.num_rows() == base_.num_rows() && base_.num_cols() + comp_.num_cols() <= 96 && !bq_col.empty()) {
                auto cols = ptrs_of(base_); for (auto& c : comp_.columns) cols.push_back(c.ptr());
                auto qc = bq_col; auto qp = bq_pt;
                for (unsigned c : cc) qc.push_back((unsigned)base_.num_cols() + c);
                qp.insert(qp.end(), cp.begin(), cp.end());
                const auto all = horner_cols<Fp>(base_.planner(), base_.num_rows(), cols, qc, qp);
                bv.assign(all.begin(), all.begin() + bq_col.size()); cv.assign(all.begin() + bq_col.size(), all.end());
                comp_done = true;
            }
        } else if (ext_ && comp_.num_rows() == ext_->num_rows() && ext_->num_cols() + comp_.num_cols() <= 96 && !eq_col.empty()) {
            auto cols = ptrs_of(*ext_); for (auto& c : comp_.columns) cols.push_back(c.ptr());
            auto qc = eq_col; auto qp = eq_pt;
            for (unsigned c : cc) qc.push_back((unsigned)ext_->num_cols() + c);
            qp.insert(qp.end(), cp.begin(), cp.end());
            const auto all = horner_cols<FqT>(base_.planner(), ext_->num_rows(), cols, qc, qp);
            ev.assign(all.begin(), all.begin() + eq_col.size()); cv.assign(all.begin() + eq_col.size(), all.end());
            bv = horner(base_, bq_col, bq_pt);
            comp_done = true;
        }
        if (!comp_done) {
            bv = horner(base_, bq_col, bq_pt);
            if (ext_) ev = horner(*ext_, eq_col, eq_pt);
            cv = horner(comp_, cc, cp);
        }
        size_t bi = 0, ei = 0;
        std::vector<FqVal> execution;
        for (auto& a : args_) execution.push_back(a.first < nbase_ ? bv[bi++] : ev[ei++]);
        ood_exec_ = execution; ood_comp_ = cv; have_ood_ = true;
        return {ood_exec_, ood_comp_};
    }
    GpuVec<FqT> into_deep_poly(const DeepCompositionCoeffs& coeffs) {           // src/composer.rs:89-188
        std::vector<const void*> bp, ep;
        for (auto& c : base_.columns) bp.push_back(c.ptr());
        auto& second = FqT::words == 1 ? bp : ep;                    // Fq = Fp: everything is a base column
        if (ext_) for (auto& c : ext_->columns) second.push_back(c.ptr());
        for (auto& c : comp_.columns) second.push_back(c.ptr());
        unsigned log_n = 0; while (((size_t)1 << log_n) < n_) log_n++;
        return compose(coeffs, bp, ep, n_, [&](Planner& pl, const Terms& t, void* out) {
            return ms_deep_compose(pl.ctx(), FqT::id, log_n, nullptr, bp.empty() ? nullptr : bp.data(), (unsigned)bp.size(), ep.empty() ? nullptr : ep.data(), (unsigned)ep.size(),
                                   t.pts.data(), t.npoints, t.tcol.data(), t.tpoint.data(), t.al.data(), t.od.data(), (unsigned)t.tcol.size(), t.da.data(), t.db.data(), out); });
    }
    // into_deep_poly(coeffs) followed by into_bit_reversed_evaluations(lde_domain) (src/prover.rs:149-152) in one step: the polynomial's
    // values on the LDE domain (domain_size points, offset 7) computed from the committed LDE matrices themselves (ms_deep_rows) -- the
    // same field elements, without the coset transforms, the inverse transform and the LDE.  Goldilocks fields.
    GpuVec<FqT> into_deep_evaluations(const DeepCompositionCoeffs& coeffs, const Matrix<Fp>& base_lde, const Matrix<FqT>* ext_lde, const Matrix<FqT>& comp_lde) {
        const size_t N = base_lde.num_rows();
        if (comp_lde.num_rows() != N || (ext_lde && ext_lde->num_rows() != N)) throw std::invalid_argument("into_deep_evaluations: LDE matrices of different heights");
        std::vector<const void*> bp, ep;
        for (auto& c : base_lde.columns) bp.push_back(c.ptr());
        auto& second = FqT::words == 1 ? bp : ep;
        if (ext_lde) for (auto& c : ext_lde->columns) second.push_back(c.ptr());
        for (auto& c : comp_lde.columns) second.push_back(c.ptr());
        unsigned log_N = 0; while (((size_t)1 << log_N) < N) log_N++;
        return compose(coeffs, bp, ep, N, [&](Planner& pl, const Terms& t, void* out) {
            return ms_deep_rows(pl.ctx(), FqT::id, log_N, nullptr, 0, N, bp.empty() ? nullptr : bp.data(), (unsigned)bp.size(), ep.empty() ? nullptr : ep.data(), (unsigned)ep.size(),
                                t.pts.data(), t.npoints, t.tcol.data(), t.tpoint.data(), t.al.data(), t.od.data(), (unsigned)t.tcol.size(), t.da.data(), t.db.data(), out); });
    }
private:
    struct Terms { std::vector<unsigned> tcol, tpoint; std::vector<uint64_t> pts, al, od, da, db; unsigned npoints = 0; };
    template <class Call>
    GpuVec<FqT> compose(const DeepCompositionCoeffs& coeffs, const std::vector<const void*>&, const std::vector<const void*>&, size_t out_len, Call call) {
        if (!have_ood_) get_ood_evals();
        Planner& pl = base_.planner();
        std::vector<FqVal> points;
        auto pid = [&](const FqVal& p) { for (size_t k = 0; k < points.size(); k++) if (points[k] == p) return (unsigned)k; points.push_back(p); return (unsigned)points.size() - 1; };
        const FqVal z_n = fq::pow(z_, comp_.num_cols());
        const unsigned next = ext_ ? (unsigned)ext_->num_cols() : 0;
        Terms t;
        std::vector<FqVal> talpha, tood;
        for (unsigned c = 0; c < comp_.num_cols(); c++) { t.tcol.push_back(nbase_ + next + c); t.tpoint.push_back(pid(z_n)); talpha.push_back(coeffs.composition_trace.at(c)); tood.push_back(ood_comp_[c]); }
        for (size_t k = 0; k < args_.size(); k++) { t.tcol.push_back(args_[k].first); t.tpoint.push_back(pid(point(args_[k].second))); talpha.push_back(coeffs.execution_trace.at(k)); tood.push_back(ood_exec_[k]); }
        auto flat = [](const std::vector<FqVal>& v) { std::vector<uint64_t> o; for (auto& q : v) fq::push_words<FqT>(o, q); return o; };
        t.pts = flat(points); t.al = flat(talpha); t.od = flat(tood); t.da = flat({coeffs.degree[0]}); t.db = flat({coeffs.degree[1]});
        t.npoints = (unsigned)points.size();
        GpuVec<FqT> out(pl, out_len);
        check(call(pl, t, out.ptr()));
        return out;
    }
    FqVal point(int offset) const { return fq::mul_base(z_, gl::pow(offset >= 0 ? g_ : g_inv_, (uint64_t)(offset >= 0 ? offset : -offset))); }
    template <class CF> static std::vector<const void*> ptrs_of(const Matrix<CF>& m) { std::vector<const void*> in; for (auto& c : m.columns) in.push_back(c.ptr()); return in; }
    template <class CF>
    static std::vector<FqVal> horner_cols(Planner& pl, size_t nrows, const std::vector<const void*>& in, const std::vector<unsigned>& qcol, const std::vector<FqVal>& qpt) {
        std::vector<FqVal> res;
        if (qcol.empty()) return res;
        std::vector<uint64_t> pts, out(qcol.size() * FqT::words);
        for (auto& p : qpt) fq::push_words<FqT>(pts, p);
        check(ms_horner_eval(pl.ctx(), CF::id, FqT::id, nrows, in.data(), (unsigned)in.size(), qcol.data(), pts.data(), (unsigned)qcol.size(), out.data()));
        for (size_t k = 0; k < qcol.size(); k++) res.push_back(fq::from_words<FqT>(&out[k * FqT::words]));
        return res;
    }
    template <class CF>
    std::vector<FqVal> horner(const Matrix<CF>& m, const std::vector<unsigned>& qcol, const std::vector<FqVal>& qpt) {
        if (qcol.empty()) return {};
        return horner_cols<CF>(m.planner(), m.num_rows(), ptrs_of(m), qcol, qpt);
    }
    std::vector<std::pair<unsigned, int>> args_;
    size_t n_; FqVal z_; const Matrix<Fp>& base_; const Matrix<FqT>* ext_; const Matrix<FqT>& comp_;
    uint64_t g_ = 1, g_inv_ = 1; unsigned nbase_ = 0;
    std::vector<FqVal> ood_exec_, ood_comp_; bool have_ood_ = false;
};

// ---- RPO-256 front-ends (gpu/src/plan.rs:32-174); digests are 4 Fp elements, Montgomery form
class GpuRpo256ColumnMajor {                      // ::new(n, requires_padding); update(col) per column; finish()
public:
    GpuRpo256ColumnMajor(Planner& pl, size_t n) : pl_(&pl), n_(n) {}
    void update(const GpuVec<Fp>& col) { if (col.len() != n_) throw std::invalid_argument("column length differs from n"); cols_.push_back(col.ptr()); }
    GpuVec<Fp> finish() {
        GpuVec<Fp> digests(*pl_, n_ * 4);
        check(ms_rpo256_rows(pl_->ctx(), n_, cols_.data(), (unsigned)cols_.size(), digests.ptr()));
        pl_->sync();
        return digests;
    }
private:
    Planner* pl_; size_t n_; std::vector<const void*> cols_;
};
inline GpuVec<Fp> rpo256_rows_row_major(const GpuVec<Fp>& rows, unsigned ncols = 8) {      // GpuRpo256RowMajor: update(rows) + finish()
    GpuVec<Fp> digests(rows.planner(), rows.len() / ncols * 4);
    check(ms_rpo256_rows_row_major(rows.planner().ctx(), rows.len() / ncols, ncols, rows.ptr(), digests.ptr()));
    return digests;
}
inline GpuVec<Fp> gen_rpo_merkle_tree(const GpuVec<Fp>& leaves) {                          // nodes, [n][4] Fp
    GpuVec<Fp> nodes(leaves.planner(), leaves.len());
    check(ms_rpo256_merkle(leaves.planner().ctx(), leaves.len() / 4, leaves.ptr(), nodes.ptr()));
    return nodes;
}

}  // namespace ms
