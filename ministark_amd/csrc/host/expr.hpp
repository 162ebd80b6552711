// expr.hpp -- host side of the fused constraint evaluator in C++: the reference's expression DAG
// `Expr<AlgebraicItem<FieldVariant<Fp, Fq>>>` (src/expression.rs:33-40, src/constraints.rs:21-28) and its
// lowering to the register program of include/ministark_hip.h ("constraint program").  `a - b` is
// Add(a, Neg(b)) (src/expression.rs:573-580); leaves X, Constant, Challenge, Hint, Trace(col, offset),
// Periodic.  `eval(...)` mirrors `eval_gpu::eval` / `eval_cpu::eval` (src/eval_gpu.rs:46-54,
// src/eval_cpu.rs:33-42).  Only bookkeeping happens here (hash-consing as reuse_shared_nodes does,
// src/expression.rs:186-357; typing as eval_cpu.rs:306-428; register allocation); every field operation
// on data runs in the library.  Goldilocks Fp / Fq3 (the 252-bit instantiation lives in the Python mirror).
#pragma once
#include <map>
#include <memory>
#include "ministark.hpp"

namespace ms {
namespace expr {

enum Op : uint32_t {
    OP_X_P = 0, OP_CONST_P, OP_CONST_Q, OP_TRACE_P, OP_TRACE_Q, OP_PERIODIC_P, OP_PERIODIC_Q, OP_NEG_P, OP_NEG_Q,
    OP_ADD_PP, OP_ADD_QQ, OP_ADD_QP, OP_MUL_PP, OP_MUL_QQ, OP_MUL_QP, OP_INV_P, OP_INV_Q, OP_POW_P, OP_POW_Q,
    OP_EMBED, OP_STORE_Q, OP_STORE_P
};
enum Kind { K_X, K_CONST_P, K_CONST_Q, K_CHALLENGE, K_HINT, K_TRACE, K_PERIODIC, K_NEG, K_ADD, K_MUL, K_DIV, K_POW };

struct Node;
using E = std::shared_ptr<const Node>;
struct Node {
    Kind kind;
    E a, b;
    uint64_t v[3] = {0, 0, 0};          // constant (canonical integers)
    uint32_t idx = 0;                   // challenge / hint index, trace column, exponent, periodic interval
    int32_t off = 0;                    // trace offset
    std::vector<uint64_t> coeffs;       // periodic column polynomial
};
inline E mk(Kind k, E a = nullptr, E b = nullptr) { auto n = std::make_shared<Node>(); n->kind = k; n->a = std::move(a); n->b = std::move(b); return n; }
inline E X() { return mk(K_X); }
inline E Constant(uint64_t c) { auto n = std::make_shared<Node>(); n->kind = K_CONST_P; n->v[0] = c % gl::P; return n; }
inline E ConstantQ(uint64_t c0, uint64_t c1, uint64_t c2) { auto n = std::make_shared<Node>(); n->kind = K_CONST_Q; n->v[0] = c0 % gl::P; n->v[1] = c1 % gl::P; n->v[2] = c2 % gl::P; return n; }
inline E Challenge(uint32_t i) { auto n = std::make_shared<Node>(); n->kind = K_CHALLENGE; n->idx = i; return n; }
inline E Hint(uint32_t i) { auto n = std::make_shared<Node>(); n->kind = K_HINT; n->idx = i; return n; }
inline E Trace(uint32_t col, int32_t offset = 0) { auto n = std::make_shared<Node>(); n->kind = K_TRACE; n->idx = col; n->off = offset; return n; }
// PeriodicColumn::new(coeffs, interval_size): polynomial coefficients (canonical) over one interval of the trace
inline E Periodic(std::vector<uint64_t> coeffs, uint32_t interval = 0) {
    auto n = std::make_shared<Node>(); n->kind = K_PERIODIC; n->idx = interval ? interval : (uint32_t)coeffs.size();
    for (auto& c : coeffs) c %= gl::P;
    n->coeffs = std::move(coeffs); return n;
}
inline E operator+(const E& a, const E& b) { return mk(K_ADD, a, b); }
inline E operator-(const E& a) { return mk(K_NEG, a); }
inline E operator-(const E& a, const E& b) { return mk(K_ADD, a, mk(K_NEG, b)); }       // expression.rs:573-580
inline E operator*(const E& a, const E& b) { return mk(K_MUL, a, b); }
inline E operator/(const E& a, const E& b) { return mk(K_DIV, a, b); }
inline E operator+(const E& a, uint64_t c) { return a + Constant(c); }
inline E operator-(const E& a, uint64_t c) { return a - Constant(c); }
inline E operator*(const E& a, uint64_t c) { return a * Constant(c); }
inline E pow(const E& a, uint32_t e) { auto n = std::make_shared<Node>(); n->kind = K_POW; n->a = a; n->idx = e; return n; }

struct Instr { uint32_t op, dst, a, b; };
struct Program {
    std::vector<Instr> instrs;
    std::vector<uint64_t> consts;                         // u64 words, Montgomery
    std::map<uint32_t, uint32_t> challenge_slots, hint_slots;   // index -> word offset in consts
    std::vector<std::pair<std::vector<uint64_t>, uint32_t>> periodic;
    bool fq_is_ext = true;
    unsigned max_p = 0, max_q = 0;
};

// Trace columns < num_base_columns are Fp, the rest Fq (eval_cpu.rs:103-134).  Challenges and hints are Fq
// (eval_cpu.rs:111-113); with fq_is_ext = false (Fq = Fp AIRs such as examples/fib) they are Fp.
inline Program compile_expr(const E& root_expr, unsigned num_base_columns, bool fq_is_ext = true) {
    Program prog;
    prog.fq_is_ext = fq_is_ext;
    struct VNode { uint32_t op; int a, b; bool q; uint32_t imm0; int32_t imm1; };
    std::vector<VNode> nodes;
    std::map<std::vector<int64_t>, int> memo;
    std::map<const Node*, int> result_of;
    auto emit = [&](uint32_t op, bool q, int a = -1, int b = -1, uint32_t imm0 = 0, int32_t imm1 = 0) { nodes.push_back({op, a, b, q, imm0, imm1}); return (int)nodes.size() - 1; };
    auto const_slot = [&](std::initializer_list<uint64_t> words) { const uint32_t off = (uint32_t)prog.consts.size(); for (auto w : words) prog.consts.push_back(w); return off; };
    // iterative post-order (DAGs can be deep)
    std::vector<std::pair<const Node*, bool>> stack{{root_expr.get(), false}};
    while (!stack.empty()) {
        auto [e, ready] = stack.back();
        stack.pop_back();
        if (result_of.count(e)) continue;
        const Node* ka = e->a.get();
        const Node* kb = e->b.get();
        if (!ready && (ka || kb)) {
            stack.push_back({e, true});
            if (kb && !result_of.count(kb)) stack.push_back({kb, false});
            if (ka && !result_of.count(ka)) stack.push_back({ka, false});
            continue;
        }
        int a = ka ? result_of.at(ka) : -1, b = kb ? result_of.at(kb) : -1;
        std::vector<int64_t> key{(int64_t)e->kind};
        switch (e->kind) {
        case K_ADD: case K_MUL: key.push_back(std::min(a, b)); key.push_back(std::max(a, b)); break;
        case K_NEG: key.push_back(a); break;
        case K_DIV: key.push_back(a); key.push_back(b); break;
        case K_POW: key.push_back(a); key.push_back(e->idx); break;
        case K_CONST_P: key.push_back((int64_t)e->v[0]); break;
        case K_CONST_Q: key.push_back((int64_t)e->v[0]); key.push_back((int64_t)e->v[1]); key.push_back((int64_t)e->v[2]); break;
        case K_TRACE: key.push_back(e->idx); key.push_back(e->off); break;
        case K_PERIODIC: key.push_back(e->idx); for (auto c : e->coeffs) key.push_back((int64_t)c); break;
        case K_CHALLENGE: case K_HINT: key.push_back(e->idx); break;
        default: break;
        }
        auto hit = memo.find(key);
        if (hit != memo.end()) { result_of[e] = hit->second; continue; }
        int v = -1;
        switch (e->kind) {
        case K_X: v = emit(OP_X_P, false); break;
        case K_CONST_P: v = emit(OP_CONST_P, false, -1, -1, const_slot({gl::to_mont(e->v[0])})); break;
        case K_CONST_Q: v = emit(OP_CONST_Q, true, -1, -1, const_slot({gl::to_mont(e->v[0]), gl::to_mont(e->v[1]), gl::to_mont(e->v[2])})); break;
        case K_CHALLENGE: case K_HINT: {
            auto& table = e->kind == K_CHALLENGE ? prog.challenge_slots : prog.hint_slots;
            if (!table.count(e->idx)) table[e->idx] = fq_is_ext ? const_slot({0, 0, 0}) : const_slot({0});
            v = emit(fq_is_ext ? OP_CONST_Q : OP_CONST_P, fq_is_ext, -1, -1, table[e->idx]);
        } break;
        case K_TRACE:
            if (e->idx < num_base_columns) v = emit(OP_TRACE_P, false, -1, -1, e->idx, e->off);
            else if (fq_is_ext) v = emit(OP_TRACE_Q, true, -1, -1, e->idx - num_base_columns, e->off);
            else v = emit(OP_TRACE_P, false, -1, -1, e->idx, e->off);
            break;
        case K_PERIODIC: {
            uint32_t pid = (uint32_t)prog.periodic.size();
            for (uint32_t j = 0; j < prog.periodic.size(); j++) if (prog.periodic[j].first == e->coeffs && prog.periodic[j].second == e->idx) pid = j;
            if (pid == prog.periodic.size()) prog.periodic.push_back({e->coeffs, e->idx});
            v = emit(OP_PERIODIC_P, false, -1, -1, pid);
        } break;
        case K_NEG: v = emit(nodes[a].q ? OP_NEG_Q : OP_NEG_P, nodes[a].q, a); break;
        case K_ADD: case K_MUL: case K_DIV: {
            if (e->kind == K_DIV) {                       // x / y = x * y^-1, 0^-1 = 0 (eval_cpu.rs:440-442)
                std::vector<int64_t> ik{-1, b};
                auto ih = memo.find(ik);
                if (ih == memo.end()) ih = memo.emplace(ik, emit(nodes[b].q ? OP_INV_Q : OP_INV_P, nodes[b].q, b)).first;
                b = ih->second;
            }
            const uint32_t base = e->kind == K_ADD ? OP_ADD_PP : OP_MUL_PP;
            if (!nodes[a].q && !nodes[b].q) v = emit(base, false, a, b);
            else if (nodes[a].q && nodes[b].q) v = emit(base + 1, true, a, b);
            else { if (!nodes[a].q) std::swap(a, b); v = emit(base + 2, true, a, b); }
        } break;
        case K_POW: v = emit(nodes[a].q ? OP_POW_Q : OP_POW_P, nodes[a].q, a, -1, e->idx); break;
        }
        memo[key] = v;
        result_of[e] = v;
    }
    int root = result_of.at(root_expr.get());
    if (fq_is_ext && !nodes[root].q) root = emit(OP_EMBED, true, root);     // the result is always Fq (eval_cpu.rs:262-275)
    // ---- register allocation: linear scan over the (already topological) node list
    std::vector<int> last_use(nodes.size(), -1);
    for (int k = 0; k < (int)nodes.size(); k++) { if (nodes[k].a >= 0) last_use[nodes[k].a] = k; if (nodes[k].b >= 0) last_use[nodes[k].b] = k; }
    last_use[root] = (int)nodes.size();
    std::vector<uint32_t> free_p, free_q, reg(nodes.size(), 0);
    unsigned next_p = 0, next_q = 0;
    for (int k = 0; k < (int)nodes.size(); k++) {
        const VNode& n = nodes[k];
        for (int opnd : {n.a, n.b == n.a ? -1 : n.b})
            if (opnd >= 0 && last_use[opnd] == k) (nodes[opnd].q ? free_q : free_p).push_back(reg[opnd]);
        auto& fr = n.q ? free_q : free_p;
        uint32_t r;
        if (!fr.empty()) { r = fr.back(); fr.pop_back(); } else r = n.q ? next_q++ : next_p++;
        reg[k] = r;
        switch (n.op) {
        case OP_X_P: prog.instrs.push_back({n.op, r, 0, 0}); break;
        case OP_CONST_P: case OP_CONST_Q: case OP_PERIODIC_P: case OP_PERIODIC_Q: prog.instrs.push_back({n.op, r, n.imm0, 0}); break;
        case OP_TRACE_P: case OP_TRACE_Q: prog.instrs.push_back({n.op, r, n.imm0, (uint32_t)n.imm1}); break;
        case OP_POW_P: case OP_POW_Q: prog.instrs.push_back({n.op, r, reg[n.a], n.imm0}); break;
        default: prog.instrs.push_back({n.op, r, reg[n.a], n.b >= 0 ? reg[n.b] : 0});
        }
        if (last_use[k] < 0) fr.push_back(r);
    }
    prog.instrs.push_back({fq_is_ext ? (uint32_t)OP_STORE_Q : (uint32_t)OP_STORE_P, 0, reg[root], 0});
    prog.max_p = next_p; prog.max_q = next_q;
    if (prog.max_p > 256 || prog.max_q > 128) throw std::invalid_argument("constraint program needs too many registers (limits 256 Fp / 128 Fq)");
    return prog;
}

// `Constraint::degree` (src/constraints.rs:32-41, 152-158, 407-455): an upper bound (numerator, denominator) on the degree in X,
// with the reference's own (loose) arithmetic.
inline std::pair<size_t, size_t> degree(const E& e, size_t trace_degree) {
    switch (e->kind) {
    case K_CONST_P: case K_CONST_Q: case K_CHALLENGE: case K_HINT: return {0, 0};
    case K_TRACE: return {trace_degree, 0};
    case K_X: return {1, 0};
    case K_PERIODIC: return {(e->coeffs.size() - 1) * ((trace_degree + 1) / e->idx), 0};      // PeriodicColumn::degree (:135-141)
    case K_NEG: return degree(e->a, trace_degree);
    case K_POW: { auto d = degree(e->a, trace_degree); return {d.first * e->idx, d.second * e->idx}; }
    default: break;
    }
    const auto a = degree(e->a, trace_degree), b = degree(e->b, trace_degree);
    if (e->kind == K_ADD) return {std::max(a.first + b.second, b.first + a.second), a.second + b.second};
    if (e->kind == K_MUL) return {a.first + b.first, a.second + b.second};
    return {a.first + b.second, a.second + b.first};                                           // K_DIV
}
inline size_t ceil_power_of_two(size_t v) {                                                    // src/utils.rs:76-82
    if (v == 0) return 1;
    if ((v & (v - 1)) == 0) return v;
    size_t r = 1; while (r <= v) r <<= 1; return r;
}
// `Constraint::blowup_factor` (src/constraints.rs:162-166, 340-347) -- note the division by trace_len - 1
inline size_t constraint_blowup_factor(const E& c, size_t trace_len) {
    const auto d = degree(c, trace_len - 1);
    return ceil_power_of_two(d.first > d.second ? d.first - d.second : 0) / (trace_len - 1);
}
// `AirConfig::composition_constraint` (src/air.rs:50-82): sum_i c_i (X^adj_i alpha_i + beta_i) with
// adj_i = (trace_len ce_blowup - 1) - (deg num_i - deg den_i); CompositionCoeff(i) = (Challenge(2 i), Challenge(2 i + 1)).
struct Composition { E expr; unsigned ce_blowup_factor; unsigned num_coeffs; };
inline Composition composition_constraint(size_t trace_len, const std::vector<E>& constraints) {
    size_t ce = 0;
    for (auto& c : constraints) ce = std::max(ce, constraint_blowup_factor(c, trace_len));
    // every constraint of degree < trace_len / 2: trace_len * 0 - 1 underflows (the reference panics on the subtraction, the Python
    // mirror asserts) -- an error here too, not a wrapped degree
    if (ce == 0) throw std::invalid_argument("composition_constraint: ce_blowup_factor is 0 (every constraint has degree < trace_len / 2)");
    const size_t composition_degree = trace_len * ce - 1;
    E comp;
    for (size_t i = 0; i < constraints.size(); i++) {
        const auto d = degree(constraints[i], trace_len - 1);
        const size_t ev = d.first > d.second ? d.first - d.second : 0;
        if (ev > composition_degree) throw std::invalid_argument("constraint degree exceeds the composition degree");
        if (composition_degree - ev > 0xFFFFFFFFull) throw std::invalid_argument("composition_constraint: degree adjustment does not fit an exponent");
        E term = constraints[i] * (pow(X(), (uint32_t)(composition_degree - ev)) * Challenge((uint32_t)(2 * i)) + Challenge((uint32_t)(2 * i + 1)));
        comp = comp ? comp + term : term;
    }
    return {comp, (unsigned)ce, (unsigned)(2 * constraints.size())};
}

// The verifier's side of the same DAG (src/verifier.rs:106-116: composition_constraint.graph_eval at the out-of-domain
// point with the opened trace values): one scalar per node, host arithmetic, Fq = Fp AIRs.  All values canonical integers.
// `trace_at(column, offset)` returns T_column(x g^offset).  Twenty lines of bookkeeping over gl::, so the C++ example can
// check what the device produced against the relation the verifier enforces.
template <class TraceAt>
inline uint64_t eval_at_point(const E& root_expr, uint64_t x, size_t trace_len, TraceAt trace_at, const std::vector<uint64_t>& challenges, const std::vector<uint64_t>& hints) {
    std::map<const Node*, uint64_t> val;
    std::vector<std::pair<const Node*, bool>> stack{{root_expr.get(), false}};
    while (!stack.empty()) {
        auto [e, ready] = stack.back();
        stack.pop_back();
        if (val.count(e)) continue;
        const Node* ka = e->a.get();
        const Node* kb = e->b.get();
        if (!ready && (ka || kb)) {
            stack.push_back({e, true});
            if (kb && !val.count(kb)) stack.push_back({kb, false});
            if (ka && !val.count(ka)) stack.push_back({ka, false});
            continue;
        }
        const uint64_t a = ka ? val.at(ka) : 0, b = kb ? val.at(kb) : 0;
        uint64_t v = 0;
        switch (e->kind) {
        case K_X: v = x; break;
        case K_CONST_P: v = e->v[0]; break;
        case K_CHALLENGE: v = challenges.at(e->idx); break;
        case K_HINT: v = hints.at(e->idx); break;
        case K_TRACE: v = trace_at(e->idx, e->off); break;
        case K_PERIODIC: { const uint64_t y = gl::pow(x, trace_len / e->idx); for (size_t i = e->coeffs.size(); i-- > 0;) v = gl::add(gl::mul(v, y), e->coeffs[i]); break; }
        case K_NEG: v = gl::neg(a); break;
        case K_ADD: v = gl::add(a, b); break;
        case K_MUL: v = gl::mul(a, b); break;
        case K_DIV: v = gl::mul(a, gl::inv(b)); break;
        case K_POW: v = gl::pow(a, e->idx); break;
        default: throw std::invalid_argument("eval_at_point: extension-field leaf in an Fq = Fp expression");
        }
        val[e] = v;
    }
    return val.at(root_expr.get());
}

// eval_periodic_column (src/eval_cpu.rs:233-256): evaluations of the column's polynomial on
// coset(interval_size * blowup, offset^(trace_len / interval_size))
inline GpuVec<Fp> periodic_lde(Planner& pl, const std::vector<uint64_t>& coeffs, uint32_t interval, uint64_t domain_offset, size_t trace_len, unsigned lde_step) {
    const size_t size = (size_t)interval * lde_step;
    std::vector<uint64_t> a(size, 0);
    for (size_t i = 0; i < coeffs.size(); i++) a[i] = gl::to_mont(coeffs[i]);
    GpuVec<Fp> v(pl, a);
    GpuFft<Fp> f(pl, Radix2EvaluationDomain(size, gl::pow(domain_offset, trace_len / interval)));
    f.encode(v);
    f.execute();
    return v;
}

// eval_cpu::eval(expr, challenges, hints, lde_step, domain_offset, x_lde, base, ext) -> n elements of Fq.
// challenges / hints: Montgomery words, Fq::words per element.  Fq = Fq3 (fq_is_ext) or Fp.
template <class Fq>
inline GpuVec<Fq> eval(const Program& prog, Planner& pl, const std::vector<uint64_t>& challenges, const std::vector<uint64_t>& hints,
                       unsigned lde_step, uint64_t domain_offset, size_t n, const std::vector<const GpuVec<Fp>*>& base_cols,
                       const std::vector<const GpuVec<Fq3>*>& ext_cols = {}, bool bit_reversed = false) {
    if (prog.fq_is_ext != (Fq::words == 3)) throw std::invalid_argument("program was compiled for the other Fq");
    std::vector<uint64_t> consts = prog.consts;
    auto fill = [&](const std::map<uint32_t, uint32_t>& table, const std::vector<uint64_t>& vals) {
        for (auto& kv : table) {
            if ((size_t)(kv.first + 1) * Fq::words > vals.size()) throw std::invalid_argument("missing challenge / hint value");
            for (unsigned w = 0; w < Fq::words; w++) consts[kv.second + w] = vals[(size_t)kv.first * Fq::words + w];
        }
    };
    fill(prog.challenge_slots, challenges);
    fill(prog.hint_slots, hints);
    std::vector<GpuVec<Fp>> per;
    for (auto& p : prog.periodic) per.push_back(periodic_lde(pl, p.first, p.second, domain_offset, n / lde_step, lde_step));
    std::vector<const void*> bp, ep, pp;
    std::vector<unsigned> plen;
    for (auto c : base_cols) bp.push_back(c->ptr());
    for (auto c : ext_cols) ep.push_back(c->ptr());
    for (auto& p : per) { pp.push_back(p.ptr()); plen.push_back((unsigned)p.len()); }
    GpuVec<Fq> out(pl, n);
    unsigned log_n = 0; while (((size_t)1 << log_n) < n) log_n++;
    const uint64_t off = gl::to_mont(domain_offset);
    // bit_reversed: the columns (their first n entries) and the result are in the committed, bit-reversed layout --
    // instead of the reference's bit_reverse_ce_trace round trip (src/prover.rs:88-91, 126-129)
    check(ms_eval_program_ex(pl.ctx(), (const uint32_t*)prog.instrs.data(), (unsigned)prog.instrs.size(), consts.empty() ? nullptr : consts.data(), (unsigned)consts.size(),
                             log_n, lde_step, &off, nullptr, bp.empty() ? nullptr : bp.data(), (unsigned)bp.size(), ep.empty() ? nullptr : ep.data(), (unsigned)ep.size(),
                             pp.empty() ? nullptr : pp.data(), plen.empty() ? nullptr : plen.data(), (unsigned)pp.size(), Fq::id, out.ptr(),
                             bit_reversed ? MS_EVAL_BIT_REVERSED : 0u));
    pl.sync();
    return out;
}

}  // namespace expr
}  // namespace ms
