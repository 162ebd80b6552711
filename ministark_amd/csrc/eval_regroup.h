// Host-side rewriting of a constraint program, step 4 (after eval_opt.h): SUMS OF PRODUCTS WITH ONE REDUCTION PER SUM.
//
// A composition constraint is sum_i c_i(trace) * (x^adj_i * alpha_i + beta_i) / zerofier_i(x)  (src/air.rs:50-82): every term is a
// product of  (u) wave-uniform values -- constants, challenges, hints --,  (x) values of x alone -- inverse-denominator tables, x^e,
// periodic columns --  and  (t) trace-dependent values.  The specialised kernel is bound by the vector ALU's ISSUE rate (DESIGN.md
// 9.2: 51 Montgomery products of 18 instructions and 68 modular additions of 6-7 per point for the reference's fib AIR), and over
// the 252-bit field a product is ~250 instructions of which the reduction is ~150.  This pass expands the program's result into
// monomials  coefficient(u) * X-part * T-part,  groups them by X-part, and emits
//
//     for every X-part g:   D_g = reduce( sum_k  C_(g,k) * T_(g,k) )      C folded on the HOST from the uniform values (exact),
//                                                                        T = a trace value or a product of trace values
//     result = reduce( sum_g  D_g * X_g )
//
// with the sums accumulated UNREDUCED (eval_kernels.h Acc6 / Acc19: six multiply-adds per term over Goldilocks, the 81 digit
// products over the 252-bit field) and one Montgomery reduction per sum.  Field arithmetic is exact and every value canonical, so
// the output words are those of the original program (tests: every parity test of the evaluator runs through this pass; the
// constraint fuzzers compare it with the C oracle on random programs).  P-typed programs only (Fq = Fp); anything the pass does not
// understand, or that would not get cheaper, is left alone.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>
#include "eval_kernels.h"
#include "eval_opt.h"

namespace mseval {

struct Regrouped {
    bool active = false;
    std::vector<Instr> prog;
    unsigned maxp = 0, maxq = 0;
    unsigned old_cost = 0, new_cost = 0;      // rough vector-instruction counts per point, before / after
};

// host arithmetic on the uniform values (Montgomery words, as the kernels compute)
struct HostGL {
    typedef uint64_t T;
    static T load(const uint64_t* p) { return p[0]; }
    static T zero() { return 0; }
    static T one() { return gl::ONE_MONT; }
    static T add(T a, T b) { return gl::add(a, b); }
    static T neg(T a) { return gl::neg(a); }
    static T mul(T a, T b) { return gl::mont_mul(a, b); }
    static T inv(T a) { return gl::mont_inv(a); }
    static T powu(T a, uint32_t e) { return gl::mont_pow(a, e); }
    static bool is_zero(T a) { return a == 0; }
    // what OP_ACC_MACC reads: 22 / 22 / 20-bit limbs as y0 | y1 << 32, y2
    static void limbs(T c, std::vector<uint64_t>& out) {
        out.push_back((c & 0x3FFFFFull) | (((c >> 22) & 0x3FFFFFull) << 32));
        out.push_back(c >> 44);
    }
    static void words(T c, std::vector<uint64_t>& out) { out.push_back(c); }
    static constexpr unsigned COST_MUL = 18, COST_ADD = 7, COST_MACC = 6, COST_MACP = 11, COST_RED = 60, MAX_TERMS = ACC_MAX_TERMS_GL;
    // a sum of up to DIRECT_INNER terms C * T (DIRECT_OUTER terms D * X) is cheaper as products and additions than through an accumulator
    // and its reduction: 18 + 7 per term against 6 (11) + 60 / terms
    static constexpr unsigned DIRECT_INNER = 3, DIRECT_OUTER = 4;
};
struct Host252 {
    typedef f252::E T;
    static T load(const uint64_t* p) { T r; memcpy(r.l, p, 32); return r; }
    static T zero() { return f252::zero(); }
    static T one() { return f252::one(); }
    static T add(const T& a, const T& b) { return f252::add(a, b); }
    static T neg(const T& a) { return f252::neg(a); }
    static T mul(const T& a, const T& b) { return f252::mul(a, b); }
    static T inv(const T& a) { return f252::inv(a); }
    static T powu(const T& a, uint32_t e) { return f252::pow_u64(a, e); }
    static bool is_zero(const T& a) { return f252::is_zero(a); }
    static void limbs(const T& c, std::vector<uint64_t>& out) {             // nine digits, two per word
        uint32_t d[9];
        f252::digits9(c, d);
        for (int k = 0; k < 4; k++) out.push_back((uint64_t)d[2 * k] | ((uint64_t)d[2 * k + 1] << 32));
        out.push_back(d[8]);
    }
    static void words(const T& c, std::vector<uint64_t>& out) { out.insert(out.end(), c.l, c.l + 4); }
    static constexpr unsigned COST_MUL = 250, COST_ADD = 30, COST_MACC = 105, COST_MACP = 125, COST_RED = 150, MAX_TERMS = ACC_MAX_TERMS_252;
    static constexpr unsigned DIRECT_INNER = 0, DIRECT_OUTER = 1;            // a reduction is most of a product here: always accumulate
};

namespace regroup_detail {
enum Cls : unsigned char { U = 0, X = 1, TR = 2 };
struct Mono { int sign; std::vector<int> u, x, t; };
static constexpr size_t MAX_MONOS = 768, MAX_CROSS = 96;
}  // namespace regroup_detail

// prog: the per-point program (validated; after split_periodic / split_inversions).  consts: every constant word the program can name;
// new constants are APPENDED to it.  max_regs: P registers the launch can hold (256).
template <class F>
static inline Regrouped regroup_sums_of_products(const Instr* prog, unsigned ninstr, std::vector<uint64_t>& consts, bool force = false, unsigned max_regs = 256) {
    using namespace regroup_detail;
    typedef typename F::T FT;
    Regrouped R;
    if ((ninstr < 8 && !force) || ninstr > 4096) return R;
    // ---- the program as a DAG (node k = instruction k); give up on anything that is not a plain P-typed instruction
    std::vector<int> na(ninstr, -1), nb(ninstr, -1), defp(256, -1), uses(ninstr, 0);
    std::vector<unsigned char> cls(ninstr, TR);
    int root = -1;
    unsigned nstores = 0;
    for (unsigned k = 0; k < ninstr; k++) {
        const Instr I = prog[k];
        switch (I.op) {
        case OP_CONST_P: cls[k] = U; break;
        case OP_X_P: case OP_XPOW_P: case OP_TABLE_P: case OP_PERIODIC_P: cls[k] = X; break;
        case OP_TRACE_P: cls[k] = TR; break;
        case OP_NEG_P: case OP_INV_P: case OP_POW_P:
            if (I.a >= 256 || defp[I.a] < 0) return R;
            na[k] = defp[I.a]; cls[k] = cls[na[k]];
            if (I.op == OP_POW_P && I.b == 0) cls[k] = U;                    // a^0 = 1
            break;
        case OP_ADD_PP: case OP_MUL_PP:
            if (I.a >= 256 || I.b >= 256 || defp[I.a] < 0 || defp[I.b] < 0) return R;
            na[k] = defp[I.a]; nb[k] = defp[I.b]; cls[k] = std::max(cls[na[k]], cls[nb[k]]);
            break;
        case OP_STORE_P:
            if (I.b != 0 || I.a >= 256 || defp[I.a] < 0) return R;
            root = defp[I.a]; nstores++;
            continue;
        default: return R;                                                   // Q-typed, accumulators, anything unknown
        }
        if (na[k] >= 0) uses[na[k]]++;
        if (nb[k] >= 0) uses[nb[k]]++;
        if (I.dst >= 256) return R;
        defp[I.dst] = (int)k;
    }
    if (nstores != 1 || root < 0 || cls[root] != TR) return R;
    // ---- uniform values on the host
    std::vector<char> u_done(ninstr, 0);
    std::vector<FT> u_val(ninstr);
    auto eval_u = [&](auto&& self, int k) -> FT {
        if (u_done[k]) return u_val[k];
        const Instr I = prog[k];
        FT v = F::zero();
        switch (I.op) {
        case OP_CONST_P: v = F::load(&consts[I.a]); break;
        case OP_NEG_P: v = F::neg(self(self, na[k])); break;
        case OP_INV_P: v = F::inv(self(self, na[k])); break;
        case OP_POW_P: v = I.b == 0 ? F::one() : F::powu(self(self, na[k]), I.b); break;
        case OP_ADD_PP: v = F::add(self(self, na[k]), self(self, nb[k])); break;
        case OP_MUL_PP: v = F::mul(self(self, na[k]), self(self, nb[k])); break;
        default: break;
        }
        u_done[k] = 1; u_val[k] = v;
        return v;
    };
    // ---- expansion into monomials.  A node is OPENED (its sum / product taken apart) only where it is used once -- a shared
    // sub-expression is computed once as it is and enters as an atom --, and a product of two sums is opened only when one side is a
    // single monomial or carries no trace value (so that (a + b)(c + d) over the trace stays two additions and one product).
    bool too_big = false;
    auto atom = [&](int k) { Mono m; m.sign = 1; (cls[k] == U ? m.u : cls[k] == X ? m.x : m.t).push_back(k); return std::vector<Mono>{m}; };
    auto expand = [&](auto&& self, int k, bool is_root) -> std::vector<Mono> {
        if (too_big) return {};
        if (cls[k] == U) return atom(k);
        const Instr I = prog[k];
        const bool open = is_root || uses[k] <= 1;
        if (!open) return atom(k);
        if (I.op == OP_ADD_PP) {
            std::vector<Mono> a = self(self, na[k], false), b = self(self, nb[k], false);
            a.insert(a.end(), b.begin(), b.end());
            if (a.size() > MAX_MONOS) too_big = true;
            return a;
        }
        if (I.op == OP_NEG_P) {
            std::vector<Mono> a = self(self, na[k], false);
            for (auto& m : a) m.sign = -m.sign;
            return a;
        }
        if (I.op == OP_MUL_PP) {
            std::vector<Mono> a = self(self, na[k], false), b = self(self, nb[k], false);
            auto has_t = [](const std::vector<Mono>& v) { for (auto& m : v) if (!m.t.empty()) return true; return false; };
            const bool ok = a.size() * b.size() <= MAX_CROSS && (a.size() == 1 || b.size() == 1 || !has_t(a) || !has_t(b));
            if (!ok) return atom(k);
            std::vector<Mono> out;
            out.reserve(a.size() * b.size());
            for (auto& ma : a) for (auto& mb : b) {
                Mono m;
                m.sign = ma.sign * mb.sign;
                m.u = ma.u; m.u.insert(m.u.end(), mb.u.begin(), mb.u.end());
                m.x = ma.x; m.x.insert(m.x.end(), mb.x.begin(), mb.x.end());
                m.t = ma.t; m.t.insert(m.t.end(), mb.t.begin(), mb.t.end());
                out.push_back(std::move(m));
            }
            if (out.size() > MAX_MONOS) too_big = true;
            return out;
        }
        return atom(k);                                                      // leaves, inverses, powers
    };
    std::vector<Mono> monos = expand(expand, root, true);
    if (too_big || (monos.size() < 2 && !force)) return R;
    // ---- group by X-part, then by T-part; coefficients summed on the host
    typedef std::vector<int> Key;
    std::map<Key, std::map<Key, FT>> groups;
    for (auto& m : monos) {
        std::sort(m.x.begin(), m.x.end());
        std::sort(m.t.begin(), m.t.end());
        FT c = F::one();
        for (int u : m.u) c = F::mul(c, eval_u(eval_u, u));
        if (m.sign < 0) c = F::neg(c);
        auto& g = groups[m.x];
        auto it = g.find(m.t);
        if (it == g.end()) g.emplace(m.t, c); else it->second = F::add(it->second, c);
    }
    for (auto& g : groups)
        for (auto it = g.second.begin(); it != g.second.end();) { if (F::is_zero(it->second)) it = g.second.erase(it); else ++it; }
    for (auto it = groups.begin(); it != groups.end();) { if (it->second.empty()) it = groups.erase(it); else ++it; }
    if (groups.empty()) return R;
    // ---- emit in SSA form on virtual registers
    std::vector<Instr> out;                                                  // operands are virtual registers until the allocation below
    std::vector<int> vreg_of(ninstr, -1);
    int nv = 0;
    auto emit_node = [&](auto&& self, int k) -> int {                       // the original computation of an atom (and what it needs)
        if (vreg_of[k] >= 0) return vreg_of[k];
        Instr I = prog[k];
        if (na[k] >= 0) I.a = (uint32_t)self(self, na[k]);
        if (nb[k] >= 0) I.b = (uint32_t)self(self, nb[k]);
        I.dst = (uint32_t)nv;
        out.push_back(I);
        return vreg_of[k] = nv++;
    };
    int one_reg = -1;
    auto get_one = [&]() {
        if (one_reg < 0) {
            const uint32_t slot = (uint32_t)consts.size();
            F::words(F::one(), consts);
            out.push_back(Instr{OP_CONST_P, (uint32_t)nv, slot, 0});
            one_reg = nv++;
        }
        return one_reg;
    };
    std::map<Key, int> prod_reg;                                             // products of atoms, shared by prefix
    auto product = [&](const Key& atoms) -> int {                           // -1: the empty product
        if (atoms.empty()) return -1;
        auto hit = prod_reg.find(atoms);
        if (hit != prod_reg.end()) return hit->second;
        int acc = emit_node(emit_node, atoms[0]);
        Key pre{atoms[0]};
        for (size_t i = 1; i < atoms.size(); i++) {
            pre.push_back(atoms[i]);
            auto h = prod_reg.find(pre);
            if (h != prod_reg.end()) { acc = h->second; continue; }
            const int r = emit_node(emit_node, atoms[i]);
            out.push_back(Instr{OP_MUL_PP, (uint32_t)nv, (uint32_t)acc, (uint32_t)r});
            acc = nv++;
            prod_reg[pre] = acc;
        }
        prod_reg[atoms] = acc;
        return acc;
    };
    unsigned cost = 0;
    std::vector<std::pair<int, int>> outer;                                  // (D_g, X_g or -1)
    for (auto& g : groups) {
        // a group of ONE term with coefficient 1 needs no accumulator
        int d = -1;
        if (g.second.size() == 1 && !g.second.begin()->first.empty()) {
            const FT c = g.second.begin()->second;
            FT m1 = F::add(c, F::neg(F::one()));
            if (F::is_zero(m1)) d = product(g.second.begin()->first);
        }
        if (d < 0 && g.second.size() <= F::DIRECT_INNER) {                  // few terms: products with the constants and additions
            for (auto& term : g.second) {
                const uint32_t slot = (uint32_t)consts.size();
                F::words(term.second, consts);
                out.push_back(Instr{OP_CONST_P, (uint32_t)nv, slot, 0});
                int v = nv++;
                const int t = product(term.first);
                if (t >= 0) { out.push_back(Instr{OP_MUL_PP, (uint32_t)nv, (uint32_t)t, (uint32_t)v}); v = nv++; }
                if (d < 0) d = v;
                else { out.push_back(Instr{OP_ADD_PP, (uint32_t)nv, (uint32_t)d, (uint32_t)v}); d = nv++; }
            }
        }
        if (d < 0) {
            std::vector<int> partial;
            unsigned in_acc = 0;
            for (auto& term : g.second) {
                if (in_acc == 0) out.push_back(Instr{OP_ACC_ZERO, 0, 0, 0});
                int t = product(term.first);
                if (t < 0) t = get_one();
                const uint32_t slot = (uint32_t)consts.size();
                F::limbs(term.second, consts);
                out.push_back(Instr{OP_ACC_MACC, 0, (uint32_t)t, slot});
                cost += F::COST_MACC;
                if (++in_acc == F::MAX_TERMS) { out.push_back(Instr{OP_ACC_RED, (uint32_t)nv, 0, 0}); partial.push_back(nv++); in_acc = 0; cost += F::COST_RED; }
            }
            if (in_acc) { out.push_back(Instr{OP_ACC_RED, (uint32_t)nv, 0, 0}); partial.push_back(nv++); cost += F::COST_RED; }
            d = partial[0];
            for (size_t i = 1; i < partial.size(); i++) { out.push_back(Instr{OP_ADD_PP, (uint32_t)nv, (uint32_t)d, (uint32_t)partial[i]}); d = nv++; cost += F::COST_ADD; }
        }
        outer.emplace_back(d, product(g.first));
    }
    int result;
    if (outer.size() <= F::DIRECT_OUTER) {
        result = -1;
        for (auto& o : outer) {
            int v = o.first;
            if (o.second >= 0) { out.push_back(Instr{OP_MUL_PP, (uint32_t)nv, (uint32_t)v, (uint32_t)o.second}); v = nv++; }
            if (result < 0) result = v;
            else { out.push_back(Instr{OP_ADD_PP, (uint32_t)nv, (uint32_t)result, (uint32_t)v}); result = nv++; }
        }
    } else {
        std::vector<int> partial;
        unsigned in_acc = 0;
        for (auto& o : outer) {
            if (in_acc == 0) out.push_back(Instr{OP_ACC_ZERO, 0, 0, 0});
            out.push_back(Instr{OP_ACC_MACP, 0, (uint32_t)o.first, (uint32_t)(o.second >= 0 ? o.second : get_one())});
            cost += F::COST_MACP;
            if (++in_acc == F::MAX_TERMS) { out.push_back(Instr{OP_ACC_RED, (uint32_t)nv, 0, 0}); partial.push_back(nv++); in_acc = 0; cost += F::COST_RED; }
        }
        if (in_acc) { out.push_back(Instr{OP_ACC_RED, (uint32_t)nv, 0, 0}); partial.push_back(nv++); cost += F::COST_RED; }
        result = partial[0];
        for (size_t i = 1; i < partial.size(); i++) { out.push_back(Instr{OP_ADD_PP, (uint32_t)nv, (uint32_t)result, (uint32_t)partial[i]}); result = nv++; cost += F::COST_ADD; }
    }
    out.push_back(Instr{OP_STORE_P, 0, (uint32_t)result, 0});
    // ---- is it cheaper?  (vector instructions per point, the currency of this kernel)
    auto op_cost = [](uint32_t op) -> unsigned {
        switch (op) {
        case OP_MUL_PP: return F::COST_MUL;
        case OP_ADD_PP: case OP_NEG_P: return F::COST_ADD;
        case OP_INV_P: return 80 * F::COST_MUL;
        case OP_XPOW_P: case OP_X_P: return 2 * F::COST_MUL;
        default: return 0;
        }
    };
    unsigned old_cost = 0;
    for (unsigned k = 0; k < ninstr; k++) if (cls[k] != U) old_cost += prog[k].op == OP_POW_P ? F::COST_MUL * 2 * (32 - (unsigned)__builtin_clz(prog[k].b | 1)) : op_cost(prog[k].op);
    for (auto& I : out) if (I.op != OP_ACC_MACC && I.op != OP_ACC_MACP && I.op != OP_ACC_RED) cost += I.op == OP_POW_P ? F::COST_MUL * 2 * (32 - (unsigned)__builtin_clz(I.b | 1)) : op_cost(I.op);
    R.old_cost = old_cost; R.new_cost = cost;
    if (cost * 10 > old_cost * 9 && !force) return R;                       // less than 10 % to gain: keep the program as it came (force: the fuzzers)
    // ---- virtual -> physical registers (a register is free again after the last instruction that reads it)
    auto reads = [](const Instr& I, uint32_t* r) -> int {
        switch (I.op) {
        case OP_NEG_P: case OP_INV_P: case OP_POW_P: case OP_STORE_P: case OP_ACC_MACC: r[0] = I.a; return 1;
        case OP_ADD_PP: case OP_MUL_PP: case OP_ACC_MACP: r[0] = I.a; r[1] = I.b; return 2;
        default: return 0;
        }
    };
    auto writes = [](const Instr& I) { return I.op != OP_STORE_P && I.op != OP_ACC_ZERO && I.op != OP_ACC_MACC && I.op != OP_ACC_MACP; };
    std::vector<int> last(nv, -1);
    for (size_t k = 0; k < out.size(); k++) { uint32_t r[2]; const int n = reads(out[k], r); for (int i = 0; i < n; i++) last[r[i]] = (int)k; }
    std::vector<int> phys(nv, -1);
    std::vector<int> free_regs;
    unsigned next = 0, maxp = 0;
    for (size_t k = 0; k < out.size(); k++) {
        Instr& I = out[k];
        uint32_t r[2];
        const int n = reads(I, r);
        const uint32_t va = I.a, vb = I.b;
        if (n >= 1) I.a = (uint32_t)phys[va];
        if (n == 2) I.b = (uint32_t)phys[vb];
        // operands whose last reader this is give their registers back BEFORE the destination is chosen (dst may reuse one)
        for (int i = 0; i < n; i++) { const uint32_t v = i == 0 ? va : vb; if (last[v] == (int)k && phys[v] >= 0 && !(i == 1 && vb == va)) free_regs.push_back(phys[v]); }
        if (writes(I)) {
            const uint32_t v = I.dst;
            int p;
            if (!free_regs.empty()) { std::sort(free_regs.begin(), free_regs.end(), std::greater<int>()); p = free_regs.back(); free_regs.pop_back(); }
            else p = (int)next++;
            if ((unsigned)p >= max_regs) return R;
            phys[v] = p;
            I.dst = (uint32_t)p;
            if ((unsigned)p + 1 > maxp) maxp = (unsigned)p + 1;
            if (last[v] < 0) free_regs.push_back(p);                         // never read (cannot happen for emitted nodes, harmless)
        }
    }
    R.prog.swap(out);
    R.maxp = maxp;
    R.active = true;
    return R;
}


// ---- the same for Goldilocks programs with Fq3 values (Q-typed opcodes): the brainfuck-shaped AIRs (examples/brainfuck/air.rs:26-27) ----
// A monomial is P-typed when all its factors are; inside a Q-typed sum it enters embedded, as ADD_QP / EMBED do.  Coefficients are
// folded on the host in the field they live in; a Q group accumulates in AccQ (eval_kernels.h): 9 / 3 / 3 / 1 multiply-add sextets
// per term for (Q value, Q constant) / (Q, P) / (P, Q) / (P, P), three reductions per sum -- against six Montgomery products and
// fifteen modular additions for ONE extension-field product.
namespace regroup_detail {
struct UV { bool q; uint64_t p; gl::Fq3 v; };
static inline gl::Fq3 uv_q(const UV& a) { return a.q ? a.v : gl::Fq3{a.p, 0, 0}; }
static inline UV uv_mul(const UV& a, const UV& b) {
    if (!a.q && !b.q) return UV{false, gl::mont_mul(a.p, b.p), {}};
    if (a.q && b.q) return UV{true, 0, gl::mont_mul(a.v, b.v)};
    return a.q ? UV{true, 0, gl::mont_mul_fp(a.v, b.p)} : UV{true, 0, gl::mont_mul_fp(b.v, a.p)};
}
static inline UV uv_add(const UV& a, const UV& b) {
    if (!a.q && !b.q) return UV{false, gl::add(a.p, b.p), {}};
    return UV{true, 0, gl::add(uv_q(a), uv_q(b))};
}
static inline UV uv_neg(const UV& a) { return a.q ? UV{true, 0, gl::neg(a.v)} : UV{false, gl::neg(a.p), {}}; }
static inline bool uv_zero(const UV& a) { return a.q ? (a.v.c0 | a.v.c1 | a.v.c2) == 0 : a.p == 0; }
static inline bool uv_is_one(const UV& a) { return a.q ? (a.v.c0 == gl::ONE_MONT && (a.v.c1 | a.v.c2) == 0) : a.p == gl::ONE_MONT; }
}  // namespace regroup_detail

static inline Regrouped regroup_sums_of_products_q(const Instr* prog, unsigned ninstr, std::vector<uint64_t>& consts, bool force = false) {
    using namespace regroup_detail;
    Regrouped R;
    if ((ninstr < 8 && !force) || ninstr > 4096) return R;
    // ---- DAG
    std::vector<int> na(ninstr, -1), nb(ninstr, -1), defp(256, -1), defq(128, -1), uses(ninstr, 0);
    std::vector<unsigned char> cls(ninstr, TR), isq(ninstr, 0);
    int root = -1;
    unsigned nstores = 0;
    bool root_q = false;
    for (unsigned k = 0; k < ninstr; k++) {
        const Instr I = prog[k];
        if (I.op >= OP_ACC_ZERO) return R;
        unsigned opnd[2][2];
        const int nop = (I.op == OP_XPOW_P) ? 0 : op_operands(I, opnd);
        int d[2] = {-1, -1};
        for (int o = 0; o < nop; o++) {
            if (opnd[o][1] >= (opnd[o][0] ? 128u : 256u)) return R;
            d[o] = opnd[o][0] ? defq[opnd[o][1]] : defp[opnd[o][1]];
            if (d[o] < 0) return R;
        }
        if (op_is_store(I.op)) {
            if (I.b != 0 || nop != 1) return R;
            root = d[0]; root_q = I.op == OP_STORE_Q; nstores++;
            continue;
        }
        isq[k] = op_is_q_dst(I.op) ? 1 : 0;
        switch (I.op) {
        case OP_CONST_P: case OP_CONST_Q: cls[k] = U; break;
        case OP_X_P: case OP_XPOW_P: case OP_TABLE_P: case OP_TABLE_Q: case OP_PERIODIC_P: case OP_PERIODIC_Q: cls[k] = X; break;
        case OP_TRACE_P: case OP_TRACE_Q: cls[k] = TR; break;
        default: {
            if (nop == 0) return R;
            unsigned char c = U;
            for (int o = 0; o < nop; o++) c = std::max(c, cls[d[o]]);
            if ((I.op == OP_POW_P || I.op == OP_POW_Q) && I.b == 0) c = U;
            cls[k] = c;
        }
        }
        na[k] = d[0]; nb[k] = d[1];
        if (na[k] >= 0) uses[na[k]]++;
        if (nb[k] >= 0) uses[nb[k]]++;
        if (I.dst >= (isq[k] ? 128u : 256u)) return R;
        (isq[k] ? defq : defp)[I.dst] = (int)k;
    }
    if (nstores != 1 || root < 0 || cls[root] != TR) return R;
    // ---- uniform values on the host
    std::vector<char> u_done(ninstr, 0);
    std::vector<UV> u_val(ninstr);
    auto eval_u = [&](auto&& self, int k) -> UV {
        if (u_done[k]) return u_val[k];
        const Instr I = prog[k];
        UV v{false, 0, {}};
        switch (I.op) {
        case OP_CONST_P: v = UV{false, consts[I.a], {}}; break;
        case OP_CONST_Q: v = UV{true, 0, gl::Fq3{consts[I.a], consts[I.a + 1], consts[I.a + 2]}}; break;
        case OP_NEG_P: case OP_NEG_Q: v = uv_neg(self(self, na[k])); break;
        case OP_ADD_PP: case OP_ADD_QQ: case OP_ADD_QP: v = uv_add(self(self, na[k]), self(self, nb[k])); break;
        case OP_MUL_PP: case OP_MUL_QQ: case OP_MUL_QP: v = uv_mul(self(self, na[k]), self(self, nb[k])); break;
        case OP_INV_P: v = UV{false, gl::mont_inv(self(self, na[k]).p), {}}; break;
        case OP_INV_Q: v = UV{true, 0, gl::mont_inv(uv_q(self(self, na[k])))}; break;
        case OP_POW_P: v = UV{false, I.b == 0 ? gl::ONE_MONT : gl::mont_pow(self(self, na[k]).p, I.b), {}}; break;
        case OP_POW_Q: v = UV{true, 0, I.b == 0 ? gl::Fq3{gl::ONE_MONT, 0, 0} : gl::mont_pow(uv_q(self(self, na[k])), I.b)}; break;
        case OP_EMBED: v = UV{true, 0, uv_q(self(self, na[k]))}; break;
        default: break;
        }
        if (isq[k] && !v.q) v = UV{true, 0, uv_q(v)};
        u_done[k] = 1; u_val[k] = v;
        return v;
    };
    // ---- expansion (same rules as the P-typed pass)
    bool too_big = false;
    auto atom = [&](int k) { Mono m; m.sign = 1; (cls[k] == U ? m.u : cls[k] == X ? m.x : m.t).push_back(k); return std::vector<Mono>{m}; };
    auto expand = [&](auto&& self, int k, bool is_root) -> std::vector<Mono> {
        if (too_big) return {};
        if (cls[k] == U) return atom(k);
        const Instr I = prog[k];
        if (!(is_root || uses[k] <= 1)) return atom(k);
        switch (I.op) {
        case OP_ADD_PP: case OP_ADD_QQ: case OP_ADD_QP: {
            std::vector<Mono> a = self(self, na[k], false), b = self(self, nb[k], false);
            a.insert(a.end(), b.begin(), b.end());
            if (a.size() > MAX_MONOS) too_big = true;
            return a;
        }
        case OP_NEG_P: case OP_NEG_Q: {
            std::vector<Mono> a = self(self, na[k], false);
            for (auto& m : a) m.sign = -m.sign;
            return a;
        }
        case OP_EMBED: return self(self, na[k], false);
        case OP_MUL_PP: case OP_MUL_QQ: case OP_MUL_QP: {
            std::vector<Mono> a = self(self, na[k], false), b = self(self, nb[k], false);
            auto has_t = [](const std::vector<Mono>& v) { for (auto& m : v) if (!m.t.empty()) return true; return false; };
            if (!(a.size() * b.size() <= MAX_CROSS && (a.size() == 1 || b.size() == 1 || !has_t(a) || !has_t(b)))) return atom(k);
            std::vector<Mono> out;
            for (auto& ma : a) for (auto& mb : b) {
                Mono m;
                m.sign = ma.sign * mb.sign;
                m.u = ma.u; m.u.insert(m.u.end(), mb.u.begin(), mb.u.end());
                m.x = ma.x; m.x.insert(m.x.end(), mb.x.begin(), mb.x.end());
                m.t = ma.t; m.t.insert(m.t.end(), mb.t.begin(), mb.t.end());
                out.push_back(std::move(m));
            }
            if (out.size() > MAX_MONOS) too_big = true;
            return out;
        }
        default: return atom(k);
        }
    };
    std::vector<Mono> monos = expand(expand, root, true);
    if (too_big || (monos.size() < 2 && !force)) return R;
    typedef std::vector<int> Key;
    std::map<Key, std::map<Key, UV>> groups;
    for (auto& m : monos) {
        std::sort(m.x.begin(), m.x.end());
        std::sort(m.t.begin(), m.t.end());
        UV c{false, gl::ONE_MONT, {}};
        for (int u : m.u) c = uv_mul(c, eval_u(eval_u, u));
        if (m.sign < 0) c = uv_neg(c);
        auto& g = groups[m.x];
        auto it = g.find(m.t);
        if (it == g.end()) g.emplace(m.t, c); else it->second = uv_add(it->second, c);
    }
    for (auto& g : groups)
        for (auto it = g.second.begin(); it != g.second.end();) { if (uv_zero(it->second)) it = g.second.erase(it); else ++it; }
    for (auto it = groups.begin(); it != groups.end();) { if (it->second.empty()) it = groups.erase(it); else ++it; }
    if (groups.empty()) return R;
    // ---- emission on typed virtual registers
    std::vector<Instr> out;
    std::vector<unsigned char> vq;                                           // type of every virtual register
    std::vector<int> vreg_of(ninstr, -1);
    auto new_v = [&](bool q) { vq.push_back(q ? 1 : 0); return (int)vq.size() - 1; };
    unsigned cost = 0;
    auto cost_of = [](const Instr& I) -> unsigned {
        auto powc = [](uint32_t e, unsigned m) { return m * 2 * (32 - (unsigned)__builtin_clz(e | 1)); };
        switch (I.op) {
        case OP_MUL_PP: return 18; case OP_ADD_PP: case OP_NEG_P: case OP_ADD_QP: return 7;
        case OP_MUL_QQ: return 215; case OP_MUL_QP: return 56; case OP_ADD_QQ: case OP_NEG_Q: return 21;
        case OP_INV_P: return 1400; case OP_INV_Q: return 2200; case OP_POW_P: return powc(I.b, 18); case OP_POW_Q: return powc(I.b, 215);
        case OP_X_P: case OP_XPOW_P: return 36;
        case OP_ACC_MACC: return 6; case OP_ACC_MACP: return 11; case OP_ACC_RED: return 60; case OP_ACCQ_RED: return 180;
        case OP_ACCQ_MACC: { static const unsigned c[4] = {6, 18, 18, 54}; return c[I.dst & 3]; }
        case OP_ACCQ_MACP: return (I.dst & 1) ? 23 : 11;
        default: return 0;
        }
    };
    auto push = [&](const Instr& I) { out.push_back(I); cost += cost_of(I); };
    auto emit_node = [&](auto&& self, int k) -> int {
        if (vreg_of[k] >= 0) return vreg_of[k];
        Instr I = prog[k];
        if (na[k] >= 0) I.a = (uint32_t)self(self, na[k]);
        if (nb[k] >= 0) I.b = (uint32_t)self(self, nb[k]);
        const int v = new_v(isq[k]);
        I.dst = (uint32_t)v;
        push(I);
        return vreg_of[k] = v;
    };
    int one_reg = -1;
    auto get_one = [&]() {
        if (one_reg < 0) { const uint32_t slot = (uint32_t)consts.size(); consts.push_back(gl::ONE_MONT); one_reg = new_v(false); push(Instr{OP_CONST_P, (uint32_t)one_reg, slot, 0}); }
        return one_reg;
    };
    std::map<Key, int> prod_reg;
    // product of atoms: the P atoms by Montgomery products, the Q atoms by extension products, then one mixed product; -1 = empty
    auto product = [&](const Key& atoms) -> int {
        if (atoms.empty()) return -1;
        auto hit = prod_reg.find(atoms);
        if (hit != prod_reg.end()) return hit->second;
        int pp = -1, qq = -1;
        for (int a : atoms) {
            const int r = emit_node(emit_node, a);
            if (vq[r]) { if (qq < 0) qq = r; else { const int v = new_v(true); push(Instr{OP_MUL_QQ, (uint32_t)v, (uint32_t)qq, (uint32_t)r}); qq = v; } }
            else { if (pp < 0) pp = r; else { const int v = new_v(false); push(Instr{OP_MUL_PP, (uint32_t)v, (uint32_t)pp, (uint32_t)r}); pp = v; } }
        }
        int res = qq < 0 ? pp : qq;
        if (qq >= 0 && pp >= 0) { res = new_v(true); push(Instr{OP_MUL_QP, (uint32_t)res, (uint32_t)qq, (uint32_t)pp}); }
        prod_reg[atoms] = res;
        return res;
    };
    auto limbs_p = [&](uint64_t c) { consts.push_back((c & 0x3FFFFFull) | (((c >> 22) & 0x3FFFFFull) << 32)); consts.push_back(c >> 44); };
    auto const_slot = [&](const UV& c) -> uint32_t {                        // what OP_ACC(Q)_MACC reads
        const uint32_t slot = (uint32_t)consts.size();
        if (!c.q) limbs_p(c.p);
        else { limbs_p(c.v.c0); limbs_p(c.v.c1); limbs_p(c.v.c2); limbs_p(gl::dbl(c.v.c1)); limbs_p(gl::dbl(c.v.c2)); }
        return slot;
    };
    struct Outer { int d, x; };
    std::vector<Outer> outer;
    for (auto& g : groups) {
        bool gq = false;
        std::vector<std::pair<int, const UV*>> terms;                        // (T register or -1, coefficient)
        for (auto& term : g.second) {
            const int t = product(term.first);
            terms.emplace_back(t, &term.second);
            if (term.second.q || (t >= 0 && vq[t])) gq = true;
        }
        int d = -1;
        if (terms.size() == 1 && terms[0].first >= 0 && uv_is_one(*terms[0].second)) d = terms[0].first;
        else if (!gq && terms.size() <= 3) {                                // few P terms: products and additions
            for (auto& tm : terms) {
                const uint32_t slot = (uint32_t)consts.size();
                consts.push_back(tm.second->p);
                int v = new_v(false);
                push(Instr{OP_CONST_P, (uint32_t)v, slot, 0});
                if (tm.first >= 0) { const int w = new_v(false); push(Instr{OP_MUL_PP, (uint32_t)w, (uint32_t)tm.first, (uint32_t)v}); v = w; }
                if (d < 0) d = v; else { const int w = new_v(false); push(Instr{OP_ADD_PP, (uint32_t)w, (uint32_t)d, (uint32_t)v}); d = w; }
            }
        } else {
            std::vector<int> partial;
            unsigned in_acc = 0;
            const unsigned lim = gq ? ACC_MAX_TERMS_Q : ACC_MAX_TERMS_GL;
            auto flush = [&]() {
                const int v = new_v(gq);
                push(gq ? Instr{OP_ACCQ_RED, (uint32_t)v, 0, 0} : Instr{OP_ACC_RED, (uint32_t)v, 0, 0});
                partial.push_back(v); in_acc = 0;
            };
            for (auto& tm : terms) {
                if (in_acc == 0) push(gq ? Instr{OP_ACCQ_ZERO, 0, 0, 0} : Instr{OP_ACC_ZERO, 0, 0, 0});
                const int t = tm.first >= 0 ? tm.first : get_one();
                const uint32_t slot = const_slot(*tm.second);
                if (gq) push(Instr{OP_ACCQ_MACC, (uint32_t)((vq[t] ? 1 : 0) | (tm.second->q ? 2 : 0)), (uint32_t)t, slot});
                else push(Instr{OP_ACC_MACC, 0, (uint32_t)t, slot});
                if (++in_acc == lim) flush();
            }
            if (in_acc) flush();
            d = partial[0];
            for (size_t i = 1; i < partial.size(); i++) { const int w = new_v(gq); push(Instr{gq ? OP_ADD_QQ : OP_ADD_PP, (uint32_t)w, (uint32_t)d, (uint32_t)partial[i]}); d = w; }
        }
        outer.push_back(Outer{d, product(g.first)});
    }
    // ---- the outer sum  result = sum_g D_g * X_g
    auto times = [&](int d, int x) -> int {                                 // one product as instructions
        if (x < 0) return d;
        if (!vq[d] && !vq[x]) { const int v = new_v(false); push(Instr{OP_MUL_PP, (uint32_t)v, (uint32_t)d, (uint32_t)x}); return v; }
        const int v = new_v(true);
        if (vq[d] && vq[x]) push(Instr{OP_MUL_QQ, (uint32_t)v, (uint32_t)d, (uint32_t)x});
        else if (vq[d]) push(Instr{OP_MUL_QP, (uint32_t)v, (uint32_t)d, (uint32_t)x});
        else push(Instr{OP_MUL_QP, (uint32_t)v, (uint32_t)x, (uint32_t)d});
        return v;
    };
    int result = -1;
    bool any_q = false;
    for (auto& o : outer) if (vq[o.d] || (o.x >= 0 && vq[o.x])) any_q = true;
    if (outer.size() <= 2 || (!any_q && outer.size() <= 4)) {
        for (auto& o : outer) {
            const int v = times(o.d, o.x);
            if (result < 0) { result = v; continue; }
            const int w = new_v(vq[result] || vq[v]);
            if (vq[result] && vq[v]) push(Instr{OP_ADD_QQ, (uint32_t)w, (uint32_t)result, (uint32_t)v});
            else if (vq[result]) push(Instr{OP_ADD_QP, (uint32_t)w, (uint32_t)result, (uint32_t)v});
            else if (vq[v]) push(Instr{OP_ADD_QP, (uint32_t)w, (uint32_t)v, (uint32_t)result});
            else push(Instr{OP_ADD_PP, (uint32_t)w, (uint32_t)result, (uint32_t)v});
            result = w;
        }
    } else {
        std::vector<int> partial;
        unsigned in_acc = 0;
        const unsigned lim = any_q ? ACC_MAX_TERMS_Q : ACC_MAX_TERMS_GL;
        auto flush = [&]() {
            const int v = new_v(any_q);
            push(any_q ? Instr{OP_ACCQ_RED, (uint32_t)v, 0, 0} : Instr{OP_ACC_RED, (uint32_t)v, 0, 0});
            partial.push_back(v); in_acc = 0;
        };
        UV one{false, gl::ONE_MONT, {}};
        uint32_t one_slot = 0;
        bool have_one_slot = false;
        for (auto& o : outer) {
            if (in_acc == 0) push(any_q ? Instr{OP_ACCQ_ZERO, 0, 0, 0} : Instr{OP_ACC_ZERO, 0, 0, 0});
            int d = o.d, x = o.x;
            if (!any_q) push(Instr{OP_ACC_MACP, 0, (uint32_t)d, (uint32_t)(x >= 0 ? x : get_one())});
            else {
                if (x >= 0 && vq[d] && vq[x]) { d = times(d, x); x = -1; }   // Q x Q (rare: an extension-valued table): reduced product, then added
                if (x < 0) {
                    if (!have_one_slot) { one_slot = const_slot(one); have_one_slot = true; }
                    push(Instr{OP_ACCQ_MACC, (uint32_t)(vq[d] ? 1 : 0), (uint32_t)d, one_slot});
                } else if (vq[d]) push(Instr{OP_ACCQ_MACP, 1, (uint32_t)d, (uint32_t)x});
                else if (vq[x]) push(Instr{OP_ACCQ_MACP, 1, (uint32_t)x, (uint32_t)d});
                else push(Instr{OP_ACCQ_MACP, 0, (uint32_t)d, (uint32_t)x});
            }
            if (++in_acc == lim) flush();
        }
        if (in_acc) flush();
        result = partial[0];
        for (size_t i = 1; i < partial.size(); i++) { const int w = new_v(any_q); push(Instr{any_q ? OP_ADD_QQ : OP_ADD_PP, (uint32_t)w, (uint32_t)result, (uint32_t)partial[i]}); result = w; }
    }
    if (root_q && !vq[result]) { const int w = new_v(true); push(Instr{OP_EMBED, (uint32_t)w, (uint32_t)result, 0}); result = w; }
    if (!root_q && vq[result]) return R;                                    // cannot happen for a validated program
    out.push_back(Instr{root_q ? (uint32_t)OP_STORE_Q : (uint32_t)OP_STORE_P, 0, (uint32_t)result, 0});
    unsigned old_cost = 0;
    for (unsigned k = 0; k < ninstr; k++) if (cls[k] != U) old_cost += cost_of(prog[k]);
    R.old_cost = old_cost; R.new_cost = cost;
    if (cost * 10 > old_cost * 9 && !force) return R;
    // ---- virtual -> physical, one pool per register file
    auto reads = [&](const Instr& I, uint32_t* r) -> int {
        switch (I.op) {
        case OP_ACC_MACC: case OP_ACCQ_MACC: r[0] = I.a; return 1;
        case OP_ACC_MACP: case OP_ACCQ_MACP: r[0] = I.a; r[1] = I.b; return 2;
        case OP_ACC_ZERO: case OP_ACCQ_ZERO: case OP_ACC_RED: case OP_ACCQ_RED: case OP_XPOW_P: return 0;
        default: {
            unsigned opnd[2][2];
            const int n = op_operands(I, opnd);
            for (int i = 0; i < n; i++) r[i] = opnd[i][1];
            return n;
        }
        }
    };
    auto writes = [](const Instr& I) {
        return !op_is_store(I.op) && I.op != OP_ACC_ZERO && I.op != OP_ACC_MACC && I.op != OP_ACC_MACP && I.op != OP_ACCQ_ZERO && I.op != OP_ACCQ_MACC && I.op != OP_ACCQ_MACP;
    };
    const int nv = (int)vq.size();
    std::vector<int> last(nv, -1), phys(nv, -1);
    for (size_t k = 0; k < out.size(); k++) { uint32_t r[2]; const int n = reads(out[k], r); for (int i = 0; i < n; i++) last[r[i]] = (int)k; }
    std::vector<int> free_regs[2];
    unsigned next[2] = {0, 0}, maxr[2] = {0, 0};
    const unsigned cap[2] = {256, 128};
    for (size_t k = 0; k < out.size(); k++) {
        Instr& I = out[k];
        uint32_t r[2];
        const int n = reads(I, r);
        const uint32_t v0 = n >= 1 ? r[0] : 0, v1 = n == 2 ? r[1] : 0;
        if (n >= 1) I.a = (uint32_t)phys[v0];
        if (n == 2) I.b = (uint32_t)phys[v1];
        for (int i = 0; i < n; i++) { const uint32_t v = i == 0 ? v0 : v1; if (last[v] == (int)k && !(i == 1 && v1 == v0)) free_regs[vq[v]].push_back(phys[v]); }
        if (writes(I)) {
            const uint32_t v = I.dst;
            const int f = vq[v];
            int p;
            if (!free_regs[f].empty()) { std::sort(free_regs[f].begin(), free_regs[f].end(), std::greater<int>()); p = free_regs[f].back(); free_regs[f].pop_back(); }
            else p = (int)next[f]++;
            if ((unsigned)p >= cap[f]) return R;
            phys[v] = p;
            I.dst = (uint32_t)p;
            maxr[f] = std::max(maxr[f], (unsigned)p + 1);
            if (last[v] < 0) free_regs[f].push_back(p);
        }
    }
    R.prog.swap(out);
    R.maxp = maxr[0]; R.maxq = maxr[1];
    R.active = true;
    return R;
}

}  // namespace mseval
