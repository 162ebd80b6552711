// Host-side rewriting of a constraint program, step 3b (after eval_opt.h split_inversions): ONE inverse table for all the
// denominators  X - a  whose roots differ by a power of the trace generator.
//
// The boundary and terminal constraints of an AIR divide by (X - g^r) for rows r of the trace (src/constraints.rs; the reference's fib
// AIR: (X - 1) and (X - g^-1), examples/fib/main.rs:73-140), and g = w^lde_step is a power of the evaluation domain's generator.  With
// rho = a_b / a_t:
//
//      X - a_t  =  (rho X - a_b) / rho ,        so        1 / (x_i - a_t)  =  rho / (rho x_i - a_b) ,
//
// and when rho = g^k the point rho x_i is the domain point k trace rows further on: the inverse of table t at point i is rho times the
// inverse of table b at row offset k -- the rotation OP_TRACE_* already reads columns with (eval_kernels.h ev_row, any layout).  Table t
// is never computed nor inverted (a batch inversion is 7.5 products per element over Goldilocks and a Fermat chain of the 252-bit
// field's per 256); the per-point program reads table b twice and multiplies by a constant, which step 4 (eval_regroup.h) folds into
// the coefficients it builds on the host.  Zeros agree: x_i = a_t makes both denominators zero and 0^-1 = 0 on both sides.
// Exact field identities on canonical values: the output words do not change (tests: the evaluator's parity tests and the fuzzers run
// with the step on and, MS_EVAL_SHARE_TABLES=0, off).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include "eval_opt.h"
#include "eval_regroup.h"      // HostGL / Host252

namespace mseval {

static constexpr int SHIFT_MAX_ROWS = 16;      // |k| searched

// The tables of S whose denominator is X + (a value built from constants alone): has_c[t], c[t] (Montgomery), and the instruction that stores it.
template <class F>
static inline void tables_x_plus_c(const InvSplit& S, unsigned first_table, unsigned elem_words_p, const std::vector<uint64_t>& consts,
                                   std::vector<char>& has_c, std::vector<typename F::T>& c, std::vector<int>& store_of) {
    typedef typename F::T FT;
    const unsigned nd = (unsigned)S.denom.size(), ntab = (unsigned)S.table_words.size();
    std::vector<int> na(nd, -1), nb(nd, -1), defp(256, -1), node_of(ntab, -1);
    has_c.assign(ntab, 0); c.assign(ntab, F::zero()); store_of.assign(ntab, -1);
    for (unsigned k = 0; k < nd; k++) {
        const Instr I = S.denom[k];
        unsigned opnd[2][2];
        const int nop = I.op == OP_XPOW_P ? 0 : op_operands(I, opnd);
        if (nop >= 1 && opnd[0][0] == 0 && opnd[0][1] < 256) na[k] = defp[opnd[0][1]];
        if (nop >= 2 && opnd[1][0] == 0 && opnd[1][1] < 256) nb[k] = defp[opnd[1][1]];
        if (I.op == OP_STORE_P) {
            if (I.b >= first_table + 1 && I.b - 1 - first_table < ntab) { store_of[I.b - 1 - first_table] = (int)k; node_of[I.b - 1 - first_table] = na[k]; }
            continue;
        }
        if (op_is_store(I.op) || op_is_q_dst(I.op)) continue;
        if (I.dst < 256) defp[I.dst] = (int)k;
    }
    auto uniform = [&](auto&& self, int k, FT& v) -> bool {          // a value built from constants alone
        if (k < 0) return false;
        const Instr I = S.denom[k];
        FT a, b;
        switch (I.op) {
        case OP_CONST_P: v = F::load(&consts[I.a]); return true;
        case OP_NEG_P: if (!self(self, na[k], a)) return false; v = F::neg(a); return true;
        case OP_ADD_PP: if (!self(self, na[k], a) || !self(self, nb[k], b)) return false; v = F::add(a, b); return true;
        case OP_MUL_PP: if (!self(self, na[k], a) || !self(self, nb[k], b)) return false; v = F::mul(a, b); return true;
        default: return false;
        }
    };
    for (unsigned t = 0; t < ntab; t++) {
        if (S.table_words[t] != elem_words_p || store_of[t] < 0 || node_of[t] < 0) continue;
        const int k = node_of[t];
        if (S.denom[k].op != OP_ADD_PP || na[k] < 0 || nb[k] < 0) continue;
        FT v;
        if (S.denom[na[k]].op == OP_X_P && uniform(uniform, nb[k], v)) { c[t] = v; has_c[t] = 1; }
        else if (S.denom[nb[k]].op == OP_X_P && uniform(uniform, na[k], v)) { c[t] = v; has_c[t] = 1; }
    }
}

// S: what split_inversions returned (active).  first_table: periodic slot of table 0.  g: the trace generator w_n^lde_step (Montgomery).
// maxp: P registers in use (one more is taken).  Returns the number of tables removed.
template <class F>
static inline unsigned share_shifted_tables(InvSplit& S, unsigned first_table, unsigned elem_words_p, std::vector<uint64_t>& consts,
                                            const typename F::T& g, unsigned& maxp, bool debug) {
    typedef typename F::T FT;
    if (!S.active || S.table_words.size() < 2 || maxp >= 256) return 0;
    const unsigned nd = (unsigned)S.denom.size(), ntab = (unsigned)S.table_words.size();
    // ---- root a_t = -c of every table whose denominator is X + c
    std::vector<char> has_root;
    std::vector<FT> root;
    std::vector<int> store_of;
    tables_x_plus_c<F>(S, first_table, elem_words_p, consts, has_root, root, store_of);
    for (unsigned t = 0; t < ntab; t++) { root[t] = F::neg(root[t]); if (F::is_zero(root[t])) has_root[t] = 0; }
    // ---- g^k, k = -SHIFT_MAX_ROWS .. SHIFT_MAX_ROWS
    std::vector<FT> gp(2 * SHIFT_MAX_ROWS + 1);
    gp[SHIFT_MAX_ROWS] = F::one();
    const FT gi = F::inv(g);
    for (int k = 1; k <= SHIFT_MAX_ROWS; k++) { gp[SHIFT_MAX_ROWS + k] = F::mul(gp[SHIFT_MAX_ROWS + k - 1], g); gp[SHIFT_MAX_ROWS - k] = F::mul(gp[SHIFT_MAX_ROWS - k + 1], gi); }
    auto equal = [](const FT& a, const FT& b) { return F::is_zero(F::add(a, F::neg(b))); };
    // ---- every table with a root is either a base or a rotation of an earlier base
    struct Map { int base; int k; FT rho; };
    std::vector<Map> map(ntab, Map{-1, 0, F::zero()});
    std::vector<unsigned> bases;
    unsigned removed = 0;
    for (unsigned t = 0; t < ntab; t++) {
        if (!has_root[t]) continue;
        bool found = false;
        for (unsigned b : bases) {
            for (int k = -SHIFT_MAX_ROWS; k <= SHIFT_MAX_ROWS && !found; k++) {
                if (k == 0) continue;
                if (equal(root[b], F::mul(root[t], gp[SHIFT_MAX_ROWS + k]))) { map[t] = Map{(int)b, k, gp[SHIFT_MAX_ROWS + k]}; found = true; }
            }
            if (found) break;
            if (equal(root[b], root[t])) { map[t] = Map{(int)b, 0, F::one()}; found = true; break; }      // the same denominator twice
        }
        if (found) removed++; else bases.push_back(t);
    }
    if (!removed) return 0;
    // ---- new table numbers; the denominators' program loses the stores of the removed tables (what fed them is dead code there)
    std::vector<int> renum(ntab, -1);
    std::vector<unsigned> words;
    for (unsigned t = 0; t < ntab; t++) if (map[t].base < 0) { renum[t] = (int)words.size(); words.push_back(S.table_words[t]); }
    std::vector<Instr> denom;
    for (unsigned k = 0; k < nd; k++) {
        Instr I = S.denom[k];
        if (op_is_store(I.op) && I.b >= first_table + 1 && I.b - 1 - first_table < ntab) {
            const unsigned t = I.b - 1 - first_table;
            if (renum[t] < 0) continue;
            I.b = first_table + (unsigned)renum[t] + 1;
        }
        denom.push_back(I);
    }
    const uint32_t tmp = maxp;
    std::vector<Instr> main;
    for (const Instr& I0 : S.main) {
        Instr I = I0;
        if ((I.op == OP_TABLE_P || I.op == OP_TABLE_Q) && I.a >= first_table && I.a - first_table < ntab) {
            const unsigned t = I.a - first_table;
            if (map[t].base < 0) { I.a = first_table + (unsigned)renum[t]; main.push_back(I); continue; }
            const Map& m = map[t];
            main.push_back(Instr{OP_TABLE_P, I.dst, first_table + (unsigned)renum[m.base], (uint32_t)m.k});
            if (m.k != 0) {
                const uint32_t slot = (uint32_t)consts.size();
                F::words(m.rho, consts);
                main.push_back(Instr{OP_CONST_P, tmp, slot, 0});
                main.push_back(Instr{OP_MUL_PP, I.dst, I.dst, tmp});
            }
            if (debug) fprintf(stderr, "shared tables: table %u = table %u at row offset %d times a constant\n", t, (unsigned)m.base, m.k);
            continue;
        }
        main.push_back(I);
    }
    S.denom.swap(denom);
    S.main.swap(main);
    S.table_words.swap(words);
    maxp = tmp + 1;
    return removed;
}

}  // namespace mseval
