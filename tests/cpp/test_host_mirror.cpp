// GPU parity test of the C++ host mirror (ministark.hpp) against the C oracle, shapes of
// gpu/tests/shaders.rs:17-117.  Built and run by tests/test_cpp_mirror.py (-m gpu).
#include <cstdio>
#include <cstring>
#include <vector>
#include <functional>
#include "../../ministark_amd/csrc/host/ministark.hpp"
#include "../../ministark_amd/csrc/host/stages.hpp"
#include "../../ministark_amd/csrc/host/expr.hpp"
#include "../../ministark_amd/csrc/host/prover.hpp"

extern "C" {   // the checker: oracle/c/oracle.c
void oracle_ntt(uint64_t* a, unsigned log_n, unsigned V, int inverse, uint64_t offset_canon);
void oracle_lde(const uint64_t* in, uint64_t* out, unsigned log_n, unsigned log_blowup, unsigned V, uint64_t offset_canon, int bit_reversed);
void oracle_sha256_rows(const uint64_t* const* cols, unsigned ncols, unsigned V, size_t nrows, uint8_t* leaves);
void oracle_sha256_merkle(const uint8_t* leaves, size_t n, uint8_t* nodes);
void oracle_sha256(const uint8_t* msg, size_t len, uint8_t* out);
void oracle_fri_fold(const uint64_t* evals, uint64_t* out, unsigned log_n, unsigned V, unsigned ff, const uint64_t* alpha, uint64_t offset_canon);
void oracle_binary(int op, unsigned VL, unsigned VR, size_t n, uint64_t* dst, const uint64_t* lhs, const uint64_t* rhs, size_t shift);
void oracle_binary_const(int op, unsigned VL, unsigned VR, size_t n, uint64_t* dst, const uint64_t* lhs, const uint64_t* c);
void oracle_unary(int op, unsigned V, size_t n, uint64_t* dst, const uint64_t* src, unsigned e);
void oracle_fq3_mul(const uint64_t* a, const uint64_t* b, uint64_t* out);
uint64_t oracle_gl_mul(uint64_t a, uint64_t b);      // Montgomery words
uint64_t oracle_gl_add(uint64_t a, uint64_t b);
uint64_t oracle_gl_sub(uint64_t a, uint64_t b);
uint64_t oracle_gl_inv(uint64_t a);
uint64_t oracle_gl_pow(uint64_t a, uint64_t e);
uint64_t oracle_gl_to_mont(uint64_t c);
uint64_t oracle_gl_root_of_unity(unsigned log_n);
}

static std::vector<uint64_t> rnd(size_t n, uint64_t seed) {
    std::vector<uint64_t> v(n);
    uint64_t s = seed;
    for (auto& x : v) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; x = (z ^ (z >> 31)) % ms::gl::P; }
    return v;
}
#define REQUIRE(c) do { if (!(c)) { printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main() {
    ms::Planner& pl = ms::get_planner();
    // fft_with_64_bit_field / ifft: 2048, 4096, 65536; subgroup and coset(GENERATOR = 7)
    for (unsigned log_n : {11u, 12u, 16u}) for (uint64_t off : {1ull, 7ull}) for (int inv : {0, 1}) {
        auto x = rnd((size_t)1 << log_n, log_n * 10 + off + inv);
        ms::GpuVec<ms::Fp> v(pl, x);
        ms::Radix2EvaluationDomain d((size_t)1 << log_n, off);
        if (inv) { ms::GpuIfft<ms::Fp> f(pl, d); f.encode(v); f.execute(); } else { ms::GpuFft<ms::Fp> f(pl, d); f.encode(v); f.execute(); }
        oracle_ntt(x.data(), log_n, 1, inv, off);
        REQUIRE(v.to_host() == x);
    }
    {   // fft_with_extension_field
        auto x = rnd((size_t)3 << 12, 99);
        ms::GpuVec<ms::Fq3> v(pl, x);
        ms::GpuFft<ms::Fq3> f(pl, ms::Radix2EvaluationDomain(4096, 7)); f.encode(v); f.execute();
        oracle_ntt(x.data(), 12, 3, 0, 7);
        REQUIRE(v.to_host() == x);
    }
    {   // Matrix: interpolate / evaluate round trip, fused LDE, commitment
        const unsigned log_n = 12, log_b = 3, ncols = 5;
        std::vector<std::vector<uint64_t>> cols;
        ms::Matrix<ms::Fp> m;
        for (unsigned c = 0; c < ncols; c++) { cols.push_back(rnd((size_t)1 << log_n, 500 + c)); m.columns.emplace_back(pl, cols.back()); }
        ms::Radix2EvaluationDomain td((size_t)1 << log_n);
        ms::Matrix<ms::Fp> back = m.interpolate(td).evaluate(td);
        for (unsigned c = 0; c < ncols; c++) REQUIRE(back.columns[c].to_host() == cols[c]);
        ms::Matrix<ms::Fp> lde = m.lde(log_b, 7, true);
        std::vector<std::vector<uint64_t>> want(ncols, std::vector<uint64_t>((size_t)1 << (log_n + log_b)));
        std::vector<const uint64_t*> wp;
        for (unsigned c = 0; c < ncols; c++) { oracle_lde(cols[c].data(), want[c].data(), log_n, log_b, 1, 7, 1); REQUIRE(lde.columns[c].to_host() == want[c]); wp.push_back(want[c].data()); }
        const size_t N = (size_t)1 << (log_n + log_b);
        std::vector<uint8_t> leaves(N * 32), nodes(N * 32);
        oracle_sha256_rows(wp.data(), ncols, 1, N, leaves.data());
        oracle_sha256_merkle(leaves.data(), N, nodes.data());
        auto root = ms::MerkleTree::from_matrix(lde).root();
        REQUIRE(memcmp(root.data(), nodes.data() + 32, 32) == 0);
        // bit_reversed_evaluate of the interpolated polynomials = the fused LDE; chunks() of a column
        ms::Matrix<ms::Fp> polys = m.interpolate(td);
        ms::Matrix<ms::Fp> lde2 = polys.bit_reversed_evaluate(ms::Radix2EvaluationDomain(N, 7));
        for (unsigned c = 0; c < ncols; c++) REQUIRE(lde2.columns[c].to_host() == want[c]);
        ms::Matrix<ms::Fp> parts = ms::Matrix<ms::Fp>::from_chunks(m.columns[0], 4);
        for (unsigned c = 0; c < 4; c++) { auto pc = parts.columns[c].to_host(); for (size_t j = 0; j < pc.size(); j++) REQUIRE(pc[j] == cols[0][j * 4 + c]); }
        // sum_columns
        auto sum = m.sum_columns().to_host();
        for (size_t i = 0; i < sum.size(); i++) {
            unsigned __int128 acc = 0;
            for (unsigned c = 0; c < ncols; c++) acc += cols[c][i];
            REQUIRE(sum[i] == (uint64_t)(acc % ms::gl::P));
        }
    }
    {   // error behaviour: wrong column length -> exception (the reference's assert_eq!)
        ms::GpuVec<ms::Fp> v(pl, 1024);
        ms::GpuFft<ms::Fp> f(pl, ms::Radix2EvaluationDomain(2048));
        bool threw = false;
        try { f.encode(v); } catch (const std::invalid_argument&) { threw = true; }
        REQUIRE(threw);
    }
    {   // element-wise stages (gpu/src/stage.rs): same names / argument order, against the C oracle
        const size_t n = 4096;
        auto a = rnd(3 * n, 1), b = rnd(n, 2), c = rnd(3 * n, 3);
        ms::GpuVec<ms::Fq3> A(pl, a), C(pl, c), D(pl, n);
        ms::GpuVec<ms::Fp> B(pl, b);
        std::vector<uint64_t> want(3 * n);
        ms::MulIntoStage<ms::Fq3, ms::Fp>(pl, n).encode(D, A, B, 5);
        oracle_binary(1, 3, 1, n, want.data(), a.data(), b.data(), 5);
        REQUIRE(D.to_host() == want);
        ms::AddAssignStage<ms::Fq3>(pl, n).encode(D, C, -3);
        oracle_binary(0, 3, 3, n, want.data(), want.data(), c.data(), n - 3);
        REQUIRE(D.to_host() == want);
        const std::vector<uint64_t> k{ms::gl::to_mont(12345), ms::gl::to_mont(6), ms::gl::to_mont(7)};
        ms::MulAssignConstStage<ms::Fq3>(pl, n).encode(D, k);
        oracle_binary_const(1, 3, 3, n, want.data(), want.data(), k.data());
        REQUIRE(D.to_host() == want);
        ms::InverseInPlaceStage<ms::Fq3>(pl, n).encode(D);
        oracle_unary(1, 3, n, want.data(), want.data(), 0);
        REQUIRE(D.to_host() == want);
        ms::ExpIntoStage<ms::Fp>(pl, n); ms::GpuVec<ms::Fp> Eo(pl, n);
        ms::ExpIntoStage<ms::Fp>(pl, n).encode(Eo, B, 11);
        std::vector<uint64_t> wb(n);
        oracle_unary(2, 1, n, wb.data(), b.data(), 11);
        REQUIRE(Eo.to_host() == wb);
        bool threw = false;
        try { ms::NegInPlaceStage<ms::Fp>(pl, 1000); } catch (const std::invalid_argument&) { threw = true; }     // stage.rs:55-59
        REQUIRE(threw);
    }
    {   // apply_drp (src/fri.rs:526-567), Fq3 layer folded by 8, and the FRI layer commitment
        const unsigned log_n = 12, ff = 8;
        auto ev = rnd((size_t)3 << log_n, 77);
        const std::vector<uint64_t> alpha = rnd(3, 78);
        ms::GpuVec<ms::Fq3> layer(pl, ev);
        auto next = ms::apply_drp(layer, alpha, ff, 1);
        std::vector<uint64_t> want((size_t)3 << (log_n - 3));
        oracle_fri_fold(ev.data(), want.data(), log_n, 3, ff, alpha.data(), 1);
        REQUIRE(next.to_host() == want);
        auto tree = ms::MerkleTree::from_fri_layer(layer, ff);
        REQUIRE(tree.num_leaves() == ((size_t)1 << log_n) / ff);
    }
    {   // constraint evaluation: DAG -> program -> device, against a direct evaluation with the oracle's field ops
        using namespace ms::expr;
        const unsigned log_n = 16, lde_step = 4, ncols = 3;
        const size_t n = (size_t)1 << log_n;
        const uint64_t offset = 7;
        std::vector<std::vector<uint64_t>> cols;
        std::vector<ms::GpuVec<ms::Fp>> dev;
        for (unsigned c = 0; c < ncols; c++) { cols.push_back(rnd(n, 900 + c)); dev.emplace_back(pl, cols.back()); }
        const std::vector<uint64_t> ch = rnd(2, 950);
        E x = X();
        E e = (Trace(0, 1) - Trace(0) * Trace(1)) / (pow(x, (uint32_t)(n / lde_step)) - 1) * (Challenge(0) * pow(x, 3) + Challenge(1)) + pow(Trace(2, -1), 5) + x * 9;
        Program prog = compile_expr(e, ncols, false);
        auto out = eval<ms::Fp>(prog, pl, ch, {}, lde_step, offset, n, {&dev[0], &dev[1], &dev[2]}).to_host();
        const uint64_t w = oracle_gl_root_of_unity(log_n), h = oracle_gl_to_mont(offset);
        auto mm = oracle_gl_mul; auto aa = oracle_gl_add; auto ss = oracle_gl_sub;
        for (size_t i : {(size_t)0, (size_t)1, (size_t)12345, n / 2, n - 1}) {
            const uint64_t xi = mm(h, oracle_gl_pow(w, i));
            auto tr = [&](unsigned c, long o) { return cols[c][(i + n + o * (long)lde_step) % n]; };
            const uint64_t num = ss(tr(0, 1), mm(tr(0, 0), tr(1, 0)));
            const uint64_t zer = oracle_gl_inv(ss(oracle_gl_pow(xi, n / lde_step), oracle_gl_to_mont(1)));
            const uint64_t lin = aa(mm(ch[0], oracle_gl_pow(xi, 3)), ch[1]);
            const uint64_t want = aa(aa(mm(mm(num, zer), lin), oracle_gl_pow(tr(2, -1), 5)), mm(xi, oracle_gl_to_mont(9)));
            REQUIRE(out[i] == want);
        }
        // the mirror's view of the kernel cache: nothing may have been left to the interpreter by a failed compilation
        const ms_jit_stats js = pl.jit_stats();
        REQUIRE(js.compile_failures == 0);
        REQUIRE(js.kernels_compiled + js.kernels_from_disk <= 8);
    }
    {   // extension-column scan (examples/brainfuck/trace.rs:131-145): running product with masked rows
        const size_t n = 10000;
        auto f = rnd(3 * n, 31);
        for (size_t i = 0; i < n; i += 4) { f[3 * i] = ms::gl::to_mont(1); f[3 * i + 1] = f[3 * i + 2] = 0; }
        const std::vector<uint64_t> init = rnd(3, 32);
        auto got = ms::running_product(ms::GpuVec<ms::Fq3>(pl, f), init).to_host();
        uint64_t st[3] = {init[0], init[1], init[2]};
        for (size_t i = 0; i < n; i++) {
            REQUIRE(got[3 * i] == st[0] && got[3 * i + 1] == st[1] && got[3 * i + 2] == st[2]);
            uint64_t nx[3]; oracle_fq3_mul(&f[3 * i], st, nx); memcpy(st, nx, 24);
        }
    }
    {   // Queries::new (src/trace.rs:113-157): rows + batched openings; the opening must hash back to the root
        const unsigned log_n = 10;
        const size_t n = (size_t)1 << log_n;
        ms::Matrix<ms::Fp> base; ms::Matrix<ms::Fq3> comp;
        std::vector<std::vector<uint64_t>> bc, cc;
        for (unsigned c = 0; c < 4; c++) { bc.push_back(rnd(n, 40 + c)); base.columns.emplace_back(pl, bc.back()); }
        for (unsigned c = 0; c < 2; c++) { cc.push_back(rnd(3 * n, 50 + c)); comp.columns.emplace_back(pl, cc.back()); }
        auto tb = ms::MerkleTree::from_matrix(base), tc = ms::MerkleTree::from_matrix(comp);
        const std::vector<size_t> positions{5, 4, 1023, 77, 5, 512};
        ms::Queries<ms::Fq3> q(base, nullptr, comp, tb, nullptr, tc, positions);
        for (size_t k = 0; k < positions.size(); k++) {
            for (unsigned c = 0; c < 4; c++) REQUIRE(q.base_trace_values[k * 4 + c] == bc[c][positions[k]]);
            for (unsigned c = 0; c < 2; c++) for (unsigned v = 0; v < 3; v++) REQUIRE(q.composition_trace_values[k * 6 + 3 * c + v] == cc[c][3 * positions[k] + v]);
        }
        // MerkleTreeImpl::verify (src/merkle.rs:208-287) with the oracle's SHA-256
        using D = ms::MerkleTree::Digest;
        auto merge = [](const D& l, const D& r) { uint8_t buf[64]; memcpy(buf, l.data(), 32); memcpy(buf + 32, r.data(), 32); D o; oracle_sha256(buf, 64, o.data()); return o; };
        auto verify = [&](const D& root, const ms::MerkleTree::MerkleView& pf, std::vector<size_t> idx) {
            std::sort(idx.begin(), idx.end()); idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
            const size_t nl = (size_t)1 << pf.height;
            std::deque<std::pair<size_t, D>> nq, lq;
            for (size_t k = 0; k < idx.size(); k++) lq.push_back({idx[k], pf.initial_leaves.at(k)});
            size_t si = 0, ni = 0;
            while (!lq.empty()) {
                auto [index, leaf] = lq.front(); lq.pop_front();
                if (!lq.empty() && (index ^ 1) == lq.front().first) { nq.push_back({(nl + index) >> 1, merge(leaf, lq.front().second)}); lq.pop_front(); continue; }
                const D& sib = pf.sibling_leaves.at(si++);
                nq.push_back({(nl + index) >> 1, index % 2 == 0 ? merge(leaf, sib) : merge(sib, leaf)});
            }
            while (!nq.empty()) {
                auto [index, hsh] = nq.front(); nq.pop_front();
                if (index == 1) return nq.empty() && hsh == root;
                if (!nq.empty() && (index ^ 1) == nq.front().first) { nq.push_back({index >> 1, merge(hsh, nq.front().second)}); nq.pop_front(); continue; }
                const D& sib = pf.nodes.at(ni++);
                nq.push_back({index >> 1, index % 2 == 0 ? merge(hsh, sib) : merge(sib, hsh)});
            }
            return false;
        };
        REQUIRE(verify(tb.root(), q.base_trace_proof, positions));
        REQUIRE(verify(tc.root(), q.composition_trace_proof, positions));
        auto bad = q.base_trace_proof; bad.nodes.at(0)[0] ^= 1;
        REQUIRE(!verify(tb.root(), bad, positions));
        // the same openings through a shared arena (one download), including a gather that does not fit and falls back
        for (size_t cap : {(size_t)1 << 20, (size_t)1024}) {
            ms::GatherArena arena(pl, cap);
            ms::Queries<ms::Fq3> qa(base, nullptr, comp, tb, nullptr, tc, positions, &arena);
            REQUIRE(qa.base_trace_values.empty());                     // deferred until fetch()
            auto extra = tb.prove_launch({3, 9}, &arena);
            qa.fetch();
            REQUIRE(qa.base_trace_values == q.base_trace_values && qa.composition_trace_values == q.composition_trace_values);
            REQUIRE(qa.base_trace_proof.nodes == q.base_trace_proof.nodes && qa.base_trace_proof.initial_leaves == q.base_trace_proof.initial_leaves && qa.base_trace_proof.sibling_leaves == q.base_trace_proof.sibling_leaves);
            REQUIRE(qa.composition_trace_proof.nodes == q.composition_trace_proof.nodes && qa.composition_trace_proof.sibling_leaves == q.composition_trace_proof.sibling_leaves);
            REQUIRE(verify(tb.root(), extra.fetch(), {3, 9}));
            auto late = tb.prove_launch({700}, &arena);                  // reserved after the first download: fetched incrementally
            REQUIRE(verify(tb.root(), late.fetch(), {700}));
        }
    }
    {   // DeepPolyComposer (src/composer.rs:43-188), Fq = Fp: the DEEP polynomial Q satisfies, at a random r,
        //   Q(r) * prod_k (r - z_k) = (a + b r) * sum_t alpha_t (P_t(r) - P_t(z_t)) * prod_{k != t} (r - z_k)
        const unsigned log_n = 10;
        const size_t n = (size_t)1 << log_n;
        ms::Matrix<ms::Fp> base, comp;
        for (unsigned c = 0; c < 3; c++) base.columns.emplace_back(pl, rnd(n, 60 + c));
        for (unsigned c = 0; c < 2; c++) comp.columns.emplace_back(pl, rnd(n, 70 + c));
        std::vector<std::pair<unsigned, int>> args{{0, 0}, {0, 1}, {1, 0}, {2, 1}};
        ms::FqVal z{{987654321987ull, 0, 0}};
        ms::DeepPolyComposer<ms::Fp> composer(args, n, z, base, nullptr, comp);
        auto ood = composer.get_ood_evals();
        REQUIRE(ood.first.size() == args.size() && ood.second.size() == 2);
        ms::DeepCompositionCoeffs co;
        for (size_t k = 0; k < args.size(); k++) co.execution_trace.push_back({{1000 + k, 0, 0}});
        for (size_t k = 0; k < 2; k++) co.composition_trace.push_back({{2000 + k, 0, 0}});
        co.degree[0] = {{5, 0, 0}}; co.degree[1] = {{11, 0, 0}};
        ms::Matrix<ms::Fp> qm; qm.columns.push_back(composer.into_deep_poly(co));
        // evaluate everything at r with the library's Horner entry point (checked against the oracle elsewhere)
        const uint64_t r = 0x123456789abcdefull % ms::gl::P;
        auto at = [&](const ms::Matrix<ms::Fp>& m, unsigned col, uint64_t point) {
            const void* in[8]; for (size_t c = 0; c < m.num_cols(); c++) in[c] = m.columns[c].ptr();
            const uint64_t p = ms::gl::to_mont(point); uint64_t o = 0;
            ms::check(ms_horner_eval(pl.ctx(), MS_GOLDILOCKS_FP, MS_GOLDILOCKS_FP, m.num_rows(), in, (unsigned)m.num_cols(), &col, &p, 1, &o));
            return ms::fq::from_mont(o);
        };
        auto mulp = ms::gl::mul; auto addp = ms::fq::addp;
        auto subp = [](uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + ms::gl::P - b) % ms::gl::P); };
        const uint64_t g = ms::Radix2EvaluationDomain(n).group_gen, z0 = z.c[0], zg = mulp(z0, g), zn = ms::gl::pow(z0, 2);
        const uint64_t zs[3] = {z0, zg, zn};
        struct T { const ms::Matrix<ms::Fp>* m; unsigned col, pt; uint64_t alpha, ood; };
        std::vector<T> terms;
        for (size_t k = 0; k < args.size(); k++) terms.push_back({&base, args[k].first, args[k].second ? 1u : 0u, 1000 + k, ood.first[k].c[0]});
        for (unsigned c = 0; c < 2; c++) terms.push_back({&comp, c, 2u, 2000 + c, ood.second[c].c[0]});
        uint64_t rhs = 0, D = 1;
        for (uint64_t zk : zs) D = mulp(D, subp(r, zk));
        for (auto& t : terms) {
            REQUIRE(at(*t.m, t.col, zs[t.pt]) == t.ood);
            uint64_t v = mulp(t.alpha, subp(at(*t.m, t.col, r), t.ood));
            for (unsigned k = 0; k < 3; k++) if (k != t.pt) v = mulp(v, subp(r, zs[k]));
            rhs = addp(rhs, v);
        }
        rhs = mulp(rhs, addp(5, mulp(11, r)));
        REQUIRE(mulp(at(qm, 0, r), D) == rhs);
        // into_deep_evaluations (ms_deep_rows: the polynomial's values taken from the committed LDEs) == into_deep_poly + its LDE, src/prover.rs:149-152
        ms::Radix2EvaluationDomain lde_dom(4 * n, 7);
        ms::Matrix<ms::Fp> bl = base.bit_reversed_evaluate(lde_dom), cl = comp.bit_reversed_evaluate(lde_dom);
        const auto want = qm.bit_reversed_evaluate(lde_dom).columns[0].to_host();
        const auto got = composer.into_deep_evaluations(co, bl, nullptr, cl).to_host();
        REQUIRE(got == want);
    }
    {   // proof of work (src/random.rs:48-55): the nonce found has the leading zero bits, no smaller one does
        std::array<uint8_t, 32> seed{};
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)(i * 7 + 1);
        const unsigned bits = 12;
        const uint64_t nonce = ms::grind_proof_of_work(pl, seed, bits);
        auto lz = [&](uint64_t nn) { uint8_t msg[40], dg[32]; memcpy(msg, seed.data(), 32); for (int k = 0; k < 8; k++) msg[32 + k] = (uint8_t)(nn >> (56 - 8 * k)); oracle_sha256(msg, 40, dg); unsigned z = 0; for (int k = 0; k < 32; k++) { if (dg[k] == 0) { z += 8; continue; } z += (unsigned)__builtin_clz(dg[k]) - 24; break; } return z; };
        REQUIRE(nonce >= 1 && lz(nonce) >= bits);
        for (uint64_t m = 1; m < nonce; m++) REQUIRE(lz(m) < bits);
    }
    printf("cpp host mirror ok\n");
    return 0;
}
