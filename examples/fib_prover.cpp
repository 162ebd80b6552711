// fib_prover -- the data-parallel phases of the reference's examples/fib (examples/fib/main.rs) on an MI355X,
// written against the C++ host mirror (ministark_amd/csrc/host/*.hpp) the way the Rust example is written
// against ministark + ministark-gpu:
//     gen_trace(n)            examples/fib/main.rs:175-224   (8 columns, multiplicative Fibonacci, n/8 rows)
//     FibAirConfig::constraints  :73-140                    (8 boundary, 1 terminal, 8 transition constraints)
//     ProofOptions::new(32, 4, 8, 8, 64)  :227              (32 queries, blow-up 4, 8 grinding bits, FRI fold 8, remainder <= 64)
//     default_prove           src/prover.rs:25-174
// The Fiat-Shamir channel is replaced by a fixed pseudo-random stream (it hashes a few digests on the host);
// everything that touches column data runs on the device.  The composition constraint is AirConfig::composition_constraint's
// (src/air.rs:50-82: degree adjustment X^adj alpha + beta per constraint; ce_blowup_factor = 1 for this AIR), evaluated on the
// constraint-evaluation coset = the first n rows of the committed bit-reversed LDE; FRI ends with set_remainder and the layer
// openings of into_proof.  Self-check (rows <= 2^18): over the whole 4n-point LDE coset the composition polynomial of a valid
// trace has degree < n -- its upper 3n coefficients must vanish.  At every size and on every repetition the two relations the
// verifier enforces on the opened values are recomputed with host scalars: the constraints at z against the composition trace
// at z^ce, and the DEEP composition at the 32 query positions against the first FRI layer.  `tamper` as the third argument
// alters one opened value (ood = one out-of-domain evaluation, row = one queried trace cell) first: the run must then FAIL.
//   build: g++ -O2 -std=c++17 examples/fib_prover.cpp ministark_amd/libministark_hip.so -Wl,-rpath,$PWD/ministark_amd -o fib_prover
//   run:   ./fib_prover [log2(rows) = 21] [repetitions = 3] [tamper-ood | tamper-row]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include "../ministark_amd/csrc/host/ministark.hpp"
#include "../ministark_amd/csrc/host/stages.hpp"
#include "../ministark_amd/csrc/host/expr.hpp"
#include "../ministark_amd/csrc/host/prover.hpp"

using namespace ms;
using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t) { return std::chrono::duration<double, std::milli>(Clock::now() - t).count(); }

static uint64_t splitmix(uint64_t& s) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return (z ^ (z >> 31)) % gl::P; }

int main(int argc, char** argv) {
    const unsigned log_rows = argc > 1 ? (unsigned)atoi(argv[1]) : 21;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    const std::string tamper = argc > 3 ? argv[3] : "";
    const unsigned log_blowup = 2, fold = 8, num_queries = 32, grinding_bits = 8, max_remainder = 64;
    const size_t n = (size_t)1 << log_rows, N = n << log_blowup;
    Planner& pl = get_planner();

    // ---- gen_trace (host, sequential by nature)
    auto t0 = Clock::now();
    std::vector<std::vector<uint64_t>> cols(8, std::vector<uint64_t>(n));
    {
        uint64_t v[8];
        v[0] = 1; v[1] = 2; v[2] = gl::mul(v[0], v[1]);
        for (int k = 3; k < 8; k++) v[k] = gl::mul(v[k - 2], v[k - 1]);
        for (size_t r = 0; r < n; r++) {
            for (int k = 0; k < 8; k++) cols[k][r] = gl::to_mont(v[k]);
            uint64_t w[8];
            w[0] = gl::mul(v[6], v[7]); w[1] = gl::mul(v[7], w[0]);
            for (int k = 2; k < 8; k++) w[k] = gl::mul(w[k - 2], w[k - 1]);
            memcpy(v, w, sizeof v);
        }
    }
    const uint64_t claimed = cols[7][n - 1];                      // FibClaim(trace.last_value()), Montgomery word
    Matrix<Fp> trace;
    for (auto& c : cols) trace.columns.emplace_back(pl, c);
    printf("trace: %zu rows x 8 columns generated and uploaded in %.1f ms\n", n, ms_since(t0));

    // ---- FibAirConfig::constraints + a random linear combination as the composition constraint
    using namespace ms::expr;
    Radix2EvaluationDomain trace_dom(n), lde_dom(N, 7);
    const uint64_t first_x = 1, last_x = gl::pow(trace_dom.group_gen, n - 1);
    E X_ = X();
    auto curr = [](unsigned c) { return Trace(c, 0); };
    auto next = [](unsigned c) { return Trace(c, 1); };
    std::vector<E> constraints;
    {
        uint64_t v[8];
        v[0] = 1; v[1] = 2; v[2] = 2;
        for (int k = 3; k < 8; k++) v[k] = gl::mul(v[k - 2], v[k - 1]);
        for (unsigned k = 0; k < 8; k++) constraints.push_back((curr(k) - Constant(v[k])) / (X_ - Constant(first_x)));       // boundary
    }
    constraints.push_back((curr(7) - Hint(0)) / (X_ - Constant(last_x)));                                                  // terminal
    {
        E zer = (X_ - Constant(last_x)) / (pow(X_, (uint32_t)n) - Constant(1));                                             // all rows but the last
        E tr[8] = {next(0) - curr(6) * curr(7), next(1) - curr(7) * next(0), next(2) - next(0) * next(1), next(3) - next(1) * next(2),
                   next(4) - next(2) * next(3), next(5) - next(3) * next(4), next(6) - next(4) * next(5), next(7) - next(5) * next(6)};
        for (auto& t : tr) constraints.push_back(t * zer);
    }
    const Composition comp = composition_constraint(n, constraints);                                                       // src/air.rs:50-82
    const unsigned ce = comp.ce_blowup_factor;
    const size_t n_ce = n * ce;
    Program prog = compile_expr(comp.expr, 8, false);
    uint64_t seed = 0x6d696e69;
    std::vector<uint64_t> challenges(comp.num_coeffs);
    for (auto& c : challenges) c = gl::to_mont(splitmix(seed));
    const std::vector<uint64_t> hints{claimed};
    Radix2EvaluationDomain ce_dom(n_ce, 7);
    printf("composition constraint: %zu constraints, ce_blowup_factor %u -> %zu instructions, %u registers\n", constraints.size(), ce, prog.instrs.size(), prog.max_p);

    bool checked = false;
    for (int rep = 0; rep < reps + 1; rep++) {                // rep 0 warms plans / the specialised kernel up
        double ph[7];
        auto all = Clock::now(), t = all;
        // 1. base trace: interpolate, LDE, commit                                   prover.rs:50-55
        Matrix<Fp> base_polys = trace.interpolate(trace_dom);
        Matrix<Fp> base_lde = base_polys.bit_reversed_evaluate(lde_dom);
        MerkleTree base_tree = MerkleTree::from_matrix(base_lde);
        auto base_root = base_tree.root();
        ph[0] = ms_since(t); t = Clock::now();
        // 2. constraint evaluation on the constraint-evaluation coset: the first n_ce rows of the committed bit-reversed LDE are
        //    that coset in its own bit-reversed order, so the evaluator works on them where they lie (the reference re-orders
        //    all columns into natural order and back: bit_reverse_ce_trace)                    prover.rs:88-107
        std::vector<const GpuVec<Fp>*> bc;
        for (auto& c : base_lde.columns) bc.push_back(&c);
        if (!checked && log_rows <= 18) {                     // a valid trace gives a composition polynomial of degree < n
            Matrix<Fp> full;
            full.columns.push_back(eval<Fp>(prog, pl, challenges, hints, 1u << log_blowup, 7, N, bc, {}, true));
            full.bit_reverse_rows();
            full.into_polynomials(lde_dom);
            auto coeffs = full.columns[0].to_host();
            for (size_t i = n_ce; i < N; i++) if (coeffs[i] != 0) { printf("FAILED: composition coefficient %zu is not zero\n", i); return 1; }
            bool any = false;
            for (size_t i = 0; i < n_ce; i++) any |= coeffs[i] != 0;
            if (!any) { printf("FAILED: composition polynomial is identically zero\n"); return 1; }
            checked = true;
            t = Clock::now();
        }
        GpuVec<Fp> comp_evals = eval<Fp>(prog, pl, challenges, hints, ce, 7, n_ce, bc, {}, true);
        ph[1] = ms_since(t); t = Clock::now();
        // 3. composition trace: coefficients, split into ce columns, LDE, commit     prover.rs:110-124
        Matrix<Fp> cm;
        cm.columns.push_back(std::move(comp_evals));
        cm.bit_reverse_rows();                                // one column back to natural order for the inverse transform
        cm.into_polynomials(ce_dom);
        Matrix<Fp> comp_polys = Matrix<Fp>::from_chunks(cm.columns[0], ce);
        Matrix<Fp> comp_lde = comp_polys.bit_reversed_evaluate(lde_dom);
        MerkleTree comp_tree = MerkleTree::from_matrix(comp_lde);
        auto comp_root = comp_tree.root();
        ph[2] = ms_since(t); t = Clock::now();
        // 4. DEEP composition                                                        prover.rs:136-153
        std::vector<std::pair<unsigned, int>> args;
        for (unsigned c = 0; c < 8; c++) { args.push_back({c, 0}); args.push_back({c, 1}); }
        FqVal z{{splitmix(seed), 0, 0}};
        DeepPolyComposer<Fp> composer(args, n, z, base_polys, nullptr, comp_polys);
        auto ood = composer.get_ood_evals();
        DeepCompositionCoeffs dc;
        for (size_t k = 0; k < args.size(); k++) dc.execution_trace.push_back({{splitmix(seed), 0, 0}});
        for (unsigned k = 0; k < ce; k++) dc.composition_trace.push_back({{splitmix(seed), 0, 0}});
        dc.degree[0] = {{splitmix(seed), 0, 0}}; dc.degree[1] = {{splitmix(seed), 0, 0}};
        // prover.rs:149-152: into_deep_poly + into_bit_reversed_evaluations, computed on the committed LDEs where they lie (ms_deep_rows:
        // the same evaluations; tests/cpp/test_host_mirror.cpp checks the two routes against each other)
        Matrix<Fp> deep_lde;
        deep_lde.columns.push_back(composer.into_deep_evaluations(dc, base_lde, nullptr, comp_lde));
        ph[3] = ms_since(t); t = Clock::now();
        // 5. FRI layers + remainder                                                  fri.rs:179-249
        std::vector<GpuVec<Fp>> layers;                       // FriLayer { merkle_tree, evaluations } (fri.rs:218-221)
        std::vector<MerkleTree> trees;
        GpuVec<Fp> layer = std::move(deep_lde.columns[0]);
        std::array<uint8_t, 32> last_root = comp_root;
        while (layer.len() > (size_t)max_remainder << log_blowup) {
            trees.push_back(MerkleTree::from_fri_layer(layer, fold));
            last_root = trees.back().root();
            const std::vector<uint64_t> alpha{gl::to_mont(splitmix(seed))};
            GpuVec<Fp> next_layer = apply_drp(layer, alpha, fold, 1);
            layers.push_back(std::move(layer));
            layer = std::move(next_layer);
        }
        // set_remainder (fri.rs:232-248): bit_reverse, iNTT over the subgroup of the remainder's size, the first len / blowup coefficients
        Matrix<Fp> rem;
        rem.columns.push_back(layer.clone());
        rem.bit_reverse_rows();
        rem.into_polynomials(Radix2EvaluationDomain(layer.len()));
        auto remainder = rem.columns[0].to_host();
        if (log_rows <= 18)                                   // fri.rs:244: a valid trace leaves nothing above len / blowup
            for (size_t i = layer.len() >> log_blowup; i < remainder.size(); i++) if (remainder[i] != 0) { printf("FAILED: FRI remainder coefficient %zu is not zero\n", i); return 1; }
        remainder.resize(std::max<size_t>(layer.len() >> log_blowup, 1));
        ph[4] = ms_since(t); t = Clock::now();
        // 6. proof of work, queries, FRI openings                                    prover.rs:160-173
        const uint64_t nonce = grind_proof_of_work(pl, last_root, grinding_bits);
        std::vector<size_t> positions(num_queries);
        for (auto& p : positions) p = (size_t)(splitmix(seed) % N);
        GatherArena arena(pl);                                // every gather of this phase lands in one buffer, downloaded once
        Queries<Fp> q(base_lde, nullptr, comp_lde, base_tree, nullptr, comp_tree, positions, &arena);
        std::set<size_t> uniq(positions.begin(), positions.end());
        std::vector<size_t> pos(uniq.begin(), uniq.end());
        size_t opened = 0;
        std::vector<size_t> pos0; std::vector<uint64_t> rows0;
        {                                                     // fri_prover.into_proof(&query_positions): fri.rs:148-165
            std::vector<Pending> row_gathers; std::vector<MerkleTree::PendingView> view_gathers;     // every gather first, then the downloads
            for (size_t l = 0; l < layers.size(); l++) {
                pos = fold_positions(pos, fold);
                if (l == 0) pos0 = pos;
                row_gathers.push_back(fri_layer_rows_launch(layers[l], fold, pos, &arena));
                view_gathers.push_back(trees[l].prove_launch(pos, &arena));
            }
            q.fetch();
            for (size_t l = 0; l < layers.size(); l++) {
                auto rows = row_gathers[l].fetch<uint64_t>();
                auto view = view_gathers[l].fetch();
                opened += rows.size() / fold + view.nodes.size();
                if (l == 0) rows0 = std::move(rows);
            }
        }
        ph[5] = ms_since(t);
        ph[6] = ms_since(all);
        if (q.base_trace_values.size() != positions.size() * 8 || ood.first.size() != args.size() || opened == 0) { printf("FAILED: query / OOD shapes\n"); return 1; }
        // 7. what the verifier checks with these values, at every size and on every repetition (host scalars, outside the clock)
        {
            if (tamper == "tamper-ood") ood.first[5].c[0] = gl::add(ood.first[5].c[0], 1);
            if (tamper == "tamper-row") q.base_trace_values[17 * 8 + 2] = gl::add(q.base_trace_values[17 * 8 + 2], 1);
            // (i) the composition constraint at z from the opened trace values == sum_k z^k C_k(z^ce)          verifier.rs:106-128
            std::vector<uint64_t> ch;
            for (auto c : challenges) ch.push_back(gl::from_mont(c));
            auto trace_at = [&](unsigned col, int off) { for (size_t k = 0; k < args.size(); k++) if (args[k].first == col && args[k].second == off) return ood.first[k].c[0]; throw std::runtime_error("trace argument not opened"); };
            const uint64_t lhs = eval_at_point(comp.expr, z.c[0], n, trace_at, ch, {gl::from_mont(claimed)});
            uint64_t rhs = 0;
            for (unsigned k = ce; k-- > 0;) rhs = gl::add(gl::mul(rhs, z.c[0]), ood.second[k].c[0]);
            if (lhs != rhs) { printf("FAILED: out-of-domain consistency (constraints at z vs composition trace at z^%u)\n", ce); return 1; }
            // (ii) the DEEP composition at the query positions from the opened rows == the first FRI layer there   verifier.rs:146-190, composer.rs:201-258
            const uint64_t w = lde_dom.group_gen, g = trace_dom.group_gen, z_n = gl::pow(z.c[0], ce);
            for (size_t i = 0; i < positions.size(); i++) {
                size_t p = positions[i], nat = 0;
                for (unsigned b = 0; b < log_rows + log_blowup; b++) nat |= ((p >> b) & 1) << (log_rows + log_blowup - 1 - b);
                const uint64_t x = gl::mul(7, gl::pow(w, nat));
                uint64_t acc = 0;
                for (size_t k = 0; k < args.size(); k++) {
                    const uint64_t v = gl::from_mont(q.base_trace_values[i * 8 + args[k].first]), pt = gl::mul(z.c[0], gl::pow(g, (uint64_t)args[k].second));
                    acc = gl::add(acc, gl::mul(dc.execution_trace[k].c[0], gl::mul(gl::sub(v, ood.first[k].c[0]), gl::inv(gl::sub(x, pt)))));
                }
                for (unsigned k = 0; k < ce; k++) {
                    const uint64_t v = gl::from_mont(q.composition_trace_values[i * ce + k]);
                    acc = gl::add(acc, gl::mul(dc.composition_trace[k].c[0], gl::mul(gl::sub(v, ood.second[k].c[0]), gl::inv(gl::sub(x, z_n)))));
                }
                acc = gl::mul(acc, gl::add(dc.degree[0].c[0], gl::mul(dc.degree[1].c[0], x)));
                const size_t slot = std::lower_bound(pos0.begin(), pos0.end(), p / fold) - pos0.begin();
                if (slot >= pos0.size() || pos0[slot] != p / fold || gl::from_mont(rows0[slot * fold + p % fold]) != acc) { printf("FAILED: DEEP composition at query %zu (position %zu)\n", i, p); return 1; }
            }
        }
        if (rep == 0) {
            printf("roots: base %02x%02x%02x%02x.. composition %02x%02x%02x%02x..  FRI layers %zu  remainder %zu coefficients  nonce %llu\n", base_root[0], base_root[1], base_root[2],
                   base_root[3], comp_root[0], comp_root[1], comp_root[2], comp_root[3], layers.size(), remainder.size(), (unsigned long long)nonce);
            continue;
        }
        printf("rep %d: base LDE+commit %.2f | evaluation %.2f | composition %.2f | DEEP %.2f | FRI %.2f | PoW+queries+openings %.2f | total %.2f ms\n", rep, ph[0], ph[1], ph[2],
               ph[3], ph[4], ph[5], ph[6]);
    }
    printf("fib prover pipeline ok\n");
    return 0;
}
