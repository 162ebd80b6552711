// GPU parity test of the C++ host mirror (ministark.hpp) against the C oracle, shapes of
// gpu/tests/shaders.rs:17-117.  Built and run by tests/test_cpp_mirror.py (-m gpu).
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../ministark_amd/csrc/host/ministark.hpp"

extern "C" {   // the checker: oracle/c/oracle.c
void oracle_ntt(uint64_t* a, unsigned log_n, unsigned V, int inverse, uint64_t offset_canon);
void oracle_lde(const uint64_t* in, uint64_t* out, unsigned log_n, unsigned log_blowup, unsigned V, uint64_t offset_canon, int bit_reversed);
void oracle_sha256_rows(const uint64_t* const* cols, unsigned ncols, unsigned V, size_t nrows, uint8_t* leaves);
void oracle_sha256_merkle(const uint8_t* leaves, size_t n, uint8_t* nodes);
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
    printf("cpp host mirror ok\n");
    return 0;
}
