// On-disk cache of the specialised constraint kernels' code objects (host only).
//
// The reference ships its shaders precompiled (gpu/src/plan.rs:30 `include_bytes!("metal/shaders.metallib")`): a prover process pays
// nothing at run time.  Here the constraint program of an AIR becomes a kernel through hiprtc (eval_jit.h) -- 1 to 4.5 s the first
// time, against a 9 ms proof -- so the gfx950 code object is kept on disk and a later PROCESS loads it with hipModuleLoadData:
//
//   directory  $MS_JIT_CACHE, else $XDG_CACHE_HOME/ministark_hip, else $HOME/.cache/ministark_hip; MS_JIT_CACHE=0 (or "off") disables
//   key        SHA-256 over: format tag, compiler options, hiprtc version, every embedded header (name + text = the library's own
//              device sources, so a rebuilt library with different kernels never reads an old entry), the generated source
//   file       <key>.co = "MSJITCO1" | u64 payload bytes | SHA-256(payload) | payload; written to a temporary name and renamed
//   size       an entry is 20 .. 90 KB (one per distinct generated source: per AIR and launch shape, not per proof); nothing is evicted --
//              remove the directory to start over
//   reading    magic, size and digest are checked; an entry that fails any of them (truncated, corrupted, foreign) is removed
//              and the program is compiled again -- a cache entry can make a run faster, never different
#pragma once
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace msjit {

// FIPS 180-4 SHA-256, host side (keys and payload digests only; the row / tree hashing of the product is sha256_kernels.h)
struct Sha256 {
    uint32_t h[8];
    uint8_t buf[64];
    uint64_t len = 0;
    Sha256() {
        static const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
        memcpy(h, iv, sizeof h);
    }
    static uint32_t rr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
            0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
            0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
            0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
            0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
            0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int t = 0; t < 16; t++) w[t] = (uint32_t)p[4 * t] << 24 | (uint32_t)p[4 * t + 1] << 16 | (uint32_t)p[4 * t + 2] << 8 | p[4 * t + 3];
        for (int t = 16; t < 64; t++) {
            const uint32_t s0 = rr(w[t - 15], 7) ^ rr(w[t - 15], 18) ^ (w[t - 15] >> 3), s1 = rr(w[t - 2], 17) ^ rr(w[t - 2], 19) ^ (w[t - 2] >> 10);
            w[t] = w[t - 16] + s0 + w[t - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int t = 0; t < 64; t++) {
            const uint32_t t1 = hh + (rr(e, 6) ^ rr(e, 11) ^ rr(e, 25)) + ((e & f) ^ (~e & g)) + K[t] + w[t];
            const uint32_t t2 = (rr(a, 2) ^ rr(a, 13) ^ rr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const void* data, size_t n) {
        const uint8_t* p = (const uint8_t*)data;
        size_t fill = (size_t)(len & 63);
        len += n;
        if (fill) {
            const size_t take = std::min(n, 64 - fill);
            memcpy(buf + fill, p, take);
            p += take; n -= take; fill += take;
            if (fill < 64) return;
            block(buf);
        }
        for (; n >= 64; p += 64, n -= 64) block(p);
        if (n) memcpy(buf, p, n);
    }
    // a length-prefixed field: ("ab", "c") and ("a", "bc") hash differently
    void field(const void* data, size_t n) { const uint64_t l = n; update(&l, 8); update(data, n); }
    void field(const std::string& s) { field(s.data(), s.size()); }
    void final(uint8_t out[32]) {
        const uint64_t bits = len * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while ((len & 63) != 56) update(&zero, 1);
        uint8_t be[8];
        for (int i = 0; i < 8; i++) be[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(be, 8);
        for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
    }
};

static inline std::string hex(const uint8_t* d, size_t n) {
    static const char* x = "0123456789abcdef";
    std::string s(2 * n, '0');
    for (size_t i = 0; i < n; i++) { s[2 * i] = x[d[i] >> 4]; s[2 * i + 1] = x[d[i] & 15]; }
    return s;
}

static inline bool mkdirs(const std::string& path) {
    for (size_t i = 1; i <= path.size(); i++) {
        if (i != path.size() && path[i] != '/') continue;
        const std::string p = path.substr(0, i);
        if (mkdir(p.c_str(), 0700) != 0 && errno != EEXIST) return false;
    }
    struct stat st;
    return stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

// "" = no disk cache (switched off, or no directory can be made).  Read at every call: a test may point it elsewhere.
static inline std::string cache_dir() {
    std::string d;
    if (const char* e = getenv("MS_JIT_CACHE")) {
        if (!*e || !strcmp(e, "0") || !strcmp(e, "off")) return "";
        d = e;
    } else if (const char* x = getenv("XDG_CACHE_HOME")) { if (*x) d = std::string(x) + "/ministark_hip"; }
    if (d.empty()) {
        const char* home = getenv("HOME");
        if (!home || !*home) return "";
        d = std::string(home) + "/.cache/ministark_hip";
    }
    return mkdirs(d) ? d : "";
}

static constexpr char kMagic[8] = {'M', 'S', 'J', 'I', 'T', 'C', 'O', '1'};

// -> true and the payload when <dir>/<key>.co is a whole, unmodified entry; a damaged one is removed (*damaged = true)
static inline bool disk_load(const std::string& dir, const std::string& key, std::vector<char>& code, bool* damaged) {
    *damaged = false;
    const std::string path = dir + "/" + key + ".co";
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    char head[8 + 8 + 32];
    if (fread(head, 1, sizeof head, f) == sizeof head && !memcmp(head, kMagic, 8)) {
        uint64_t n;
        memcpy(&n, head + 8, 8);
        if (n > 0 && n < ((uint64_t)1 << 30)) {
            code.resize((size_t)n);
            char extra;
            if (fread(code.data(), 1, (size_t)n, f) == n && fread(&extra, 1, 1, f) == 0) {
                Sha256 s;
                s.update(code.data(), code.size());
                uint8_t dg[32];
                s.final(dg);
                ok = !memcmp(dg, head + 16, 32);
            }
        }
    }
    fclose(f);
    if (!ok) { code.clear(); *damaged = true; (void)unlink(path.c_str()); }
    return ok;
}

static inline bool disk_store(const std::string& dir, const std::string& key, const std::vector<char>& code) {
    char tmpl[64];
    snprintf(tmpl, sizeof tmpl, "/.tmp-%ld-XXXXXX", (long)getpid());
    std::string tmp = dir + tmpl;
    const int fd = mkstemp(&tmp[0]);
    if (fd < 0) return false;
    FILE* f = fdopen(fd, "wb");
    if (!f) { close(fd); (void)unlink(tmp.c_str()); return false; }
    Sha256 s;
    s.update(code.data(), code.size());
    uint8_t dg[32];
    s.final(dg);
    const uint64_t n = code.size();
    bool ok = fwrite(kMagic, 1, 8, f) == 8 && fwrite(&n, 1, 8, f) == 8 && fwrite(dg, 1, 32, f) == 32 && fwrite(code.data(), 1, code.size(), f) == code.size();
    ok = (fclose(f) == 0) && ok;
    // rename is atomic: a concurrent reader sees the old entry, none, or the whole new one
    if (ok) ok = rename(tmp.c_str(), (dir + "/" + key + ".co").c_str()) == 0;
    if (!ok) (void)unlink(tmp.c_str());
    return ok;
}

}  // namespace msjit
