// differential test: the same gl_dev.h source on host (clang) vs device, bit-for-bit (weak values included)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "gl.h"
#include "gl_dev.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
template <bool INV, int E, bool VC> __global__ void k(const uint64_t* a, uint64_t* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    uint64_t u = a[2 * i], v = a[2 * i + 1]; if (VC) v = gld::canon(v);
    gld::bfly<INV, E, VC>(u, v); o[2 * i] = u; o[2 * i + 1] = v;
}
template <bool INV> __global__ void kd(const uint64_t* a, uint64_t* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    uint64_t x[16];
    #pragma unroll
    for (int q = 0; q < 16; q++) x[q] = a[16 * i + q] % gl::P;
    gld::dft16<INV>(x);
    #pragma unroll
    for (int q = 0; q < 16; q++) o[16 * i + q] = gld::canon(x[q]);
}
static std::vector<uint64_t> vals(int n) {
    const uint64_t P = gl::P;
    std::vector<uint64_t> v; const uint64_t edge[] = {0, 1, P - 1, P, P + 1, ~0ull, ~0ull - 1, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFF00000000ull, 0xFFFFFFFEFFFFFFFFull};
    for (uint64_t a : edge) for (uint64_t b : edge) { v.push_back(a); v.push_back(b); }
    uint64_t s = 7; while ((int)v.size() < n) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; v.push_back(z); }
    return v;
}
template <bool INV, int E, bool VC> static void test(const std::vector<uint64_t>& in, uint64_t* din, uint64_t* dout) {
    int n = in.size() / 2; std::vector<uint64_t> out(in.size());
    hipLaunchKernelGGL((k<INV, E, VC>), dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
    CK(hipMemcpy(out.data(), dout, in.size() * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; i++) { uint64_t u = in[2 * i], v = in[2 * i + 1]; if (VC) v = gld::canon(v); gld::bfly<INV, E, VC>(u, v);
        if (u != out[2 * i] || v != out[2 * i + 1]) { if (bad++ < 2) printf("  bfly INV=%d E=%d VC=%d in=(%016llx,%016llx) dev=(%016llx,%016llx) host=(%016llx,%016llx)\n", INV, E, VC,
            (unsigned long long)in[2*i], (unsigned long long)in[2*i+1], (unsigned long long)out[2*i], (unsigned long long)out[2*i+1], (unsigned long long)u, (unsigned long long)v); } }
    printf("bfly INV=%d E=%d VC=%d: %s (%d bad)\n", INV, E, VC, bad ? "DIFF" : "same", bad);
}
int main() {
    std::vector<uint64_t> in = vals(1 << 14);
    uint64_t *din, *dout; CK(hipMalloc(&din, in.size() * 8)); CK(hipMalloc(&dout, in.size() * 8));
    CK(hipMemcpy(din, in.data(), in.size() * 8, hipMemcpyHostToDevice));
    test<false, 0, true>(in, din, dout); test<false, 0, false>(in, din, dout);
    test<false, 1, false>(in, din, dout); test<false, 2, false>(in, din, dout); test<false, 3, false>(in, din, dout); test<false, 4, false>(in, din, dout);
    test<false, 5, false>(in, din, dout); test<false, 6, false>(in, din, dout); test<false, 7, false>(in, din, dout);
    test<true, 1, false>(in, din, dout); test<true, 3, false>(in, din, dout); test<true, 6, false>(in, din, dout);
    // whole network
    for (int inv = 0; inv < 2; inv++) {
        int n = in.size() / 16; std::vector<uint64_t> out(in.size());
        if (inv) hipLaunchKernelGGL(kd<true>, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n); else hipLaunchKernelGGL(kd<false>, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
        CK(hipMemcpy(out.data(), dout, in.size() * 8, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < n; i++) { uint64_t x[16]; for (int q = 0; q < 16; q++) x[q] = in[16 * i + q] % gl::P; if (inv) gld::dft16<true>(x); else gld::dft16<false>(x);
            for (int q = 0; q < 16; q++) x[q] = gld::canon(x[q]);
            for (int q = 0; q < 16; q++) if (x[q] != out[16 * i + q]) { if (bad++ < 3) printf("  dft16 inv=%d group %d q=%d dev=%016llx host=%016llx\n", inv, i, q, (unsigned long long)out[16*i+q], (unsigned long long)x[q]); } }
        printf("dft16 inv=%d: %s (%d bad)\n", inv, bad ? "DIFF" : "same", bad);
    }
    return 0;
}
