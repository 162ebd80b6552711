// Second instruction-rate probe (gfx950): which ops are 2-cycle and which 4-cycle per wave64.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
#define ITERS 4096
#define REP8(x) x x x x x x x x
#define ASM8(S, IN) asm volatile(S("%0") S("%1") S("%2") S("%3") S("%4") S("%5") S("%6") S("%7") \
    : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(IN), "s"(sm), "v"(tt32) : "vcc", "s10", "s11");
#define RATE_KERNEL(name, TYPE, S)                                                        \
__global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) {               \
    uint32_t t = threadIdx.x + seed;                                                        \
    TYPE a0=(TYPE)t,a1=(TYPE)(t+1),a2=(TYPE)(t+2),a3=(TYPE)(t+3),a4=(TYPE)(t+4),a5=(TYPE)(t+5),a6=(TYPE)(t+6),a7=(TYPE)(t+7); \
    TYPE tt = (TYPE)(t * 2654435761u) | 1;                                                  \
    uint64_t sm = 0x5555aaaa3333ccccull ^ seed; uint32_t tt32 = t * 40503u | 1;                                             \
    asm volatile("s_mov_b64 vcc, %0" :: "s"(sm) : "vcc");                                   \
    for (int i = 0; i < ITERS; i++) { REP8(ASM8(S, tt)) }                                   \
    TYPE x = a0; x = x + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                  \
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)x;                                      \
}
#define S_MOV(x)      "v_mov_b32 " x ", %8\n"
#define S_SUB(x)      "v_sub_u32 " x ", " x ", %8\n"
#define S_AND(x)      "v_and_b32 " x ", " x ", %8\n"
#define S_XOR(x)      "v_xor_b32 " x ", " x ", %8\n"
#define S_LSHL(x)     "v_lshlrev_b32 " x ", 3, " x "\n"
#define S_LSHR(x)     "v_lshrrev_b32 " x ", 3, " x "\n"
#define S_NOT(x)      "v_not_b32 " x ", " x "\n"
#define S_CND32(x)    "v_cndmask_b32 " x ", " x ", %8, vcc\n"
#define S_CND64(x)    "v_cndmask_b32_e64 " x ", " x ", %8, %9\n"
#define S_CMP32(x)    "v_cmp_lt_u32 vcc, " x ", %8\n"
#define S_CMP32S(x)   "v_cmp_lt_u32_e64 s[10:11], " x ", %8\n"
#define S_ADD64E(x)   "v_add_u32_e64 " x ", " x ", %8\n"
#define S_SUBCO(x)    "v_sub_co_u32 " x ", vcc, " x ", %8\n"
#define S_LSHLADD(x)  "v_lshl_add_u32 " x ", " x ", 3, %8\n"
#define S_BFE(x)      "v_bfe_u32 " x ", " x ", 3, 7\n"
#define S_PERM(x)     "v_perm_b32 " x ", " x ", %8, %8\n"
#define S_MIN(x)      "v_min_u32 " x ", " x ", %8\n"
#define S_ADDCOS(x)   "v_add_co_u32_e64 " x ", s[10:11], " x ", %8\n"
#define S_MAD64S(x)   "v_mad_u64_u32 " x ", s[10:11], %10, %10, " x "\n"
#define S_MADI64(x)   "v_mad_i64_i32 " x ", s[10:11], %10, %10, " x "\n"
#define S_SUBREV(x)   "v_subrev_u32 " x ", " x ", %8\n"
#define S_OR3(x)      "v_or3_b32 " x ", " x ", %8, %8\n"
#define S_ANDOR(x)    "v_and_or_b32 " x ", " x ", %8, %8\n"
#define S_XAD(x)      "v_xad_u32 " x ", " x ", %8, %8\n"
#define S_MULLO(x)    "v_mul_lo_u32 " x ", " x ", %8\n"
#define S_DPP(x)      "v_mov_b32_dpp " x ", " x " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define S_ADDDPP(x)   "v_add_u32_dpp " x ", " x ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define S_CMPCND(x)   "v_cmp_lt_u32 vcc, " x ", %8\n v_cndmask_b32 " x ", " x ", %8, vcc\n"
#define S_CMPCNDS(x)  "v_cmp_lt_u32_e64 s[10:11], " x ", %8\n v_cndmask_b32_e64 " x ", " x ", %8, s[10:11]\n"
#define S_CND64V(x)   "v_cndmask_b32_e64 " x ", " x ", %8, vcc\n"
#define S_ADDCCND(x)  "v_add_co_u32 " x ", vcc, " x ", %8\n v_cndmask_b32 " x ", " x ", %8, vcc\n"
#define S_ADDCADDC(x) "v_add_co_u32 " x ", vcc, " x ", %8\n v_addc_co_u32 " x ", vcc, 0, " x ", vcc\n"
RATE_KERNEL(k_cmpcnd, uint32_t, S_CMPCND)
RATE_KERNEL(k_cmpcnds, uint32_t, S_CMPCNDS)
RATE_KERNEL(k_cnd64v, uint32_t, S_CND64V)
RATE_KERNEL(k_addccnd, uint32_t, S_ADDCCND)
RATE_KERNEL(k_addcaddc, uint32_t, S_ADDCADDC)
RATE_KERNEL(k_mov, uint32_t, S_MOV)
RATE_KERNEL(k_sub, uint32_t, S_SUB)
RATE_KERNEL(k_and, uint32_t, S_AND)
RATE_KERNEL(k_xor, uint32_t, S_XOR)
RATE_KERNEL(k_lshl, uint32_t, S_LSHL)
RATE_KERNEL(k_lshr, uint32_t, S_LSHR)
RATE_KERNEL(k_not, uint32_t, S_NOT)
RATE_KERNEL(k_cnd32, uint32_t, S_CND32)
RATE_KERNEL(k_cnd64, uint32_t, S_CND64)
RATE_KERNEL(k_cmp32, uint32_t, S_CMP32)
RATE_KERNEL(k_cmp32s, uint32_t, S_CMP32S)
RATE_KERNEL(k_add64e, uint32_t, S_ADD64E)
RATE_KERNEL(k_subco, uint32_t, S_SUBCO)
RATE_KERNEL(k_lshladd, uint32_t, S_LSHLADD)
RATE_KERNEL(k_bfe, uint32_t, S_BFE)
RATE_KERNEL(k_perm, uint32_t, S_PERM)
RATE_KERNEL(k_min, uint32_t, S_MIN)
RATE_KERNEL(k_addcos, uint32_t, S_ADDCOS)
RATE_KERNEL(k_mad64s, uint64_t, S_MAD64S)
RATE_KERNEL(k_madi64, uint64_t, S_MADI64)
RATE_KERNEL(k_subrev, uint32_t, S_SUBREV)
RATE_KERNEL(k_or3, uint32_t, S_OR3)
RATE_KERNEL(k_andor, uint32_t, S_ANDOR)
RATE_KERNEL(k_xad, uint32_t, S_XAD)
RATE_KERNEL(k_mullo, uint32_t, S_MULLO)
RATE_KERNEL(k_dpp, uint32_t, S_DPP)
RATE_KERNEL(k_adddpp, uint32_t, S_ADDDPP)

template <typename K>
static void run_rate(const char* name, K kern, uint32_t* d_out, int blocks_per_cu) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, (uint32_t)r);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    double wave_insts = blocks * 4.0 * ITERS * 64.0;
    printf("RATE %-22s waves/SIMD=%d %8.3f ms  => %6.2f cycles/wave-inst/SIMD @2.4GHz\n", name, blocks_per_cu, best,
           best * 1e-3 * 2.4e9 / (wave_insts / 1024.0));
    fflush(stdout);
}
int main() {
    uint32_t* d; CK(hipMalloc(&d, 256 * 8 * 256 * 4));
    run_rate("cmp_vcc+cnd_vcc (x2 inst)", k_cmpcnd, d, 8);
    run_rate("cmp_sgpr+cnd_sgpr (x2)", k_cmpcnds, d, 8);
    run_rate("v_cndmask_e64 vcc", k_cnd64v, d, 8);
    run_rate("add_co+cnd_vcc (x2)", k_addccnd, d, 8);
    run_rate("add_co+addc (x2)", k_addcaddc, d, 8);
    for (int w : {8}) { if (w) break;
    run_rate("v_mov_b32", k_mov, d, w); run_rate("v_sub_u32", k_sub, d, w); run_rate("v_and_b32", k_and, d, w);
    run_rate("v_xor_b32", k_xor, d, w); run_rate("v_lshlrev_b32", k_lshl, d, w); run_rate("v_lshrrev_b32", k_lshr, d, w);
    run_rate("v_not_b32", k_not, d, w); run_rate("v_cndmask_b32 vcc", k_cnd32, d, w); run_rate("v_cndmask_b32_e64 sgpr", k_cnd64, d, w);
    run_rate("v_cmp_lt_u32 vcc", k_cmp32, d, w); run_rate("v_cmp_lt_u32_e64 sgpr", k_cmp32s, d, w);
    run_rate("v_add_u32_e64", k_add64e, d, w); run_rate("v_sub_co_u32", k_subco, d, w); run_rate("v_lshl_add_u32", k_lshladd, d, w);
    run_rate("v_bfe_u32", k_bfe, d, w); run_rate("v_perm_b32", k_perm, d, w); run_rate("v_min_u32", k_min, d, w);
    run_rate("v_add_co_u32_e64 sgpr", k_addcos, d, w); run_rate("v_mad_u64_u32 sgpr", k_mad64s, d, w); run_rate("v_mad_i64_i32", k_madi64, d, w);
    run_rate("v_subrev_u32", k_subrev, d, w); run_rate("v_or3_b32", k_or3, d, w); run_rate("v_and_or_b32", k_andor, d, w);
    run_rate("v_xad_u32", k_xad, d, w); run_rate("v_mul_lo_u32", k_mullo, d, w);
    run_rate("v_mov_b32_dpp", k_dpp, d, w); run_rate("v_add_u32_dpp", k_adddpp, d, w);
    }
    return 0;
}
