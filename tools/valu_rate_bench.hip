// Issue-rate microbenchmark for the integer VALU instructions the Goldilocks kernels are built from (gfx950).
// Each kernel runs ITER iterations of 32 independent instructions of one kind in every wave; the host reports
// cycles per wave-instruction per SIMD at W waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O2 -o build/valu_rate_bench tools/valu_rate_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define ITER 20000

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

#define KERNEL(name, body)                                                                  \
    __global__ __launch_bounds__(256) void name(unsigned* out, unsigned seed) {            \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 ^ 0x55; \
        unsigned long long q0 = a0 * 77ull + 1, q1 = a1 * 99ull + 3, q2 = 5;                \
        for (int it = 0; it < ITER; it++) {                                                 \
            asm volatile(REP32(body) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(q0), "+v"(q1), "+v"(q2)::"vcc", "scc", "s10", "s11"); \
        }                                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (unsigned)q0 + (unsigned)q1 + (unsigned)q2; \
    }

KERNEL(k_add_u32, "v_add_u32 %0, %1, %2\n")
KERNEL(k_add3_u32, "v_add3_u32 %0, %1, %2, %3\n")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %1, %2\n")
KERNEL(k_mul_u24_dpp, "v_mul_u32_u24_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %1, %2, %3\n")
KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %4, s[10:11], %1, %2, %5\n")
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %2\n")
KERNEL(k_mul_hi_u32, "v_mul_hi_u32 %0, %1, %2\n")
KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %4, %5, 0, %6\n")
KERNEL(k_cmp_lt_u64, "v_cmp_lt_u64 vcc, %4, %5\n")
KERNEL(k_cmp_lt_u32, "v_cmp_lt_u32 vcc, %1, %2\n")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %1, %2, vcc\n")
KERNEL(k_add_co_pair, "v_add_co_u32 %0, vcc, %1, %2\n v_addc_co_u32 %3, vcc, %1, %2, vcc\n")
KERNEL(k_lshlrev_b64, "v_lshlrev_b64 %4, 3, %5\n")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %1, %2, 7\n")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_cmp_cnd_chain, "v_cmp_lt_u64 vcc, %4, %5\n v_cndmask_b32 %0, %1, %2, vcc\n")
KERNEL(k_dot4_u8, "v_dot4_u32_u8 %0, %1, %2, %3\n")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %1, %2\n")
// which 32-bit instructions share v_add_u32's double rate
KERNEL(k_sub_u32, "v_sub_u32 %0, %1, %2\n")
KERNEL(k_and_b32, "v_and_b32 %0, %1, %2\n")
KERNEL(k_xor_b32, "v_xor_b32 %0, %1, %2\n")
KERNEL(k_lshlrev_b32, "v_lshlrev_b32 %0, 5, %1\n")
KERNEL(k_lshrrev_b32, "v_lshrrev_b32 %0, 5, %1\n")
KERNEL(k_ashrrev_i32, "v_ashrrev_i32 %0, 5, %1\n")
KERNEL(k_mov_b32, "v_mov_b32 %0, %1\n")
KERNEL(k_cndmask_s, "v_cndmask_b32_e64 %0, %1, %2, s[10:11]\n")
KERNEL(k_bfe_u32, "v_bfe_u32 %0, %1, 5, 12\n")
KERNEL(k_and_or_b32, "v_and_or_b32 %0, %1, %2, %3\n")
KERNEL(k_lshl_add_u32, "v_lshl_add_u32 %0, %1, 5, %2\n")
KERNEL(k_lshl_or_b32, "v_lshl_or_b32 %0, %1, 5, %2\n")
KERNEL(k_perm_b32, "v_perm_b32 %0, %1, %2, %3\n")
KERNEL(k_bfi_b32, "v_bfi_b32 %0, %1, %2, %3\n")
KERNEL(k_min_u32, "v_min_u32 %0, %1, %2\n")
KERNEL(k_fma_f32, "v_fma_f32 %0, %1, %2, %3\n")
KERNEL(k_add_f32, "v_add_f32 %0, %1, %2\n")
KERNEL(k_mul_f32, "v_mul_f32 %0, %1, %2\n")
KERNEL(k_pk_fma_f32, "v_pk_fma_f32 %4, %5, %6, %5\n")
KERNEL(k_fma_f64, "v_fma_f64 %4, %5, %6, %5\n")
KERNEL(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %1\n")
KERNEL(k_add_co_e64, "v_add_co_u32_e64 %0, s[10:11], %1, %2\n")
KERNEL(k_addc_only, "v_addc_co_u32 %0, vcc, %1, %2, vcc\n")
KERNEL(k_mad_i64_i32, "v_mad_i64_i32 %4, s[10:11], %1, %2, %5\n")
KERNEL(k_mul_i32_i24, "v_mul_i32_i24 %0, %1, %2\n")
KERNEL(k_add_lshl_u32, "v_add_lshl_u32 %0, %1, %2, 3\n")
KERNEL(k_xad_u32, "v_xad_u32 %0, %1, %2, %3\n")
KERNEL(k_or3_b32, "v_or3_b32 %0, %1, %2, %3\n")
KERNEL(k_sub_e64, "v_sub_u32_e64 %0, %1, %2\n")
KERNEL(k_add_sdwa, "v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n")
KERNEL(k_mix_add_mad, "v_add_u32 %0, %1, %2\n v_mad_u64_u32 %4, s[10:11], %1, %2, %5\n")
KERNEL(k_mix_3add_mad, "v_add_u32 %0, %1, %2\n v_add_u32 %3, %1, %2\n v_xor_b32 %0, %1, %2\n v_mad_u64_u32 %4, s[10:11], %1, %2, %5\n")
KERNEL(k_salu, "s_add_u32 s10, s10, 3\n")
KERNEL(k_mix_salu_valu, "s_add_u32 s10, s10, 3\n v_mad_u32_u24 %0, %1, %2, %3\n")
KERNEL(k_mix_salu_fast, "s_add_u32 s10, s10, 3\n v_add_u32 %0, %1, %2\n")

// DEPENDENT chains: every instruction reads the previous one's result -- one wave alone then shows the result latency, which is
// what a latency chain (a lone wave walking a sponge) pays per instruction
KERNEL(k_dep_mad_u64_acc, "v_mad_u64_u32 %4, s[10:11], %1, %2, %4\n")
KERNEL(k_dep_lshl_add_u64, "v_lshl_add_u64 %4, %4, 0, %5\n")
KERNEL(k_dep_add_u32, "v_add_u32 %0, %0, %1\n")
KERNEL(k_dep_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1\n")
KERNEL(k_dep_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2\n")
KERNEL(k_dep_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_dep_two_chains, "v_mad_u64_u32 %4, s[10:11], %1, %2, %4\n v_mad_u64_u32 %5, s[10:11], %1, %3, %5\n")

typedef void (*kern_t)(unsigned*, unsigned);
struct Entry {
    const char* name;
    kern_t k;
    int per_rep;  // instructions per asm repetition
};

int main() {
    Entry tab[] = {{"v_add_u32", k_add_u32, 1},          {"v_add3_u32", k_add3_u32, 1},        {"v_mul_u32_u24", k_mul_u24, 1},
                   {"v_mul_u32_u24_dpp", k_mul_u24_dpp, 1}, {"v_mad_u32_u24", k_mad_u24, 1},     {"v_mad_u64_u32", k_mad_u64_u32, 1},
                   {"v_mul_lo_u32", k_mul_lo_u32, 1},    {"v_mul_hi_u32", k_mul_hi_u32, 1},    {"v_lshl_add_u64", k_lshl_add_u64, 1},
                   {"v_cmp_lt_u64", k_cmp_lt_u64, 1},    {"v_cmp_lt_u32", k_cmp_lt_u32, 1},    {"v_cndmask_b32", k_cndmask, 1},
                   {"v_add_co+v_addc_co", k_add_co_pair, 2}, {"v_lshlrev_b64", k_lshlrev_b64, 1}, {"v_alignbit_b32", k_alignbit, 1},
                   {"v_mov_b32_dpp", k_mov_dpp, 1},      {"v_cmp_lt_u64+v_cndmask", k_cmp_cnd_chain, 2}, {"v_dot4_u32_u8", k_dot4_u8, 1},
                   {"v_pk_add_u16", k_pk_add_u16, 1},
                   {"v_sub_u32", k_sub_u32, 1}, {"v_and_b32", k_and_b32, 1}, {"v_xor_b32", k_xor_b32, 1},
                   {"v_lshlrev_b32", k_lshlrev_b32, 1}, {"v_lshrrev_b32", k_lshrrev_b32, 1}, {"v_ashrrev_i32", k_ashrrev_i32, 1},
                   {"v_mov_b32", k_mov_b32, 1}, {"v_cndmask_b32 (sgpr mask)", k_cndmask_s, 1}, {"v_bfe_u32", k_bfe_u32, 1},
                   {"v_and_or_b32", k_and_or_b32, 1}, {"v_lshl_add_u32", k_lshl_add_u32, 1}, {"v_lshl_or_b32", k_lshl_or_b32, 1},
                   {"v_perm_b32", k_perm_b32, 1}, {"v_bfi_b32", k_bfi_b32, 1}, {"v_min_u32", k_min_u32, 1},
                   {"v_fma_f32", k_fma_f32, 1}, {"v_add_f32", k_add_f32, 1}, {"v_mul_f32", k_mul_f32, 1},
                   {"v_pk_fma_f32", k_pk_fma_f32, 1}, {"v_fma_f64", k_fma_f64, 1}, {"v_cvt_f32_u32", k_cvt_f32_u32, 1},
                   {"v_add_co_u32 (sgpr carry)", k_add_co_e64, 1}, {"v_addc_co_u32", k_addc_only, 1}, {"v_mad_i64_i32", k_mad_i64_i32, 1},
                   {"v_mul_i32_i24", k_mul_i32_i24, 1}, {"v_add_lshl_u32", k_add_lshl_u32, 1}, {"v_xad_u32", k_xad_u32, 1},
                   {"v_or3_b32", k_or3_b32, 1}, {"v_sub_u32_e64", k_sub_e64, 1}, {"v_add_u32_sdwa", k_add_sdwa, 1},
                   {"v_add_u32 + v_mad_u64_u32", k_mix_add_mad, 2}, {"3 simple + v_mad_u64_u32", k_mix_3add_mad, 4},
                   {"dep: v_mad_u64_u32 (acc)", k_dep_mad_u64_acc, 1}, {"dep: v_lshl_add_u64", k_dep_lshl_add_u64, 1}, {"dep: v_add_u32", k_dep_add_u32, 1},
                   {"dep: v_mul_lo_u32", k_dep_mul_lo_u32, 1}, {"dep: v_mad_u32_u24", k_dep_mad_u32_u24, 1}, {"dep: v_mov_b32_dpp", k_dep_mov_dpp, 1},
                   {"dep: 2 chains of mad_u64", k_dep_two_chains, 2},
                   {"s_add_u32", k_salu, 1}, {"s_add_u32 + v_mad_u32_u24", k_mix_salu_valu, 2}, {"s_add_u32 + v_add_u32", k_mix_salu_fast, 2}};
    setvbuf(stdout, NULL, _IOLBF, 0);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    printf("device %s, %d CUs, %.2f GHz\n", prop.name, cus, ghz);
    unsigned* out;
    hipMalloc(&out, (size_t)cus * 16 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-26s %10s %10s %10s   (cycles per wave-instruction per SIMD)\n", "instruction", "1 wave", "2 waves", "4 waves");
    for (auto& en : tab) {
        printf("%-26s", en.name);
        for (int wps : {1, 2, 4}) {
            const int blocks = cus * wps;  // 256 threads = 4 waves, one per SIMD
            en.k<<<blocks, 256>>>(out, 1);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            en.k<<<blocks, 256>>>(out, 2);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)ITER * 32 * en.per_rep * wps;
            printf(" %10.2f", ms * 1e-3 * ghz * 1e9 / instr_per_simd);
        }
        printf("\n");
    }
    return 0;
}
