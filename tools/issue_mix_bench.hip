// Issue-slot microbenchmark for instruction MIXES (gfx950): what a SIMD's waves can overlap and what they cannot.
// tools/valu_rate_bench.hip measures one opcode at a time; the quotient kernel (csrc/kernels_quotient.hip) is a mix of
// v_mad_u64_u32, scalar bookkeeping, LDS reads (a per-lane ds_read_b64 for the cell, two BROADCAST ds_read_b128 for the record),
// waits and branches.  Every row runs a block of `mads` multiply-adds plus the named extras in all four SIMDs of every CU at
// 1 / 2 / 4 waves per SIMD and prints cycles per block per SIMD; a block of 12 multiply-adds alone costs 12 x 4.2 = 50 cycles,
// so a row that stays at 50 with more than one wave per SIMD overlaps its extras completely.
// Build: hipcc --offload-arch=gfx950 -O2 -o build/issue_mix_bench tools/issue_mix_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define ITER 4000
#define REP8(x) x x x x x x x x

// twelve multiply-adds into six 64-bit sums, operands in the asm's own registers (two records' worth share the sums)
#define MAD12                                            \
    "v_mad_u64_u32 v[100:101], s[10:11], %0, %1, v[100:101]\n" \
    "v_mad_u64_u32 v[102:103], s[10:11], %0, %2, v[102:103]\n" \
    "v_mad_u64_u32 v[104:105], s[10:11], %0, %3, v[104:105]\n" \
    "v_mad_u64_u32 v[106:107], s[10:11], %1, %2, v[106:107]\n" \
    "v_mad_u64_u32 v[108:109], s[10:11], %1, %3, v[108:109]\n" \
    "v_mad_u64_u32 v[110:111], s[10:11], %2, %3, v[110:111]\n" \
    "v_mad_u64_u32 v[112:113], s[10:11], %0, %1, v[112:113]\n" \
    "v_mad_u64_u32 v[114:115], s[10:11], %0, %2, v[114:115]\n" \
    "v_mad_u64_u32 v[116:117], s[10:11], %0, %3, v[116:117]\n" \
    "v_mad_u64_u32 v[118:119], s[10:11], %1, %2, v[118:119]\n" \
    "v_mad_u64_u32 v[120:121], s[10:11], %1, %3, v[120:121]\n" \
    "v_mad_u64_u32 v[122:123], s[10:11], %2, %3, v[122:123]\n"
#define MAD4                                             \
    "v_mad_u64_u32 v[100:101], s[10:11], %0, %1, v[100:101]\n" \
    "v_mad_u64_u32 v[102:103], s[10:11], %0, %2, v[102:103]\n" \
    "v_mad_u64_u32 v[104:105], s[10:11], %0, %3, v[104:105]\n" \
    "v_mad_u64_u32 v[106:107], s[10:11], %1, %2, v[106:107]\n"

// %4 = per-lane LDS byte address (lane * 8, conflict free), %5 = one address for the whole wave (broadcast)
#define CLOBBERS                                                                                                                   \
    "vcc", "scc", "s10", "s11", "s12", "s13", "s14", "s15", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", \
        "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107",  \
        "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", \
        "memory"

#define KERNEL(name, body)                                                                                   \
    __global__ __launch_bounds__(256) void name(unsigned* out, unsigned seed) {                              \
        __shared__ unsigned lds[8192];                                                                       \
        for (unsigned i = threadIdx.x; i < 8192; i += 256) lds[i] = i * seed;                                \
        __syncthreads();                                                                                     \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 ^ 0x55;                  \
        unsigned lane_addr = (unsigned)(uintptr_t)lds + (threadIdx.x & 63u) * 8u + (threadIdx.x >> 6) * 4096u; \
        unsigned wave_addr = (unsigned)(uintptr_t)lds + (threadIdx.x >> 6) * 4096u + 2048u;                  \
        unsigned lane16_addr = (unsigned)(uintptr_t)lds + (threadIdx.x & 63u) * 16u + (threadIdx.x >> 6) * 4096u; \
        for (int it = 0; it < ITER; it++) {                                                                  \
            asm volatile(REP8(body) "s_waitcnt lgkmcnt(0)\n"                                                 \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                                            \
                         : "v"(lane_addr), "v"(wave_addr), "v"(lane16_addr)                                  \
                         : CLOBBERS);                                                                        \
        }                                                                                                    \
        unsigned r;                                                                                          \
        asm volatile("v_xor_b32 %0, v100, v122\n v_xor_b32 %0, %0, v64\n v_xor_b32 %0, %0, v72" : "=v"(r)::CLOBBERS); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + r;                                  \
    }

#define S3 "s_add_u32 s12, s12, 3\n s_and_b32 s13, s12, 15\n s_lshl_b32 s14, s13, 2\n"
#define CELL "ds_read_b64 v[64:65], %4\n"
#define REC "ds_read_b128 v[68:71], %5\n ds_read_b128 v[72:75], %5 offset:16\n"

KERNEL(k_mad12, MAD12)
KERNEL(k_mad12_s3, MAD12 S3)
KERNEL(k_mad12_s6, MAD12 S3 S3)
KERNEL(k_mad12_s12, MAD12 S3 S3 S3 S3)
KERNEL(k_mad12_cell, MAD12 CELL)
KERNEL(k_mad12_rec, MAD12 REC)
KERNEL(k_mad12_cell_rec, MAD12 CELL REC)
KERNEL(k_mad12_cell_rec_s3, MAD12 CELL REC S3)
KERNEL(k_mad12_rec_b64, MAD12 "ds_read_b64 v[68:69], %5\n ds_read_b64 v[70:71], %5 offset:8\n ds_read_b64 v[72:73], %5 offset:16\n ds_read_b64 v[74:75], %5 offset:24\n")
KERNEL(k_mad12_rec_b32x8, MAD12 "ds_read2_b32 v[68:69], %5 offset1:1\n ds_read2_b32 v[70:71], %5 offset0:2 offset1:3\n ds_read2_b32 v[72:73], %5 offset0:4 offset1:5\n ds_read2_b32 v[74:75], %5 offset0:6 offset1:7\n")
KERNEL(k_cell_only, CELL)
KERNEL(k_rec_only, REC)
KERNEL(k_b128_lane_only, "ds_read_b128 v[68:71], %6\n")
KERNEL(k_b128_bcast_only, "ds_read_b128 v[68:71], %5\n")
KERNEL(k_b64_bcast_only, "ds_read_b64 v[68:69], %5\n")
KERNEL(k_b32_bcast_only, "ds_read_b32 v68, %5\n")
KERNEL(k_mad12_branch, MAD12 "s_branch 0\n")  // a taken branch to the next instruction
KERNEL(k_mad12_cbranch_nt, MAD12 "s_cmp_eq_u32 s12, 77\n s_cbranch_scc1 0\n")  // compare + not-taken branch
KERNEL(k_mad12_cbranch_t, MAD12 "s_cmp_lg_u32 s12, 77\n s_cbranch_scc1 0\n")   // compare + taken branch
KERNEL(k_mad12_rfl, MAD12 "v_readfirstlane_b32 s12, %0\n")
KERNEL(k_mad12_rfl_use, MAD12 "v_readfirstlane_b32 s12, %0\n s_and_b32 s13, s12, 15\n s_cmp_eq_u32 s13, 99\n s_cbranch_scc1 0\n")
KERNEL(k_mad12_wait, MAD12 "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_mad12_nop, MAD12 "s_nop 0\n")
KERNEL(k_mad12_nop4, MAD12 "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
KERNEL(k_mad12_mov4, MAD12 "v_mov_b64 v[76:77], v[64:65]\n v_mov_b64 v[78:79], v[68:69]\n v_mov_b64 v[80:81], v[70:71]\n v_mov_b64 v[82:83], v[72:73]\n")
KERNEL(k_mad12_sdwa, MAD12 "v_add_u32_sdwa v76, %4, v68 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n")
// the plain record as the kernel issues it: address add, cell read, record read, 12 multiply-adds, a wait, two scalar instructions
KERNEL(k_plain_record,
       "v_add_u32_sdwa v76, %4, v68 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n" CELL MAD12 "s_add_i32 s12, s12, 4\n s_cmp_gt_u32 s12, 3\n"
       "s_waitcnt lgkmcnt(2)\n" REC)
// ... and with the record's eight dwords as SCALAR operands (no record read from LDS at all)
KERNEL(k_plain_record_sgpr_weights,
       "v_add_u32_sdwa v76, %4, v68 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n" CELL MAD12 "s_add_i32 s12, s12, 4\n s_cmp_gt_u32 s12, 3\n"
       "s_waitcnt lgkmcnt(0)\n")
// carry chains of the fold (piece end): v_add_co / v_addc_co through vcc and through an SGPR pair with the s_nop the hazard asks for
KERNEL(k_fold_like, MAD4 "v_add_co_u32 v76, vcc, %0, %1\n v_addc_co_u32 v77, vcc, %2, %3, vcc\n v_sub_co_u32_e64 v78, s[14:15], %0, %1\n s_nop 1\n v_subb_co_u32_e64 v79, s[14:15], %2, %3, s[14:15]\n")

typedef void (*kern_t)(unsigned*, unsigned);
struct Entry {
    const char* name;
    kern_t k;
    int mads, other;  // multiply-adds and other instructions per block
};

int main() {
    Entry tab[] = {
        {"12 mad", k_mad12, 12, 0},
        {"12 mad + 3 salu", k_mad12_s3, 12, 3},
        {"12 mad + 6 salu", k_mad12_s6, 12, 6},
        {"12 mad + 12 salu", k_mad12_s12, 12, 12},
        {"12 mad + cell (ds_read_b64 per lane)", k_mad12_cell, 12, 1},
        {"12 mad + record (2 bcast ds_read_b128)", k_mad12_rec, 12, 2},
        {"12 mad + cell + record", k_mad12_cell_rec, 12, 3},
        {"12 mad + cell + record + 3 salu", k_mad12_cell_rec_s3, 12, 6},
        {"12 mad + record as 4 bcast ds_read_b64", k_mad12_rec_b64, 12, 4},
        {"12 mad + record as 4 bcast ds_read2_b32", k_mad12_rec_b32x8, 12, 4},
        {"cell read alone", k_cell_only, 0, 1},
        {"record read alone (2 bcast b128)", k_rec_only, 0, 2},
        {"ds_read_b128 per lane alone", k_b128_lane_only, 0, 1},
        {"ds_read_b128 broadcast alone", k_b128_bcast_only, 0, 1},
        {"ds_read_b64 broadcast alone", k_b64_bcast_only, 0, 1},
        {"ds_read_b32 broadcast alone", k_b32_bcast_only, 0, 1},
        {"12 mad + s_branch (taken)", k_mad12_branch, 12, 1},
        {"12 mad + s_cmp + s_cbranch not taken", k_mad12_cbranch_nt, 12, 2},
        {"12 mad + s_cmp + s_cbranch taken", k_mad12_cbranch_t, 12, 2},
        {"12 mad + v_readfirstlane", k_mad12_rfl, 12, 1},
        {"12 mad + v_readfirstlane -> s_and, s_cmp, s_cbranch", k_mad12_rfl_use, 12, 4},
        {"12 mad + s_waitcnt (nothing pending)", k_mad12_wait, 12, 1},
        {"12 mad + s_nop", k_mad12_nop, 12, 1},
        {"12 mad + 4 s_nop", k_mad12_nop4, 12, 4},
        {"12 mad + 4 v_mov_b64", k_mad12_mov4, 12, 4},
        {"12 mad + v_add_u32_sdwa", k_mad12_sdwa, 12, 1},
        {"plain record (sdwa, cell, 12 mad, 2 salu, wait, record)", k_plain_record, 12, 7},
        {"plain record, weights not from LDS", k_plain_record_sgpr_weights, 12, 5},
        {"4 mad + 2 carry pairs (vcc; sgpr + s_nop 1)", k_fold_like, 4, 5},
    };
    setvbuf(stdout, NULL, _IOLBF, 0);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    printf("device %s, %d CUs, %.2f GHz; cycles per block per SIMD (all four SIMDs of every CU busy; the LDS is the CU's)\n", prop.name, cus, ghz);
    unsigned* out;
    hipMalloc(&out, (size_t)cus * 16 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-58s %5s %9s %9s %9s\n", "block", "instr", "1 wave", "2 waves", "4 waves");
    for (auto& en : tab) {
        printf("%-58s %5d", en.name, en.mads + en.other);
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
            const double blocks_per_simd = (double)ITER * 8 * wps;
            printf(" %9.1f", ms * 1e-3 * ghz * 1e9 / blocks_per_simd);
        }
        printf("\n");
    }
    return 0;
}
