// Micro-benchmark (round 5): issue cost of the VALU / DPP / LDS-crossbar instructions the LK tracker is made of, on gfx950.
// Each test runs UNROLL copies of one instruction per loop iteration on 8 independent register chains (throughput) or on one chain
// (latency), one wave per SIMD (grid = 256 CUs x 4 waves) and, for throughput, also 4 waves per SIMD.  Reported: cycles per wave-instruction
// from s_memtime-independent wall time (hipEvents) at the clock the chip sustains for a plain v_add_u32 stream (normalised: v_add_u32 = 2.0).
//   hipcc --offload-arch=gfx950 -O2 -o valu_cost valu_cost.hip && ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <cstdlib>

#define ITERS 20000

#define CHAIN8(OP)  OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY32(OP) CHAIN8(OP) CHAIN8(OP) CHAIN8(OP) CHAIN8(OP)

#define DEF_TEST(NAME, ASM_T, ASM_L)                                                                                        \
    __global__ __launch_bounds__(256) void t_##NAME(unsigned *out, int lat) {                                              \
        unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
        unsigned b = 0x00030001u + threadIdx.x, c = 0x00010002u;                                                           \
        unsigned long long d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;                         \
        typedef unsigned u4 __attribute__((ext_vector_type(4)));                                                           \
        u4 q0 = {a0, a1, a2, a3}, q1 = q0, q2 = q0, q3 = q0; (void) q0; (void) q1; (void) q2; (void) q3;                       \
        __shared__ unsigned lds_words[1024];                                                                                \
        lds_words[threadIdx.x] = a0;                                                                                       \
        __syncthreads();                                                                                                   \
        (void) d0; (void) d1; (void) d2; (void) d3; (void) d4; (void) d5; (void) d6; (void) d7;                            \
        if (lat) {                                                                                                         \
            for (int i = 0; i < ITERS; i++) { ASM_L }                                                                      \
        } else {                                                                                                           \
            for (int i = 0; i < ITERS; i++) { ASM_T }                                                                      \
        }                                                                                                                  \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned) (d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7); \
    }

// 32 instructions per loop body: 8 chains x 4 (throughput) or chain 0 x 32 (latency)
#define T32(FMT, ...) asm volatile(FMT FMT FMT FMT : __VA_ARGS__);
#define V8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define D8 "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)

#define VOP_T(INS) T32(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n", V8 : "v"(b))
#define VOP_L(INS) T32(INS " %0, %0, %8\n" INS " %0, %0, %8\n" INS " %0, %0, %8\n" INS " %0, %0, %8\n" INS " %0, %0, %8\n" INS " %0, %0, %8\n" INS " %0, %0, %8\n" INS " %0, %0, %8\n", V8 : "v"(b))
#define VOP3_T(INS) T32(INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n" INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9\n", V8 : "v"(b), "v"(c))
#define VOP3_L(INS) T32(INS " %0, %0, %8, %9\n" INS " %0, %0, %8, %9\n" INS " %0, %0, %8, %9\n" INS " %0, %0, %8, %9\n" INS " %0, %0, %8, %9\n" INS " %0, %0, %8, %9\n" INS " %0, %0, %8, %9\n" INS " %0, %0, %8, %9\n", V8 : "v"(b), "v"(c))
// accumulate form: d = op(b, c, d)
#define ACC_T(INS) T32(INS " %0, %8, %9, %0\n" INS " %1, %8, %9, %1\n" INS " %2, %8, %9, %2\n" INS " %3, %8, %9, %3\n" INS " %4, %8, %9, %4\n" INS " %5, %8, %9, %5\n" INS " %6, %8, %9, %6\n" INS " %7, %8, %9, %7\n", V8 : "v"(b), "v"(c))
#define ACC_L(INS) T32(INS " %0, %8, %9, %0\n" INS " %0, %8, %9, %0\n" INS " %0, %8, %9, %0\n" INS " %0, %8, %9, %0\n" INS " %0, %8, %9, %0\n" INS " %0, %8, %9, %0\n" INS " %0, %8, %9, %0\n" INS " %0, %8, %9, %0\n", V8 : "v"(b), "v"(c))
#define ACC2_T(INS) T32(INS " %0, %8, %9\n" INS " %1, %8, %9\n" INS " %2, %8, %9\n" INS " %3, %8, %9\n" INS " %4, %8, %9\n" INS " %5, %8, %9\n" INS " %6, %8, %9\n" INS " %7, %8, %9\n", V8 : "v"(b), "v"(c))
#define ACC2_L(INS) T32(INS " %0, %8, %9\n" INS " %0, %8, %9\n" INS " %0, %8, %9\n" INS " %0, %8, %9\n" INS " %0, %8, %9\n" INS " %0, %8, %9\n" INS " %0, %8, %9\n" INS " %0, %8, %9\n", V8 : "v"(b), "v"(c))
#define VOP1_T(INS) T32(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n" INS " %4, %4\n" INS " %5, %5\n" INS " %6, %6\n" INS " %7, %7\n", V8 : "v"(b))
#define VOP1_L(INS) T32(INS " %0, %0\n" INS " %0, %0\n" INS " %0, %0\n" INS " %0, %0\n" INS " %0, %0\n" INS " %0, %0\n" INS " %0, %0\n" INS " %0, %0\n", V8 : "v"(b))
#define DPP_T(CTRL) T32("v_add_u32_dpp %0, %0, %0 " CTRL "\n v_add_u32_dpp %1, %1, %1 " CTRL "\n v_add_u32_dpp %2, %2, %2 " CTRL "\n v_add_u32_dpp %3, %3, %3 " CTRL "\n v_add_u32_dpp %4, %4, %4 " CTRL "\n v_add_u32_dpp %5, %5, %5 " CTRL "\n v_add_u32_dpp %6, %6, %6 " CTRL "\n v_add_u32_dpp %7, %7, %7 " CTRL "\n", V8 : "v"(b))
#define DPP_L(CTRL) T32("v_add_u32_dpp %0, %0, %0 " CTRL "\n s_nop 1\n v_add_u32_dpp %0, %0, %0 " CTRL "\n s_nop 1\n v_add_u32_dpp %0, %0, %0 " CTRL "\n s_nop 1\n v_add_u32_dpp %0, %0, %0 " CTRL "\n s_nop 1\n v_add_u32_dpp %0, %0, %0 " CTRL "\n s_nop 1\n v_add_u32_dpp %0, %0, %0 " CTRL "\n s_nop 1\n v_add_u32_dpp %0, %0, %0 " CTRL "\n s_nop 1\n v_add_u32_dpp %0, %0, %0 " CTRL "\n s_nop 1\n", V8 : "v"(b))
// 64-bit
#define D1_T(INS) T32(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n" INS " %4, %4\n" INS " %5, %5\n" INS " %6, %6\n" INS " %7, %7\n", D8 : "v"(b))
#define D2_T(INS) T32(INS " %0, %0, %0\n" INS " %1, %1, %1\n" INS " %2, %2, %2\n" INS " %3, %3, %3\n" INS " %4, %4, %4\n" INS " %5, %5, %5\n" INS " %6, %6, %6\n" INS " %7, %7, %7\n", D8 : "v"(b))
#define D2_L(INS) T32(INS " %0, %0, %0\n" INS " %0, %0, %0\n" INS " %0, %0, %0\n" INS " %0, %0, %0\n" INS " %0, %0, %0\n" INS " %0, %0, %0\n" INS " %0, %0, %0\n" INS " %0, %0, %0\n", D8 : "v"(b))
// f64 <- i32 and f32 <- f64 conversions (mixed widths)
#define CVT64_T T32("v_cvt_f64_i32 %0, %8\n v_cvt_f64_i32 %1, %8\n v_cvt_f64_i32 %2, %8\n v_cvt_f64_i32 %3, %8\n v_cvt_f64_i32 %4, %8\n v_cvt_f64_i32 %5, %8\n v_cvt_f64_i32 %6, %8\n v_cvt_f64_i32 %7, %8\n", D8 : "v"(b))
#define CVT32_T T32("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %8\n v_cvt_f32_f64 %2, %8\n v_cvt_f32_f64 %3, %8\n v_cvt_f32_f64 %4, %8\n v_cvt_f32_f64 %5, %8\n v_cvt_f32_f64 %6, %8\n v_cvt_f32_f64 %7, %8\n", V8 : "v"(d0))
#define RDL_T T32("v_readlane_b32 s20, %0, 63\n v_readlane_b32 s21, %1, 63\n v_readlane_b32 s22, %2, 63\n v_readlane_b32 s23, %3, 63\n v_readlane_b32 s24, %4, 63\n v_readlane_b32 s25, %5, 63\n v_readlane_b32 s26, %6, 63\n v_readlane_b32 s27, %7, 63\n", V8 : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
#define SWAP_T(INS) T32(INS " %0, %1\n" INS " %2, %3\n" INS " %4, %5\n" INS " %6, %7\n" INS " %0, %1\n" INS " %2, %3\n" INS " %4, %5\n" INS " %6, %7\n", V8 : "v"(b))
#define BPERM_T T32("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n", V8 : "v"(b))
#define NOP_T T32("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n", V8 : "v"(b))
#define SALU_T T32("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1\n", V8 : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc")
// a VALU stream with SALU interleaved 1:1 (does SALU steal VALU issue slots of the same wave?)
#define MIX_T T32("v_add_u32 %0, %0, %8\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %8\n s_add_u32 s21, s21, 1\n v_add_u32 %2, %2, %8\n s_add_u32 s22, s22, 1\n v_add_u32 %3, %3, %8\n s_add_u32 s23, s23, 1\n", V8 : "v"(b) : "s20", "s21", "s22", "s23", "scc")

DEF_TEST(add_u32, VOP_T("v_add_u32"), VOP_L("v_add_u32"))
DEF_TEST(mul_f32, VOP_T("v_mul_f32"), VOP_L("v_mul_f32"))
DEF_TEST(and_b32, VOP_T("v_and_b32"), VOP_L("v_and_b32"))
DEF_TEST(ashr_i32, VOP_T("v_ashrrev_i32"), VOP_L("v_ashrrev_i32"))
DEF_TEST(mul_lo_u32, VOP_T("v_mul_lo_u32"), VOP_L("v_mul_lo_u32"))
DEF_TEST(mad_i32_i24, VOP3_T("v_mad_i32_i24"), VOP3_L("v_mad_i32_i24"))
DEF_TEST(add3_u32, VOP3_T("v_add3_u32"), VOP3_L("v_add3_u32"))
DEF_TEST(perm_b32, VOP3_T("v_perm_b32"), VOP3_L("v_perm_b32"))
DEF_TEST(alignbit, VOP3_T("v_alignbit_b32"), VOP3_L("v_alignbit_b32"))
DEF_TEST(bfi_b32, VOP3_T("v_bfi_b32"), VOP3_L("v_bfi_b32"))
DEF_TEST(lshl_or, VOP3_T("v_lshl_or_b32"), VOP3_L("v_lshl_or_b32"))
DEF_TEST(dot2_i32_i16, ACC_T("v_dot2_i32_i16"), ACC_L("v_dot2_i32_i16"))
DEF_TEST(dot2c_i32_i16, ACC2_T("v_dot2c_i32_i16"), ACC2_L("v_dot2c_i32_i16"))
DEF_TEST(dot4_i32_i8, ACC_T("v_dot4_i32_i8"), ACC_L("v_dot4_i32_i8"))
DEF_TEST(pk_add_u16, VOP_T("v_pk_add_u16"), VOP_L("v_pk_add_u16"))
DEF_TEST(pk_mul_lo_u16, VOP_T("v_pk_mul_lo_u16"), VOP_L("v_pk_mul_lo_u16"))
DEF_TEST(pk_mad_u16, VOP3_T("v_pk_mad_u16"), VOP3_L("v_pk_mad_u16"))
DEF_TEST(pk_add_f32, D2_T("v_pk_add_f32"), D2_L("v_pk_add_f32"))
DEF_TEST(pk_mul_f32, D2_T("v_pk_mul_f32"), D2_L("v_pk_mul_f32"))
DEF_TEST(rndne_f32, VOP1_T("v_rndne_f32"), VOP1_L("v_rndne_f32"))
DEF_TEST(floor_f32, VOP1_T("v_floor_f32"), VOP1_L("v_floor_f32"))
DEF_TEST(cvt_i32_f32, VOP1_T("v_cvt_i32_f32"), VOP1_L("v_cvt_i32_f32"))
DEF_TEST(cvt_f32_i32, VOP1_T("v_cvt_f32_i32"), VOP1_L("v_cvt_f32_i32"))
DEF_TEST(rcp_f32, VOP1_T("v_rcp_f32"), VOP1_L("v_rcp_f32"))
DEF_TEST(sqrt_f32, VOP1_T("v_sqrt_f32"), VOP1_L("v_sqrt_f32"))
DEF_TEST(mov_b32, VOP1_T("v_mov_b32"), VOP1_L("v_mov_b32"))
DEF_TEST(add_f64, D2_T("v_add_f64"), D2_L("v_add_f64"))
DEF_TEST(mul_f64, D2_T("v_mul_f64"), D2_L("v_mul_f64"))
DEF_TEST(cvt_f64_i32, CVT64_T, CVT64_T)
DEF_TEST(cvt_f32_f64, CVT32_T, CVT32_T)
DEF_TEST(readlane, RDL_T, RDL_T)
DEF_TEST(dpp_quad, DPP_T("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"), DPP_L("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"))
DEF_TEST(dpp_row_mirror, DPP_T("row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1"), DPP_L("row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1"))
DEF_TEST(dpp_row_bcast15, DPP_T("row_bcast:15 row_mask:0xa bank_mask:0xf"), DPP_L("row_bcast:15 row_mask:0xa bank_mask:0xf"))
DEF_TEST(dpp_row_bcast31, DPP_T("row_bcast:31 row_mask:0xc bank_mask:0xf"), DPP_L("row_bcast:31 row_mask:0xc bank_mask:0xf"))
DEF_TEST(permlane32_swap, SWAP_T("v_permlane32_swap_b32"), SWAP_T("v_permlane32_swap_b32"))
DEF_TEST(permlane16_swap, SWAP_T("v_permlane16_swap_b32"), SWAP_T("v_permlane16_swap_b32"))
DEF_TEST(ds_bpermute, BPERM_T, BPERM_T)
DEF_TEST(s_nop0, NOP_T, NOP_T)
DEF_TEST(salu_add, SALU_T, SALU_T)
DEF_TEST(valu_salu_mix, MIX_T, MIX_T)

// the same streams with part of the wave switched off: does a SIMD-32 skip the pass of an all-inactive half (or quarter)?
#define DEF_EXEC_TEST(NAME, MASK_LO, MASK_HI, ASM_T)                                                                        \
    __global__ __launch_bounds__(256) void t_##NAME(unsigned *out, int lat) {                                              \
        unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
        unsigned b = 0x00030001u + threadIdx.x, c = 0x00010002u;                                                           \
        unsigned long long d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;                         \
        unsigned *optr = out + (blockIdx.x * 256 + threadIdx.x);                                                            \
        asm volatile("" : "+v"(optr), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b), "+v"(c)); \
        asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));                 \
        unsigned long long saved_exec;                                                                                     \
        asm volatile("s_mov_b64 %0, exec\n s_mov_b32 exec_lo, " MASK_LO "\n s_mov_b32 exec_hi, " MASK_HI "\n" : "=s"(saved_exec)); \
        for (int i = 0; i < ITERS; i++) { ASM_T }                                                                          \
        asm volatile("s_mov_b64 exec, %0\n" :: "s"(saved_exec));                                                          \
        *optr = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned) (d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7); \
    }
DEF_EXEC_TEST(add_lo32, "0xffffffff", "0", VOP_T("v_add_u32"))
DEF_EXEC_TEST(add_hi32, "0", "0xffffffff", VOP_T("v_add_u32"))
DEF_EXEC_TEST(add_lo16, "0xffff", "0", VOP_T("v_add_u32"))
DEF_EXEC_TEST(add_lane0, "1", "0", VOP_T("v_add_u32"))
DEF_EXEC_TEST(dot2_lo32, "0xffffffff", "0", ACC_T("v_dot2_i32_i16"))
DEF_EXEC_TEST(dot2_lo16, "0xffff", "0", ACC_T("v_dot2_i32_i16"))
DEF_EXEC_TEST(dot2_lane0, "1", "0", ACC_T("v_dot2_i32_i16"))
DEF_EXEC_TEST(cvt_lane0, "1", "0", VOP1_T("v_cvt_f32_i32"))
DEF_EXEC_TEST(mulf64_lane0, "1", "0", D2_T("v_mul_f64"))
DEF_EXEC_TEST(pkmulf32_lane0, "1", "0", D2_T("v_pk_mul_f32"))
DEF_EXEC_TEST(rcp_lane0, "1", "0", VOP1_T("v_rcp_f32"))
DEF_EXEC_TEST(dpp_lo32, "0xffffffff", "0", DPP_T("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"))

DEF_TEST(fma_f32, VOP3_T("v_fma_f32"), VOP3_L("v_fma_f32"))
DEF_TEST(sub_f32, VOP_T("v_sub_f32"), VOP_L("v_sub_f32"))
DEF_TEST(add_f32, VOP_T("v_add_f32"), VOP_L("v_add_f32"))
DEF_TEST(max_f32, VOP_T("v_max_f32"), VOP_L("v_max_f32"))
DEF_TEST(lshlrev_b32, VOP_T("v_lshlrev_b32"), VOP_L("v_lshlrev_b32"))
DEF_TEST(or_b32, VOP_T("v_or_b32"), VOP_L("v_or_b32"))
DEF_TEST(xor_b32, VOP_T("v_xor_b32"), VOP_L("v_xor_b32"))
DEF_TEST(sub_u32, VOP_T("v_sub_u32"), VOP_L("v_sub_u32"))
DEF_TEST(min_u32, VOP_T("v_min_u32"), VOP_L("v_min_u32"))
DEF_TEST(max_i32, VOP_T("v_max_i32"), VOP_L("v_max_i32"))
DEF_TEST(mul_u32_u24, VOP_T("v_mul_u32_u24"), VOP_L("v_mul_u32_u24"))
DEF_TEST(mul_i32_i24, VOP_T("v_mul_i32_i24"), VOP_L("v_mul_i32_i24"))
DEF_TEST(lshl_add_u32, VOP3_T("v_lshl_add_u32"), VOP3_L("v_lshl_add_u32"))
DEF_TEST(and_or_b32, VOP3_T("v_and_or_b32"), VOP3_L("v_and_or_b32"))
DEF_TEST(bfe_u32, VOP3_T("v_bfe_u32"), VOP3_L("v_bfe_u32"))
DEF_TEST(mad_u32_u24, VOP3_T("v_mad_u32_u24"), VOP3_L("v_mad_u32_u24"))
DEF_TEST(ldexp_f32, VOP_T("v_ldexp_f32"), VOP_L("v_ldexp_f32"))
DEF_TEST(add_u16, VOP_T("v_add_u16"), VOP_L("v_add_u16"))
DEF_TEST(mul_lo_u16, VOP_T("v_mul_lo_u16"), VOP_L("v_mul_lo_u16"))
DEF_TEST(pk_fma_f32, T32("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7\n", D8 : "v"(b)), T32("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %0, %0, %0, %0\n", D8 : "v"(b)))
DEF_TEST(fma_f64, T32("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n v_fma_f64 %4, %4, %4, %4\n v_fma_f64 %5, %5, %5, %5\n v_fma_f64 %6, %6, %6, %6\n v_fma_f64 %7, %7, %7, %7\n", D8 : "v"(b)), T32("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %0, %0, %0, %0\n", D8 : "v"(b)))
#define CMP_T(INS) T32(INS " vcc, %0, %8\n" INS " vcc, %1, %8\n" INS " vcc, %2, %8\n" INS " vcc, %3, %8\n" INS " vcc, %4, %8\n" INS " vcc, %5, %8\n" INS " vcc, %6, %8\n" INS " vcc, %7, %8\n", V8 : "v"(b) : "vcc")
DEF_TEST(cmp_lt_f32, CMP_T("v_cmp_lt_f32"), CMP_T("v_cmp_lt_f32"))
DEF_TEST(cmp_lt_u32, CMP_T("v_cmp_lt_u32"), CMP_T("v_cmp_lt_u32"))
DEF_TEST(cmp_lt_f64, T32("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %2, %3\n v_cmp_lt_f64 vcc, %4, %5\n v_cmp_lt_f64 vcc, %6, %7\n v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %2, %3\n v_cmp_lt_f64 vcc, %4, %5\n v_cmp_lt_f64 vcc, %6, %7\n", D8 : "v"(b) : "vcc"), VOP_T("v_add_u32"))
DEF_TEST(cndmask, T32("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n", V8 : "v"(b) : "vcc"), VOP_T("v_add_u32"))
DEF_TEST(cvt_f64_f32, T32("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %8\n v_cvt_f64_f32 %2, %8\n v_cvt_f64_f32 %3, %8\n v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %8\n v_cvt_f64_f32 %6, %8\n v_cvt_f64_f32 %7, %8\n", D8 : "v"(b)), VOP_T("v_add_u32"))
DEF_TEST(readfirstlane, T32("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n v_readfirstlane_b32 s24, %4\n v_readfirstlane_b32 s25, %5\n v_readfirstlane_b32 s26, %6\n v_readfirstlane_b32 s27, %7\n", V8 : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"), VOP_T("v_add_u32"))
DEF_TEST(sdwa_add, T32("v_add_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n", V8 : "v"(b)), VOP_T("v_add_u32"))
DEF_TEST(mov_dpp, T32("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n", V8 : "v"(b)), VOP_T("v_add_u32"))
DEF_TEST(ds_read_b32, T32("ds_read_b32 %0, %8\n ds_read_b32 %1, %8\n ds_read_b32 %2, %8\n ds_read_b32 %3, %8\n ds_read_b32 %4, %8\n ds_read_b32 %5, %8\n ds_read_b32 %6, %8\n ds_read_b32 %7, %8\n s_waitcnt lgkmcnt(0)\n", V8 : "v"(b & 1020)), VOP_T("v_add_u32"))
DEF_TEST(ds_read_b128, T32("ds_read_b128 %0, %8\n ds_read_b128 %1, %8\n ds_read_b128 %2, %8\n ds_read_b128 %3, %8\n s_waitcnt lgkmcnt(0)\n ds_read_b128 %0, %8\n ds_read_b128 %1, %8\n ds_read_b128 %2, %8\n ds_read_b128 %3, %8\n s_waitcnt lgkmcnt(0)\n", "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(b & 1008), "v"(b), "v"(b), "v"(b), "v"(b)), VOP_T("v_add_u32"))
// VALU stream with LDS reads interleaved (do LDS instructions take VALU issue slots?)
DEF_TEST(valu_ds_mix, T32("v_add_u32 %0, %0, %8\n ds_read_b32 %4, %9\n v_add_u32 %1, %1, %8\n ds_read_b32 %5, %9\n v_add_u32 %2, %2, %8\n ds_read_b32 %6, %9\n v_add_u32 %3, %3, %8\n ds_read_b32 %7, %9\n s_waitcnt lgkmcnt(0)\n", V8 : "v"(b), "v"(b & 1020)), VOP_T("v_add_u32"))

DEF_EXEC_TEST(dot2_m24, "0xffffff", "0", ACC_T("v_dot2_i32_i16"))
DEF_EXEC_TEST(dot2_m17, "0x1ffff", "0", ACC_T("v_dot2_i32_i16"))
DEF_EXEC_TEST(dot2_m16_16, "0xffff", "0xffff", ACC_T("v_dot2_i32_i16"))
DEF_EXEC_TEST(dot2_m1_1, "1", "1", ACC_T("v_dot2_i32_i16"))
DEF_EXEC_TEST(dot2_m32_1, "0xffffffff", "1", ACC_T("v_dot2_i32_i16"))
DEF_EXEC_TEST(dot2_mhi16, "0xffff0000", "0", ACC_T("v_dot2_i32_i16"))
DEF_EXEC_TEST(dot2_malt, "0x55555555", "0x55555555", ACC_T("v_dot2_i32_i16"))
DEF_EXEC_TEST(dot2_m63, "0xffffffff", "0x7fffffff", ACC_T("v_dot2_i32_i16"))
DEF_EXEC_TEST(dot2_m5, "0x1f", "0", ACC_T("v_dot2_i32_i16"))
DEF_EXEC_TEST(dot2_m48, "0xffffffff", "0xffff", ACC_T("v_dot2_i32_i16"))
DEF_EXEC_TEST(mulf64_m32, "0xffffffff", "0", D2_T("v_mul_f64"))
DEF_EXEC_TEST(mulf64_m17, "0x1ffff", "0", D2_T("v_mul_f64"))
DEF_EXEC_TEST(mulf32_lane0, "1", "0", VOP_T("v_mul_f32"))
DEF_EXEC_TEST(perm_lane0, "1", "0", VOP3_T("v_perm_b32"))
DEF_EXEC_TEST(lshl_lane0, "1", "0", VOP_T("v_lshlrev_b32"))
DEF_TEST(lshrrev_b32, VOP_T("v_lshrrev_b32"), VOP_L("v_lshrrev_b32"))
DEF_TEST(lshlrev_imm, T32("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7\n", V8 : "v"(b)), VOP_T("v_add_u32"))
DEF_TEST(ashrrev_imm, T32("v_ashrrev_i32 %0, 9, %0\n v_ashrrev_i32 %1, 9, %1\n v_ashrrev_i32 %2, 9, %2\n v_ashrrev_i32 %3, 9, %3\n v_ashrrev_i32 %4, 9, %4\n v_ashrrev_i32 %5, 9, %5\n v_ashrrev_i32 %6, 9, %6\n v_ashrrev_i32 %7, 9, %7\n", V8 : "v"(b)), VOP_T("v_add_u32"))
DEF_TEST(cndmask_sgpr, T32("v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]\n", V8 : "v"(b) : "s20", "s21"), VOP_T("v_add_u32"))
DEF_TEST(mul_f32_dpp, T32("v_mul_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %4, %4, %4 row_mirror row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %5, %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %6, %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %7, %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n", V8 : "v"(b)), VOP_T("v_add_u32"))
DEF_TEST(add_co_u32, T32("v_add_co_u32 %0, vcc, %0, %8\n v_add_co_u32 %1, vcc, %1, %8\n v_add_co_u32 %2, vcc, %2, %8\n v_add_co_u32 %3, vcc, %3, %8\n v_add_co_u32 %4, vcc, %4, %8\n v_add_co_u32 %5, vcc, %5, %8\n v_add_co_u32 %6, vcc, %6, %8\n v_add_co_u32 %7, vcc, %7, %8\n", V8 : "v"(b) : "vcc"), VOP_T("v_add_u32"))
DEF_TEST(mul_hi_u32, VOP_T("v_mul_hi_u32"), VOP_L("v_mul_hi_u32"))
DEF_TEST(mad_u64_u32, T32("v_mad_u64_u32 %0, vcc, %8, %8, %0\n v_mad_u64_u32 %1, vcc, %8, %8, %1\n v_mad_u64_u32 %2, vcc, %8, %8, %2\n v_mad_u64_u32 %3, vcc, %8, %8, %3\n v_mad_u64_u32 %4, vcc, %8, %8, %4\n v_mad_u64_u32 %5, vcc, %8, %8, %5\n v_mad_u64_u32 %6, vcc, %8, %8, %6\n v_mad_u64_u32 %7, vcc, %8, %8, %7\n", D8 : "v"(b) : "vcc"), VOP_T("v_add_u32"))
DEF_TEST(fmac_f32, VOP_T("v_fmac_f32"), VOP_L("v_fmac_f32"))

struct Test {
    const char *name;
    void (*fn)(unsigned *, int);
    int per_body; // instructions of the kind named per loop body
    bool has_lat;
};

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned *out;
    hipMalloc(&out, 1024 * 1024 * sizeof(unsigned));
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    std::vector<Test> tests = {
#define T(N, P, L) {#N, t_##N, P, L},
        T(add_u32, 32, true) T(mul_f32, 32, true) T(and_b32, 32, true) T(ashr_i32, 32, true) T(mul_lo_u32, 32, true) T(mad_i32_i24, 32, true)
        T(add3_u32, 32, true) T(perm_b32, 32, true) T(alignbit, 32, true) T(bfi_b32, 32, true) T(lshl_or, 32, true) T(dot2_i32_i16, 32, true)
        T(dot2c_i32_i16, 32, true) T(dot4_i32_i8, 32, true) T(pk_add_u16, 32, true) T(pk_mul_lo_u16, 32, true) T(pk_mad_u16, 32, true)
        T(pk_add_f32, 32, true) T(pk_mul_f32, 32, true) T(rndne_f32, 32, true) T(floor_f32, 32, true) T(cvt_i32_f32, 32, true)
        T(cvt_f32_i32, 32, true) T(rcp_f32, 32, true) T(sqrt_f32, 32, true) T(mov_b32, 32, true) T(add_f64, 32, true) T(mul_f64, 32, true)
        T(cvt_f64_i32, 32, false) T(cvt_f32_f64, 32, false) T(readlane, 32, false) T(dpp_quad, 32, true) T(dpp_row_mirror, 32, true)
        T(dpp_row_bcast15, 32, true) T(dpp_row_bcast31, 32, true) T(permlane32_swap, 32, false) T(permlane16_swap, 32, false)
        T(ds_bpermute, 32, false) T(s_nop0, 32, false) T(salu_add, 32, false) T(valu_salu_mix, 16, false)
        T(add_lo32, 32, false) T(add_hi32, 32, false) T(add_lo16, 32, false) T(add_lane0, 32, false) T(dot2_lo32, 32, false) T(dot2_lo16, 32, false)
        T(dot2_lane0, 32, false) T(cvt_lane0, 32, false) T(mulf64_lane0, 32, false) T(pkmulf32_lane0, 32, false) T(rcp_lane0, 32, false) T(dpp_lo32, 32, false)
        T(fma_f32, 32, true) T(sub_f32, 32, true) T(add_f32, 32, true) T(max_f32, 32, true) T(lshlrev_b32, 32, true) T(or_b32, 32, true) T(xor_b32, 32, true)
        T(sub_u32, 32, true) T(min_u32, 32, true) T(max_i32, 32, true) T(mul_u32_u24, 32, true) T(mul_i32_i24, 32, true) T(lshl_add_u32, 32, true)
        T(and_or_b32, 32, true) T(bfe_u32, 32, true) T(mad_u32_u24, 32, true) T(ldexp_f32, 32, true) T(add_u16, 32, true) T(mul_lo_u16, 32, true)
        T(pk_fma_f32, 32, true) T(fma_f64, 32, true) T(cmp_lt_f32, 32, false) T(cmp_lt_u32, 32, false) T(cmp_lt_f64, 32, false) T(cndmask, 32, false)
        T(cvt_f64_f32, 32, false) T(readfirstlane, 32, false) T(sdwa_add, 32, false) T(mov_dpp, 32, false) T(ds_read_b32, 32, false)
        T(ds_read_b128, 32, false) T(valu_ds_mix, 16, false)
        T(dot2_m24, 32, false) T(dot2_m17, 32, false) T(dot2_m16_16, 32, false) T(dot2_m1_1, 32, false) T(dot2_m32_1, 32, false) T(dot2_mhi16, 32, false)
        T(dot2_malt, 32, false) T(dot2_m63, 32, false) T(dot2_m5, 32, false) T(dot2_m48, 32, false) T(mulf64_m32, 32, false) T(mulf64_m17, 32, false)
        T(mulf32_lane0, 32, false) T(perm_lane0, 32, false) T(lshl_lane0, 32, false) T(lshrrev_b32, 32, true) T(lshlrev_imm, 32, false) T(ashrrev_imm, 32, false)
        T(cndmask_sgpr, 32, false) T(mul_f32_dpp, 32, false) T(add_co_u32, 32, false) T(mul_hi_u32, 32, true) T(mad_u64_u32, 32, false) T(fmac_f32, 32, true)
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const Test &t, int waves_per_simd, int lat) {
        const int blocks = cus * waves_per_simd; // 256 threads = 4 waves = one per SIMD
        hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, lat); // warm
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 2; r++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, lat);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        return (double) best * 1e-3; // seconds
    };
    // reference: v_add_u32 throughput at 1 wave per SIMD = 2 cycles per wave-instruction
    const double ref = run(tests[0], 1, 0) / ((double) ITERS * 32);
    printf("v_add_u32: %.3f ns per wave-instruction at one wave per SIMD => clock %.2f GHz if 2 cycles\n", ref * 1e9, 2.0 / (ref * 1e9));
    printf("%-18s %10s %10s %10s %10s   (ns per wave-instruction and SIMD; x2.4 = cycles at 2.4 GHz)\n", "instruction", "thr 1w", "thr 4w", "thr 8w", "latency");
    const char *only = getenv("UB_FROM");
    bool on = only == nullptr;
    for (const Test &t : tests) {
        if (!on && std::string(t.name) == only) on = true;
        if (!on) continue;
        const double n = (double) ITERS * t.per_body;
        const double t1 = run(t, 1, 0) / n, t4 = run(t, 4, 0) / n / 4.0, t8 = run(t, 8, 0) / n / 8.0, tl = t.has_lat ? run(t, 1, 1) / n : 0.0;
        printf("%-18s %10.3f %10.3f %10.3f %10.3f\n", t.name, t1 * 1e9, t4 * 1e9, t8 * 1e9, tl * 1e9);
    }
    return 0;
}
