// F2/F3: pyramidal Lucas-Kanade exactly as cv::calcOpticalFlowPyrLK is parameterised by the reference
// (tracking/tracking.cc:385-393, 487-496: win 21x21, maxLevel 3, (COUNT+EPS,30,0.01), USE_INITIAL_FLOW,
// minEigThreshold 1e-4), plus the forward/backward/border cull of tracking.cc:396-403 / 499-506.
// Arithmetic definition: SURVEY.md Appendix B.4/B.5 (OpenCV LKTrackerInvoker): Q14 bilinear weights, patch samples
// with 5 fractional bits, Scharr derivatives, window sums accumulated as EXACT integers (wave-reduced in int64), one
// conversion to float, 2x2 solve in float without FMA contraction -> bit-identical to the CPU restatement.
//
// MI355X mapping: ONE 64-lane wavefront per feature (one workgroup = one wave, so LDS staging needs no cross-wave
// barrier and thousands of features from many camera streams fill the 256 CUs).
//   * per level the 24x24 u8 neighbourhood of the previous image is staged in LDS (reflect-101 applied at load),
//     Scharr derivatives of the 22x22 support are computed ON THE FLY into LDS (the derivative planes are never
//     materialised in HBM: -5.3 B/px of traffic per frame vs OpenCV's layout),
//   * each lane owns a 7-pixel horizontal run of the 21x21 window (63 lanes x 7 = 441): it reads its 4x10-byte patch of
//     the staged tile as aligned dwords (+ v_alignbyte), computes its Scharr derivatives and I/Ix/Iy samples in
//     registers and keeps them there for all <=30 Gauss-Newton iterations,
//   * the next image is read through a 32x32 u8 LDS tile (re-staged only when the window leaves it); the lane's 2x8
//     bytes of it are CACHED IN REGISTERS and re-fetched only when the integer window position changes, so a typical
//     iteration touches no LDS at all (v1 spent 39% of its wave cycles in LDS issue stalls: rocprofv3 SQ_WAIT_INST_LDS),
//   * A11/A12/A22 and b1/b2 are per-lane int32 partials (bounded: 7*4080^2 < 2^27, 7*8160*4080 < 2^28) reduced exactly:
//     DPP butterflies in 32 bits up to 16 / 8 lanes, then 16-bit halves through row_mirror/row_bcast DPP stages;
//   * the kernel is VALU-issue bound (rocprofv3: ~11k VALU instructions per point in v2), so the integer math is packed:
//     pixels, Scharr terms and Q14 weights all fit 16 bits -> v_pk_{add,sub,mul_lo}_u16 for the derivative stencil and
//     v_dot2_i32_i16 for every bilinear blend (2 taps per instruction) and for the window products; the patch sample is
//     folded into the blend's rounding constant (c0 = 256 - 512*I) so a residual costs 2 dot2 + 1 shift.
//   * (round 6) the Scharr terms carry a factor 4 so that a blended derivative is the HIGH half of its dot2 result (packed by one v_perm, no
//     shift); the three window sums of a level come out of one reduction tree (v_permlane32_swap + v_permlane16_swap); the eigenvalue test
//     needs no division (threshold on the numerator) and the square root no range scaling; the verdict on a point (border, forward-backward
//     distance, undistortion: FP64 that every lane of the wave computed alike) is k_lk_finish's, a thread per point.
//     6.3 k vector instructions per tracked point (7.1 k in rounds 4-5), 93 VGPRs, no scratch, 5 waves per SIMD.
// Algorithmic HBM bytes per point and direction: 4 levels x (24^2 + 22^2) B; everything else is
// LDS/VGPR traffic.
#include <cfloat>

#include "dev_camera.h"
#include "icg_internal.h"

using namespace icgd;

#define LK_IT 24   // I tile side (22 support + 1 halo each side)
#define LK_IS 28   // I tile row stride in bytes (7 dwords: every lane reads 4 aligned dwords per row)
#define LK_JT 32   // J tile side
#define LK_JS 36   // J tile row stride in bytes (9 dwords: 3 aligned dwords cover any 8-byte run)
#define LK_JM 5    // J tile margin around the 22x22 support
#define LK_MAX_ITERS 30
#ifndef LK_WAVES_PER_EU
#define LK_WAVES_PER_EU 5 // second __launch_bounds__ argument of k_lk_track_fb (the resource remarks of the build are quoted in DESIGN.md section 4)
#endif

struct lk_smem {
    unsigned int I[LK_IT * LK_IS / 4];
    unsigned int J[LK_JT * LK_JS / 4 + 1];
};

// Exact wave-wide integer sums.  The per-lane partials are bounded (see the kernel comment), so the first butterfly
// stages run in 32 bits as fused DPP adds (quad_perm xor1, xor2, row_half_mirror[, row_mirror]).
__device__ __forceinline__ int dpp_add_xor1(int v) { return v + __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false); }
__device__ __forceinline__ int dpp_add_xor2(int v) { return v + __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false); }
__device__ __forceinline__ int dpp_add_half_mirror(int v) { return v + __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false); }
__device__ __forceinline__ int dpp_add_mirror(int v) { return v + __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false); }

// Exact wave-wide sum of bounded int32 partials, returned as (float) of the exact integer (what the CPU restatement's
// (float)(int64 sum) produces).  Stages: 32-bit DPP butterflies while the group sums still fit int32 (8 lanes for
// |partial| <= 2^28, 16 lanes for <= 2^27); the group sums are then split into a signed high and an unsigned low 16-bit
// half, which both survive the remaining cross-row stages (row_mirror / row_bcast15 / row_bcast31, one fused
// v_add_u32_dpp each) without overflow; lane 63 holds both totals, hi*65536 + lo is formed exactly in double and one
// v_cvt_f32_f64 rounds once (RNE) like i64->f32.  No scalar carry chains, no 64-bit DPP moves.
// (round 5) the totals are combined in f32, not f64: |hi| < 2^24 and lo < 2^24 convert exactly, hi * 65536 is exact, and the one rounding of
// the fused multiply-add is RNE of the exact integer hi * 65536 + lo — what i64 -> f32 gives.  Two conversions + one v_fma on the lane that
// holds the totals, ONE v_readlane of the result (was: two readlanes, two v_cvt_f64_i32, v_ldexp_f64, v_add_f64, v_cvt_f32_f64).
__device__ __forceinline__ float lk_combine_f32(int hi, int lo) { return __builtin_fmaf((float) hi, 65536.f, (float) lo); }
__device__ __forceinline__ float lk_readlane_f32(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum_tail_f32(int hi, int lo) {
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, false); // row_bcast15: rows 1,3 += lane 15 of rows 0,2
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, false);
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xC, 0xF, false); // row_bcast31: rows 2,3 += lane 31
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xC, 0xF, false);
    return lk_readlane_f32(lk_combine_f32(hi, lo), 63);
}
// |per-lane partial| <= 2^28: sums of 8 lanes fit in int32
__device__ __forceinline__ float wave_sum_i32x8_f32(int v) {
    v = dpp_add_xor1(v);
    v = dpp_add_xor2(v);
    v = dpp_add_half_mirror(v);
    int hi = v >> 16, lo = v & 0xffff;
    hi     = dpp_add_mirror(hi);
    lo     = dpp_add_mirror(lo);
    return wave_sum_tail_f32(hi, lo);
}
// |per-lane partial| <= 2^27: sums of 16 lanes fit in int32
__device__ __forceinline__ float wave_sum_i32x16_f32(int v) {
    v = dpp_add_xor1(v);
    v = dpp_add_xor2(v);
    v = dpp_add_half_mirror(v);
    v = dpp_add_mirror(v);
    return wave_sum_tail_f32(v >> 16, v & 0xffff);
}

// TWO exact wave-wide sums from ONE reduction tree (round 5).  v_permlane32_swap (gfx950) exchanges the upper half of the first register with
// the lower half of the second: one add later lanes 0..31 hold the 2-lane sums of v1 and lanes 32..63 those of v2, and every further DPP
// stage works on both values at once; lane 31 ends with the total of v1, lane 63 with the total of v2.
typedef unsigned int lk_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int lk_fold32(int v1, int v2) {
    const lk_u2 r = __builtin_amdgcn_permlane32_swap((unsigned int) v1, (unsigned int) v2, false, false);
    return (int) (r.x + r.y);
}
__device__ __forceinline__ void wave_sum2_finish_f32(int hi, int lo, float &f1, float &f2) {
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, false); // row_bcast15: lanes 31 / 63 = totals of lanes 0..31 / 32..63
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, false);
    const float t = lk_combine_f32(hi, lo);
    f1 = lk_readlane_f32(t, 31);
    f2 = lk_readlane_f32(t, 63);
}
// |per-lane partial| <= 2^28: sums of 8 lanes fit in int32
__device__ __forceinline__ void wave_sum2_i32x8_f32(int v1, int v2, float &f1, float &f2) {
    int v = lk_fold32(v1, v2);
    v     = dpp_add_xor1(v);
    v     = dpp_add_xor2(v);
    int hi = v >> 16, lo = v & 0xffff;
    hi     = dpp_add_half_mirror(hi);
    lo     = dpp_add_half_mirror(lo);
    hi     = dpp_add_mirror(hi);
    lo     = dpp_add_mirror(lo);
    wave_sum2_finish_f32(hi, lo, f1, f2);
}
// |per-lane partial| <= 2^27: sums of 16 lanes fit in int32
__device__ __forceinline__ void wave_sum2_i32x16_f32(int v1, int v2, float &f1, float &f2) {
    int v = lk_fold32(v1, v2);
    v     = dpp_add_xor1(v);
    v     = dpp_add_xor2(v);
    v     = dpp_add_half_mirror(v);
    int hi = v >> 16, lo = v & 0xffff;
    hi     = dpp_add_mirror(hi);
    lo     = dpp_add_mirror(lo);
    wave_sum2_finish_f32(hi, lo, f1, f2);
}

// THREE exact wave-wide sums from ONE reduction tree (round 6): after v_permlane32_swap has put v1 | v2 and v3 | 0 side by side,
// v_permlane16_swap (gfx950) interleaves the two registers by rows of 16 lanes — row 0 carries v1, row 1 v3, row 2 v2 — and the four DPP
// stages that are left work on all of them at once; every lane of a row ends with the total of its row.  20 instructions for the three
// window sums of the level set-up (were 31).  |per-lane partial| <= 2^27: sums of 16 lanes fit in int32.
__device__ __forceinline__ void wave_sum3_i32x16_f32(int v1, int v2, int v3, float &f1, float &f2, float &f3) {
    const int x       = lk_fold32(v1, v2); // lanes 0..31: 2-lane sums of v1, lanes 32..63: of v2
    const int y       = lk_fold32(v3, 0);  // lanes 0..31: 2-lane sums of v3, lanes 32..63: 0
    const lk_u2 r     = __builtin_amdgcn_permlane16_swap((unsigned int) x, (unsigned int) y, false, false); // odd rows of x <-> even rows of y
    int v             = (int) (r.x + r.y); // rows: v1, v3, v2, 0 (4-lane sums)
    v                 = dpp_add_xor1(v);
    v                 = dpp_add_xor2(v);   // 16-lane sums
    int hi = v >> 16, lo = v & 0xffff;
    hi                = dpp_add_half_mirror(hi);
    lo                = dpp_add_half_mirror(lo);
    hi                = dpp_add_mirror(hi);
    lo                = dpp_add_mirror(lo);
    const float t     = lk_combine_f32(hi, lo);
    f1                = lk_readlane_f32(t, 0);
    f3                = lk_readlane_f32(t, 16);
    f2                = lk_readlane_f32(t, 32);
}

__device__ __forceinline__ int lk_descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

// The Q14 weights as the packed pairs the blends consume, W0 = (w00, w01), W1 = (w10, w11) (round 5).  rint() by the magic constant: for
// 0 <= x < 2^22 the low 16 bits of bits(x + 1.5 * 2^23) are rint(x) (the addition rounds to nearest even on an integer grid and the
// constant is even) — one full-rate v_add_f32 instead of v_rndne_f32 + v_cvt_i32_f32; w11 = 2^14 - w00 - w01 - w10 is formed on the biased
// bit patterns modulo 2^16 (the three biases 0x4B400000 have zero low halves) and v_perm picks the low halves.  Same weights, bit for bit.
__device__ __forceinline__ void lk_weights_pk(float a, float b, unsigned int &W0, unsigned int &W1) {
    const float A = a * (float) (1 << 14), A1 = (float) (1 << 14) - A, b1 = 1.f - b;
    const unsigned int r00 = __float_as_uint(A1 * b1 + 12582912.f), r01 = __float_as_uint(A * b1 + 12582912.f);
    const unsigned int r10 = __float_as_uint(A1 * b + 12582912.f);
    const unsigned int r11 = 0x4000u - (r00 + r01 + r10);
    W0 = __builtin_amdgcn_perm(r01, r00, 0x05040100u);
    W1 = __builtin_amdgcn_perm(r11, r10, 0x05040100u);
}

__device__ __forceinline__ void lk_weights(float a, float b, int &w00, int &w01, int &w10, int &w11) {
    // rint((1-a)(1-b) 2^14) etc.; the power-of-two scale commutes with every rounding, so it is applied to a once:
    // fl(fl(1-a) fl(1-b)) 2^14 == fl((2^14 - a 2^14) fl(1-b))   (bit-identical, 8 fewer VALU ops per iteration)
    const float A = a * (float) (1 << 14), A1 = (float) (1 << 14) - A, b1 = 1.f - b;
    w00 = (int) rintf(A1 * b1);
    w01 = (int) rintf(A * b1);
    w10 = (int) rintf(A1 * b);
    w11 = (1 << 14) - w00 - w01 - w10;
}

// 4 consecutive pixels of row `row` starting at image column x (reflect-101 outside the image), packed little-endian
__device__ __forceinline__ unsigned int lk_load4(const unsigned char *row, int x, int W) {
    if (x >= 0 && x + 3 < W) {
        typedef unsigned int __attribute__((aligned(1))) u32u;
        return *reinterpret_cast<const u32u *>(row + x);
    }
    unsigned int v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) v |= (unsigned int) row[icg_reflect101(x + k, W)] << (8 * k);
    return v;
}
__device__ __forceinline__ int lk_byte(unsigned int w, int k) { return (int) ((w >> (8 * k)) & 0xffu); }

typedef unsigned int __attribute__((aligned(1))) lk_u32u;

// Tile loads are split into "issue all global loads into registers" and "write them to LDS", so that the loads of the
// previous-image tile and of the next-image tile of a level fly together (one HBM/L2 round trip per level instead of
// two or more back-to-back ones: the wave has few siblings to hide latency behind at 84 VGPRs).
// 24x24 I tile: dword i = lane + 64*q -> (row i/6, dword column i%6), tile (r,c) <-> image (ipx-1+c, ipy-1+r)
__device__ __forceinline__ void lk_load_I(unsigned int (&v)[3], const unsigned char *I, int W, int H, int pitch, int ipx, int ipy,
                                          int lane) {
    const bool inside = ipx - 1 >= 0 && ipx - 1 + LK_IT <= W && ipy - 1 >= 0 && ipy - 1 + LK_IT <= H; // wave-uniform
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int i = lane + 64 * q;
        const int r = i / 6, cd = i - r * 6;
        v[q] = 0;
        if (i < LK_IT * 6) {
            if (inside)
                v[q] = *reinterpret_cast<const lk_u32u *>(I + (size_t) (ipy - 1 + r) * pitch + (ipx - 1 + 4 * cd));
            else
                v[q] = lk_load4(I + (size_t) icg_reflect101(ipy - 1 + r, H) * pitch, ipx - 1 + 4 * cd, W);
        }
    }
}
__device__ __forceinline__ void lk_store_I(lk_smem &S, const unsigned int (&v)[3], int lane) {
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int i = lane + 64 * q;
        const int r = i / 6, cd = i - r * 6;
        if (i < LK_IT * 6) S.I[(r * LK_IS >> 2) + cd] = v[q];
    }
}
// 32x32 J tile: lane -> (row lane>>1, 16 pixels at column (lane&1)*16) = 4 packed dwords, tile (r,c) <-> image (jx0+c, jy0+r)
__device__ __forceinline__ void lk_load_J(unsigned int (&v)[4], const unsigned char *J, int W, int H, int pitch, int jx0, int jy0,
                                          int lane) {
    const int r = lane >> 1, c0 = (lane & 1) * 16;
    const bool inside = jx0 >= 0 && jx0 + LK_JT <= W && jy0 >= 0 && jy0 + LK_JT <= H; // wave-uniform
    if (inside) {
        const unsigned char *row = J + (size_t) (jy0 + r) * pitch + (jx0 + c0);
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const lk_u32u *>(row + 4 * q);
    } else {
        const unsigned char *row = J + (size_t) icg_reflect101(jy0 + r, H) * pitch;
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = lk_load4(row, jx0 + c0 + 4 * q, W);
    }
}
__device__ __forceinline__ void lk_store_J(lk_smem &S, const unsigned int (&v)[4], int lane) {
    const int r = lane >> 1, c0 = (lane & 1) * 16;
    unsigned int *dst = &S.J[(r * LK_JS + c0) >> 2];
#pragma unroll
    for (int q = 0; q < 4; q++) dst[q] = v[q];
}
__device__ __forceinline__ void lk_stage_J(lk_smem &S, const unsigned char *J, int W, int H, int pitch, int jx0, int jy0,
                                           int lane) {
    unsigned int v[4];
    lk_load_J(v, J, W, H, pitch, jx0, jy0, lane);
    lk_store_J(S, v, lane);
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// packed 16-bit helpers (all operands stay inside 16 bits: pixels <= 255, Scharr terms <= 4080, Q14 weights <= 16384)
__device__ __forceinline__ unsigned int pk_add(unsigned int a, unsigned int b) {
    return __builtin_bit_cast(unsigned int, (u16x2) (__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned int pk_sub(unsigned int a, unsigned int b) {
    return __builtin_bit_cast(unsigned int, (u16x2) (__builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned int pk_mul(unsigned int a, unsigned short k) {
    return __builtin_bit_cast(unsigned int, (u16x2) (__builtin_bit_cast(u16x2, a) * k));
}
// exact a.lo*b.lo + a.hi*b.hi + c on signed 16-bit halves (v_dot2_i32_i16)
__device__ __forceinline__ int dot2(unsigned int a, unsigned int b, int c) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}
// the same with a SEPARATE accumulator register (VOP3P v_dot2_i32_i16).  The builtin is selected as v_dot2c, whose accumulator is tied to the
// destination: an accumulator that must survive (c0[k] in every iteration, the rounding constants of the set-up) costs a v_mov per use.
__device__ __forceinline__ int dot2_keep(unsigned int a, unsigned int b, int c) {
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// the same with a wave-uniform accumulator from a scalar register (one SGPR operand is allowed on the constant bus): the rounding constants
// of the set-up's blends cost no VGPR this way (round 6: pinned in VGPRs they were the two registers k_lk_track_fb spilled at 5 waves)
__device__ __forceinline__ int dot2_keep_s(unsigned int a, unsigned int b, int c_uniform) {
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
    return d;
}
// a.lo*b.lo + a.hi*b.hi with the accumulator 0 as an inline constant
__device__ __forceinline__ int dot2_zero(unsigned int a, unsigned int b) {
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// (lo16(a), lo16(b)) as one packed register
__device__ __forceinline__ unsigned int pk_lo16(int a, int b) {
    return __builtin_amdgcn_perm((unsigned int) b, (unsigned int) a, 0x05040100u);
}
// (hi16(a), hi16(b)) as one packed register
__device__ __forceinline__ unsigned int pk_hi16(int a, int b) {
    return __builtin_amdgcn_perm((unsigned int) b, (unsigned int) a, 0x07060302u);
}
// the pair one 16-bit element further: (a.hi, b.lo)
__device__ __forceinline__ unsigned int pk_shift(unsigned int a, unsigned int b) { return __builtin_amdgcn_alignbit(b, a, 16); }

// the lane's two 8-pixel rows of the window at integer position (inx, iny), from the staged tile (3 aligned dwords + a
// byte funnel shift per row), expanded to the 7 horizontally adjacent u16 pixel pairs the bilinear blend consumes
__device__ __forceinline__ void lk_fetch_J(const lk_smem &S, int row0, int o, unsigned int (&JA)[7], unsigned int (&JB)[7]) {
    const int idx = o >> 2, sh = o & 3;
    const unsigned int *r0 = &S.J[(row0 * LK_JS >> 2) + idx];
    const unsigned int *r1 = r0 + (LK_JS >> 2);
    const unsigned int d0 = r0[0], d1 = r0[1], d2 = r0[2];
    const unsigned int e0 = r1[0], e1 = r1[1], e2 = r1[2];
    const unsigned int a0 = __builtin_amdgcn_alignbyte(d1, d0, sh), a1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
    const unsigned int b0 = __builtin_amdgcn_alignbyte(e1, e0, sh), b1 = __builtin_amdgcn_alignbyte(e2, e1, sh);
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const unsigned int sel = 0x0c000c00u | (unsigned int) k | ((unsigned int) (k + 1) << 16); // (byte k, 0, byte k+1, 0)
        JA[k] = __builtin_amdgcn_perm(a1, a0, sel);
        JB[k] = __builtin_amdgcn_perm(b1, b0, sel);
    }
}

// One calcOpticalFlowPyrLK point, executed cooperatively by a full wave. Returns status.
__device__ bool lk_track_wave(const icg_pyr_desc &P, const unsigned char *slotI, const unsigned char *slotJ,
                              float2 prevPt, float2 &nextIO, lk_smem &S, int lane, float *err_out) {
    const float FLT_SCALE = 1.f / (1 << 20);
    const double eps2     = 0.01 * 0.01;
    bool status           = true;
    float errv            = 0.f;
    float2 nextStore      = nextIO;
    const int maxLevel    = P.n_levels - 1;
    const bool active     = lane < 63;
    const int ly          = active ? lane / 3 : 0;
    const int lx0         = active ? (lane - ly * 3) * 7 : 0;
    const unsigned int am = active ? ~0u : 0u; // lane 63 owns no pixels: its derivative weights are zeroed (Ix = Iy = 0 -> no contribution to any sum)
    // rounding constants of the set-up's blends, kept in scalar registers for dot2_keep_s (opaque to the optimizer: not re-materialised per use)
    int rnd15 = 1 << 15, rnd8 = 1 << 8;
    asm volatile("" : "+s"(rnd15), "+s"(rnd8));

    for (int level = maxLevel; level >= 0; --level) {
        const int W = P.w[level], H = P.h[level], pitch = P.pitch[level];
        const unsigned char *I = slotI + P.off[level];
        const unsigned char *J = slotJ + P.off[level];
        const float scale      = (float) (1. / (1 << level));
        float prevx = prevPt.x * scale, prevy = prevPt.y * scale;
        float nptx, npty;
        if (level == maxLevel) {
            nptx = nextStore.x * scale;
            npty = nextStore.y * scale;
        } else {
            nptx = nextStore.x * 2.f;
            npty = nextStore.y * 2.f;
        }
        nextStore = make_float2(nptx, npty);

        prevx -= (float) ICG_LK_HALF;
        prevy -= (float) ICG_LK_HALF;
        const int ipx = (int) floorf(prevx), ipy = (int) floorf(prevy);
        if (ipx < -ICG_LK_WIN || ipx >= W || ipy < -ICG_LK_WIN || ipy >= H) {
            if (level == 0) {
                status = false;
                errv   = 0.f;
            }
            continue;
        }
        unsigned int W0, W1;
        lk_weights_pk(prevx - ipx, prevy - ipy, W0, W1);

        // ---- stage the 24x24 neighbourhood of the previous image AND the 32x32 tile of the next image around the
        //      level's starting estimate in one go (both address sets are known here) ----
        int jx0 = -1000000, jy0 = -1000000;
        {
            const int inx0 = (int) floorf(nptx - (float) ICG_LK_HALF), iny0 = (int) floorf(npty - (float) ICG_LK_HALF);
            const bool jok = !(inx0 < -ICG_LK_WIN || inx0 >= W || iny0 < -ICG_LK_WIN || iny0 >= H); // else iteration 0 bails out
            unsigned int vi[3] = {0, 0, 0}, vj[4] = {0, 0, 0, 0};
            lk_load_I(vi, I, W, H, pitch, ipx, ipy, lane);
            if (jok) {
                jx0 = inx0 - LK_JM;
                jy0 = iny0 - LK_JM;
                lk_load_J(vj, J, W, H, pitch, jx0, jy0, lane);
            }
            __syncthreads(); // previous level's LDS readers are done
            lk_store_I(S, vi, lane);
            if (jok) lk_store_J(S, vj, lane);
            __syncthreads();
        }

        // ---- per lane: 4 rows x 10 bytes -> I samples and on-the-fly Scharr derivatives of its 7-pixel run ----
        // c0[k] = 256 - 512*Ival[k]: the patch sample folded into the rounding constant of the J blend, so that
        //         diff = ((Jblend + 256) >> 9) - Ival == (Jblend + c0) >> 9 exactly (Ival is an integer)
        // IXP/IYP: the derivative samples as packed i16 pairs (k, k+1), k = 0,2,4,6 (upper half of the last pair is 0)
        int c0[7];
        unsigned int IXP[4], IYP[4];
        int sA11 = 0, sA12 = 0, sA22 = 0;
        float A11, A12, A22;
        {
            const int idx = lx0 >> 2, sh = lx0 & 3;
            unsigned int Pp[4][5]; // pixel pairs (col 2m, 2m+1) of tile rows ly..ly+3, cols lx0..lx0+9
            // pair m = bytes sh + 2m, sh + 2m + 1 of the row's four dwords: for every sh in 0..3 it lies inside ONE pair of adjacent dwords that
            // does not depend on sh — (d1:d0) for m = 0, 1, (d2:d1) for m = 2, 3, (d3:d2) for m = 4 — so v_perm takes it straight from the
            // loaded dwords with two per-lane selectors (round 6: the three v_alignbyte per row that normalised the row first are gone)
            const unsigned int sel0 = 0x0c000c00u | (unsigned int) sh | ((unsigned int) (sh + 1) << 16), sel1 = sel0 + 0x00020002u;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const unsigned int *p = &S.I[((ly + r) * LK_IS >> 2) + idx];
                const unsigned int d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3];
                Pp[r][0] = __builtin_amdgcn_perm(d1, d0, sel0);
                Pp[r][1] = __builtin_amdgcn_perm(d1, d0, sel1);
                Pp[r][2] = __builtin_amdgcn_perm(d2, d1, sel0);
                Pp[r][3] = __builtin_amdgcn_perm(d2, d1, sel1);
                Pp[r][4] = __builtin_amdgcn_perm(d3, d2, sel0);
            }
            // derivative at support position (c = lx0+j, r = ly+rr): 3x3 neighbourhood = tile rows rr..rr+2, cols j..j+2.
            // The derivative plane is ZERO outside the image: X = ipx+lx0+j in [0,W), Y = ipy+ly+rr in [0,H).
            const bool all_in = ipx >= 0 && ipx + 22 <= W && ipy >= 0 && ipy + 22 <= H; // wave-uniform fast path
            unsigned int DX[2][4], DY[2][4];
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                unsigned int T0[5], T1[5];
#pragma unroll
                for (int m = 0; m < 5; m++) {
                    T0[m] = pk_add(pk_mul(pk_add(Pp[rr][m], Pp[rr + 2][m]), 12), pk_mul(Pp[rr + 1][m], 40)); // 4 * (3*(p0+p2) + 10*p1)
                    T1[m] = pk_sub(Pp[rr + 2][m], Pp[rr][m]);                                                // p2 - p0
                }
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    DX[rr][m] = pk_sub(T0[m + 1], T0[m]);                                                       // 4 * (t0[j+2] - t0[j])
                    DY[rr][m] = pk_add(pk_mul(pk_add(T1[m], T1[m + 1]), 12), pk_mul(pk_shift(T1[m], T1[m + 1]), 40)); // 4 * (3*(t1[j]+t1[j+2]) + 10*t1[j+1])
                }
            }
            if (!all_in) { // windows that reach over the image border: a real (wave-uniform) branch — written as masks on the common path the
                           // compiler turned it into 46 v_cndmask + 16 v_and per level for every window (round 5: the asm statement keeps it a branch)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const int Y           = ipy + ly + rr;
                    const unsigned int RM = (Y >= 0 && Y < H) ? ~0u : 0u;
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        const int X0 = ipx + lx0 + 2 * m, X1 = X0 + 1;
                        const unsigned int CM = ((X0 >= 0 && X0 < W) ? 0x0000ffffu : 0u) | ((X1 >= 0 && X1 < W) ? 0xffff0000u : 0u);
                        DX[rr][m] &= CM & RM;
                        DY[rr][m] &= CM & RM;
                    }
                }
            }
            // The Scharr terms above carry a factor 4 (12, 40 for 3, 10: |4 d| <= 16 320 still fits 16 bits), so the blended derivative
            // (sum w d + 2^13) >> 14 is the HIGH half of sum w (4 d) + 2^15: one v_perm packs two of them, no shift (round 6: 14 shifts per level).
            int bx[7], by[7];
            const unsigned int W0d = W0 & am, W1d = W1 & am; // (lane 63: (0 + 2^15) >> 16 = 0)
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const int m = k >> 1;
                // (v[k], v[k+1]) pairs of the two derivative rows and of tile rows 1, 2 at columns k+1, k+2
                const unsigned int dx0 = (k & 1) ? pk_shift(DX[0][m], DX[0][m + 1]) : DX[0][m];
                const unsigned int dx1 = (k & 1) ? pk_shift(DX[1][m], DX[1][m + 1]) : DX[1][m];
                const unsigned int dy0 = (k & 1) ? pk_shift(DY[0][m], DY[0][m + 1]) : DY[0][m];
                const unsigned int dy1 = (k & 1) ? pk_shift(DY[1][m], DY[1][m + 1]) : DY[1][m];
                const int m1 = (k + 1) >> 1;
                const unsigned int i1 = ((k + 1) & 1) ? pk_shift(Pp[1][m1], Pp[1][m1 + 1]) : Pp[1][m1];
                const unsigned int i2 = ((k + 1) & 1) ? pk_shift(Pp[2][m1], Pp[2][m1 + 1]) : Pp[2][m1];
                bx[k]        = dot2(dx1, W1d, dot2_keep_s(dx0, W0d, rnd15));
                by[k]        = dot2(dy1, W1d, dot2_keep_s(dy0, W0d, rnd15));
                // c0 = 256 - 512 * ((blend + 256) >> 9) = 256 - ((blend + 256) & ~511)
                c0[k]        = 256 - (dot2(i2, W1, dot2_keep_s(i1, W0, rnd8)) & ~511);
            }
#pragma unroll
            for (int m = 0; m < 4; m++) {
                IXP[m] = m < 3 ? pk_hi16(bx[2 * m], bx[2 * m + 1]) : (unsigned int) bx[6] >> 16;
                IYP[m] = m < 3 ? pk_hi16(by[2 * m], by[2 * m + 1]) : (unsigned int) by[6] >> 16;
            }
            sA11 = dot2(IXP[3], IXP[3], dot2(IXP[2], IXP[2], dot2(IXP[1], IXP[1], dot2(IXP[0], IXP[0], 0))));
            sA12 = dot2(IXP[3], IYP[3], dot2(IXP[2], IYP[2], dot2(IXP[1], IYP[1], dot2(IXP[0], IYP[0], 0))));
            sA22 = dot2(IYP[3], IYP[3], dot2(IYP[2], IYP[2], dot2(IYP[1], IYP[1], dot2(IYP[0], IYP[0], 0))));
            wave_sum3_i32x16_f32(sA11, sA12, sA22, A11, A12, A22);
            A11 *= FLT_SCALE, A12 *= FLT_SCALE, A22 *= FLT_SCALE;
        }
        float D            = A11 * A22 - A12 * A12;
        // (the argument of the root: A = n 2^-20 rounded to 24 bits, so a difference is zero or at least 2^-20 — zero or >= 2^-40, icg_sqrt_unscaled's domain)
        const float eig2 = A22 + A11 - icg_sqrt_unscaled((A11 - A22) * (A11 - A22) + 4.f * A12 * A12);
        // minEig = eig2 / (2 * 21 * 21) < 1e-4f, without the division (round 6: ten instructions per level): a correctly rounded quotient is
        // non-decreasing in its numerator, so the test is eig2 < T with T the smallest float whose quotient by 882 reaches 1e-4f —
        // T = 0x3DB4A233 (0.088199995; fl(T / 882) = 1e-4f, fl(pred(T) / 882) = 9.999998e-5: searched over the floats around 0.0882)
        static_assert(ICG_LK_WIN == 21, "the threshold constant below is for a 21 x 21 window");
        const bool weak = eig2 < __uint_as_float(0x3DB4A233u) || D < FLT_EPSILON;
        if (weak) {
            if (level == 0) status = false;
            continue;
        }
        D = 1.f / D;
        // the 2^-20 of the window sums folded into the inverse determinant (round 5): b1, b2 stay unscaled integers-as-floats, and scaling by a
        // power of two commutes with every rounding of (A12 b2 - A22 b1) D (no intermediate leaves the normal range: |b| >= 1 or 0,
        // A = n 2^-20, 2^-28 <= D <= 2^23) — the step is the same float, bit for bit, two multiplications per iteration cheaper
        const float Ds = D * FLT_SCALE;
        nptx -= (float) ICG_LK_HALF;
        npty -= (float) ICG_LK_HALF;
        float pdx = 0.f, pdy = 0.f;
        int cinx = -1000000, ciny = -1000000; // integer window position whose pixel pairs are cached in JA/JB
        unsigned int JA[7], JB[7];
#pragma unroll
        for (int k = 0; k < 7; k++) JA[k] = JB[k] = 0;

        // The Gauss-Newton iterations, as EPOCHS of constant integer window position: the lane's pixel pairs JA/JB are fetched once per epoch
        // and are loop-invariant inside it (written as one loop with a conditional re-fetch, they were loop-carried through the condition
        // and the compiler copied all 14 registers twice per iteration: 28 of ~130 VALU instructions).  Same checks in the same order.
        // Round 5 (instruction diet by the measured issue costs, profiles/ubench: conversions, roundings, compares, DPP, dot2, FP64 and
        // everything VOP3 cost twice a plain add / mul): the fractional position a = npt - floor(npt) is carried from
        // the epoch test of the previous iteration (the window stays in the epoch iff 0 <= a < 1 on both axes: one subtraction and one
        // unsigned compare per axis instead of floor, convert, compare and convert back), the weights come packed from lk_weights_pk,
        // c0[k] is read in place (dot2_keep), b1 and b2 come out of ONE reduction tree, the oscillation test compares in f32
        // (|s| < 0.01 in double  <=>  |s| <= 0.01f for a float s: 0.01f < 0.01 < nextafter(0.01f)).
        int j = 0;
        bool more = true;
        while (more) {
            const float fx = floorf(nptx), fy = floorf(npty);
            const int inx = (int) fx, iny = (int) fy;
            if (inx < -ICG_LK_WIN || inx >= W || iny < -ICG_LK_WIN || iny >= H) {
                if (level == 0) status = false;
                break;
            }
            if (!(inx >= jx0 && inx <= jx0 + (LK_JT - 22) && iny >= jy0 && iny <= jy0 + (LK_JT - 22))) {
                jx0 = inx - LK_JM;
                jy0 = iny - LK_JM;
                __syncthreads();
                lk_stage_J(S, J, W, H, pitch, jx0, jy0, lane);
                __syncthreads();
            }
            lk_fetch_J(S, iny - jy0 + ly, (inx - jx0) + lx0, JA, JB);
            cinx = inx;
            ciny = iny;
            float ax = nptx - fx, ay = npty - fy; // == nptx - (float) inx
            for (;;) {
                lk_weights_pk(ax, ay, W0, W1);
                int diff[7];
#pragma unroll
                for (int k = 0; k < 7; k++) diff[k] = dot2(JB[k], W1, dot2_keep(JA[k], W0, c0[k])) >> 9;
                int sb1 = 0, sb2 = 0;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const unsigned int dp = m < 3 ? pk_lo16(diff[2 * m], diff[2 * m + 1]) : (unsigned int) diff[6]; // IXP[3].hi == 0
                    // (the first product takes its zero accumulator as an inline constant: v_dot2c's tied accumulator cost a v_mov 0 per sum)
                    sb1 = m ? dot2(dp, IXP[m], sb1) : dot2_zero(dp, IXP[m]);
                    sb2 = m ? dot2(dp, IYP[m], sb2) : dot2_zero(dp, IYP[m]);
                }
                float b1, b2;
                wave_sum2_i32x8_f32(sb1, sb2, b1, b2);
                const float dx = (float) ((A12 * b2 - A22 * b1) * Ds);
                const float dy = (float) ((A12 * b1 - A11 * b2) * Ds);
                nptx += dx;
                npty += dy;
                nextStore = make_float2(nptx + (float) ICG_LK_HALF, npty + (float) ICG_LK_HALF);
                // ddx^2 + ddy^2 <= eps^2 in double (dx^2 is exact in double, so the fused form rounds the same sum once, as the two-step form
                // does).  Round 6: the float sum of the float squares is within 2^-22 of it, relatively — outside a band of 1e-5 around eps^2
                // it decides the test, and only inside the band the FP64 form runs (four half-rate instructions of every iteration before).
                const float n2 = dx * dx + dy * dy;
                bool converged = n2 < 9.9999e-5f;
                if (!converged && !(n2 > 1.00001e-4f)) { // wave-uniform; a real branch (the empty asm keeps the compiler from computing both forms always)
                    asm volatile("");
                    converged = __builtin_fma((double) dx, (double) dx, (double) dy * (double) dy) <= eps2;
                }
                if (converged) {
                    more = false;
                    break;
                }
                if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
                    nextStore.x -= dx * 0.5f;
                    nextStore.y -= dy * 0.5f;
                    more = false;
                    break;
                }
                pdx = dx;
                pdy = dy;
                if (++j >= LK_MAX_ITERS) {
                    more = false;
                    break;
                }
                ax = nptx - fx;
                ay = npty - fy;
                // floor(npt) unchanged on both axes <=> 0 <= a < 1 <=> bits(a) < bits(1.0f) as unsigned (a negative a has the sign bit set);
                // when a rounding makes a == 1.0f for a position still inside the pixel, the next epoch finds the same pixels again
                if (!(__float_as_uint(ax) < 0x3f800000u && __float_as_uint(ay) < 0x3f800000u)) break; // next epoch: bounds test, tile, fetch
            }
        }

        if (status && level == 0) {
            // OpenCV >= 3.4 epilogue (taken because the reference passes an err vector, tracking.cc:381,386,391)
            const float ex = nextStore.x - (float) ICG_LK_HALF, ey = nextStore.y - (float) ICG_LK_HALF;
            const int inx = (int) floorf(ex), iny = (int) floorf(ey);
            if (inx < -ICG_LK_WIN || inx >= W || iny < -ICG_LK_WIN || iny >= H) {
                status = false;
                continue;
            }
            if (err_out) {
                if (inx != cinx || iny != ciny) {
                    if (!(inx >= jx0 && inx <= jx0 + (LK_JT - 22) && iny >= jy0 && iny <= jy0 + (LK_JT - 22))) {
                        jx0 = inx - LK_JM;
                        jy0 = iny - LK_JM;
                        __syncthreads();
                        lk_stage_J(S, J, W, H, pitch, jx0, jy0, lane);
                        __syncthreads();
                    }
                    lk_fetch_J(S, iny - jy0 + ly, (inx - jx0) + lx0, JA, JB);
                }
                lk_weights_pk(ex - inx, ey - iny, W0, W1);
                int se = 0;
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    const int diff = dot2(JB[k], W1, dot2_keep(JA[k], W0, c0[k])) >> 9;
                    se += diff < 0 ? -diff : diff;
                }
                se &= (int) am; // lane 63 owns no pixels
                errv = wave_sum_i32x16_f32(se) * 1.f / (float) (32 * ICG_LK_WIN * ICG_LK_WIN);
            }
        }
    }
    nextIO = nextStore;
    if (err_out && lane == 0) *err_out = errv;
    return status;
}

__global__ __launch_bounds__(64) void k_lk_track(icg_pyr_desc P, int n, const int32_t *prev_slot, const int32_t *next_slot,
                                                 const float2 *prev_pts, float2 *next_pts, unsigned char *status, float *err) {
    __shared__ lk_smem S;
    const int i = icg_xcd_chunked(blockIdx.x, n);
    if (i >= n) return;
    const int lane = threadIdx.x;
    const unsigned char *sI = P.base + (size_t) prev_slot[i] * P.slot_bytes;
    const unsigned char *sJ = P.base + (size_t) next_slot[i] * P.slot_bytes;
    float2 nx = next_pts[i];
    bool st   = lk_track_wave(P, sI, sJ, prev_pts[i], nx, S, lane, err ? err + i : nullptr);
    if (lane == 0) {
        next_pts[i] = nx;
        status[i]   = st ? 1 : 0;
    }
}

__global__ __launch_bounds__(64, LK_WAVES_PER_EU) void k_lk_track_fb(icg_pyr_desc P, int n, const int32_t *prev_slot,
                                                    const int32_t *next_slot, const float2 *prev_pts,
                                                    const float2 *guess_pts, float2 *out_pts, unsigned char *status,
                                                    float2 *out_bwd, int img_w, int img_h,
                                                    int seg_cap, const int32_t *seg_count) {
    __shared__ lk_smem S;
    const int i = icg_xcd_chunked(blockIdx.x, n);
    if (i >= n) return;
    // segmented call (device-resident tracker, tracker.hip): the points of stream s are entries [s * seg_cap, s * seg_cap + seg_count[s])
    // of every array; the stage kernel that ran before this launch left the count in device memory — no host round trip sizes the grid
    if (seg_count) {
        const int s = i / seg_cap;
        if (i - s * seg_cap >= seg_count[s]) return; // wave-uniform
    }
    const int lane = threadIdx.x;
    const unsigned char *sP = P.base + (size_t) prev_slot[i] * P.slot_bytes;
    const unsigned char *sN = P.base + (size_t) next_slot[i] * P.slot_bytes;
    const float2 p0 = prev_pts[i];
    float2 fwd      = guess_pts[i];
    float2 bwd      = p0;
    bool st_f = false, st_b = false;
    // forward then backward through ONE copy of the tracker body (16 KB of code instead of 32 KB in the shared I-cache)
    for (int dir = 0; dir < 2; dir++) {
        if (dir) {
            // the cull of tracking.cc:396-403 needs the backward track only for points that are still alive
            // (isOnBorder compares in double: (double) f < 5.0 <=> f < 5.0f and (double) f > w - 5.0 <=> f > (float) (w - 5) for a float f and
            // an image narrower than 2^24 — the float form keeps two FP64 constants out of the registers for the whole forward pass)
            const bool border = fwd.x < 5.f || fwd.y < 5.f || fwd.x > (float) (img_w - 5) || fwd.y > (float) (img_h - 5);
            if (!st_f || border) break; // wave-uniform
        }
        const unsigned char *a = dir ? sN : sP, *b = dir ? sP : sN;
        const float2 from      = dir ? fwd : p0;
        float2 io              = dir ? bwd : fwd;
        const bool st          = lk_track_wave(P, a, b, from, io, S, lane, nullptr);
        if (dir) {
            bwd  = io;
            st_b = st;
        } else {
            fwd  = io;
            st_f = st;
        }
    }
    // The verdict on the point — isOnBorder, ptsDistance, undistortPoints — is k_lk_finish's, a THREAD per point: as the tail of this kernel it
    // was ~390 wave-instructions per point that every lane computed alike, most of them FP64 (five divisions of the undistortion's fixed-point
    // iteration, a square root): 5.5 % of the kernel's vector instructions and more of its time (round 6).
    if (lane == 0) {
        out_pts[i] = fwd;
        out_bwd[i] = bwd;
        status[i]  = (unsigned char) ((st_f && st_b) ? 1u : 0u); // (the backward pass is skipped for a lost or border point: st_b stays false)
    }
}

// isOnBorder (tracking.cc:847-849) and ptsDistance (tracking.cc:841-845) on the forward / backward results of k_lk_track_fb, then
// undistortPoints of the forward result; one thread per point.  bwd_undist holds the backward result on entry and, when the caller asked for
// undistorted points, the undistorted point on exit (the segmented tracker passes its undistorted-point array; the C entry point a scratch
// array or the caller's).
__global__ __launch_bounds__(256) void k_lk_finish(int n, const float2 *prev_pts, const float2 *out_pts, float2 *bwd_undist, unsigned char *status,
                                                   int want_undist, int has_cam, icg_camera cam, int img_w, int img_h, int seg_cap,
                                                   const int32_t *seg_count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (seg_count) {
        const int s = i / seg_cap;
        if (i - s * seg_cap >= seg_count[s]) return;
    }
    const float2 p0 = prev_pts[i], fwd = out_pts[i], bwd = bwd_undist[i];
    const bool border = (double) fwd.x < 5.0 || (double) fwd.y < 5.0 || ((double) fwd.x > (img_w - 5.0)) || ((double) fwd.y > (img_h - 5.0));
    const double ddx = (double) (bwd.x - p0.x), ddy = (double) (bwd.y - p0.y);
    const double dist = sqrt(ddx * ddx + ddy * ddy);
    status[i] = (unsigned char) ((status[i] && !border && dist < 0.5) ? 1u : 0u);
    if (want_undist) bwd_undist[i] = has_cam ? cam_undistort(cam, fwd) : fwd;
}

// order-preserving compaction indices (what reduceVector keeps, tracking.cc:831-839); single workgroup scan.
__global__ __launch_bounds__(1024) void k_keep_indices(int n, const unsigned char *status, int32_t *keep_idx, int32_t *n_keep) {
    __shared__ int wave_tot[16];
    __shared__ int base;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) base = 0;
    __syncthreads();
    for (int start = 0; start < n; start += 1024) {
        const int i   = start + t;
        const bool k  = i < n && status[i];
        const unsigned long long m = __ballot(k);
        const int pre = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wv; w++) off += wave_tot[w];
        if (k) keep_idx[off + pre] = i;
        __syncthreads();
        if (t == 0) {
            int s = 0;
            for (int w = 0; w < 16; w++) s += wave_tot[w];
            base += s;
        }
        __syncthreads();
    }
    if (t == 0) *n_keep = base;
}

// ---- host side ---------------------------------------------------------------------------------------------
static int check_slots(icg_ctx *ctx, int n, const int32_t *a, const int32_t *b) {
    for (int i = 0; i < n; i++)
        if (a[i] < 0 || a[i] >= ctx->cfg.n_slots || b[i] < 0 || b[i] >= ctx->cfg.n_slots)
            return icg_fail(ctx, ICG_ERR_INVALID, "slot index out of range at point %d", i);
    return 0;
}

extern "C" int icg_lk_track(icg_ctx *ctx, int n, const int32_t *prev_slot, const int32_t *next_slot, const float *prev_pts,
                            float *next_pts, uint8_t *status, float *err) {
    if (!ctx || n < 0) return ICG_ERR_INVALID;
    if (n == 0) return ICG_OK;
    if (!prev_slot || !next_slot || !prev_pts || !next_pts || !status) return ICG_ERR_INVALID;
    if (n > ctx->cfg.max_points) return icg_fail(ctx, ICG_ERR_CAPACITY, "%d points > max_points %d", n, ctx->cfg.max_points);
    int rc = check_slots(ctx, n, prev_slot, next_slot);
    if (rc) return rc;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    icg_call c(ctx);
    if ((rc = c.reserve((size_t) n * 64))) return rc;
    const int32_t *d_ps = c.in_zc(prev_slot, (size_t) n);
    const int32_t *d_ns = c.in_zc(next_slot, (size_t) n);
    const float2 *d_pp  = (const float2 *) c.in_zc(prev_pts, 2 * (size_t) n);
    float2 *d_np        = (float2 *) c.out_zc(next_pts, 2 * (size_t) n); // in/out: initial flow in, result out
    memcpy(d_np, next_pts, sizeof(float) * 2 * (size_t) n);
    unsigned char *d_st = c.out_zc(status, (size_t) n);
    float *d_err        = err ? c.out_zc(err, (size_t) n) : nullptr;
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "lk_track");
        hipLaunchKernelGGL(k_lk_track, dim3(icg_xcd_grid(n)), dim3(64), 0, ctx->stream, icg_make_pyr_desc(ctx), n, d_ps, d_ns, d_pp, d_np,
                           d_st, d_err);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

extern "C" int icg_lk_track_fb(icg_ctx *ctx, int n, const int32_t *prev_slot, const int32_t *next_slot,
                               const float *prev_pts, const float *guess_pts, float *out_pts, uint8_t *status,
                               float *out_undist, int32_t *keep_idx, int32_t *n_keep) {
    if (!ctx || n < 0) return ICG_ERR_INVALID;
    if (n == 0) {
        if (n_keep) *n_keep = 0;
        return ICG_OK;
    }
    if (!prev_slot || !next_slot || !prev_pts || !guess_pts || !out_pts || !status) return ICG_ERR_INVALID;
    if ((keep_idx == nullptr) != (n_keep == nullptr)) return ICG_ERR_INVALID;
    if (out_undist && !ctx->has_cam) return icg_fail(ctx, ICG_ERR_INVALID, "out_undist requested but camera not set");
    if (n > ctx->cfg.max_points) return icg_fail(ctx, ICG_ERR_CAPACITY, "%d points > max_points %d", n, ctx->cfg.max_points);
    int rc = check_slots(ctx, n, prev_slot, next_slot);
    if (rc) return rc;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    icg_call c(ctx);
    if ((rc = c.reserve((size_t) n * 104))) return rc;
    const int32_t *d_ps = c.in_zc(prev_slot, (size_t) n);
    const int32_t *d_ns = c.in_zc(next_slot, (size_t) n);
    const float2 *d_pp  = (const float2 *) c.in_zc(prev_pts, 2 * (size_t) n);
    const float2 *d_gs  = (const float2 *) c.in_zc(guess_pts, 2 * (size_t) n);
    float2 *d_out       = (float2 *) c.out_zc(out_pts, 2 * (size_t) n);
    unsigned char *d_st = c.out_zc(status, (size_t) n);
    // (the backward results travel from k_lk_track_fb to k_lk_finish in the undistorted-point array, or in device scratch when none was asked for)
    float2 *d_und       = out_undist ? (float2 *) c.out_zc(out_undist, 2 * (size_t) n) : (float2 *) c.out<float>(nullptr, 2 * (size_t) n);
    int32_t *d_keep     = keep_idx ? c.out_zc(keep_idx, (size_t) n) : nullptr;
    int32_t *d_nkeep    = keep_idx ? c.out_zc(n_keep, 1) : nullptr;
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "lk_track_fb");
        hipLaunchKernelGGL(k_lk_track_fb, dim3(icg_xcd_grid(n)), dim3(64), 0, ctx->stream, icg_make_pyr_desc(ctx), n, d_ps, d_ns, d_pp, d_gs, d_out,
                           d_st, d_und, ctx->cfg.width, ctx->cfg.height, 0, (const int32_t *) nullptr);
    }
    {
        icg_prof_scope ps(ctx, "lk_finish");
        hipLaunchKernelGGL(k_lk_finish, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, d_pp, (const float2 *) d_out, d_und, d_st,
                           out_undist ? 1 : 0, ctx->has_cam ? 1 : 0, ctx->cam, ctx->cfg.width, ctx->cfg.height, 0, (const int32_t *) nullptr);
    }
    if (keep_idx) {
        icg_prof_scope ps(ctx, "keep_indices");
        hipLaunchKernelGGL(k_keep_indices, dim3(1), dim3(1024), 0, ctx->stream, n, d_st, d_keep, d_nkeep);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

// Segmented launch for the device-resident tracker (tracker.hip): every array is device memory laid out as n_seg segments of seg_cap
// entries, d_count[s] of them valid; asynchronous on the context's stream (the tracker's next stage kernel is stream-ordered behind it).
int icg_lk_launch_segments(icg_ctx *ctx, int n_seg, int seg_cap, const int32_t *d_count, const int32_t *d_prev_slot, const int32_t *d_next_slot,
                           const float2 *d_prev, const float2 *d_guess, float2 *d_out, uint8_t *d_status, float2 *d_undist) {
    if (!ctx->has_cam) return icg_fail(ctx, ICG_ERR_INVALID, "camera not set");
    const int n = n_seg * seg_cap;
    {
        icg_prof_scope ps(ctx, "lk_track_fb");
        hipLaunchKernelGGL(k_lk_track_fb, dim3(icg_xcd_grid(n)), dim3(64), 0, ctx->stream, icg_make_pyr_desc(ctx), n, d_prev_slot, d_next_slot, d_prev,
                           d_guess, d_out, d_status, d_undist, ctx->cfg.width, ctx->cfg.height, seg_cap, d_count);
    }
    {
        icg_prof_scope ps(ctx, "lk_finish");
        hipLaunchKernelGGL(k_lk_finish, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, d_prev, (const float2 *) d_out, d_undist, d_status, 1, 1,
                           ctx->cam, ctx->cfg.width, ctx->cfg.height, seg_cap, d_count);
    }
    ICG_HIP(ctx, hipGetLastError());
    return ICG_OK;
}
