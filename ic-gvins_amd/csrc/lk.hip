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
//     DPP butterflies in 32 bits up to 16 / 8 lanes, then v_readlane + 64-bit scalar adds (uniform result in SGPRs);
//     every multiply has 24-bit operands -> full-rate v_mul_i32_i24 / v_mad_i32_i24.
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

struct lk_smem {
    unsigned int I[LK_IT * LK_IS / 4];
    unsigned int J[LK_JT * LK_JS / 4 + 1];
};

// Exact wave-wide integer sums, result uniform (SGPRs).  The per-lane partials are bounded (see the kernel comment), so
// the first butterfly stages run in 32 bits as fused DPP adds (quad_perm xor1, xor2, row_half_mirror[, row_mirror]); the
// 8 (or 4) group sums are then read with v_readlane and added as 64-bit SCALAR integers — no LDS crossbar round trips.
__device__ __forceinline__ int dpp_add_xor1(int v) { return v + __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false); }
__device__ __forceinline__ int dpp_add_xor2(int v) { return v + __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false); }
__device__ __forceinline__ int dpp_add_half_mirror(int v) { return v + __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false); }
__device__ __forceinline__ int dpp_add_mirror(int v) { return v + __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false); }

// |per-lane partial| <= 2^28: sums of 8 lanes fit in int32
__device__ __forceinline__ long long wave_sum_i32x8(int v) {
    v = dpp_add_xor1(v);
    v = dpp_add_xor2(v);
    v = dpp_add_half_mirror(v);
    long long s = 0;
#pragma unroll
    for (int g = 0; g < 8; g++) s += (long long) __builtin_amdgcn_readlane(v, g * 8);
    return s;
}
// |per-lane partial| <= 2^27: sums of 16 lanes fit in int32
__device__ __forceinline__ long long wave_sum_i32x16(int v) {
    v = dpp_add_xor1(v);
    v = dpp_add_xor2(v);
    v = dpp_add_half_mirror(v);
    v = dpp_add_mirror(v);
    long long s = 0;
#pragma unroll
    for (int g = 0; g < 4; g++) s += (long long) __builtin_amdgcn_readlane(v, g * 16);
    return s;
}

__device__ __forceinline__ int lk_descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

__device__ __forceinline__ void lk_weights(float a, float b, int &w00, int &w01, int &w10, int &w11) {
    w00 = (int) rintf((1.f - a) * (1.f - b) * (float) (1 << 14));
    w01 = (int) rintf(a * (1.f - b) * (float) (1 << 14));
    w10 = (int) rintf((1.f - a) * b * (float) (1 << 14));
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

// 32x32 u8 tile of the next image: lane -> (row lane>>1, 16 pixels at column (lane&1)*16) = 4 packed dwords
__device__ __forceinline__ void lk_stage_J(lk_smem &S, const unsigned char *J, int W, int H, int pitch, int jx0, int jy0,
                                           int lane) {
    const int r  = lane >> 1;
    const int c0 = (lane & 1) * 16;
    const unsigned char *row = J + (size_t) icg_reflect101(jy0 + r, H) * pitch;
    unsigned int *dst        = &S.J[(r * LK_JS + c0) >> 2];
#pragma unroll
    for (int q = 0; q < 4; q++) dst[q] = lk_load4(row, jx0 + c0 + 4 * q, W);
}

// the lane's two 8-pixel rows of the window at integer position (inx, iny), from the staged tile (3 aligned dwords + a
// byte funnel shift per row)
__device__ __forceinline__ void lk_fetch_J(const lk_smem &S, int row0, int o, unsigned int &a0, unsigned int &a1,
                                           unsigned int &b0, unsigned int &b1) {
    const int idx = o >> 2, sh = o & 3;
    const unsigned int *r0 = &S.J[(row0 * LK_JS >> 2) + idx];
    const unsigned int *r1 = r0 + (LK_JS >> 2);
    const unsigned int d0 = r0[0], d1 = r0[1], d2 = r0[2];
    const unsigned int e0 = r1[0], e1 = r1[1], e2 = r1[2];
    a0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
    a1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
    b0 = __builtin_amdgcn_alignbyte(e1, e0, sh);
    b1 = __builtin_amdgcn_alignbyte(e2, e1, sh);
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
        int w00, w01, w10, w11;
        lk_weights(prevx - ipx, prevy - ipy, w00, w01, w10, w11);

        // ---- stage the 24x24 neighbourhood of the previous image (tile (r,c) <-> image (ipx-1+c, ipy-1+r)) ----
        __syncthreads(); // previous level's LDS readers are done
        for (int i = lane; i < LK_IT * 6; i += 64) {
            const int r = i / 6, cd = i - r * 6;
            const unsigned char *row = I + (size_t) icg_reflect101(ipy - 1 + r, H) * pitch;
            S.I[(r * LK_IS >> 2) + cd] = lk_load4(row, ipx - 1 + 4 * cd, W);
        }
        __syncthreads();

        // ---- per lane: 4 rows x 10 bytes -> I samples and on-the-fly Scharr derivatives of its 7-pixel run ----
        int iv[7], ix[7], iy[7];
        int sA11 = 0, sA12 = 0, sA22 = 0;
        {
            // rows ly..ly+3, byte columns lx0..lx0+9 of the tile
            const int idx = lx0 >> 2, sh = lx0 & 3;
            unsigned int wv[4][3];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const unsigned int *p = &S.I[((ly + r) * LK_IS >> 2) + idx];
                const unsigned int d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3];
                wv[r][0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
                wv[r][1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
                wv[r][2] = __builtin_amdgcn_alignbyte(d3, d2, sh);
            }
#define LK_T(r, j) lk_byte(wv[r][(j) >> 2], (j) &3)
            // derivative at support position (c = lx0+j, r = ly+rr): 3x3 neighbourhood = tile rows rr..rr+2, cols j..j+2
            int dxv[2][8], dyv[2][8];
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const int Y = ipy + ly + rr;
                const bool yin = Y >= 0 && Y < H;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int X = ipx + lx0 + j;
                    const int p00 = LK_T(rr, j), p01 = LK_T(rr, j + 1), p02 = LK_T(rr, j + 2);
                    const int p10 = LK_T(rr + 1, j), p12 = LK_T(rr + 1, j + 2);
                    const int p20 = LK_T(rr + 2, j), p21 = LK_T(rr + 2, j + 1), p22 = LK_T(rr + 2, j + 2);
                    const int t0m = 3 * (p00 + p20) + 10 * p10;
                    const int t0p = 3 * (p02 + p22) + 10 * p12;
                    const int t1m = p20 - p00, t1c = p21 - p01, t1p = p22 - p02;
                    const bool in = yin && X >= 0 && X < W; // derivative plane is ZERO outside the image
                    dxv[rr][j]    = in ? (t0p - t0m) : 0;
                    dyv[rr][j]    = in ? (3 * (t1m + t1p) + 10 * t1c) : 0;
                }
            }
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const int a0 = LK_T(1, k + 1), b0 = LK_T(1, k + 2), a1 = LK_T(2, k + 1), b1 = LK_T(2, k + 2);
                iv[k] = lk_descale(__mul24(a0, w00) + __mul24(b0, w01) + __mul24(a1, w10) + __mul24(b1, w11), 14 - 5);
                ix[k] = lk_descale(__mul24(dxv[0][k], w00) + __mul24(dxv[0][k + 1], w01) + __mul24(dxv[1][k], w10) + __mul24(dxv[1][k + 1], w11), 14);
                iy[k] = lk_descale(__mul24(dyv[0][k], w00) + __mul24(dyv[0][k + 1], w01) + __mul24(dyv[1][k], w10) + __mul24(dyv[1][k + 1], w11), 14);
                if (!active) {
                    iv[k] = 0;
                    ix[k] = 0;
                    iy[k] = 0;
                }
                sA11 += __mul24(ix[k], ix[k]);
                sA12 += __mul24(ix[k], iy[k]);
                sA22 += __mul24(iy[k], iy[k]);
            }
#undef LK_T
        }
        const long long iA11 = wave_sum_i32x16(sA11), iA12 = wave_sum_i32x16(sA12), iA22 = wave_sum_i32x16(sA22);
        const float A11 = (float) iA11 * FLT_SCALE, A12 = (float) iA12 * FLT_SCALE, A22 = (float) iA22 * FLT_SCALE;
        float D            = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float) (2 * ICG_LK_WIN * ICG_LK_WIN);
        if (minEig < 1e-4f || D < FLT_EPSILON) {
            if (level == 0) status = false;
            continue;
        }
        D = 1.f / D;
        nptx -= (float) ICG_LK_HALF;
        npty -= (float) ICG_LK_HALF;
        float pdx = 0.f, pdy = 0.f;
        int jx0 = -1000000, jy0 = -1000000;
        int cinx = -1000000, ciny = -1000000;     // integer window position whose bytes are cached in ja/jb
        unsigned int ja0 = 0, ja1 = 0, jb0 = 0, jb1 = 0; // the lane's 2 x 8 bytes of the next image

        for (int j = 0; j < LK_MAX_ITERS; j++) {
            const int inx = (int) floorf(nptx), iny = (int) floorf(npty);
            if (inx < -ICG_LK_WIN || inx >= W || iny < -ICG_LK_WIN || iny >= H) {
                if (level == 0) status = false;
                break;
            }
            if (inx != cinx || iny != ciny) {
                if (!(inx >= jx0 && inx <= jx0 + (LK_JT - 22) && iny >= jy0 && iny <= jy0 + (LK_JT - 22))) {
                    jx0 = inx - LK_JM;
                    jy0 = iny - LK_JM;
                    __syncthreads();
                    lk_stage_J(S, J, W, H, pitch, jx0, jy0, lane);
                    __syncthreads();
                }
                lk_fetch_J(S, iny - jy0 + ly, (inx - jx0) + lx0, ja0, ja1, jb0, jb1);
                cinx = inx;
                ciny = iny;
            }
            lk_weights(nptx - inx, npty - iny, w00, w01, w10, w11);
            int sb1 = 0, sb2 = 0;
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const int a0 = lk_byte(k < 4 ? ja0 : ja1, k & 3), b0 = lk_byte(k + 1 < 4 ? ja0 : ja1, (k + 1) & 3);
                const int a1 = lk_byte(k < 4 ? jb0 : jb1, k & 3), b1 = lk_byte(k + 1 < 4 ? jb0 : jb1, (k + 1) & 3);
                const int diff = lk_descale(__mul24(a0, w00) + __mul24(b0, w01) + __mul24(a1, w10) + __mul24(b1, w11), 14 - 5) - iv[k];
                sb1 += __mul24(diff, ix[k]);
                sb2 += __mul24(diff, iy[k]);
            }
            const long long ib1 = wave_sum_i32x8(sb1), ib2 = wave_sum_i32x8(sb2);
            const float b1 = (float) ib1 * FLT_SCALE, b2 = (float) ib2 * FLT_SCALE;
            const float dx = (float) ((A12 * b2 - A22 * b1) * D);
            const float dy = (float) ((A12 * b1 - A11 * b2) * D);
            nptx += dx;
            npty += dy;
            nextStore = make_float2(nptx + (float) ICG_LK_HALF, npty + (float) ICG_LK_HALF);
            if ((double) dx * dx + (double) dy * dy <= eps2) break;
            if (j > 0 && fabs((double) (dx + pdx)) < 0.01 && fabs((double) (dy + pdy)) < 0.01) {
                nextStore.x -= dx * 0.5f;
                nextStore.y -= dy * 0.5f;
                break;
            }
            pdx = dx;
            pdy = dy;
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
                    lk_fetch_J(S, iny - jy0 + ly, (inx - jx0) + lx0, ja0, ja1, jb0, jb1);
                }
                lk_weights(ex - inx, ey - iny, w00, w01, w10, w11);
                int se = 0;
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    const int a0 = lk_byte(k < 4 ? ja0 : ja1, k & 3), b0 = lk_byte(k + 1 < 4 ? ja0 : ja1, (k + 1) & 3);
                    const int a1 = lk_byte(k < 4 ? jb0 : jb1, k & 3), b1 = lk_byte(k + 1 < 4 ? jb0 : jb1, (k + 1) & 3);
                    int diff = lk_descale(__mul24(a0, w00) + __mul24(b0, w01) + __mul24(a1, w10) + __mul24(b1, w11), 14 - 5) - iv[k];
                    if (!active) diff = 0;
                    se += diff < 0 ? -diff : diff;
                }
                const long long ie = wave_sum_i32x16(se);
                errv               = (float) ie * 1.f / (float) (32 * ICG_LK_WIN * ICG_LK_WIN);
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
    const int i = blockIdx.x;
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

__global__ __launch_bounds__(64) void k_lk_track_fb(icg_pyr_desc P, int n, const int32_t *prev_slot,
                                                    const int32_t *next_slot, const float2 *prev_pts,
                                                    const float2 *guess_pts, float2 *out_pts, unsigned char *status,
                                                    int has_cam, icg_camera cam, float2 *out_undist, int img_w, int img_h) {
    __shared__ lk_smem S;
    const int i = blockIdx.x;
    if (i >= n) return;
    const int lane = threadIdx.x;
    const unsigned char *sP = P.base + (size_t) prev_slot[i] * P.slot_bytes;
    const unsigned char *sN = P.base + (size_t) next_slot[i] * P.slot_bytes;
    const float2 p0 = prev_pts[i];
    float2 fwd      = guess_pts[i];
    const bool st_f = lk_track_wave(P, sP, sN, p0, fwd, S, lane, nullptr);
    float2 bwd      = p0;
    const bool st_b = lk_track_wave(P, sN, sP, fwd, bwd, S, lane, nullptr);
    if (lane == 0) {
        // isOnBorder (tracking.cc:847-849) and ptsDistance (tracking.cc:841-845)
        const bool border = (double) fwd.x < 5.0 || (double) fwd.y < 5.0 || ((double) fwd.x > (img_w - 5.0)) ||
                            ((double) fwd.y > (img_h - 5.0));
        const double ddx = (double) (bwd.x - p0.x), ddy = (double) (bwd.y - p0.y);
        const double dist = sqrt(ddx * ddx + ddy * ddy);
        out_pts[i]        = fwd;
        status[i]         = (st_f && st_b && !border && dist < 0.5) ? 1 : 0;
        if (out_undist) out_undist[i] = has_cam ? cam_undistort(cam, fwd) : fwd;
    }
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
    {
        icg_prof_scope ps(ctx, "lk_track");
        hipLaunchKernelGGL(k_lk_track, dim3(n), dim3(64), 0, ctx->stream, icg_make_pyr_desc(ctx), n, d_ps, d_ns, d_pp, d_np,
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
    if ((rc = c.reserve((size_t) n * 96))) return rc;
    const int32_t *d_ps = c.in_zc(prev_slot, (size_t) n);
    const int32_t *d_ns = c.in_zc(next_slot, (size_t) n);
    const float2 *d_pp  = (const float2 *) c.in_zc(prev_pts, 2 * (size_t) n);
    const float2 *d_gs  = (const float2 *) c.in_zc(guess_pts, 2 * (size_t) n);
    float2 *d_out       = (float2 *) c.out_zc(out_pts, 2 * (size_t) n);
    unsigned char *d_st = c.out_zc(status, (size_t) n);
    float2 *d_und       = out_undist ? (float2 *) c.out_zc(out_undist, 2 * (size_t) n) : nullptr;
    int32_t *d_keep     = keep_idx ? c.out_zc(keep_idx, (size_t) n) : nullptr;
    int32_t *d_nkeep    = keep_idx ? c.out_zc(n_keep, 1) : nullptr;
    {
        icg_prof_scope ps(ctx, "lk_track_fb");
        hipLaunchKernelGGL(k_lk_track_fb, dim3(n), dim3(64), 0, ctx->stream, icg_make_pyr_desc(ctx), n, d_ps, d_ns, d_pp,
                           d_gs, d_out, d_st, ctx->has_cam ? 1 : 0, ctx->cam, d_und, ctx->cfg.width, ctx->cfg.height);
    }
    if (keep_idx) {
        icg_prof_scope ps(ctx, "keep_indices");
        hipLaunchKernelGGL(k_keep_indices, dim3(1), dim3(1024), 0, ctx->stream, n, d_st, d_keep, d_nkeep);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}
