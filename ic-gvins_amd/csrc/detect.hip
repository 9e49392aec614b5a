// F7: Tracking::featuresDetection on device — mask discs, gridded Shi-Tomasi (cv::goodFeaturesToTrack) and
// cv::cornerSubPix, batched over frames of many streams.
//
// Reference: tracking/tracking.cc:576-688 (mask :609-620, ROI per block :632-645, GFTT :647, subpix :651).
// Arithmetic definition: SURVEY.md Appendix B.7/B.8 with the formulation pinned in oracle/orc_detect.cc:
// exact-integer Sobel x (1/3060) in float, covariance products in float, 3x3 box sums accumulated in double in raster
// order (reflect-101 at the ROI edge), min-eigenvalue in float; threshold 0.01*max over unmasked ROI pixels; 3x3
// non-maximum test inside the ROI; candidates ordered by (response desc, raster address desc); greedy minimum-distance
// selection; sub-pixel refinement with sequential double accumulation (bit-identical to the CPU restatement).
//
// Kernels (HBM/L2-bound stencils, no MFMA shape):
//   k_min_eig_nms one wave per 60 x 64 block of a ROI: the mask of the block from the DISC LIST (no mask plane: the existing features whose
//                 discs reach the block are found by the wave itself, a 64-bit row set per lane), then only the RUNS of rows that hold an
//                 unmasked pixel are streamed down the image columns: response (separable exact box sums), per-ROI masked maximum by an
//                 order-preserving uint atomicMax, 3x3 NMS + mask in registers -> every local maximum (key = response bits << 32 | raster
//                 index) appended per ROI; no response plane in HBM
//   k_select      one workgroup per ROI: quality threshold 0.01*max, then repeated block-wide arg-max over live candidates +
//                 min-distance kill (equivalent to sort + greedy grid test, needs no sort and no capacity cap)
//   k_subpix      one wavefront per corner: 13x13 bilinear patch and the 121 gradient terms in parallel through LDS,
//                 the five 121-term sums each sequentially in raster order (IEEE order == CPU order), one lane per sum
#include <cfloat>

#include "icg_internal.h"


__device__ __forceinline__ unsigned int f32_order_key(float f) {
    unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_from_order_key(unsigned int k) {
    unsigned int b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

// ---------------------------------------------------------------------------------------------------------
// The mask (tracking.cc:609-620: a 255-plane with a filled cv::circle of radius track_min_pixel_distance_ zeroed at every existing feature)
// is never materialised.  Rounds 1-3 wrote it as a byte plane (k_mask_discs: 0.9 MB per frame written in spans and read back by the
// detector, 1.4-1.6 us per frame of pure HBM time and a launch); now every wave of k_min_eig_nms tests its 60 x 64 block against the
// discs directly.  A pixel (x, y) is masked iff some feature centre (cx, cy) = (rint(px), rint(py)) has |x - cx| <= halfw[|y - cy|],
// halfw = the span table of OpenCV's midpoint circle (drawing.cpp Circle(), circle_halfwidths below).  That table is non-increasing in
// |dy| (checked on the host when it is built), so per COLUMN distance a = |x - cx| the masked rows are the interval |y - cy| <= vh[a] with
// vh[a] = max{dy : halfw[dy] >= a}: one LDS lookup and a 64-bit interval mask per (lane, disc) instead of 64 span tests.  The wave finds
// the discs that reach its block itself: the job's ~300 centres in chunks of 64 (a lane each, bounding-box test, ballot), then a
// wave-uniform loop over the hits with the centre moved to SGPRs by readlane.
// ---------------------------------------------------------------------------------------------------------
// k_min_eig_nms: Shi-Tomasi response, masked per-ROI maximum AND the 3x3 non-maximum test in ONE pass over the u8 image —
// the f32 response plane of the two-kernel formulation (4 B/px written by k_min_eig, 4 B/px re-read by k_candidates) is
// never materialised.  The quality threshold 0.01*max is only known when every tile of the ROI is done, but the local-maximum
// test does not depend on it ("thresholded value equals the 3x3 maximum of the thresholded map" == v > thresh && v >= every
// RAW neighbour: a neighbour at or below the threshold is below v, one above it keeps its value), so the kernel appends every
// unmasked, non-zero local maximum and k_select drops the ones at or below the threshold.
//
// One WAVE per 60 x 64 block of NMS outputs (fe_block below), a lane per image column (64 lanes = 60 outputs + 2 halo columns each side);
// for every run [a, b] of rows that hold an unmasked owned pixel the wave STREAMS down its columns (image rows a-3 .. b+3 -> product rows
// a-2 .. b+2 -> response rows a-1 .. b+1 -> NMS rows a .. b) with three-row rolling windows in registers; the left/right neighbours' values
// move through wave-shift DPP, no LDS, no barrier:
//   image row        one dword per lane (3 pixels by one v_perm with the lane's selector: reflect-101 at true image borders folded in),
//                    running Sobel differences d = p[x+1]-p[x-1], s = p[x-1]+2p[x]+p[x+1]
//   product row      gx = d0+2d1+d2, gy = s2-s0, the three float products; horizontal 3-sums in double
//   response row     vertical 3-sum in double (exact in any order: every term is a float in [s^2,(1020 s)^2], s = 1/3060, so all
//                    partial sums are multiples of 2^-47 below 1), min-eigenvalue in float, masked maximum (ordered-uint key)
//   NMS row          centre >= max of the 8 raw neighbours, unmasked, non-zero -> wave-aggregated append
// Reflect-101 at the ROI edge (the covariance maps are ROI-sized Mats in OpenCV): a lane left/right of the ROI computes the
// products of the mirrored column (same Sobel, bit-identical), the product row above/below the ROI is the mirrored row of the
// rolling window.  Coordinates further out belong to partial blocks: clamped, their results never used.
#define FE_TW 60    // NMS output columns per wave
#define FE_TH 16    // FE_TH * FE_WAVES = 64 NMS output rows per wave (the block height; the two factors are what is left of the
#define FE_WAVES 4  // 60 x 16 tiles of rounds 2-5) — and FE_WAVES waves = FE_WAVES neighbouring blocks per workgroup
#define FE_MAX_RADIUS 1022 // largest disc radius the LDS span table holds (min_dist; 45 at C2)
#ifndef FE_RESIDENT_PER_XCD
#define FE_RESIDENT_PER_XCD 192 // workgroups of k_min_eig_nms per XCD (x 8 XCDs x 4 waves = the kernel's residency; fewer when the grid is smaller)
#endif

// lane i <- lane i-1 / lane i+1 across the whole wave (wave_shr:1 / wave_shl:1); the edge lanes receive 0 and are halo
__device__ __forceinline__ float from_left(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float from_right(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130, 0xF, 0xF, true));
}

// wave-wide OR, total in lane 63: xor 1, xor 2, row_half_mirror, row_mirror inside every row of 16 lanes, then row_bcast15 / row_bcast31 (one
// fused DPP operand each; lanes a broadcast does not reach OR in 0)
__device__ __forceinline__ unsigned int wave_or_to_lane63(unsigned int v) {
    v |= (unsigned int) __builtin_amdgcn_update_dpp(0, (int) v, 0xB1, 0xF, 0xF, false);
    v |= (unsigned int) __builtin_amdgcn_update_dpp(0, (int) v, 0x4E, 0xF, 0xF, false);
    v |= (unsigned int) __builtin_amdgcn_update_dpp(0, (int) v, 0x141, 0xF, 0xF, false);
    v |= (unsigned int) __builtin_amdgcn_update_dpp(0, (int) v, 0x140, 0xF, 0xF, false);
    v |= (unsigned int) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xA, 0xF, false);
    v |= (unsigned int) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xC, 0xF, false);
    return v;
}

// one 60 x 64 BLOCK per wave (round 6; rounds 2-5 and the first half of round 6 gave every 60 x 16 tile a wave of its own):
//   1. the mask of the whole block from the disc list — ONE pass over the job's centres for 64 rows (a 64-bit row set per lane) instead of
//      four passes for 16 rows each: the bounding-box chunks fall from 16 to 4 per block and the (lane, disc) interval updates from ~44 to
//      ~16, and the wave leaves as soon as every owned pixel is masked;
//   2. the rows that hold an unmasked owned pixel, as RUNS (gaps of up to FE_GAP rows are streamed through: re-priming the rolling
//      windows costs six rows); each run [a, b] is streamed from image row a - 3 to b + 3 in chunks of FE_CH rows; a row's register is
//      reloaded with the same row of the next chunk as soon as it has been consumed.  The row step is the same for every row: the first rows of a run prime the
//      windows with values nobody reads (every output is gated by the run's own unmasked bits).
#ifndef FE_CH
#define FE_CH 4   // image rows per load chunk (and the length of a run's priming chunk: keep it 4)
#endif
#define FE_GAP 6  // masked rows between two needed rows that are streamed rather than skipped

// v_max_f32 without the two canonicalising v_max x, x the compiler puts in front of fmaxf for values that arrive through DPP (no NaN here)
__device__ __forceinline__ float fe_max(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// the rolling windows of one run of rows and the step that consumes FE_CH image rows
enum { FE_PLAIN = 0, FE_MIRRORS = 1, FE_PRIME = 2 };
struct fe_rows {
    int d0 = 0, d1 = 0, s0 = 0, s1 = 0;                                    // Sobel differences of the two image rows before
    double hA0 = 0, hA1 = 0, hB0 = 0, hB1 = 0, hC0 = 0, hC1 = 0;           // horizontal 3-sums of the two product rows before
    float eC = 0.f, sideC = 0.f, m3T = 0.f, m3C = 0.f;                     // response rows: centre and top of the NMS window
    bool unmC = false;
    float emax = -INFINITY;                                                // maximum of the unmasked owned responses (its key is formed once)

    // image row y (ROI coordinate) -> product row y - 1 -> response row y - 2 -> NMS row y - 3.
    // FE_PRIME: the first chunk of a run — its two first rows only start the Sobel differences, the other two only the product windows; no
    // response of these rows is ever read.  FE_MIRRORS: the chunk holds one of the two mirrored product rows of the ROI (row -1 := row 1 when
    // image row 2 is consumed, row rh := row rh-2 when row rh+1 is): as run-time tests in the only copy they cost eight 64-bit moves per row.
    template <int MODE>
    __device__ __forceinline__ void chunk(unsigned int (&pix)[FE_CH], int y_last /* last image row of the run */, const uint8_t *img,
                                          int base /* the lane's column offset */, int pitch, int ry, int h, unsigned int uc, int y0, int rh, int rw,
                                          unsigned int sel, float s, bool nms_col, int lane, int roi, int xcol, int32_t *cand_cnt,
                                          unsigned long long *cbase) {
        typedef unsigned int __attribute__((aligned(1))) u32u;
#pragma unroll
        for (int i = 0; i < FE_CH; i++) {
            const int y = y0 + i; // wave-uniform
            // the lane's three pixels as bytes 0, 1, 2 (one v_perm with the lane's selector; the byte operands below are SDWA selects)
            const unsigned int v = __builtin_amdgcn_perm(0u, pix[i], sel);
            // the register is free: the same row of the NEXT chunk goes into it now and has FE_CH - 1 rows of arithmetic to arrive
            {   // (wave-uniform row base + per-lane 32-bit column offset; past the run's last row the last row again: no branch, a cache hit)
                const uint8_t *rowp = img + (size_t) icg_reflect1(min(max(ry + min(y + FE_CH, y_last), -1), h), h) * pitch;
                unsigned int vb = (unsigned int) base;
                asm("" : "+v"(vb)); // (opaque per row: hoisted out of the loop, img + base becomes a 64-bit address per lane and every row a 64-bit mad)
                pix[i] = *reinterpret_cast<const u32u *>(rowp + vb);
            }
            const int a3 = (int) (v & 0xffu), b3 = (int) ((v >> 8) & 0xffu), c3 = (int) ((v >> 16) & 0xffu);
            const int dN = c3 - a3, sN = a3 + 2 * b3 + c3;
            if (MODE != FE_PRIME || i >= 2) {
                const int gxv = d0 + 2 * d1 + dN, gyv = sN - s0;
                const float dx = (float) gxv * s, dy = (float) gyv * s;
                const float cx = dx * dx, cm = dx * dy, cy = dy * dy;
                double hAn = ((double) from_left(cx) + (double) cx) + (double) from_right(cx);
                double hBn = ((double) from_left(cm) + (double) cm) + (double) from_right(cm);
                double hCn = ((double) from_left(cy) + (double) cy) + (double) from_right(cy);
                if (MODE != FE_PRIME) {
                    double tA = hA0, tB = hB0, tC = hC0;
                    if (MODE == FE_MIRRORS) {
                        if (y - 1 == rh) { // product row rh := row rh-2 (wave-uniform: a real branch)
                            asm volatile("");
                            hAn = hA0, hBn = hB0, hCn = hC0;
                        }
                        if (y == 2) { // product row -1 := row 1 (the first response row of the ROI)
                            asm volatile("");
                            tA = hAn, tB = hBn, tC = hCn;
                        }
                    }
                    const double sa = (tA + hA1) + hAn, sb = (tB + hB1) + hBn, sc = (tC + hC1) + hCn;
                    const float fa = (float) sa * 0.5f, fb = (float) sb, fc = (float) sc * 0.5f;
                    // (the argument: fa, fc, fb are multiples of 2^-48, so it is zero or at least 2^-96 — icg_sqrt_unscaled's domain)
                    const float e  = (fa + fc) - icg_sqrt_unscaled((fa - fc) * (fa - fc) + fb * fb);
                    const bool unm = (uc >> i) & 1u;
                    emax           = fmaxf(emax, unm ? e : -INFINITY);
                    const float el = from_left(e), er = from_right(e);
                    const float side = fe_max(el, er), m3 = fe_max(side, e);
                    {
                        // NMS: centre = response row y - 3
                        const int yc = y - 3;
                        const bool is_cand = (yc >= 1 && yc < rh - 1) && nms_col && unmC && eC != 0.f && eC >= fmaxf(fmaxf(m3T, sideC), m3);
                        // wave-aggregated append: one atomic per wavefront and row
                        const unsigned long long m = __ballot(is_cand);
                        if (m != 0) {
                            int slot0 = 0;
                            const int leader = __ffsll((long long) m) - 1;
                            if (lane == leader) slot0 = atomicAdd(&cand_cnt[roi], __popcll(m));
                            slot0 = __shfl(slot0, leader, 64);
                            if (is_cand)
                                cbase[slot0 + __popcll(m & ((1ull << lane) - 1ull))] =
                                    ((unsigned long long) f32_order_key(eC) << 32) | (unsigned int) (yc * rw + xcol);
                        }
                    }
                    m3T   = m3C;
                    m3C   = m3;
                    eC    = e;
                    sideC = side;
                    unmC  = unm;
                }
                hA0 = hA1, hA1 = hAn;
                hB0 = hB1, hB1 = hBn;
                hC0 = hC1, hC1 = hCn;
            }
            d0 = d1, d1 = dN;
            s0 = s1, s1 = sN;
        }
    }
};
__device__ __forceinline__ void fe_block(const int bl, const int *vh, const det_roi *rois, const uint8_t *frames, size_t slot_bytes,
                                         const int32_t *slots, int pitch, int w, int h, const float2 *mask_pts, const int32_t *mask_begin,
                                         const int32_t *mask_cnt, int radius, unsigned int *roi_max, unsigned long long *cand, size_t cand_plane,
                                         int32_t *cand_cnt, int gx, int gy, unsigned int m_roi, unsigned int m_gx) {
    const int roi = icg_div_by_magic(bl, m_roi), rem = bl - roi * (gx * gy);
    const int by = icg_div_by_magic(rem, m_gx), bx = rem - by * gx;
    const det_roi R = rois[roi];
    if (R.quota <= 0) return; // inactive entry of a dense (job, block) table (device-resident tracker); wave-uniform
    const int lane = threadIdx.x & 63;
    const int tx0 = bx * FE_TW, by0 = by * (FE_TH * FE_WAVES);
    if (tx0 >= R.rw || by0 >= R.rh) return; // wave-uniform
    const uint8_t *img = frames + (size_t) slots[R.job] * slot_bytes;
    const float s      = (float) (1.0 / 3060.0);

    // the lane's column: ROI coordinate xcol, mirrored at the ROI edge, then the three image columns with reflect-101 at the true
    // image border, fetched as byte selects of ONE dword (base <= all three <= base+3)
    const int xcol = tx0 - 2 + lane;
    const int X    = R.rx + icg_reflect1(min(max(xcol, -1), R.rw), R.rw);
    const int ca = icg_reflect1(X - 1, w), cc = icg_reflect1(X + 1, w);
    const int base = min(max(X - 1, 0), w - 4);
    const unsigned int sel = 0x0c000000u | (unsigned int) (ca - base) | ((unsigned int) (X - base) << 8) | ((unsigned int) (cc - base) << 16); // v_perm selector
    const bool own_col  = lane >= 2 && lane <= FE_TW + 1 && xcol < R.rw;   // columns whose responses this block owns
    const bool nms_col  = own_col && xcol >= 1 && xcol < R.rw - 1;
    const int mx        = R.rx + xcol; // image column of the lane
    unsigned long long *cbase = cand + (size_t) R.job * cand_plane + R.cand_base;

    // ---- 1. the mask: existing features blank discs of radius min_dist, which cover most of a tracked image ----
    const int rows = min(R.rh - by0, FE_TH * FE_WAVES); // block rows inside the ROI (>= 1 here)
    const unsigned long long rowset = rows >= 64 ? ~0ull : ((1ull << rows) - 1ull);
    unsigned long long unmasked = own_col ? rowset : 0ull; // bit k: the owned response of block row k is not masked
    {
        const int ytop = R.ry + by0;                          // image row of block row 0
        const int xl = R.rx + tx0, xr = xl + FE_TW - 1;        // image columns of the owned lanes
        const int p0 = mask_begin[R.job], p1 = p0 + mask_cnt[R.job];
        for (int pb = p0; pb < p1; pb += 64) {
            const int i = pb + lane;
            int cx = 0, cy = 0;
            bool hit = false;
            if (i < p1) {
                const float2 p = mask_pts[i];
                cx  = (int) rintf(p.x); // cvRound of the key point (tracking.cc:613, 618)
                cy  = (int) rintf(p.y);
                hit = cx + radius >= xl && cx - radius <= xr && cy + radius >= ytop && cy - radius <= ytop + rows - 1;
            }
            unsigned long long m = __ballot(hit);
            while (m) { // wave-uniform
                const int l = __ffsll((long long) m) - 1;
                m &= m - 1;
                const int ccx = __builtin_amdgcn_readlane(cx, l), ccy = __builtin_amdgcn_readlane(cy, l);
                const int adx = abs(mx - ccx);
                const int v   = vh[min(adx, radius + 1)]; // rows |y - ccy| <= v of this column are inside the disc (-1: none)
                const int lo = max(ccy - v - ytop, 0), hi = min(ccy + v - ytop, 63);
                if (v >= 0 && lo <= hi) unmasked &= ~(((2ull << hi) - 1ull) & ~((1ull << lo) - 1ull));
            }
            if (__ballot(unmasked != 0ull) == 0) return; // every owned pixel is masked: no maximum to raise, no candidate (wave-uniform)
        }
    }
    // ---- 2. the rows somebody needs (wave-wide OR of the lanes' sets, both halves through one DPP tree each) ----
    unsigned long long need;
    {
        unsigned int lo = (unsigned int) unmasked, hi = (unsigned int) (unmasked >> 32);
        lo = wave_or_to_lane63(lo);
        hi = wave_or_to_lane63(hi);
        need = ((unsigned long long) (unsigned int) __builtin_amdgcn_readlane((int) hi, 63) << 32) | (unsigned int) __builtin_amdgcn_readlane((int) lo, 63);
    }
    if (need == 0ull) return;
    typedef unsigned int __attribute__((aligned(1))) u32u;
    float emax = -INFINITY;
    while (need) { // wave-uniform: one run of rows per pass
        const int a = __ffsll((long long) need) - 1;
        // close gaps of up to FE_GAP rows above every needed row, take the run of the closed set that starts at a, then its last NEEDED row
        unsigned long long closed = need;
#pragma unroll
        for (int g = 1; g <= FE_GAP; g++) closed |= need << g;
        const unsigned long long from_a = closed >> a;
        const int len                   = (~from_a == 0ull) ? 64 : __ffsll((long long) ~from_a) - 1; // rows a .. a+len-1 of the closed set
        const unsigned long long runset = (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << a;
        const unsigned long long inrun  = need & runset;
        const int b                     = 63 - __builtin_clzll(inrun);
        need &= ~runset;
        // the lane's unmasked rows of this run, aligned so that bit t belongs to the response row of streamed row t
        // (streamed row t = ROI row by0 + a - 3 + t carries response row by0 + a - 5 + t: block row a at t = 5)
        const unsigned long long urun = (unmasked & inrun) >> a;
        const int y_first = by0 + a - 3, n_rows = b - a + 7;

        fe_rows st;
        st.emax = emax;
        unsigned int pix[FE_CH];
#pragma unroll
        for (int i = 0; i < FE_CH; i++) { // image row R.ry + y (rows further than one past the image only feed rows nobody reads)
            const uint8_t *rowp = img + (size_t) icg_reflect1(min(max(R.ry + y_first + i, -1), h), h) * pitch; // wave-uniform
            pix[i] = *reinterpret_cast<const u32u *>(rowp + (unsigned int) base);
        }
        const int y_last = y_first + n_rows - 1;
        st.chunk<FE_PRIME>(pix, y_last, img, base, pitch, R.ry, h, 0u, y_first, R.rh, R.rw, sel, s, nms_col, lane, roi, xcol, cand_cnt, cbase);
        for (int t0 = FE_CH; t0 < n_rows; t0 += FE_CH) { // wave-uniform
            // bits t0-5 .. of urun: the unmasked flags of the response rows of this chunk
            const unsigned int uc = t0 >= 5 ? (unsigned int) (urun >> (t0 - 5)) : (unsigned int) urun << (5 - t0);
            const int y0 = y_first + t0; // ROI row of the chunk's first image row
            if ((y0 <= 2 && 2 < y0 + FE_CH) || (y0 <= R.rh + 1 && R.rh + 1 < y0 + FE_CH))
                st.chunk<FE_MIRRORS>(pix, y_last, img, base, pitch, R.ry, h, uc, y0, R.rh, R.rw, sel, s, nms_col, lane, roi, xcol, cand_cnt, cbase);
            else
                st.chunk<FE_PLAIN>(pix, y_last, img, base, pitch, R.ry, h, uc, y0, R.rh, R.rw, sel, s, nms_col, lane, roi, xcol, cand_cnt, cbase);
        }
        emax = st.emax;
    }
    unsigned int key = emax == -INFINITY ? 0u : f32_order_key(emax); // (no unmasked owned pixel met: nothing to report)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned int o = __shfl_xor(key, m, 64);
        key                  = o > key ? o : key;
    }
    if (lane == 0 && key) atomicMax(&roi_max[roi], key);
}

// One wave per block; a workgroup of FE_WAVES waves takes FE_WAVES consecutive blocks of the ROI-major block grid (neighbours in a block
// row: they share image rows and the job's disc list in L2).  At most 8 x FE_RESIDENT_PER_XCD workgroups are launched: workgroup b belongs
// to XCD b & 7 (the dispatcher's round-robin) and walks that XCD's contiguous eighth of the grid (icg_xcd_chunked's partition) at a fixed
// stride.  (Rounds 2-5: one workgroup per block, a wave per 60 x 16 tile — 221 k waves per launch of 192 frames, 72 % of which left after
// the mask test.  A dynamic form — every wave pulling tiles from a per-XCD counter — was measured too: 221 k atomics on eight addresses
// serialise in L2, 3.1 ms per launch against 0.27-0.45: profiles/r06_detector_and_lk_diet.txt.)
__global__ __launch_bounds__(64 * FE_WAVES) void k_min_eig_nms(const det_roi *rois, const uint8_t *frames, size_t slot_bytes,
                                                               const int32_t *slots, int pitch, int w, int h, const float2 *mask_pts,
                                                               const int32_t *mask_begin, const int32_t *mask_cnt /* per job */, int radius,
                                                               const int32_t *vspan /*radius+2*/,
                                                               unsigned int *roi_max,
                                                               unsigned long long *cand, size_t cand_plane, int32_t *cand_cnt,
                                                               int gx, int gy, int n_blocks, unsigned int m_roi, unsigned int m_gx) {
    // vh[a], a = 0..radius, and vh[radius+1] = -1 for every column further away; staged before any wave starts (the only barrier)
    __shared__ int vh[FE_MAX_RADIUS + 2];
    for (int a = threadIdx.x; a < radius + 2; a += 64 * FE_WAVES) vh[a] = vspan[a];
    __syncthreads();
    const int n_items = (n_blocks + FE_WAVES - 1) / FE_WAVES;    // groups of FE_WAVES blocks
    const int chunk   = (n_items + 7) >> 3;                      // items per XCD range
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, G = gridDim.x >> 3; // (the grid is a multiple of 8)
    const int wv  = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)); // wave-uniform by construction: keep it in an SGPR
    const int end = min((x + 1) * chunk, n_items);
    for (int it = x * chunk + j; it < end; it += G) {
        const int bl = it * FE_WAVES + wv;
        if (bl < n_blocks)
            fe_block(bl, vh, rois, frames, slot_bytes, slots, pitch, w, h, mask_pts, mask_begin, mask_cnt, radius, roi_max, cand, cand_plane, cand_cnt, gx, gy,
                     m_roi, m_gx);
    }
}

// ---------------------------------------------------------------------------------------------------------
#define DET_MAX_PER_BLOCK 64

struct subpix_mask_t {
    float m[121];
};

struct subpix_smem { // per wave
    double terms[121][5];
    float patch[13][13];
    float cur[2];
    int stop;
};

// LDS ordering inside ONE wave: the LDS queue is in order, the compiler must not reorder across the point
#define DET_WAVE_SYNC()                                         \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  \
        __builtin_amdgcn_wave_barrier();                        \
    } while (0)

// cv::cornerSubPix (App. B.8) of ONE corner on one wave: 13x13 bilinear patch and the 121 gradient terms in parallel through the wave's
// LDS block, the five 121-term sums each sequentially in raster order (IEEE order == CPU order), one lane per sum.
__device__ __forceinline__ float2 subpix_corner(subpix_smem &S, const det_roi &R, const uint8_t *img, int pitch, const float2 cT, int lane,
                                                const subpix_mask_t &M) {
    float cIx = cT.x, cIy = cT.y;
    int iter = 0;
    while (true) {
        const float ox = cIx - 6.f, oy = cIy - 6.f;
        const int iox = (int) floorf(ox), ioy = (int) floorf(oy);
        const float fa = ox - iox, fb = oy - ioy;
        const float w00 = (1.f - fa) * (1.f - fb), w01 = fa * (1.f - fb), w10 = (1.f - fa) * fb, w11 = fa * fb;
        for (int i = lane; i < 169; i += 64) {
            int r = i / 13, c = i - r * 13;
            int x0 = min(max(iox + c, 0), R.rw - 1), x1 = min(max(iox + c + 1, 0), R.rw - 1);
            int y0 = min(max(ioy + r, 0), R.rh - 1), y1 = min(max(ioy + r + 1, 0), R.rh - 1);
            const uint8_t *p0 = img + (size_t) (R.ry + y0) * pitch + R.rx;
            const uint8_t *p1 = img + (size_t) (R.ry + y1) * pitch + R.rx;
            float s00 = p0[x0], s01 = p0[x1], s10 = p1[x0], s11 = p1[x1];
            S.patch[r][c] = s00 * w00 + s01 * w01 + s10 * w10 + s11 * w11;
        }
        DET_WAVE_SYNC();
        for (int i = lane; i < 121; i += 64) {
            int r = i / 11, c = i - r * 11;
            double m   = M.m[i];
            double tgx = S.patch[r + 1][c + 2] - S.patch[r + 1][c];
            double tgy = S.patch[r + 2][c + 1] - S.patch[r][c + 1];
            double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
            double px = c - 5, py = r - 5;
            S.terms[i][0] = gxx;
            S.terms[i][1] = gxy;
            S.terms[i][2] = gyy;
            S.terms[i][3] = gxx * px + gxy * py;
            S.terms[i][4] = gxy * px + gyy * py;
        }
        DET_WAVE_SYNC();
        // the five 121-term sums keep the CPU's sequential raster order and run side by side: sum q on every lane with lane % 5 == q.
        // (Round 5: ALL lanes add, lanes 5..63 as copies of lanes 0..4 — an FP64 add issued with 16 or fewer active lanes takes ~5x as long on
        // gfx950 as one issued with more, profiles/ubench/valu_cost_r05.txt: under `if (lane < 5)` this chain was 121 x 8.9 ns per iteration;
        // the LDS reads of the copies are broadcasts of the same five addresses.)
        double acc = 0;
        {
            const int q = lane % 5;
#pragma unroll 11
            for (int i = 0; i < 121; i++) acc += S.terms[i][q];
        }
        const double a = __shfl(acc, 0, 64), b = __shfl(acc, 1, 64), c = __shfl(acc, 2, 64);
        const double bb1 = __shfl(acc, 3, 64), bb2 = __shfl(acc, 4, 64);
        // every lane evaluates the (wave-uniform) update: no LDS round trip for the decision
        int st     = 0;
        float c2x = cIx, c2y = cIy;
        double det = a * c - b * b;
        if (fabs(det) <= DBL_EPSILON * DBL_EPSILON) {
            st = 1; // break before updating
        } else {
            double scale = 1.0 / det;
            c2x          = (float) (cIx + c * scale * bb1 - b * scale * bb2);
            c2y          = (float) (cIy - b * scale * bb1 + a * scale * bb2);
            double err   = (double) ((c2x - cIx) * (c2x - cIx) + (c2y - cIy) * (c2y - cIy));
            if (c2x < 0 || c2x >= R.rw || c2y < 0 || c2y >= R.rh)
                st = 2;
            else if (!(iter + 1 < 20 && err > 0.01 * 0.01))
                st = 2;
        }
        if (st != 1) {
            cIx = c2x;
            cIy = c2y;
        }
        ++iter;
        DET_WAVE_SYNC(); // the next iteration overwrites patch / terms
        if (st) break;
    }
    if (fabsf(cIx - cT.x) > 5 || fabsf(cIy - cT.y) > 5) {
        cIx = cT.x;
        cIy = cT.y;
    }
    return make_float2(cIx, cIy);
}

// k_select: one workgroup per ROI — quality threshold 0.01 * (masked ROI maximum), then repeated block-wide arg-max over the live candidates
// + min-distance kill (equivalent to featureselect.cpp's sort + greedy grid test; needs no sort and no capacity cap).
// k_subpix: one wave per picked corner (cornerSubPix).  (Round 4 fused the two into one 16-wave workgroup per ROI: 138 us alone on the GPU
// against 41 + 85 us for the pair, and 949 us against 240 + 263 us with twelve groups' kernels in flight — a 1024-thread workgroup with
// 89 KB of LDS waits for a whole CU; reverted to the pair.)
__global__ __launch_bounds__(256) void k_select(const det_roi *rois, unsigned long long *cand, size_t cand_plane, int32_t *cand_cnt, unsigned int *roi_max,
                                                int min_dist, float2 *picks /*roi x max_pb*/, int32_t *pick_cnt, int32_t *cnt_out, int max_pb) {
    __shared__ unsigned long long wbest[4];
    __shared__ unsigned long long best;
    const det_roi R = rois[blockIdx.x];
    const int t     = threadIdx.x;
    if (R.quota <= 0) { // inactive entry of a dense (job, block) table: no corner (its accumulators were never touched)
        if (t == 0) pick_cnt[blockIdx.x] = 0, cnt_out[blockIdx.x] = 0;
        return;
    }
    unsigned long long *C = cand + (size_t) R.job * cand_plane + R.cand_base;
    const int n = cand_cnt[blockIdx.x];
    const double md2 = (double) min_dist * (double) min_dist;
    int quota = R.quota < max_pb ? R.quota : max_pb;
    int acc   = 0;
    {
        // quality level (featureselect.cpp: threshold(eig, eig, maxVal*qualityLevel, 0, THRESH_TOZERO)): k_min_eig_nms appended
        // every unmasked local maximum, the ones that are not above 0.01 * (masked ROI maximum) are dropped here
        const unsigned int mk = roi_max[blockIdx.x];
        const double maxVal   = mk ? (double) f32_from_order_key(mk) : 0.0;
        const float thresh    = (float) (maxVal * 0.01);
        for (int i = t; i < n; i += 256)
            if (!(f32_from_order_key((unsigned int) (C[i] >> 32)) > thresh)) C[i] = 0;
        __syncthreads();
    }
    while (acc < quota) {
        unsigned long long k = 0;
        for (int i = t; i < n; i += 256) {
            unsigned long long c = C[i];
            k                    = c > k ? c : k;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            unsigned long long o = __shfl_xor(k, m, 64);
            k                    = o > k ? o : k;
        }
        if ((t & 63) == 0) wbest[t >> 6] = k;
        __syncthreads();
        if (t == 0) {
            unsigned long long b = wbest[0];
            for (int i = 1; i < 4; i++) b = wbest[i] > b ? wbest[i] : b;
            best = b;
        }
        __syncthreads();
        const unsigned long long b = best;
        if (b == 0) break; // no live candidate left
        const int idx = (int) (b & 0xffffffffu);
        const int by = idx / R.rw, bx = idx - by * R.rw;
        if (t == 0) picks[(size_t) blockIdx.x * max_pb + acc] = make_float2((float) bx, (float) by);
        acc++;
        if (min_dist >= 1) {
            for (int i = t; i < n; i += 256) {
                unsigned long long c = C[i];
                if (!c) continue;
                int ci = (int) (c & 0xffffffffu);
                int cy = ci / R.rw, cx = ci - cy * R.rw;
                float dx = (float) (cx - bx), dy = (float) (cy - by);
                if (c == b || (double) (dx * dx + dy * dy) < md2) C[i] = 0;
            }
        } else {
            for (int i = t; i < n; i += 256)
                if (C[i] == b) C[i] = 0;
        }
        __syncthreads();
    }
    if (t == 0) {
        pick_cnt[blockIdx.x] = acc; // device copy: read by every k_subpix workgroup of the ROI
        cnt_out[blockIdx.x]  = acc; // the caller's copy (pinned staging memory of the call, or the tracker's arena)
        // the ROI's accumulators are consumed (every thread read them before the first barrier above): leave them zero for the next call,
        // which then needs neither a memset nor an upload of zeros
        roi_max[blockIdx.x]  = 0;
        cand_cnt[blockIdx.x] = 0;
    }
}

__global__ __launch_bounds__(64) void k_subpix(const det_roi *rois, const uint8_t *frames, size_t slot_bytes, const int32_t *slots, int pitch,
                                               const float2 *picks, float2 *corners_out, const int32_t *pick_cnt, int max_pb, subpix_mask_t M) {
    __shared__ subpix_smem SP;
    const int roi = blockIdx.x / max_pb, ci = blockIdx.x - roi * max_pb;
    if (ci >= pick_cnt[roi]) return;
    const det_roi R    = rois[roi];
    const uint8_t *img = frames + (size_t) slots[R.job] * slot_bytes;
    const float2 r     = subpix_corner(SP, R, img, pitch, picks[(size_t) roi * max_pb + ci], (int) threadIdx.x, M);
    if (threadIdx.x == 0) corners_out[(size_t) roi * max_pb + ci] = r;
}

// ---- host side ---------------------------------------------------------------------------------------------
static void circle_halfwidths(int radius, std::vector<int32_t> &hw) {
    // spans of OpenCV drawing.cpp Circle() (midpoint algorithm) for a filled circle, per |row offset|
    hw.assign((size_t) radius + 1, -1);
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        if (dx > hw[dy]) hw[dy] = dx;
        if (dy > hw[dx]) hw[dx] = dy;
        dy++;
        err += plus;
        plus += 2;
        int m = (err <= 0) - 1;
        err -= minus & m;
        dx += m;
        minus -= m & 2;
    }
}

// vh[a] = max{dy : halfw[dy] >= a} for a = 0..radius, vh[radius+1] = -1: the rows of a disc column (see the note on the mask above).
// Returns false if the span table is not non-increasing (then the interval form would not describe the disc; cannot happen for a circle)
static bool circle_row_spans(const std::vector<int32_t> &hw, std::vector<int32_t> &vh) {
    const int radius = (int) hw.size() - 1;
    vh.assign((size_t) radius + 2, -1);
    for (int a = 0; a <= radius; a++)
        for (int dy = 0; dy <= radius; dy++)
            if (hw[(size_t) dy] >= a) vh[(size_t) a] = dy;
    for (int a = 0; a <= radius; a++)
        for (int dy = 0; dy <= radius; dy++)
            if ((hw[(size_t) dy] >= a) != (dy <= vh[(size_t) a])) return false;
    return true;
}

static int ensure_roi_state(icg_ctx *ctx, int n_roi) {
    if (n_roi <= ctx->roi_state_cap) return 0;
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_roi_max) (void) hipFree(ctx->d_roi_max);
    if (ctx->d_cand_cnt) (void) hipFree(ctx->d_cand_cnt);
    ctx->d_roi_max = nullptr, ctx->d_cand_cnt = nullptr, ctx->roi_state_cap = 0;
    const int cap = std::max(1024, 2 * n_roi);
    ICG_HIP(ctx, hipMalloc((void **) &ctx->d_roi_max, sizeof(uint32_t) * (size_t) cap));
    ICG_HIP(ctx, hipMalloc((void **) &ctx->d_cand_cnt, sizeof(int32_t) * (size_t) cap));
    ICG_HIP(ctx, hipMemsetAsync(ctx->d_roi_max, 0, sizeof(uint32_t) * (size_t) cap, ctx->stream));
    ICG_HIP(ctx, hipMemsetAsync(ctx->d_cand_cnt, 0, sizeof(int32_t) * (size_t) cap, ctx->stream));
    ctx->roi_state_cap = cap;
    return 0;
}

static subpix_mask_t subpix_window() {
    subpix_mask_t M;
    for (int i = 0; i < 11; i++) {
        float y  = (float) (i - 5) / 5;
        float vy = std::exp(-y * y);
        for (int j = 0; j < 11; j++) {
            float x         = (float) (j - 5) / 5;
            M.m[i * 11 + j] = (float) (vy * std::exp(-x * x));
        }
    }
    return M;
}

static int ensure_detect_ws(icg_ctx *ctx) {
    if (ctx->d_cand) return 0;
    const size_t w = ctx->cfg.width, h = ctx->cfg.height, nb = ctx->cfg.max_batch;
    ICG_HIP(ctx, hipMalloc((void **) &ctx->d_cand, sizeof(unsigned long long) * w * h * nb));
    return 0;
}

extern "C" int icg_detect(icg_ctx *ctx, int n, const int32_t *slots, const icg_detect_grid *grid, const int32_t *mask_off,
                          const float *mask_pts, const int32_t *quota, int max_per_job, float *out_pts,
                          int32_t *out_count, int32_t *out_block) {
    if (!ctx || n < 0) return ICG_ERR_INVALID;
    if (n == 0) return ICG_OK;
    if (!slots || !grid || !mask_off || !quota || !out_pts || !out_count || max_per_job <= 0) return ICG_ERR_INVALID;
    if (n > ctx->cfg.max_batch) return icg_fail(ctx, ICG_ERR_CAPACITY, "detect batch %d > max_batch %d", n, ctx->cfg.max_batch);
    const int w = ctx->cfg.width, h = ctx->cfg.height, pitch = ctx->lv[0].pitch;
    const int nblk = grid->block_cols * grid->block_rows;
    if (nblk <= 0 || grid->block_w <= 6 || grid->block_h <= 6 || grid->block_cols * grid->block_w > w ||
        grid->block_rows * grid->block_h > h || grid->min_dist < 0 || grid->max_per_block <= 0 ||
        grid->max_per_block > DET_MAX_PER_BLOCK || grid->min_dist > FE_MAX_RADIUS)
        return icg_fail(ctx, ICG_ERR_INVALID, "bad detection grid");
    for (int k = 0; k < n; k++)
        if (slots[k] < 0 || slots[k] >= ctx->cfg.n_slots) return icg_fail(ctx, ICG_ERR_INVALID, "bad slot");
    const int n_mask = mask_off[n];
    if (n_mask < 0 || (n_mask > 0 && !mask_pts)) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = ensure_detect_ws(ctx);
    if (rc) return rc;

    // ROI list (tracking.cc:629-645)
    std::vector<det_roi> rois;
    for (int b = 0; b < n; b++)
        for (int k = 0; k < nblk; k++) {
            int q = quota[(size_t) b * nblk + k];
            if (q <= 0) continue;
            if (q > grid->max_per_block) q = grid->max_per_block;
            int cols = k % grid->block_cols, rows = k / grid->block_cols;
            det_roi R;
            R.job   = b;
            R.block = k;
            R.rx    = cols * grid->block_w;
            R.ry    = rows * grid->block_h;
            R.rw    = grid->block_w;
            R.rh    = grid->block_h;
            if (k != nblk - 1) {
                R.rw -= 5;
                R.rh -= 5;
            }
            R.quota     = q;
            R.cand_base = k * grid->block_w * grid->block_h;
            rois.push_back(R);
        }
    const int n_roi = (int) rois.size();
    for (int b = 0; b < n; b++) out_count[b] = 0;
    if (n_roi == 0) return ICG_OK;
    const int max_pb = grid->max_per_block;

    std::vector<int32_t> hw, vh;
    circle_halfwidths(grid->min_dist, hw);
    if (!circle_row_spans(hw, vh)) return icg_fail(ctx, ICG_ERR_INVALID, "circle span table of radius %d is not monotone", grid->min_dist);
    for (int b = 0; b < n; b++)
        if (mask_off[b] < 0 || mask_off[b] > mask_off[b + 1]) return icg_fail(ctx, ICG_ERR_INVALID, "mask_off is not non-decreasing");

    icg_call c(ctx);
    size_t need = sizeof(det_roi) * n_roi + sizeof(int32_t) * (2 * (size_t) n + 1 + vh.size()) + sizeof(float) * 2 * n_mask +
                  2 * (sizeof(float2) * max_pb + 16) * (size_t) n_roi + 8192;
    if ((rc = c.reserve(need))) return rc;
    // Every small array of the call is read or written by the kernels in the call's pinned staging memory (zero-copy over PCIe): one
    // read per workgroup / one write per corner.  The mirrored form cost an H2D and a D2H copy launch per call, and under load every
    // launch on a busy hardware queue costs ~75-100 us whatever its size (profiles/r02_queue_view.json).
    const det_roi *d_rois  = c.in_zc(rois.data(), (size_t) n_roi);
    const int32_t *d_slots = c.in_zc(slots, (size_t) n);
    // the disc centres are read by every wave of the job's ROIs (48 per ROI): device copy, not zero-copy
    const int32_t *d_vh    = c.in(vh.data(), vh.size());
    const float2 *d_mpts   = (const float2 *) c.in(mask_pts, 2 * (size_t) n_mask);
    std::vector<int32_t> mcnt((size_t) n);
    for (int b = 0; b < n; b++) mcnt[(size_t) b] = mask_off[b + 1] - mask_off[b];
    const int32_t *d_moff  = c.in(mask_off, (size_t) n);
    const int32_t *d_mcnt  = c.in(mcnt.data(), (size_t) n);
    // [roi_max | cand_cnt] per ROI live in the context and are zero between calls (k_select clears the entries it consumed)
    if ((rc = ensure_roi_state(ctx, n_roi))) return rc;
    unsigned int *d_rmax = ctx->d_roi_max;
    int32_t *d_ccnt      = ctx->d_cand_cnt;
    if ((rc = c.seal())) return rc;
    std::vector<float> h_corners((size_t) n_roi * max_pb * 2);
    std::vector<int32_t> h_cnt((size_t) n_roi);
    // the picks and their counts are re-read by k_subpix: device scratch; the refined corners and the counts the caller needs are written
    // by the kernels into the staging memory directly
    float2 *d_picks = (float2 *) c.out((float *) nullptr, (size_t) n_roi * max_pb * 2);
    int32_t *d_pcnt = c.out((int32_t *) nullptr, (size_t) n_roi);
    float2 *z_corners  = (float2 *) c.out_zc(h_corners.data(), (size_t) n_roi * max_pb * 2);
    int32_t *z_cnt     = c.out_zc(h_cnt.data(), (size_t) n_roi);

    const size_t cand_plane = (size_t) w * h;
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "detect_min_eig_nms");
        const int gx = (grid->block_w + FE_TW - 1) / FE_TW, gy = (grid->block_h + FE_TH * FE_WAVES - 1) / (FE_TH * FE_WAVES);
        hipLaunchKernelGGL(k_min_eig_nms, dim3(std::min(icg_xcd_grid((gx * gy * n_roi + FE_WAVES - 1) / FE_WAVES), 8 * FE_RESIDENT_PER_XCD)), dim3(64 * FE_WAVES), 0, ctx->stream, d_rois, ctx->d_frames,
                           ctx->slot_bytes, d_slots, pitch, w, h, d_mpts, d_moff, d_mcnt, grid->min_dist, d_vh, d_rmax, ctx->d_cand, cand_plane, d_ccnt,
                           gx, gy, gx * gy * n_roi, icg_div_magic(gx * gy), icg_div_magic(gx));
    }
    {
        icg_prof_scope ps(ctx, "detect_select");
        hipLaunchKernelGGL(k_select, dim3(n_roi), dim3(256), 0, ctx->stream, d_rois, ctx->d_cand, cand_plane, d_ccnt, d_rmax, grid->min_dist, d_picks, d_pcnt,
                           z_cnt, max_pb);
    }
    {
        const subpix_mask_t M = subpix_window();
        icg_prof_scope ps(ctx, "detect_subpix");
        hipLaunchKernelGGL(k_subpix, dim3(n_roi * max_pb), dim3(64), 0, ctx->stream, d_rois, ctx->d_frames, ctx->slot_bytes, d_slots, pitch,
                           (const float2 *) d_picks, z_corners, (const int32_t *) d_pcnt, max_pb, M);
    }
    ICG_HIP(ctx, hipGetLastError());
    if ((rc = c.finish())) return rc;

    // block-order assembly with the block origin added (tracking.cc:669-685)
    for (int r = 0; r < n_roi; r++) {
        const det_roi &R = rois[r];
        for (int i = 0; i < h_cnt[r]; i++) {
            int &cnt = out_count[R.job];
            if (cnt >= max_per_job) break;
            float x = (float) R.rx + h_corners[((size_t) r * max_pb + i) * 2];
            float y = (float) R.ry + h_corners[((size_t) r * max_pb + i) * 2 + 1];
            out_pts[((size_t) R.job * max_per_job + cnt) * 2]     = x;
            out_pts[((size_t) R.job * max_per_job + cnt) * 2 + 1] = y;
            if (out_block) out_block[(size_t) R.job * max_per_job + cnt] = R.block;
            cnt++;
        }
    }
    return ICG_OK;
}

// ---- device-resident tracker (tracker.hip) ----------------------------------------------------------------------------------------------
// The work of a detection call as the stage kernels leave it in device memory: a DENSE table of n_jobs x n_blocks ROI descriptors
// (quota <= 0: inactive), the frame slot of every job, the disc centres of every job as (begin, count) into one point array; the corners
// and their counts stay in device memory (the next stage kernel assembles them in block order).  Asynchronous on the context's stream.
int icg_detect_circle_rows(int radius, std::vector<int32_t> &vh) {
    std::vector<int32_t> hw;
    circle_halfwidths(radius, hw);
    return circle_row_spans(hw, vh) ? 0 : -1;
}

int icg_detect_launch_ind(icg_ctx *ctx, int n_jobs, const icg_detect_grid *grid, const void *d_rois, const int32_t *d_slots, const float2 *d_mask_pts,
                          const int32_t *d_mask_begin, const int32_t *d_mask_cnt, const int32_t *d_vh, float2 *d_picks, int32_t *d_pick_cnt,
                          float2 *d_corners, int32_t *d_corner_cnt) {
    const int w = ctx->cfg.width, h = ctx->cfg.height, pitch = ctx->lv[0].pitch;
    const int nblk = grid->block_cols * grid->block_rows, n_roi = n_jobs * nblk;
    if (n_jobs > ctx->cfg.max_batch) return icg_fail(ctx, ICG_ERR_CAPACITY, "detect batch %d > max_batch %d", n_jobs, ctx->cfg.max_batch);
    if (grid->max_per_block > DET_MAX_PER_BLOCK || grid->min_dist > FE_MAX_RADIUS) return icg_fail(ctx, ICG_ERR_INVALID, "bad detection grid");
    int rc = ensure_detect_ws(ctx);
    if (rc) return rc;
    if ((rc = ensure_roi_state(ctx, n_roi))) return rc;
    const size_t cand_plane = (size_t) w * h;
    {
        icg_prof_scope ps(ctx, "detect_min_eig_nms");
        const int gx = (grid->block_w + FE_TW - 1) / FE_TW, gy = (grid->block_h + FE_TH * FE_WAVES - 1) / (FE_TH * FE_WAVES);
        hipLaunchKernelGGL(k_min_eig_nms, dim3(std::min(icg_xcd_grid((gx * gy * n_roi + FE_WAVES - 1) / FE_WAVES), 8 * FE_RESIDENT_PER_XCD)), dim3(64 * FE_WAVES), 0, ctx->stream, (const det_roi *) d_rois, ctx->d_frames,
                           ctx->slot_bytes, d_slots, pitch, w, h, d_mask_pts, d_mask_begin, d_mask_cnt, grid->min_dist, d_vh, ctx->d_roi_max, ctx->d_cand,
                           cand_plane, ctx->d_cand_cnt, gx, gy, gx * gy * n_roi, icg_div_magic(gx * gy), icg_div_magic(gx));
    }
    {
        icg_prof_scope ps(ctx, "detect_select");
        hipLaunchKernelGGL(k_select, dim3(n_roi), dim3(256), 0, ctx->stream, (const det_roi *) d_rois, ctx->d_cand, cand_plane, ctx->d_cand_cnt, ctx->d_roi_max,
                           grid->min_dist, d_picks, d_pick_cnt, d_corner_cnt, grid->max_per_block);
    }
    {
        const subpix_mask_t M = subpix_window();
        icg_prof_scope ps(ctx, "detect_subpix");
        hipLaunchKernelGGL(k_subpix, dim3(n_roi * grid->max_per_block), dim3(64), 0, ctx->stream, (const det_roi *) d_rois, ctx->d_frames, ctx->slot_bytes,
                           d_slots, pitch, (const float2 *) d_picks, d_corners, (const int32_t *) d_pick_cnt, grid->max_per_block, M);
    }
    ICG_HIP(ctx, hipGetLastError());
    return ICG_OK;
}
