// F1: Tracking::preprocessing on device — BGR->gray, CLAHE(3.0, 21x21) and the LK image pyramid.
//
// Reference call sites: tracking/tracking.cc:107-142 (preprocessing), :63 (createCLAHE(3.0, Size(21,21))),
// :385-393 (calcOpticalFlowPyrLK rebuilds the pyramid of both images on every call; here it is built once per
// frame and kept resident in the frame slot for the two frames it is used in).
// Arithmetic definitions: SURVEY.md Appendix B.1 (gray, exact), B.2 (CLAHE: integer histogram/clip/redistribute,
// float LUT scale and float bilinear LUT blend in OpenCV's association order), B.3 (pyrDown, exact integer).
//
// Kernels are HBM-bound u8 stencils/histograms (no MFMA shape here):
//   k_clahe_lut   one workgroup per (tile, frame): LDS histogram with reflect-101 padded reads, clip,
//                 redistribute, 256-bin block scan -> LUT (441 x 256 B per frame, stays in L2)
//   k_clahe_apply one workgroup per (interpolation strip, 256-px chunk, frame): the <=2x8 LUTs the chunk needs are
//                 staged in LDS, rows are read/written as coalesced uchar4 per lane
//   k_pyrdown_rows one pyrDown step as a row stream: no LDS, 12-byte source windows per lane, SWAR vertical sums (see below)
// Algorithmic bytes per frame: CLAHE 3*W*H, pyramid 1.640625*W*H.
#include <algorithm>

#include "icg_internal.h"

// per-launch job list passed BY VALUE in the kernel arguments (no staging copy, no host sync needed)
#define PRE_MAX_JOBS 64
struct pre_jobs {
    const uint8_t *src[PRE_MAX_JOBS]; // gray source image per job (device memory); nullptr: the job is idle this launch
    int32_t slot[PRE_MAX_JOBS];       // destination frame slot per job
    int stride;                       // source row stride in bytes
    // device-resident tracker (tracker.hip): the slot of job b is slot_ind[b], written by the stage kernel that ran before this launch
    // (the stream's own slot pool lives on the device); < 0 = no frame for that stream in this step
    const int32_t *slot_ind;
};
__device__ __forceinline__ int pre_job_slot(const pre_jobs &j, int b) { return j.slot_ind ? j.slot_ind[b] : j.slot[b]; }

// ---------------------------------------------------------------------------------------------------------
__global__ void k_bgr2gray(const uint8_t *bgr, int w, int h, int sstride, size_t sbatch, uint8_t *gray, int gstride,
                           size_t gbatch) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    int b = blockIdx.z;
    if (x >= w) return;
    const uint8_t *p = bgr + (size_t) b * sbatch + (size_t) y * sstride + 3 * x;
    gray[(size_t) b * gbatch + (size_t) y * gstride + x] =
        (uint8_t) ((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14);
}

__global__ void k_hist256(pre_jobs jobs, int w, int h, unsigned int *hist /*n x 256*/) {
    __shared__ unsigned int sh[256];
    int b = blockIdx.y;
    sh[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t *src = jobs.src[b];
    const int stride   = jobs.stride;
    int total          = src ? w * h : 0; // (idle job: the barriers below are still reached by every thread)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int y = i / w, x = i - y * w;
        atomicAdd(&sh[src[(size_t) y * stride + x]], 1u);
    }
    __syncthreads();
    if (sh[threadIdx.x]) atomicAdd(&hist[(size_t) b * 256 + threadIdx.x], sh[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------------
struct clahe_geom {
    int w, h, tw, th, clip;
    float lut_scale, inv_tw, inv_th;
};

// one WAVE per (tile, frame): rows are read as aligned dwords (reflect-101 only on the padded right/bottom border),
// the 256-bin histogram lives in LDS, and clip / redistribute / prefix-sum run inside the wave (4 bins per lane)
// (round 5: DPP adds, 2 issue units each, instead of __shfl_xor / __shfl_up — ds_bpermute costs 10, profiles/ubench/valu_cost_r05.txt)
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false); // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false); // row_mirror: every lane holds its row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); // row_bcast15: rows 1, 3 += rows 0, 2
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); // row_bcast31: rows 2, 3 += lane 31
    return __builtin_amdgcn_readlane(v, 63);
}
// inclusive prefix sum across the wave: Hillis-Steele inside the 16-lane rows (row_shr 1, 2, 4, 8 with zero fill), then the row totals
__device__ __forceinline__ int wave_scan_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); // rows 1, 3 += lane 15 of rows 0, 2
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); // rows 2, 3 += lane 31
    return v;
}

__global__ __launch_bounds__(64) void k_clahe_lut(pre_jobs jobs, clahe_geom g, uint8_t *lut /* n x tiles^2 x 256 */, int n_items) {
    __shared__ __attribute__((aligned(16))) unsigned int hist[256];
    const int lane = threadIdx.x;
    // (frame, tile) items frame-major, XCD-chunked: a tile row is 61-byte runs of the same image rows, so neighbouring tiles share
    // every 128-byte line — with round-robin placement each line was fetched into two or three XCDs' L2s (3.2x the image bytes
    // from HBM per frame, rocprofv3 FETCH_SIZE); now a frame's tiles sit on one XCD
    const int item = icg_xcd_chunked(blockIdx.x, n_items);
    if (item >= n_items) return;
    const int b    = item / (ICG_CLAHE_TILES * ICG_CLAHE_TILES);
    const int tile = item - b * (ICG_CLAHE_TILES * ICG_CLAHE_TILES);
    const int ty = tile / ICG_CLAHE_TILES, tx = tile - ty * ICG_CLAHE_TILES;
    const uint8_t *src = jobs.src[b];
    if (!src) return; // idle job (workgroup-uniform)
    const int stride   = jobs.stride;
    reinterpret_cast<uint4 *>(hist)[lane] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    const int x_begin = tx * g.tw, x_end = x_begin + g.tw, y_begin = ty * g.th;
    const int xa  = x_begin & ~3;
    const int ndw = (((x_end + 3) & ~3) - xa) >> 2;
    const bool pad_small = g.th * ICG_CLAHE_TILES < 2 * g.h - 1; // bottom padding reflects at most once (always, for real images)
    auto load_row_dword = [&](int r, int x) -> unsigned int {
        const uint8_t *row = src + (size_t) (pad_small ? icg_reflect1(y_begin + r, g.h) : icg_reflect101(y_begin + r, g.h)) * stride;
        if (x + 3 < g.w) return *reinterpret_cast<const unsigned int *>(row + x);
        unsigned int v = 0; // right padding of the last tile column
#pragma unroll
        for (int j = 0; j < 4; j++) v |= (unsigned int) row[icg_reflect101(x + j, g.w)] << (8 * j);
        return v;
    };
    // pass A: the dwords that lie entirely inside the tile (all but the first and last of each row): no per-byte tests.
    // Round 5 (the front-end is bound by VALU issue): a FIXED lane -> (dword column, row phase) mapping — lane = 16 q + d
    // takes dword d + 1 of rows q, q + 4, q + 8, ... — so an item costs one address add, one load and the four bin addresses
    // ((v >> (8 j - 2)) & 0x3fc: a shift and a mask each) instead of a division by the row length, a row reflection, a 64-bit
    // multiply-add and a bounds test per dword (rounds 1-4: ~46 issue units per dword, now ~12).  Tiles wider than 16 interior dwords or
    // touching the padded right / bottom border (the last tile row and column only) keep the general loop.
    const int nin = ndw - 2;
    const bool fast = nin >= 1 && nin <= 16 && y_begin + g.th <= g.h && xa + 4 * ndw <= g.w; // wave-uniform
    if (fast) {
        const int d = lane & 15, q = lane >> 4;
        const bool on = d < nin;
        const uint8_t *p = src + (size_t) (y_begin + q) * stride + (xa + 4 * (on ? d + 1 : 1)); // (idle lanes re-read a valid dword)
        unsigned char *hb = reinterpret_cast<unsigned char *>(hist);
        const size_t step = 4 * (size_t) stride;
        for (int r = q; r < g.th; r += 16, p += 4 * step) { // four rows per trip: their loads are in flight together
            unsigned int v[4];
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                ok[k] = on && r + 4 * k < g.th;
                v[k]  = r + 4 * k < g.th ? *reinterpret_cast<const unsigned int *>(p + k * step) : 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (ok[k]) {
                    atomicAdd(reinterpret_cast<unsigned int *>(hb + ((v[k] << 2) & 0x3fcu)), 1u);
                    atomicAdd(reinterpret_cast<unsigned int *>(hb + ((v[k] >> 6) & 0x3fcu)), 1u);
                    atomicAdd(reinterpret_cast<unsigned int *>(hb + ((v[k] >> 14) & 0x3fcu)), 1u);
                    atomicAdd(reinterpret_cast<unsigned int *>(hb + ((v[k] >> 22) & 0x3fcu)), 1u);
                }
            }
        }
    } else {
        {
            const int items = g.th * (nin > 0 ? nin : 0);
            const unsigned int magic = nin > 0 ? ((1u << 20) + (unsigned int) nin - 1u) / (unsigned int) nin : 0u; // i/nin == (i*magic)>>20, i < 43690
            constexpr int CH = 9; // dwords in flight per lane
            for (int base = 0; base < items; base += 64 * CH) {
                unsigned int v[CH];
                bool ok[CH];
    #pragma unroll
                for (int k = 0; k < CH; k++) {
                    const int i = base + k * 64 + lane;
                    ok[k]       = i < items;
                    v[k]        = 0;
                    if (ok[k]) {
                        const int r = (int) (((unsigned int) i * magic) >> 20), d = 1 + i - r * nin;
                        v[k]        = load_row_dword(r, xa + 4 * d);
                    }
                }
    #pragma unroll
                for (int k = 0; k < CH; k++) {
                    if (ok[k]) {
    #pragma unroll
                        for (int j = 0; j < 4; j++) atomicAdd(&hist[(v[k] >> (8 * j)) & 0xff], 1u);
                    }
                }
            }
        }
    }
    // pass B: first and last dword of each row, byte-wise against the tile's column range
    {
        const int nedge = ndw >= 2 ? 2 : 1, items = g.th * nedge;
        for (int i = lane; i < items; i += 64) {
            const int r = nedge == 2 ? (i >> 1) : i, d = (nedge == 2 && (i & 1)) ? ndw - 1 : 0;
            const int x = xa + 4 * d;
            const unsigned int v = load_row_dword(r, x);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int xx = x + j;
                if (xx >= x_begin && xx < x_end) atomicAdd(&hist[(v >> (8 * j)) & 0xff], 1u);
            }
        }
    }
    __syncthreads();

    const uint4 h4 = reinterpret_cast<const uint4 *>(hist)[lane];
    int hv[4]      = {(int) h4.x, (int) h4.y, (int) h4.z, (int) h4.w};
    int excess     = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (hv[j] > g.clip) {
            excess += hv[j] - g.clip;
            hv[j] = g.clip;
        }
    }
    const int clipped   = wave_sum_i32(excess);
    const int batch_add = clipped / 256;
    const int residual  = clipped - batch_add * 256;
    int step            = residual ? 256 / residual : 1;
    if (step < 1) step = 1;
    int run = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int t = 4 * lane + j;
        hv[j] += batch_add;
        if (residual != 0 && (t % step) == 0 && (t / step) < residual) hv[j]++;
        run += hv[j];
        hv[j] = run; // inclusive prefix inside the lane
    }
    // exclusive scan of the lane totals across the wave
    const int incl = wave_scan_i32(run);
    const int excl = incl - run;
    unsigned int packed = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float f = rintf((float) (excl + hv[j]) * g.lut_scale);
        int iv        = (int) f;
        iv            = iv < 0 ? 0 : (iv > 255 ? 255 : iv);
        packed |= (unsigned int) iv << (8 * j);
    }
    reinterpret_cast<unsigned int *>(lut + ((size_t) b * ICG_CLAHE_TILES * ICG_CLAHE_TILES + tile) * 256)[lane] = packed;
}

#define CLAHE_CHUNK 256 // pixels per workgroup row segment (64 lanes x uchar4)
#define CLAHE_MAXCOLS 24
#define CLAHE_FCOLS 8   // column pairs the float form of the staged LUT holds (32 KB of LDS; 1280-wide images touch <= 6 per chunk)

// LDS holds, per interpolation column pair p (= tx1+1) and grey level v, the FOUR LUT values the bilinear blend needs:
// {L[ty1][c1][v], L[ty1][c2][v], L[ty2][c1][v], L[ty2][c2][v]}.
//   FLT = true (round 5, every chunk that touches <= CLAHE_FCOLS pairs: all real image sizes): as four FLOATS, one ds_read_b128 per pixel.
//     The front-end is bound by VALU issue and v_cvt_f32_ubyte is a half-rate instruction: converting the LUT once per
//     staged entry instead of four times per pixel and row removes 16 of the ~50 VALU instructions a lane spends per 4 pixels; the
//     result bytes are rounded by the magic constant (res in [0, 255.0001]: the low byte of bits(res + 1.5 * 2^23) is rint(res), ties to even,
//     no clamp needed) and packed by v_perm: 27 -> ~17 issue units per pixel, same bytes.
//   FLT = false: as one dword of four bytes, v_cvt_f32_ubyte0..3 per pixel (rounds 1-4; kept for chunks of very narrow tiles).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool FLT>
__device__ __forceinline__ void clahe_apply_body(const pre_jobs &jobs, const clahe_geom &g, const uint8_t *lut, uint8_t *frames, size_t slot_bytes,
                                                 int dpitch, unsigned int *slut, int strip, int chunk, int b, int t, int dslot, int p_lo, int npairs,
                                                 int x_begin) {
    const int T = ICG_CLAHE_TILES;
    int ty1 = strip - 1, ty2 = strip;
    if (ty1 < 0) ty1 = 0;
    if (ty2 > T - 1) ty2 = T - 1;
    const unsigned int *blut = reinterpret_cast<const unsigned int *>(lut + (size_t) b * T * T * 256);
    for (int i = t; i < npairs * 64; i += 256) {
        const int p = p_lo + (i >> 6), v4 = i & 63;
        int c1 = p - 1, c2 = p;
        if (c1 < 0) c1 = 0;
        if (c2 > T - 1) c2 = T - 1;
        const unsigned int A = blut[(ty1 * T + c1) * 64 + v4], B = blut[(ty1 * T + c2) * 64 + v4];
        const unsigned int C = blut[(ty2 * T + c1) * 64 + v4], D = blut[(ty2 * T + c2) * 64 + v4];
        if (FLT) { // entries 4 i .. 4 i + 3 (grey levels 4 v4 .. 4 v4 + 3 of pair i >> 6), four floats each
            float4 *dst = reinterpret_cast<float4 *>(slut) + 4 * i;
#define CL_F(w, k) ((float) (((w) >> (8 * (k))) & 0xffu))
            dst[0] = make_float4(CL_F(A, 0), CL_F(B, 0), CL_F(C, 0), CL_F(D, 0));
            dst[1] = make_float4(CL_F(A, 1), CL_F(B, 1), CL_F(C, 1), CL_F(D, 1));
            dst[2] = make_float4(CL_F(A, 2), CL_F(B, 2), CL_F(C, 2), CL_F(D, 2));
            dst[3] = make_float4(CL_F(A, 3), CL_F(B, 3), CL_F(C, 3), CL_F(D, 3));
#undef CL_F
        } else {
            // transpose 4 LUT dwords (4 consecutive grey levels each) into 4 per-level entries
            const unsigned int ab01 = __builtin_amdgcn_perm(B, A, 0x05010400u), ab23 = __builtin_amdgcn_perm(B, A, 0x07030602u);
            const unsigned int cd01 = __builtin_amdgcn_perm(D, C, 0x05010400u), cd23 = __builtin_amdgcn_perm(D, C, 0x07030602u);
            uint4 e;
            e.x = __builtin_amdgcn_perm(cd01, ab01, 0x05040100u);
            e.y = __builtin_amdgcn_perm(cd01, ab01, 0x07060302u);
            e.z = __builtin_amdgcn_perm(cd23, ab23, 0x05040100u);
            e.w = __builtin_amdgcn_perm(cd23, ab23, 0x07060302u);
            reinterpret_cast<uint4 *>(slut)[i] = e;
        }
    }
    __syncthreads();

    const int wave = t >> 6, lane = t & 63;
    const int x0 = x_begin + lane * 4;
    int pofs[4];
    float xa[4], xa1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x     = x0 + k;
        const float txf = x * g.inv_tw - 0.5f;
        const int tx1   = (int) floorf(txf);
        xa[k]           = txf - tx1;
        xa1[k]          = 1.0f - xa[k];
        int p           = tx1 + 1 - p_lo;
        p               = p < 0 ? 0 : (p > npairs - 1 ? npairs - 1 : p); // only lanes beyond the image clamp
        pofs[k]         = p << 8;
    }
    if (x0 >= g.w) return;

    // candidate rows of this strip: floor(y*inv_th - 0.5) == strip-1
    int y_lo = (strip - 1) * g.th + g.th / 2 - 2;
    int y_hi = strip * g.th + g.th / 2 + 3;
    if (y_lo < 0) y_lo = 0;
    if (y_hi > g.h) y_hi = g.h;
    const uint8_t *src = jobs.src[b];
    const int stride   = jobs.stride;
    uint8_t *dst       = frames + (size_t) dslot * slot_bytes;
    if (FLT) {
        const float4 *sf = reinterpret_cast<const float4 *>(slut);
        for (int y = y_lo + wave; y < y_hi; y += 4) {
            const float tyf = y * g.inv_th - 0.5f;
            const int tyr   = (int) floorf(tyf);
            if (tyr != strip - 1) continue;
            const float ya = tyf - tyr, ya1 = 1.0f - ya;
            const unsigned int pin = *reinterpret_cast<const unsigned int *>(src + (size_t) y * stride + x0);
            const float4 e0 = sf[pofs[0] + (pin & 0xff)], e1 = sf[pofs[1] + ((pin >> 8) & 0xff)];
            const float4 e2 = sf[pofs[2] + ((pin >> 16) & 0xff)], e3 = sf[pofs[3] + (pin >> 24)];
            // (l11*xa1 + l12*xa)*ya1 + (l21*xa1 + l22*xa)*ya — OpenCV's association order, no contraction
            const float r0 = (e0.x * xa1[0] + e0.y * xa[0]) * ya1 + (e0.z * xa1[0] + e0.w * xa[0]) * ya;
            const float r1 = (e1.x * xa1[1] + e1.y * xa[1]) * ya1 + (e1.z * xa1[1] + e1.w * xa[1]) * ya;
            const float r2 = (e2.x * xa1[2] + e2.y * xa[2]) * ya1 + (e2.z * xa1[2] + e2.w * xa[2]) * ya;
            const float r3 = (e3.x * xa1[3] + e3.y * xa[3]) * ya1 + (e3.z * xa1[3] + e3.w * xa[3]) * ya;
            const unsigned int m0 = __float_as_uint(r0 + 12582912.f), m1 = __float_as_uint(r1 + 12582912.f);
            const unsigned int m2 = __float_as_uint(r2 + 12582912.f), m3 = __float_as_uint(r3 + 12582912.f);
            const unsigned int lo = __builtin_amdgcn_perm(m1, m0, 0x0c0c0400u), hi = __builtin_amdgcn_perm(m3, m2, 0x0c0c0400u);
            *reinterpret_cast<unsigned int *>(dst + (size_t) y * dpitch + x0) = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        }
        return;
    }
    const f32x2 XA01 = {xa[0], xa[1]}, XA23 = {xa[2], xa[3]}, XB01 = {xa1[0], xa1[1]}, XB23 = {xa1[2], xa1[3]};
    for (int y = y_lo + wave; y < y_hi; y += 4) {
        const float tyf = y * g.inv_th - 0.5f;
        const int tyr   = (int) floorf(tyf);
        if (tyr != strip - 1) continue;
        const float ya = tyf - tyr, ya1 = 1.0f - ya;
        const unsigned int pin = *reinterpret_cast<const unsigned int *>(src + (size_t) y * stride + x0);
        const unsigned int e0 = slut[pofs[0] + (pin & 0xff)], e1 = slut[pofs[1] + ((pin >> 8) & 0xff)];
        const unsigned int e2 = slut[pofs[2] + ((pin >> 16) & 0xff)], e3 = slut[pofs[3] + (pin >> 24)];
#define CL_B(e, k) ((float) (((e) >> (8 * (k))) & 0xffu))
        const f32x2 A01 = {CL_B(e0, 0), CL_B(e1, 0)}, B01 = {CL_B(e0, 1), CL_B(e1, 1)};
        const f32x2 C01 = {CL_B(e0, 2), CL_B(e1, 2)}, D01 = {CL_B(e0, 3), CL_B(e1, 3)};
        const f32x2 A23 = {CL_B(e2, 0), CL_B(e3, 0)}, B23 = {CL_B(e2, 1), CL_B(e3, 1)};
        const f32x2 C23 = {CL_B(e2, 2), CL_B(e3, 2)}, D23 = {CL_B(e2, 3), CL_B(e3, 3)};
#undef CL_B
        // (l11*xa1 + l12*xa)*ya1 + (l21*xa1 + l22*xa)*ya — OpenCV's association order, no contraction
        const f32x2 r01 = (A01 * XB01 + B01 * XA01) * ya1 + (C01 * XB01 + D01 * XA01) * ya;
        const f32x2 r23 = (A23 * XB23 + B23 * XA23) * ya1 + (C23 * XB23 + D23 * XA23) * ya;
        const float res[4] = {r01.x, r01.y, r23.x, r23.y};
        unsigned int pout = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int iv = (int) rintf(res[k]);
            pout |= (unsigned int) (iv < 0 ? 0 : (iv > 255 ? 255 : iv)) << (8 * k);
        }
        *reinterpret_cast<unsigned int *>(dst + (size_t) y * dpitch + x0) = pout;
    }
}

__global__ __launch_bounds__(256) void k_clahe_apply(pre_jobs jobs, clahe_geom g, const uint8_t *lut, uint8_t *frames,
                                                     size_t slot_bytes, int dpitch) {
    // 32 KB: CLAHE_FCOLS pairs x 256 grey levels x 4 floats, or CLAHE_MAXCOLS pairs x 256 dwords (24 KB) in the byte form
    __shared__ __attribute__((aligned(16))) unsigned int slut[CLAHE_FCOLS * 256 * 4];
    static_assert(CLAHE_FCOLS * 4 >= CLAHE_MAXCOLS, "the byte form must fit the same array");
    const int strip = blockIdx.x; // ty1_raw = strip-1
    const int chunk = blockIdx.y;
    const int b     = blockIdx.z;
    const int t     = threadIdx.x;
    const int dslot = pre_job_slot(jobs, b);
    if (!jobs.src[b] || dslot < 0) return; // idle job (workgroup-uniform, before the first barrier)
    const int x_begin = chunk * CLAHE_CHUNK;
    int x_end         = x_begin + CLAHE_CHUNK;
    if (x_end > g.w) x_end = g.w;
    // pair range touched by this chunk: p = floor(x*inv_tw - 0.5) + 1
    const int p_lo   = (int) floorf(x_begin * g.inv_tw - 0.5f) + 1;
    const int p_hi   = (int) floorf((x_end - 1) * g.inv_tw - 0.5f) + 1;
    const int npairs = p_hi - p_lo + 1; // <= CLAHE_MAXCOLS by construction of the launch (checked on host)
    if (npairs <= CLAHE_FCOLS) // workgroup-uniform
        clahe_apply_body<true>(jobs, g, lut, frames, slot_bytes, dpitch, slut, strip, chunk, b, t, dslot, p_lo, npairs, x_begin);
    else
        clahe_apply_body<false>(jobs, g, lut, frames, slot_bytes, dpitch, slut, strip, chunk, b, t, dslot, p_lo, npairs, x_begin);
}

// ---------------------------------------------------------------------------------------------------------
// One pyrDown step as a ROW STREAM (round 5; the LDS tile kernels of rounds 1-4, k_pyramid3 / k_pyrdown, are gone since round 6).  The front-end as a whole is bound by VALU issue
// (~88 % of the issue slots of the co-resident kernel mix are taken), so what a streaming kernel costs the frame rate is
// its instruction count, not its HBM fraction — and the tile kernel spent 24 VALU + 16 SALU instructions per level-0 pixel
// (rocprofv3, profiles/r04_pmc_summary.json): 1.74x / 1.6x halo recomputation per level, per-item index arithmetic, three LDS round trips.
// Here a wave owns 128 output columns (a lane: output columns 2l, 2l+1 <- the 12 source bytes x-4 .. x+7, ONE global_load_dwordx3 per source
// row) and a band of output rows, and streams down the band with the horizontal [1 4 6 4 1] sums of the last five source rows in registers
// (packed u16 pairs, as below): per output row two new source rows (2 x (1 align + 3 v_dot4 + 1 and + 1 pack)), one SWAR vertical sum
// (9 instructions for both columns) and one 2-byte store.  No LDS, no barrier, vertical halo 3 source rows per band, horizontal halo none
// (the neighbours' bytes come with the lane's own 12-byte load).  ~5 issue units per source pixel and level instead of ~40.
// Borders: BORDER_REFLECT_101 of the level's own image — rows by a scalar reflect of the row index; columns by per-lane byte selectors
// computed once (v_perm picks the reflected partner out of the same 12-byte window: every partner a valid output needs lies inside it),
// only in the waves that touch the left or right image edge (wave-uniform branch).  Exact integers (tests/test_gpu_frontend.py: all levels
// against the oracle at 4 sizes incl. 333 x 257 and 1278 x 1022).
// Layout the loads rely on (static in csrc/ctx.hip, checked at icg_ctx_create): every level's pitch and slot offset are multiples of 4 (the
// 12-byte windows are read as three aligned dwords) and a source level is never the last region of a slot — the window of the last pixels of
// a row reaches up to 8 bytes past the row, on the last row of a level whose pitch equals its width into the NEXT level's region, which
// this launch may be writing: those bytes are never selected (benign for the result, worth knowing under a race detector).
#define PR_STRIP_OUT 128 // output columns per wave
struct pr_sel {
    unsigned int a, b, t, v; // selectors of (V-2 V-1 V0 V1), (V0 V1 V2 V3), the (E1, E0) candidate of V4, and V4 out of (E2, candidate)
};
typedef unsigned int pr_u3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ pr_u3 pr_load(const uint8_t *row, unsigned int wx) { // source bytes wx .. wx + 11 of a row (row: wave-uniform)
    typedef pr_u3 __attribute__((aligned(4))) u3a;
    return *reinterpret_cast<const u3a *>(row + wx);
}
template <bool EDGE> __device__ __forceinline__ unsigned int pr_hsum(const pr_u3 E, const pr_sel &S) {
    unsigned int A, B, V4;
    if (EDGE) {
        A  = __builtin_amdgcn_perm(E.y, E.x, S.a);
        B  = __builtin_amdgcn_perm(E.y, E.x, S.b);
        V4 = __builtin_amdgcn_perm(E.z, __builtin_amdgcn_perm(E.y, E.x, S.t), S.v);
    } else {
        A  = __builtin_amdgcn_alignbyte(E.y, E.x, 2);
        B  = E.y;
        V4 = E.z & 0xffu;
    }
    const unsigned int h0 = __builtin_amdgcn_udot4(B, 0x00010000u, __builtin_amdgcn_udot4(A, 0x04060401u, 0u, false), false); // + V2
    const unsigned int h1 = __builtin_amdgcn_udot4(B, 0x04060401u, V4, false);
    return h0 | (h1 << 16);
}

template <bool EDGE>
__device__ __forceinline__ void pr_band(const uint8_t *src, int sh, int spitch, uint8_t *dst, int dpitch, int i0, int i1, unsigned int wx,
                                        const pr_sel &S, unsigned int dcol, bool store) {
    // (|overshoot| <= 4 rows, sh >= 6; the row offset is forced onto the scalar unit: address = scalar row pointer + the lane's 32-bit wx)
    auto srow = [&](int r) { return src + (unsigned int) __builtin_amdgcn_readfirstlane(icg_reflect1(r, sh) * spitch); };
    unsigned int hA = pr_hsum<EDGE>(pr_load(srow(2 * i0 - 2), wx), S), hB = pr_hsum<EDGE>(pr_load(srow(2 * i0 - 1), wx), S);
    unsigned int hC = pr_hsum<EDGE>(pr_load(srow(2 * i0), wx), S);
    // the two source rows of an output row are loaded one iteration ahead (one memory round trip per band, not per output row)
    pr_u3 R1 = pr_load(srow(2 * i0 + 1), wx), R2 = pr_load(srow(2 * i0 + 2), wx);
    for (int i = i0; i < i1; i++) {
        const pr_u3 N1 = pr_load(srow(2 * i + 3), wx), N2 = pr_load(srow(2 * i + 4), wx); // (past the band: reflected, valid rows; unused)
        const unsigned int hD = pr_hsum<EDGE>(R1, S), hE = pr_hsum<EDGE>(R2, S);
        // (v + 128) >> 8 of v = hA + 4 hB + 6 hC + 4 hD + hE on both packed fields (<= 65408: no carry between them)
        unsigned int v = hA + hE;
        v              = ((hB + hD) << 2) + v;
        v += (hC << 2) + (hC + hC);
        v += 0x00800080u;
        if (store) *reinterpret_cast<unsigned short *>(dst + (size_t) i * dpitch + dcol) = (unsigned short) __builtin_amdgcn_perm(v, v, 0x0c0c0301u);
        hA = hC, hB = hD, hC = hE;
        R1 = N1, R2 = N2;
    }
}

__global__ __launch_bounds__(64) void k_pyrdown_rows(uint8_t *frames, size_t slot_bytes, pre_jobs jobs, unsigned int src_off, int sw, int sh,
                                                     int spitch, unsigned int dst_off, int dw, int dh, int dpitch, int n_strips, int n_bands,
                                                     int band_rows, int n_tasks) {
    const int task = icg_xcd_chunked(blockIdx.x, n_tasks);
    if (task >= n_tasks) return;
    const int per_job = n_strips * n_bands;
    const int job = task / per_job, rem = task - job * per_job; // (wave-uniform: scalar unit)
    const int band = rem / n_strips, strip = rem - band * n_strips;
    const int dslot = __builtin_amdgcn_readfirstlane(pre_job_slot(jobs, job)); // (wave-uniform: row pointers stay on the scalar unit)
    if (dslot < 0) return; // idle job
    uint8_t *slot = frames + (size_t) dslot * slot_bytes;
    const int lane = threadIdx.x;
    const int oj   = strip * PR_STRIP_OUT + 2 * lane; // the lane's output columns oj, oj + 1
    // lanes right of the image repeat the last lane that has a valid output (their loads stay inside the row, nothing is stored)
    const int x  = min(2 * oj, (sw - 1) & ~3);        // first source column of the lane's pair
    const int wx = max(x - 4, 0);                     // window: source bytes wx .. wx + 11 (<= 8 bytes past the row end: inside the slot)
    const bool edge = strip == 0 || 2 * (strip + 1) * PR_STRIP_OUT + 4 > sw; // wave-uniform: some lane needs a reflected column
    pr_sel S;
    S.a = S.b = S.t = S.v = 0;
    if (edge) {
        int ix[7];
#pragma unroll
        for (int k = 0; k < 7; k++) ix[k] = min(max(icg_reflect101(x - 2 + k, sw) - wx, 0), 11); // window index of V[k-2]
        // (V-2 .. V3 of a lane with a valid output lie in bytes 0..7; V4 only feeds the second output: when that one is valid its
        // reflected partner is byte 8 or one of bytes 0..7 — image.hip comment above)
        S.a = (unsigned int) (ix[0] & 7) | ((unsigned int) (ix[1] & 7) << 8) | ((unsigned int) (ix[2] & 7) << 16) | ((unsigned int) (ix[3] & 7) << 24);
        S.b = (unsigned int) (ix[2] & 7) | ((unsigned int) (ix[3] & 7) << 8) | ((unsigned int) (ix[4] & 7) << 16) | ((unsigned int) (ix[5] & 7) << 24);
        S.t = 0x0c0c0c00u | (unsigned int) (ix[6] & 7);
        S.v = ix[6] >= 8 ? (0x0c0c0c04u + (unsigned int) (ix[6] - 8)) : 0x0c0c0c00u;
    }
    const int i0 = band * band_rows, i1 = min(i0 + band_rows, dh);
    if (i0 >= i1) return;
    const uint8_t *src = slot + src_off;
    uint8_t *dst       = slot + dst_off;
    const bool store   = oj < dw; // (oj + 1 == dw: the second byte lands in the row padding)
    if (edge)
        pr_band<true>(src, sh, spitch, dst, dpitch, i0, i1, (unsigned int) wx, S, (unsigned int) oj, store);
    else
        pr_band<false>(src, sh, spitch, dst, dpitch, i0, i1, (unsigned int) wx, S, (unsigned int) oj, store);
}

// ---------------------------------------------------------------------------------------------------------
// tracking.cc:98-102 on the device (segmented preprocess only): float histogram, (float)k product in float, /256.0 and accumulation in
// double, sequentially — a lane per job
__global__ void k_hist_mean(int n, const unsigned int *hist, int w, int h, double *mean) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double acc = 0;
    for (int i = 0; i < 256; i++) {
        const float hf = (float) hist[(size_t) k * 256 + i];
        acc += hf * (float) i / 256.0;
    }
    mean[k] = acc / ((double) w * h);
}

// slots: host array (the ABI entry point) — or nullptr with d_slot_ind: the slot of job k is read by the kernels from device memory and
// images[k] == nullptr marks a job that is idle in this launch (device-resident tracker).  hist_mean: host output (synchronous) or
// d_hist_mean: device output (asynchronous).
static int preprocess_impl(icg_ctx *ctx, int n, const int32_t *slots, const int32_t *d_slot_ind, const uint8_t *const *images, int stride, int channels,
                           int src_on_device, double *hist_mean, double *d_hist_mean) {
    if (!ctx || n < 0 || (n > 0 && ((!slots && !d_slot_ind) || !images))) return ICG_ERR_INVALID;
    if (n == 0) return ICG_OK;
    if (n > ctx->cfg.max_batch) return icg_fail(ctx, ICG_ERR_CAPACITY, "preprocess batch %d > max_batch %d", n, ctx->cfg.max_batch);
    if (channels != 1 && channels != 3) return icg_fail(ctx, ICG_ERR_INVALID, "channels must be 1 or 3");
    const int w = ctx->cfg.width, h = ctx->cfg.height;
    if (stride < w * channels) return icg_fail(ctx, ICG_ERR_INVALID, "stride too small");
    for (int k = 0; k < n && slots; k++)
        if (slots[k] < 0 || slots[k] >= ctx->cfg.n_slots || !images[k]) return icg_fail(ctx, ICG_ERR_INVALID, "bad slot/image %d", k);
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const bool want_hist = hist_mean || d_hist_mean;

    // CLAHE geometry
    const int T = ICG_CLAHE_TILES;
    int ew = w, eh = h;
    if (w % T != 0 || h % T != 0) {
        ew = w + (T - w % T);
        eh = h + (T - h % T);
    }
    clahe_geom g;
    g.w = w;
    g.h = h;
    g.tw = ew / T;
    g.th = eh / T;
    int area    = g.tw * g.th;
    g.lut_scale = 255.0f / area;
    g.clip      = (int) (3.0 * area / 256);
    if (g.clip < 1) g.clip = 1;
    g.inv_tw = 1.0f / g.tw;
    g.inv_th = 1.0f / g.th;
    if (CLAHE_CHUNK / g.tw + 3 > CLAHE_MAXCOLS)
        return icg_fail(ctx, ICG_ERR_INVALID, "image too small for CLAHE chunking (tile width %d)", g.tw);

    const size_t raw_batch = (size_t) ctx->raw_pitch * h;
    hipMemcpyKind kind     = src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    unsigned int *d_hist   = nullptr;
    if (want_hist) {
        if (int rcd = icg_arena_drain(ctx)) return rcd;
        ctx->arena_off = 0;
        int rc         = icg_arena_reserve(ctx, sizeof(unsigned int) * 256 * (size_t) n + 4096);
        if (rc) return rc;
        d_hist = icg_d<unsigned int>(ctx, icg_arena_alloc(ctx, sizeof(unsigned int) * 256 * (size_t) n));
        if ((rc = icg_arena_overflow_check(ctx))) return rc;
        ICG_HIP(ctx, hipMemsetAsync(d_hist, 0, sizeof(unsigned int) * 256 * (size_t) n, ctx->stream));
    }

    for (int base = 0; base < n; base += PRE_MAX_JOBS) {
        const int m = std::min(PRE_MAX_JOBS, n - base);
        pre_jobs jobs;
        memset(&jobs, 0, sizeof jobs);
        for (int k = 0; k < m && slots; k++) jobs.slot[k] = slots[base + k];
        jobs.slot_ind = d_slot_ind ? d_slot_ind + base : nullptr;
        if (channels == 3) {
            if (!ctx->d_bgr) ICG_HIP(ctx, hipMalloc((void **) &ctx->d_bgr, (size_t) w * 3 * h * ctx->cfg.max_batch));
            for (int k = 0; k < m; k++)
                if (images[base + k])
                    ICG_HIP(ctx, hipMemcpy2DAsync(ctx->d_bgr + (size_t) (base + k) * w * 3 * h, (size_t) w * 3, images[base + k], stride,
                                                  (size_t) w * 3, h, kind, ctx->stream));
            icg_prof_scope ps(ctx, "bgr2gray");
            hipLaunchKernelGGL(k_bgr2gray, dim3((w + 255) / 256, h, m), dim3(256), 0, ctx->stream,
                               ctx->d_bgr + (size_t) base * w * 3 * h, w, h, w * 3, (size_t) w * 3 * h,
                               ctx->d_raw + (size_t) base * raw_batch, ctx->raw_pitch, raw_batch);
            for (int k = 0; k < m; k++) jobs.src[k] = images[base + k] ? ctx->d_raw + (size_t) (base + k) * raw_batch : nullptr;
            jobs.stride = ctx->raw_pitch;
        } else {
            // device-resident gray frames are consumed in place when uchar4 loads are aligned; otherwise (and for host
            // frames) they are first brought into the pitched staging planes
            bool direct = src_on_device && (stride % 4 == 0);
            for (int k = 0; k < m && direct; k++) direct = (((uintptr_t) images[base + k]) % 4) == 0;
            if (direct) {
                for (int k = 0; k < m; k++) jobs.src[k] = images[base + k];
                jobs.stride = stride;
            } else {
                // contiguous frames (stride == width == staging pitch) go as ONE linear copy: the 2-D path of the runtime is several
                // times slower for pinned host memory
                const bool linear = stride == w && ctx->raw_pitch == w;
                for (int k = 0; k < m; k++) {
                    uint8_t *dst = ctx->d_raw + (size_t) (base + k) * raw_batch;
                    if (!images[base + k]) {
                        jobs.src[k] = nullptr;
                        continue;
                    }
                    if (linear)
                        ICG_HIP(ctx, hipMemcpyAsync(dst, images[base + k], (size_t) w * h, kind, ctx->stream));
                    else
                        ICG_HIP(ctx, hipMemcpy2DAsync(dst, ctx->raw_pitch, images[base + k], stride, w, h, kind, ctx->stream));
                    jobs.src[k] = dst;
                }
                jobs.stride = ctx->raw_pitch;
            }
        }
        if (want_hist) {
            icg_prof_scope ps(ctx, "hist256");
            hipLaunchKernelGGL(k_hist256, dim3(64, m), dim3(256), 0, ctx->stream, jobs, w, h, d_hist + (size_t) base * 256);
        }
        uint8_t *lut = ctx->d_lut + (size_t) base * T * T * 256;
        {
            icg_prof_scope ps(ctx, "clahe_lut");
            hipLaunchKernelGGL(k_clahe_lut, dim3(icg_xcd_grid(T * T * m)), dim3(64), 0, ctx->stream, jobs, g, lut, T * T * m);
        }
        {
            icg_prof_scope ps(ctx, "clahe_apply");
            hipLaunchKernelGGL(k_clahe_apply, dim3(T + 1, (w + CLAHE_CHUNK - 1) / CLAHE_CHUNK, m), dim3(256), 0, ctx->stream, jobs,
                               g, lut, ctx->d_frames, ctx->slot_bytes, ctx->lv[0].pitch);
        }
        for (int l = 1; l < ctx->n_levels; l++) { // (every source level is at least 43 x 43 with a pitch of 128: icg_ctx_create)
            icg_prof_scope ps(ctx, "pyrdown_rows");
            const icg_level &a = ctx->lv[l - 1], &bb = ctx->lv[l];
            const int n_strips = (bb.w + PR_STRIP_OUT - 1) / PR_STRIP_OUT;
            // bands: enough waves to fill the chip at the level's size (>= ~4 per SIMD for 64 frames at level 1), 3 halo rows per band
            const int band_rows = bb.h >= 256 ? 32 : bb.h >= 128 ? 16 : 8;
            const int n_bands   = (bb.h + band_rows - 1) / band_rows;
            const int n_tasks   = n_strips * n_bands * m;
            hipLaunchKernelGGL(k_pyrdown_rows, dim3(icg_xcd_grid(n_tasks)), dim3(64), 0, ctx->stream, ctx->d_frames, ctx->slot_bytes, jobs,
                               (unsigned int) a.off, a.w, a.h, a.pitch, (unsigned int) bb.off, bb.w, bb.h, bb.pitch, n_strips, n_bands, band_rows,
                               n_tasks);
        }
    }
    if (d_hist_mean) hipLaunchKernelGGL(k_hist_mean, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, n, d_hist, w, h, d_hist_mean);
    ICG_HIP(ctx, hipGetLastError());
    if (d_hist_mean) ctx->arena_off = 0;
    if (!hist_mean) return ICG_OK; // asynchronous: later calls on this context are stream-ordered behind these kernels

    std::vector<unsigned int> hc((size_t) n * 256);
    ICG_HIP(ctx, hipMemcpyAsync(hc.data(), d_hist, sizeof(unsigned int) * 256 * (size_t) n, hipMemcpyDeviceToHost, ctx->stream));
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    icg_prof_collect(ctx);
    // tracking.cc:98-102: float histogram, (float)k product in float, /256.0 and accumulation in double
    for (int k = 0; k < n; k++) {
        double acc = 0;
        for (int i = 0; i < 256; i++) {
            float hf = (float) hc[(size_t) k * 256 + i];
            acc += hf * (float) i / 256.0;
        }
        hist_mean[k] = acc / ((double) w * h);
    }
    ctx->arena_off = 0;
    return ICG_OK;
}

extern "C" int icg_frames_preprocess(icg_ctx *ctx, int n, const int32_t *slots, const uint8_t *const *images, int stride, int channels,
                                     int src_on_device, double *hist_mean) {
    if (n > 0 && !slots) return ICG_ERR_INVALID;
    return preprocess_impl(ctx, n, slots, nullptr, images, stride, channels, src_on_device, hist_mean, nullptr);
}

// device-resident tracker: slots read from device memory (d_slot_ind[k] < 0 or images[k] == nullptr: stream k idles), brightness mean
// (histogram gate) left in device memory; asynchronous on the context's stream
int icg_preprocess_launch_ind(icg_ctx *ctx, int n, const int32_t *d_slot_ind, const uint8_t *const *images, int stride, int channels,
                              int src_on_device, double *d_hist_mean) {
    return preprocess_impl(ctx, n, nullptr, d_slot_ind, images, stride, channels, src_on_device, nullptr, d_hist_mean);
}

extern "C" int icg_frame_download(icg_ctx *ctx, int slot, int level, uint8_t *dst, int dst_stride) {
    if (!ctx || !dst || slot < 0 || slot >= ctx->cfg.n_slots || level < 0 || level >= ctx->n_levels) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const icg_level &L = ctx->lv[level];
    if (dst_stride < L.w) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ICG_HIP(ctx, hipMemcpy2D(dst, dst_stride, ctx->d_frames + (size_t) slot * ctx->slot_bytes + L.off, L.pitch, L.w, L.h,
                             hipMemcpyDeviceToHost));
    return ICG_OK;
}
