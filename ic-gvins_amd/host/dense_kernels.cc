// Dense FP64 helpers of the window solvers (solver_detail.h): the reduced-system Cholesky solve and the J^T J accumulation of
// host-evaluated factors.  Written with 4-wide vector types and a FIXED summation order, and compiled for AVX2 and for baseline x86-64
// (resolved at load time): both clones perform the same IEEE operations in the same order — no FMA contraction, no reassociation — so the
// results do not depend on the machine the library runs on.
#include <cmath>
#include <cstring>

#include "solver_detail.h"

namespace icg {
namespace solver_detail {
namespace {

typedef double v4d __attribute__((vector_size(32)));
#define ICG_CLONES __attribute__((target_clones("avx2", "default")))

inline __attribute__((always_inline)) v4d load4(const double *p) {
    v4d v;
    memcpy(&v, p, sizeof v);
    return v;
}
inline __attribute__((always_inline)) void store4(double *p, v4d v) { memcpy(p, &v, sizeof v); }

// sum of a[k] * b[k], k < n, in EIGHT interleaved partial sums p[u] (u = k mod 8 over the leading multiple of 8) combined as
// ((p0+p4) + (p2+p6)) + ((p1+p5) + (p3+p7)), plus the tail in order
inline __attribute__((always_inline)) double dot8(const double *a, const double *b, int n) {
    v4d s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    int k = 0;
    for (; k + 8 <= n; k += 8) {
        s0 += load4(a + k) * load4(b + k);
        s1 += load4(a + k + 4) * load4(b + k + 4);
    }
    const v4d e = s0 + s1;
    double t     = 0;
    for (; k < n; k++) t += a[k] * b[k];
    return ((e[0] + e[2]) + (e[1] + e[3])) + t;
}

// the four inner products of two row prefixes a0, a1 with two row prefixes b0, b1 (length n), each summed exactly as dot8 sums it; the
// loads of the four rows are shared (half the loads per multiply-add of four separate dot8 calls: the solve is load bound)
inline __attribute__((always_inline)) void dot8_2x2(const double *a0, const double *a1, const double *b0, const double *b1, int n, double out[4]) {
    v4d s00 = {0, 0, 0, 0}, t00 = s00, s01 = s00, t01 = s00, s10 = s00, t10 = s00, s11 = s00, t11 = s00;
    int k = 0;
    for (; k + 8 <= n; k += 8) {
        const v4d x0 = load4(a0 + k), y0 = load4(a0 + k + 4), x1 = load4(a1 + k), y1 = load4(a1 + k + 4);
        const v4d p0 = load4(b0 + k), q0 = load4(b0 + k + 4), p1 = load4(b1 + k), q1 = load4(b1 + k + 4);
        s00 += x0 * p0, t00 += y0 * q0;
        s01 += x0 * p1, t01 += y0 * q1;
        s10 += x1 * p0, t10 += y1 * q0;
        s11 += x1 * p1, t11 += y1 * q1;
    }
    double r00 = 0, r01 = 0, r10 = 0, r11 = 0;
    for (; k < n; k++) {
        r00 += a0[k] * b0[k];
        r01 += a0[k] * b1[k];
        r10 += a1[k] * b0[k];
        r11 += a1[k] * b1[k];
    }
    const v4d e00 = s00 + t00, e01 = s01 + t01, e10 = s10 + t10, e11 = s11 + t11;
    out[0] = ((e00[0] + e00[2]) + (e00[1] + e00[3])) + r00;
    out[1] = ((e01[0] + e01[2]) + (e01[1] + e01[3])) + r01;
    out[2] = ((e10[0] + e10[2]) + (e10[1] + e10[3])) + r10;
    out[3] = ((e11[0] + e11[2]) + (e11[1] + e11[3])) + r11;
}

} // namespace

// in-place Cholesky solve of the symmetric positive definite n x n system A x = b (row-major, lower triangle used).
// 2 x 2 tiles; the part of every element that involves the columns left of its tile is an inner product of two contiguous row prefixes
// (dot8 order), four of them at a time with shared loads; 1.3 M multiply-adds at the P = 157 of a 10-keyframe GNSS/INS/visual window — as a
// scalar column sweep this solve was 40 % of a single-stream window solve.
ICG_CLONES bool choleskySolve(int n, std::vector<double> &Av, std::vector<double> &bv) {
    double *A = Av.data(), *b = bv.data();
    // column tiles outermost: for a fixed tile of two columns the tiles of all rows below are independent of each other (they read row
    // prefixes that earlier column tiles completed), so their reduce / divide latencies overlap instead of forming one chain per row
    for (int j = 0; j < n; j += 2) {
        double *C0 = A + (size_t) j * n, *C1 = C0 + n;
        const bool pair = j + 1 < n;
        // the diagonal tile
        const double d0 = C0[j] - dot8(C0, C0, j);
        if (!(d0 > 0.0) || !std::isfinite(d0)) return false;
        C0[j] = std::sqrt(d0);
        if (!pair) break;
        C1[j]           = (C1[j] - dot8(C1, C0, j)) / C0[j];
        const double d1 = (C1[j + 1] - dot8(C1, C1, j)) - C1[j] * C1[j];
        if (!(d1 > 0.0) || !std::isfinite(d1)) return false;
        C1[j + 1] = std::sqrt(d1);
        // the tiles below it
        int i = j + 2;
        for (; i + 1 < n; i += 2) {
            double *R0 = A + (size_t) i * n, *R1 = R0 + n;
            double S[4];
            dot8_2x2(R0, R1, C0, C1, j, S);
            const double l00 = (R0[j] - S[0]) / C0[j];
            const double l10 = (R1[j] - S[2]) / C0[j];
            R0[j]            = l00;
            R1[j]            = l10;
            R0[j + 1]        = ((R0[j + 1] - S[1]) - l00 * C1[j]) / C1[j + 1];
            R1[j + 1]        = ((R1[j + 1] - S[3]) - l10 * C1[j]) / C1[j + 1];
        }
        if (i < n) { // odd n: the last row on its own, same arithmetic per element
            double *R0       = A + (size_t) i * n;
            const double l00 = (R0[j] - dot8(R0, C0, j)) / C0[j];
            R0[j]            = l00;
            R0[j + 1]        = ((R0[j + 1] - dot8(R0, C1, j)) - l00 * C1[j]) / C1[j + 1];
        }
    }
    for (int r = 0; r < n; r++) b[r] = (b[r] - dot8(A + (size_t) r * n, b, r)) / A[(size_t) r * n + r];
    // L^T x = y as a column sweep: row r of L is contiguous
    for (int r = n - 1; r >= 0; r--) {
        const double *Ar = A + (size_t) r * n;
        const double x   = b[r] / Ar[r];
        b[r]             = x;
        for (int k = 0; k < r; k++) b[k] -= Ar[k] * x;
    }
    return true;
}

// T (nf x nf, upper triangle) += J^T J and g (nf) += J^T r for a dense row-major J (nr x nf): residual index outermost, so that every
// cell is the sum over k in ascending order starting from zero — the value a cell-by-cell inner product produces
ICG_CLONES void accumulateJtJ(int nr, int nf, const double *J, const double *r, double *T, double *g) {
    for (int k = 0; k < nr; k++) {
        const double *Jk = J + (size_t) k * nf;
        const double rk  = r[k];
        for (int x = 0; x < nf; x++) {
            const double jx = Jk[x];
            double *Tx      = T + (size_t) x * nf;
            const v4d jv    = {jx, jx, jx, jx};
            int y           = x;
            for (; y + 4 <= nf; y += 4) store4(Tx + y, load4(Tx + y) + jv * load4(Jk + y));
            for (; y < nf; y++) Tx[y] += jx * Jk[y];
            g[x] += jx * rk;
        }
    }
}

} // namespace solver_detail
} // namespace icg
