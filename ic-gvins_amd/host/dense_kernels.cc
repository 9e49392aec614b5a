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

} // namespace

// in-place Cholesky solve of the symmetric positive definite n x n system A x = b (row-major, lower triangle used).
// Row-oriented (Cholesky-Banachiewicz): every inner product runs over two contiguous row prefixes, 1.3 M multiply-adds at the P = 157 of a
// 10-keyframe GNSS/INS/visual window — as a scalar column sweep this solve was 40 % of a single-stream window solve.
ICG_CLONES bool choleskySolve(int n, std::vector<double> &Av, std::vector<double> &bv) {
    double *A = Av.data(), *b = bv.data();
    for (int i = 0; i < n; i++) {
        double *Ai = A + (size_t) i * n;
        for (int j = 0; j < i; j++) {
            const double *Aj = A + (size_t) j * n;
            Ai[j]            = (Ai[j] - dot8(Ai, Aj, j)) / Aj[j];
        }
        const double d = Ai[i] - dot8(Ai, Ai, i);
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        Ai[i] = std::sqrt(d);
    }
    for (int i = 0; i < n; i++) b[i] = (b[i] - dot8(A + (size_t) i * n, b, i)) / A[(size_t) i * n + i];
    // L^T x = y as a column sweep: row i of L is contiguous
    for (int i = n - 1; i >= 0; i--) {
        const double *Ai = A + (size_t) i * n;
        const double x   = b[i] / Ai[i];
        b[i]             = x;
        for (int k = 0; k < i; k++) b[k] -= Ai[k] * x;
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
