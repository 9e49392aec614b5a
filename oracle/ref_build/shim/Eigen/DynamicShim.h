// ORACLE / TEST INFRASTRUCTURE ONLY — run-time sized companion of the fixed-size interface shim in Geometry: the slice of
// Eigen's dynamic API (MatrixXd, VectorXd, row-major dynamic matrices, Map, blocks/segments, array().select(), asDiagonal(),
// LLT, SelfAdjointEigenSolver) that the reference's preintegration and marginalization sources use
// (preintegration_{base,normal,earth}.{h,cc}, factors/{residual_block_info,marginalization_info,marginalization_factor}.h),
// so that they compile UNMODIFIED from where they lie.  NOT Eigen: eager evaluation everywhere (every expression returns a
// column-major MatrixXd / VectorXd copy), inverse() by Gauss-Jordan with partial pivoting, LLT by the textbook unblocked
// Cholesky, SelfAdjointEigenSolver by cyclic Jacobi (eigenvalues ascending like Eigen; eigenvector signs / bases of repeated
// eigenvalues are NOT Eigen's — compare invariants).  Results agree with real Eigen up to rounding.
#pragma once
#include <algorithm>
#include <vector>

namespace Eigen {

const int Dynamic = -1;

template <typename D> struct DynBase;
template <typename XprT> class DynRef;
struct ArrayXd;
struct DiagXd;

typedef Matrix<double, Dynamic, Dynamic, ColMajor> MatrixXd;
typedef Matrix<double, Dynamic, 1, ColMajor> VectorXd;

// ---- read interface of everything run-time sized ---------------------------------------------------------------------
template <typename D> struct DynBase {
    const D &derived() const { return static_cast<const D &>(*this); }
    int rows() const { return derived().rows_(); }
    int cols() const { return derived().cols_(); }
    int size() const { return rows() * cols(); }
    double get(int i, int j) const { return derived().get_(i, j); }
    double operator()(int i, int j) const { return get(i, j); }
    double operator()(int i) const { return cols() == 1 ? get(i, 0) : get(0, i); }
    inline MatrixXd transpose() const;
    inline MatrixXd block(int r0, int c0, int nr, int nc) const;
    inline MatrixXd leftCols(int n) const;
    inline MatrixXd middleCols(int c0, int n) const;
    inline VectorXd segment(int i0, int n) const;
    template <int N> Matrix<double, N, 1> head() const {
        Matrix<double, N, 1> h;
        for (int i = 0; i < N; i++) h(i) = (*this)(i);
        return h;
    }
    double squaredNorm() const {
        double s = 0;
        for (int j = 0; j < cols(); j++) for (int i = 0; i < rows(); i++) s += get(i, j) * get(i, j);
        return s;
    }
    inline ArrayXd array() const;
    inline VectorXd cwiseSqrt() const;
    inline DiagXd asDiagonal() const;
};

// ---- owning dynamic matrix, either storage order -----------------------------------------------------------------------
template <int O> class Matrix<double, Dynamic, Dynamic, O> : public DynBase<Matrix<double, Dynamic, Dynamic, O>> {
public:
    int r_ = 0, c_ = 0;
    std::vector<double> d;
    Matrix() {}
    Matrix(int r, int c) : r_(r), c_(c), d((size_t) r * c, 0.0) {}
    template <typename D2> Matrix(const DynBase<D2> &o) { assign(o); }
    template <typename D2, int R, int C> Matrix(const DenseBase<D2, R, C> &o) : Matrix(R, C) {
        for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) set(i, j, o.coeff(i, j));
    }
    template <typename D2> Matrix &operator=(const DynBase<D2> &o) { assign(o); return *this; }
    template <typename D2, int R, int C> Matrix &operator=(const DenseBase<D2, R, C> &o) { return *this = Matrix(o); }
    template <typename D2> void assign(const DynBase<D2> &o) {
        Matrix t(o.rows(), o.cols());
        for (int i = 0; i < t.r_; i++) for (int j = 0; j < t.c_; j++) t.set(i, j, o.get(i, j));
        r_ = t.r_, c_ = t.c_;
        d.swap(t.d);
    }
    int rows_() const { return r_; }
    int cols_() const { return c_; }
    size_t idx(int i, int j) const { return O == RowMajor ? (size_t) i * c_ + j : (size_t) j * r_ + i; }
    double get_(int i, int j) const { return d[idx(i, j)]; }
    void set(int i, int j, double v) { d[idx(i, j)] = v; }
    using DynBase<Matrix>::operator();
    double &operator()(int i, int j) { return d[idx(i, j)]; }
    double *data() { return d.data(); }
    const double *data() const { return d.data(); }
    void resize(int r, int c) { r_ = r, c_ = c; d.assign((size_t) r * c, 0.0); }
    static Matrix Zero(int r, int c) { return Matrix(r, c); }
    static Matrix Identity(int r, int c) { Matrix m(r, c); m.setIdentity(); return m; }
    void setZero() { for (auto &v : d) v = 0; }
    void setZero(int r, int c) { resize(r, c); }
    void setIdentity() { setZero(); for (int i = 0; i < (r_ < c_ ? r_ : c_); i++) set(i, i, 1.0); }
    void setIdentity(int r, int c) { resize(r, c); setIdentity(); }
    Matrix &operator*=(double s) { for (auto &v : d) v *= s; return *this; }
    template <int BR, int BC> Block<Matrix, BR, BC> block(int r0, int c0) { return Block<Matrix, BR, BC>(*this, r0, c0); }
    using DynBase<Matrix>::block;
    using DynBase<Matrix>::leftCols;
    DynRef<Matrix> block(int r0, int c0, int nr, int nc) { return DynRef<Matrix>(*this, r0, c0, nr, nc); }
    DynRef<Matrix> leftCols(int n) { return DynRef<Matrix>(*this, 0, 0, r_, n); }
    // fixed-size interface needs (used by Block<>)
    double get(int i, int j) const { return get_(i, j); }
    inline MatrixXd inverse() const;
};

// ---- writable rectangular view (block / segment / leftCols of something mutable) ---------------------------------------
template <typename XprT> class DynRef : public DynBase<DynRef<XprT>> {
public:
    XprT &m;
    int r0, c0, nr, nc;
    DynRef(XprT &m_, int r, int c, int rows, int cols) : m(m_), r0(r), c0(c), nr(rows), nc(cols) {}
    int rows_() const { return nr; }
    int cols_() const { return nc; }
    double get_(int i, int j) const { return m.get(r0 + i, c0 + j); }
    template <typename D2> DynRef &operator=(const DynBase<D2> &o) {
        MatrixXd t(o); // evaluate first: the source may alias the destination
        assert(t.rows() == nr && t.cols() == nc);
        for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) m.set(r0 + i, c0 + j, t.get(i, j));
        return *this;
    }
    DynRef &operator=(const DynRef &o) { return operator=<DynRef>(o); }
    template <typename D2, int R, int C> DynRef &operator=(const DenseBase<D2, R, C> &o) {
        assert(R == nr && C == nc);
        double t[R * C];
        for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) t[i * C + j] = o.coeff(i, j);
        for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) m.set(r0 + i, c0 + j, t[i * C + j]);
        return *this;
    }
    template <typename D2> DynRef &operator+=(const DynBase<D2> &o) {
        MatrixXd t(o);
        for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) m.set(r0 + i, c0 + j, m.get(r0 + i, c0 + j) + t.get(i, j));
        return *this;
    }
    template <typename D2> DynRef &operator-=(const DynBase<D2> &o) {
        MatrixXd t(o);
        for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) m.set(r0 + i, c0 + j, m.get(r0 + i, c0 + j) - t.get(i, j));
        return *this;
    }
};
template <typename XprT> using RtBlock = DynRef<XprT>; // name used by the fixed-size Map::block(r, c, nr, nc)

// ---- owning dynamic column vector ----------------------------------------------------------------------------------------
template <> class Matrix<double, Dynamic, 1, ColMajor> : public DynBase<Matrix<double, Dynamic, 1, ColMajor>> {
public:
    std::vector<double> d;
    Matrix() {}
    explicit Matrix(int n) : d((size_t) n, 0.0) {}
    template <typename D2> Matrix(const DynBase<D2> &o) { assign(o); }
    inline Matrix(const ArrayXd &a);
    template <typename D2> Matrix &operator=(const DynBase<D2> &o) { assign(o); return *this; }
    template <typename D2> void assign(const DynBase<D2> &o) {
        assert(o.cols() == 1);
        std::vector<double> t((size_t) o.rows());
        for (int i = 0; i < o.rows(); i++) t[(size_t) i] = o.get(i, 0);
        d.swap(t);
    }
    int rows_() const { return (int) d.size(); }
    int cols_() const { return 1; }
    double get_(int i, int) const { return d[(size_t) i]; }
    double get(int i, int j) const { return get_(i, j); }
    void set(int i, int, double v) { d[(size_t) i] = v; }
    using DynBase<Matrix>::operator();
    double &operator()(int i) { return d[(size_t) i]; }
    double *data() { return d.data(); }
    const double *data() const { return d.data(); }
    void resize(int n) { d.assign((size_t) n, 0.0); }
    static Matrix Zero(int n) { return Matrix(n); }
    void setZero() { for (auto &v : d) v = 0; }
    Matrix &operator*=(double s) { for (auto &v : d) v *= s; return *this; }
    using DynBase<Matrix>::segment;
    DynRef<Matrix> segment(int i0, int n) { return DynRef<Matrix>(*this, i0, 0, n, 1); }
    template <int N> DynRef<Matrix> segment(int i0) { return DynRef<Matrix>(*this, i0, 0, N, 1); }
};

// ---- Map of dynamic types over external storage --------------------------------------------------------------------------
template <> class Map<VectorXd> : public DynBase<Map<VectorXd>> {
public:
    double *p;
    int n;
    Map(double *ptr, int size) : p(ptr), n(size) {}
    int rows_() const { return n; }
    int cols_() const { return 1; }
    double get_(int i, int) const { return p[i]; }
    template <typename D2> Map &operator=(const DynBase<D2> &o) {
        VectorXd t(o);
        assert(t.rows() == n);
        for (int i = 0; i < n; i++) p[i] = t.get(i, 0);
        return *this;
    }
};
template <> class Map<const VectorXd> : public DynBase<Map<const VectorXd>> {
public:
    const double *p;
    int n;
    Map(const double *ptr, int size) : p(ptr), n(size) {}
    int rows_() const { return n; }
    int cols_() const { return 1; }
    double get_(int i, int) const { return p[i]; }
};
template <int O> class Map<Matrix<double, Dynamic, Dynamic, O>> : public DynBase<Map<Matrix<double, Dynamic, Dynamic, O>>> {
public:
    double *p;
    int r_, c_;
    Map(double *ptr, int r, int c) : p(ptr), r_(r), c_(c) {}
    int rows_() const { return r_; }
    int cols_() const { return c_; }
    size_t idx(int i, int j) const { return O == RowMajor ? (size_t) i * c_ + j : (size_t) j * r_ + i; }
    double get_(int i, int j) const { return p[idx(i, j)]; }
    double get(int i, int j) const { return get_(i, j); }
    void set(int i, int j, double v) { p[idx(i, j)] = v; }
    void setZero() { for (size_t k = 0; k < (size_t) r_ * c_; k++) p[k] = 0; }
    DynRef<Map> leftCols(int n) { return DynRef<Map>(*this, 0, 0, r_, n); }
};

// ---- element-wise helpers: (v.array() > eps).select(v.array()[.inverse()], 0) -------------------------------------------
struct ArrayXd {
    std::vector<double> v;
    struct Mask {
        std::vector<char> m;
        ArrayXd select(const ArrayXd &a, double otherwise) const {
            ArrayXd r;
            r.v.resize(m.size());
            for (size_t i = 0; i < m.size(); i++) r.v[i] = m[i] ? a.v[i] : otherwise;
            return r;
        }
    };
    Mask operator>(double t) const {
        Mask k;
        k.m.resize(v.size());
        for (size_t i = 0; i < v.size(); i++) k.m[i] = v[i] > t;
        return k;
    }
    ArrayXd inverse() const {
        ArrayXd r;
        r.v.resize(v.size());
        for (size_t i = 0; i < v.size(); i++) r.v[i] = 1.0 / v[i];
        return r;
    }
};
inline VectorXd::Matrix(const ArrayXd &a) : d(a.v) {}
template <typename D> ArrayXd DynBase<D>::array() const {
    assert(cols() == 1);
    ArrayXd a;
    a.v.resize((size_t) rows());
    for (int i = 0; i < rows(); i++) a.v[(size_t) i] = get(i, 0);
    return a;
}
template <typename D> VectorXd DynBase<D>::cwiseSqrt() const {
    assert(cols() == 1);
    VectorXd r(rows());
    for (int i = 0; i < rows(); i++) r(i) = std::sqrt(get(i, 0));
    return r;
}
struct DiagXd {
    std::vector<double> v;
};
template <typename D> DiagXd DynBase<D>::asDiagonal() const {
    assert(cols() == 1);
    DiagXd g;
    g.v.resize((size_t) rows());
    for (int i = 0; i < rows(); i++) g.v[(size_t) i] = get(i, 0);
    return g;
}

// ---- DynBase members that return copies -----------------------------------------------------------------------------------
template <typename D> MatrixXd DynBase<D>::transpose() const {
    MatrixXd t(cols(), rows());
    for (int i = 0; i < rows(); i++) for (int j = 0; j < cols(); j++) t.set(j, i, get(i, j));
    return t;
}
template <typename D> MatrixXd DynBase<D>::block(int r0, int c0, int nr, int nc) const {
    MatrixXd t(nr, nc);
    for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) t.set(i, j, get(r0 + i, c0 + j));
    return t;
}
template <typename D> MatrixXd DynBase<D>::leftCols(int n) const { return block(0, 0, rows(), n); }
template <typename D> MatrixXd DynBase<D>::middleCols(int c0, int n) const { return block(0, c0, rows(), n); }
template <typename D> VectorXd DynBase<D>::segment(int i0, int n) const {
    assert(cols() == 1);
    VectorXd t(n);
    for (int i = 0; i < n; i++) t(i) = get(i0 + i, 0);
    return t;
}

// ---- arithmetic: anything with at least one run-time sized operand -> MatrixXd --------------------------------------------
template <typename A, typename B> inline MatrixXd dyn_mul(const A &a, int ar, int ak, const B &b, int bc) {
    MatrixXd r(ar, bc);
    for (int i = 0; i < ar; i++)
        for (int j = 0; j < bc; j++) {
            double s = 0;
            for (int k = 0; k < ak; k++) s += a.get(i, k) * b.get(k, j);
            r.set(i, j, s);
        }
    return r;
}
template <typename A, typename B> MatrixXd operator*(const DynBase<A> &a, const DynBase<B> &b) {
    assert(a.cols() == b.rows());
    return dyn_mul(a, a.rows(), a.cols(), b, b.cols());
}
template <typename A, typename D, int R, int C> MatrixXd operator*(const DynBase<A> &a, const DenseBase<D, R, C> &b) {
    assert(a.cols() == R);
    return dyn_mul(a, a.rows(), R, b.derived(), C);
}
template <typename D, int R, int C, typename B> MatrixXd operator*(const DenseBase<D, R, C> &a, const DynBase<B> &b) {
    assert(C == b.rows());
    return dyn_mul(a.derived(), R, C, b, b.cols());
}
template <typename A, typename B> MatrixXd operator+(const DynBase<A> &a, const DynBase<B> &b) {
    assert(a.rows() == b.rows() && a.cols() == b.cols());
    MatrixXd r(a.rows(), a.cols());
    for (int i = 0; i < r.rows(); i++) for (int j = 0; j < r.cols(); j++) r.set(i, j, a.get(i, j) + b.get(i, j));
    return r;
}
template <typename A, typename B> MatrixXd operator-(const DynBase<A> &a, const DynBase<B> &b) {
    assert(a.rows() == b.rows() && a.cols() == b.cols());
    MatrixXd r(a.rows(), a.cols());
    for (int i = 0; i < r.rows(); i++) for (int j = 0; j < r.cols(); j++) r.set(i, j, a.get(i, j) - b.get(i, j));
    return r;
}
template <typename A> MatrixXd operator-(const DynBase<A> &a) {
    MatrixXd r(a.rows(), a.cols());
    for (int i = 0; i < r.rows(); i++) for (int j = 0; j < r.cols(); j++) r.set(i, j, -a.get(i, j));
    return r;
}
template <typename A> MatrixXd operator*(double s, const DynBase<A> &a) {
    MatrixXd r(a.rows(), a.cols());
    for (int i = 0; i < r.rows(); i++) for (int j = 0; j < r.cols(); j++) r.set(i, j, s * a.get(i, j));
    return r;
}
template <typename A> MatrixXd operator*(const DynBase<A> &a, double s) {
    MatrixXd r(a.rows(), a.cols());
    for (int i = 0; i < r.rows(); i++) for (int j = 0; j < r.cols(); j++) r.set(i, j, a.get(i, j) * s);
    return r;
}
template <typename B> MatrixXd operator*(const DiagXd &g, const DynBase<B> &b) {
    assert((int) g.v.size() == b.rows());
    MatrixXd r(b.rows(), b.cols());
    for (int i = 0; i < r.rows(); i++) for (int j = 0; j < r.cols(); j++) r.set(i, j, g.v[(size_t) i] * b.get(i, j));
    return r;
}
template <typename A> MatrixXd operator*(const DynBase<A> &a, const DiagXd &g) {
    assert((int) g.v.size() == a.cols());
    MatrixXd r(a.rows(), a.cols());
    for (int i = 0; i < r.rows(); i++) for (int j = 0; j < r.cols(); j++) r.set(i, j, a.get(i, j) * g.v[(size_t) j]);
    return r;
}

template <int O> MatrixXd Matrix<double, Dynamic, Dynamic, O>::inverse() const { // Gauss-Jordan, partial pivoting
    assert(r_ == c_);
    const int n = r_;
    MatrixXd a(*this), inv = MatrixXd::Identity(n, n);
    for (int k = 0; k < n; k++) {
        int p = k;
        for (int i = k + 1; i < n; i++) if (std::fabs(a.get(i, k)) > std::fabs(a.get(p, k))) p = i;
        if (p != k) for (int j = 0; j < n; j++) { std::swap(a(k, j), a(p, j)); std::swap(inv(k, j), inv(p, j)); }
        const double piv = a.get(k, k);
        for (int j = 0; j < n; j++) { a(k, j) /= piv; inv(k, j) /= piv; }
        for (int i = 0; i < n; i++) {
            if (i == k) continue;
            const double f = a.get(i, k);
            if (f == 0.0) continue;
            for (int j = 0; j < n; j++) { a(i, j) -= f * a.get(k, j); inv(i, j) -= f * inv.get(k, j); }
        }
    }
    return inv;
}

// Cholesky A = L L^T of a fixed-size SPD matrix type (constructed from anything with get(i, j))
template <typename MatT> class LLT {
    MatT L_;
public:
    template <typename X> explicit LLT(const X &a) {
        const int n = MatT::RowsAtCompileTime;
        for (int j = 0; j < n; j++) {
            double s = a.get(j, j);
            for (int k = 0; k < j; k++) s -= L_(j, k) * L_(j, k);
            const double ljj = std::sqrt(s);
            L_(j, j) = ljj;
            for (int i = j + 1; i < n; i++) {
                double t = a.get(i, j);
                for (int k = 0; k < j; k++) t -= L_(i, k) * L_(j, k);
                L_(i, j) = t / ljj;
            }
        }
    }
    MatT matrixL() const { return L_; }
};

// symmetric eigen-decomposition by cyclic Jacobi; eigenvalues ascending, eigenvectors in the columns
template <typename MatT> class SelfAdjointEigenSolver {
    VectorXd w_;
    MatrixXd V_;
public:
    template <typename D2> explicit SelfAdjointEigenSolver(const DynBase<D2> &A0) {
        const int n = A0.rows();
        MatrixXd A(n, n), V = MatrixXd::Identity(n, n);
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) A.set(i, j, 0.5 * (A0.get(i, j) + A0.get(j, i)));
        for (int sweep = 0; sweep < 100; sweep++) {
            double off = 0, diag = 0;
            for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) (i == j ? diag : off) += A.get(i, j) * A.get(i, j);
            if (off <= 1e-32 * diag || off == 0.0) break;
            for (int p = 0; p < n - 1; p++)
                for (int q = p + 1; q < n; q++) {
                    const double apq = A.get(p, q);
                    if (apq == 0.0) continue;
                    const double theta = (A.get(q, q) - A.get(p, p)) / (2.0 * apq);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                    const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                    for (int k = 0; k < n; k++) {
                        const double akp = A.get(k, p), akq = A.get(k, q);
                        A.set(k, p, c * akp - s * akq);
                        A.set(k, q, s * akp + c * akq);
                    }
                    for (int k = 0; k < n; k++) {
                        const double apk = A.get(p, k), aqk = A.get(q, k);
                        A.set(p, k, c * apk - s * aqk);
                        A.set(q, k, s * apk + c * aqk);
                    }
                    for (int k = 0; k < n; k++) {
                        const double vkp = V.get(k, p), vkq = V.get(k, q);
                        V.set(k, p, c * vkp - s * vkq);
                        V.set(k, q, s * vkp + c * vkq);
                    }
                }
        }
        std::vector<int> order((size_t) n);
        for (int i = 0; i < n; i++) order[(size_t) i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return A.get(a, a) < A.get(b, b); });
        w_ = VectorXd(n);
        V_ = MatrixXd(n, n);
        for (int k = 0; k < n; k++) {
            w_(k) = A.get(order[(size_t) k], order[(size_t) k]);
            for (int i = 0; i < n; i++) V_.set(i, k, V.get(i, order[(size_t) k]));
        }
    }
    const VectorXd &eigenvalues() const { return w_; }
    const MatrixXd &eigenvectors() const { return V_; }
};

} // namespace Eigen
