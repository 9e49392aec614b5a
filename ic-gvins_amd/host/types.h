// Host-side value types of the MI355X-native IC-GVINS front-end.
//
// The reference's public API speaks Eigen (Vector3d, Matrix3d, Pose) and OpenCV (cv::Point2f, cv::Mat) —
// reference: ic_gvins/ic_gvins/common/types.h:32-63, tracking/*.h.  Neither library exists in this environment, so
// the same names are provided here as minimal PODs with the semantics the tracker relies on (column vectors, row-major
// 3x3 storage, float pixel coordinates).  A maintainer linking against the real libraries converts at the boundary
// with a memcpy (layouts documented per type).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace icg {

typedef unsigned long ulong;

struct Vector2d {
    double v[2]{0, 0};
    Vector2d() = default;
    Vector2d(double x, double y) : v{x, y} {}
    double &operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1]); }
};
inline Vector2d operator-(const Vector2d &a, const Vector2d &b) { return {a[0] - b[0], a[1] - b[1]}; }

struct Vector3d {
    double v[3]{0, 0, 0};
    Vector3d() = default;
    Vector3d(double x, double y, double z) : v{x, y, z} {}
    double &operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double z() const { return v[2]; }
    double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
};
inline Vector3d operator+(const Vector3d &a, const Vector3d &b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline Vector3d operator-(const Vector3d &a, const Vector3d &b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline Vector3d operator/(const Vector3d &a, double s) { return {a[0] / s, a[1] / s, a[2] / s}; }
inline Vector3d operator*(const Vector3d &a, double s) { return {a[0] * s, a[1] * s, a[2] * s}; }

// Row-major 3x3 (Eigen::Matrix3d is column-major: transpose when memcpy-ing to/from Eigen).
struct Matrix3d {
    double m[9]{1, 0, 0, 0, 1, 0, 0, 0, 1};
    double &operator()(int r, int c) { return m[r * 3 + c]; }
    double operator()(int r, int c) const { return m[r * 3 + c]; }
    Matrix3d transpose() const {
        Matrix3d t;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) t(i, j) = (*this)(j, i);
        return t;
    }
    static Matrix3d Identity() { return Matrix3d(); }
};
inline Matrix3d operator*(const Matrix3d &a, const Matrix3d &b) {
    Matrix3d r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j);
    return r;
}
inline Vector3d operator*(const Matrix3d &a, const Vector3d &v) {
    return {a(0, 0) * v[0] + a(0, 1) * v[1] + a(0, 2) * v[2], a(1, 0) * v[0] + a(1, 1) * v[1] + a(1, 2) * v[2],
            a(2, 0) * v[0] + a(2, 1) * v[1] + a(2, 2) * v[2]};
}

// common/types.h:60-63
struct Pose {
    Matrix3d R;
    Vector3d t;
};

// cv::Point2f
struct Point2f {
    float x{0}, y{0};
    Point2f() = default;
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};

// cv::Mat stand-in for 8-bit images (1 or 3 channels). Data is reference counted like cv::Mat; an image may also
// live in device memory already (device=true), in which case `data` is a HIP device pointer owned by the caller.
struct Mat {
    int rows{0}, cols{0}, chans{1};
    size_t step{0};
    uint8_t *data{nullptr};
    bool device{false};
    std::shared_ptr<std::vector<uint8_t>> storage;

    Mat() = default;
    Mat(int rows_, int cols_, int chans_ = 1) : rows(rows_), cols(cols_), chans(chans_), step((size_t) cols_ * chans_) {
        storage = std::make_shared<std::vector<uint8_t>>((size_t) rows * step);
        data    = storage->data();
    }
    // wraps caller-owned memory (host or device), no copy
    static Mat wrap(uint8_t *ptr, int rows_, int cols_, int chans_, size_t step_, bool on_device) {
        Mat m;
        m.rows = rows_, m.cols = cols_, m.chans = chans_, m.step = step_, m.data = ptr, m.device = on_device;
        return m;
    }
    bool empty() const { return data == nullptr; }
    int channels() const { return chans; }
    void copyTo(Mat &dst) const {
        if (device) { // device images are not duplicated on the host
            dst = *this;
            return;
        }
        dst = Mat(rows, cols, chans);
        for (int r = 0; r < rows; r++) memcpy(dst.data + (size_t) r * dst.step, data + (size_t) r * step, (size_t) cols * chans);
    }
};

} // namespace icg
