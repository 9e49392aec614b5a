#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

// ---- drop-in mode -------------------------------------------------------------------------------------------------------------
// Built next to the reference (its include root on the path, Eigen and OpenCV available) the value types below ARE the reference's:
// Eigen vectors / matrices, cv::Point2f and the reference's own `Pose` (common/types.h:60-63), so that icg::Tracking / Frame /
// MapPoint / Map / Camera have literally the signatures of tracking/tracking.h:51-61 and the reference's GVINS can hold an
// icg::Tracking in its `tracking_` member (oracle/ref_build/ref_gvins_icg.cc compiles exactly that and replays the estimator golden).
// Switched on by ICG_REFERENCE_TYPES (not silently by header presence: the standalone libraries must not change layout because a
// build machine happens to have Eigen installed).
#if defined(ICG_REFERENCE_TYPES)
#if !__has_include(<Eigen/Geometry>) || !__has_include(<opencv2/opencv.hpp>)
#error "ICG_REFERENCE_TYPES needs Eigen and OpenCV headers on the include path"
#endif
#include <Eigen/Geometry>
#include <opencv2/opencv.hpp>

#include "common/types.h" // the reference's Pose

namespace icg {
typedef unsigned long ulong;
using Vector2d = Eigen::Vector2d;
using Vector3d = Eigen::Vector3d;
using Matrix3d = Eigen::Matrix3d;
using Matrix4d = Eigen::Matrix4d;
using Point2f  = cv::Point2f;
using ::Pose;
} // namespace icg
#else

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

// 4x4, row-major (Tracking::pose2Tcw, tracking/tracking.h:61)
struct Matrix4d {
    double m[16]{};
    double &operator()(int r, int c) { return m[r * 4 + c]; }
    double operator()(int r, int c) const { return m[r * 4 + c]; }
    static Matrix4d Zero() { return Matrix4d(); }
};

} // namespace icg
#endif // ICG_REFERENCE_TYPES

namespace icg {

// Storage-independent access (the PODs are row-major, Eigen is column-major): always through (row, col)
inline void toRowMajor(const Matrix3d &R, double *out9) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out9[i * 3 + j] = R(i, j);
}
inline Matrix3d fromRowMajor3(const double *in9) {
    Matrix3d R;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R(i, j) = in9[i * 3 + j];
    return R;
}
inline void toArray(const Vector3d &v, double *out3) { out3[0] = v[0], out3[1] = v[1], out3[2] = v[2]; }
inline Vector3d fromArray3(const double *in3) { return Vector3d(in3[0], in3[1], in3[2]); }
// the upper 3 x 4 block of a 4 x 4 matrix, row-major (what icg_triangulate takes)
inline void toRowMajor3x4(const Matrix4d &T, double *out12) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) out12[i * 4 + j] = T(i, j);
}
// R | t as the 12 doubles the C entry points exchange
inline void poseToArray12(const Pose &p, double *out12) {
    toRowMajor(p.R, out12);
    toArray(p.t, out12 + 9);
}
inline Pose poseFromArray12(const double *in12) {
    Pose p;
    p.R = fromRowMajor3(in12);
    p.t = fromArray3(in12 + 9);
    return p;
}
inline Pose identityPose() {
    Pose p;
    p.R = Matrix3d::Identity();
    p.t = Vector3d(0, 0, 0);
    return p;
}

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
