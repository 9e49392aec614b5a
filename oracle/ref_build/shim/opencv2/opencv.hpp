// ORACLE / TEST INFRASTRUCTURE ONLY — the slice of the OpenCV C++ interface that the reference's tracker sources use
// (tracking/{tracking,frame,feature,mappoint,map,camera}.{h,cc}), so that they compile UNMODIFIED from where they lie.
// NOT OpenCV: cv::Mat is a minimal reference-counted 2-D array with ROI views; every image-processing ENTRY POINT
// (CLAHE, cvtColor, calcHist, calcOpticalFlowPyrLK, goodFeaturesToTrack, cornerSubPix, findFundamentalMat, undistortPoints,
// circle) forwards to the CPU restatement in oracle/orc_*.cc — those primitives therefore stay
// "unpinned vs OpenCV"; what this build pins is everything the reference does AROUND them (tracking.cc's control flow,
// bookkeeping, container orders, keyframe logic, triangulation gates, map maintenance).
#pragma once
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "../../../oracle.h"

#define CV_8UC1 0
#define CV_8UC3 16
#define CV_16SC2 11
#define CV_32FC1 5
#define CV_32F 5
#define CV_64FC1 6
#define CV_64F 6

namespace cv {

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    template <typename U> Point_(const Point_<U> &o) : x((T) o.x), y((T) o.y) {}
};
typedef Point_<float> Point2f;
typedef Point_<int> Point;
template <typename T> Point_<T> operator-(const Point_<T> &a, const Point_<T> &b) { return Point_<T>(a.x - b.x, a.y - b.y); }
template <typename T> Point_<T> operator+(const Point_<T> &a, const Point_<T> &b) { return Point_<T>(a.x + b.x, a.y + b.y); }

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};
struct Scalar {
    double v[4];
    Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {}
};
struct TermCriteria {
    enum { COUNT = 1, MAX_ITER = 1, EPS = 2 };
    int type, maxCount;
    double epsilon;
    TermCriteria() : type(0), maxCount(0), epsilon(0) {}
    TermCriteria(int t, int n, double e) : type(t), maxCount(n), epsilon(e) {}
};
template <typename T> using Ptr = std::shared_ptr<T>;

enum { COLOR_BGR2GRAY = 6, FM_RANSAC = 8, FILLED = -1, OPTFLOW_USE_INITIAL_FLOW = 4, INTER_LINEAR = 1, BORDER_CONSTANT = 0 };

inline int cvRound(double v) { return (int) std::lrint(v); }
inline int elemSize(int type) {
    switch (type) {
    case CV_8UC1: return 1;
    case CV_8UC3: return 3;
    case CV_16SC2: return 4;
    case CV_32FC1: return 4;
    default: return 8;
    }
}

// 2-D array with shared storage; colRange/rowRange produce views that remember their offset inside the parent
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t *data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(Size s, int type) { create(s.height, s.width, type); }
    Mat(Size s, int type, const Scalar &v) {
        create(s.height, s.width, type);
        assert(type == CV_8UC1);
        memset(data, (int) v.v[0], step * (size_t) rows);
    }
    Mat(Size s, int type, int v) : Mat(s, type, Scalar(v)) {}
    // user-allocated data (not owned)
    Mat(int r, int c, int type, void *ptr, size_t st = 0) : rows(r), cols(c), type_(type) {
        step = st ? st : (size_t) c * elemSize(type);
        data = (uint8_t *) ptr;
        full_w_ = c, full_h_ = r, origin_ = data;
    }
    void create(int r, int c, int type) {
        rows = r, cols = c, type_ = type;
        step  = (size_t) c * elemSize(type);
        buf_  = std::shared_ptr<uint8_t>(new uint8_t[step * (size_t) r + 16](), std::default_delete<uint8_t[]>());
        data  = buf_.get();
        off_x_ = off_y_ = 0;
        full_w_ = c, full_h_ = r, origin_ = data;
    }
    int type() const { return type_; }
    int channels() const { return type_ == CV_8UC3 ? 3 : (type_ == CV_16SC2 ? 2 : 1); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    Size size() const { return Size(cols, rows); }
    template <typename T> T &at(int i) { return rows == 1 ? ((T *) data)[i] : *(T *) (data + step * (size_t) i); }
    template <typename T> const T &at(int i) const { return rows == 1 ? ((const T *) data)[i] : *(const T *) (data + step * (size_t) i); }
    template <typename T> T &at(int r, int c) { return ((T *) (data + step * (size_t) r))[c]; }
    template <typename T> const T &at(int r, int c) const { return ((const T *) (data + step * (size_t) r))[c]; }
    Mat colRange(int c0, int c1) const {
        Mat m = *this;
        m.data += (size_t) c0 * elemSize(type_);
        m.cols = c1 - c0;
        m.off_x_ += c0;
        return m;
    }
    Mat rowRange(int r0, int r1) const {
        Mat m = *this;
        m.data += step * (size_t) r0;
        m.rows = r1 - r0;
        m.off_y_ += r0;
        return m;
    }
    void copyTo(Mat &dst) const {
        if (empty()) {
            dst = Mat();
            return;
        }
        dst.create(rows, cols, type_);
        for (int r = 0; r < rows; r++) memcpy(dst.data + dst.step * (size_t) r, data + step * (size_t) r, (size_t) cols * elemSize(type_));
    }
    Mat clone() const {
        Mat m;
        copyTo(m);
        return m;
    }
    // view bookkeeping for the ROI-aware entry points (goodFeaturesToTrack / cornerSubPix on block views)
    const uint8_t *origin() const { return origin_; }
    int offX() const { return off_x_; }
    int offY() const { return off_y_; }
    int fullW() const { return full_w_; }
    int fullH() const { return full_h_; }

protected:
    int type_ = CV_8UC1;
    std::shared_ptr<uint8_t> buf_;
    int off_x_ = 0, off_y_ = 0, full_w_ = 0, full_h_ = 0;
    const uint8_t *origin_ = nullptr;
};

// Mat_<double>(r, c) << a, b, c ...  (row-major fill)
template <typename T> class Mat_ : public Mat {
public:
    Mat_(int r, int c) : Mat(r, c, sizeof(T) == 8 ? CV_64FC1 : CV_32FC1) {}
    struct Init {
        Mat_ m;
        int k;
        Init &operator,(T v) {
            ((T *) (m.data + m.step * (size_t) (k / m.cols)))[k % m.cols] = v;
            k++;
            return *this;
        }
        operator Mat() const { return m; }
    };
    Init operator<<(T v) {
        Init i{*this, 0};
        return (i, v);
    }
};

// ---- entry points (declarations; definitions in ref_tracking.cc forward to oracle/orc_*.cc) ---------------------------------
class CLAHE {
public:
    CLAHE(double clip, Size tiles) : clip_(clip), tiles_(tiles) {}
    void apply(const Mat &src, Mat &dst);

private:
    double clip_;
    Size tiles_;
};
inline Ptr<CLAHE> createCLAHE(double clipLimit, Size tileGridSize) { return std::make_shared<CLAHE>(clipLimit, tileGridSize); }
void cvtColor(const Mat &src, Mat &dst, int code);
void calcHist(const Mat *images, int nimages, const int *channels, const Mat &mask, Mat &hist, int dims, const int *histSize,
              const float **ranges, bool uniform, bool accumulate);
void calcOpticalFlowPyrLK(const Mat &prevImg, const Mat &nextImg, const std::vector<Point2f> &prevPts, std::vector<Point2f> &nextPts,
                          std::vector<uint8_t> &status, std::vector<float> &err, Size winSize, int maxLevel, TermCriteria criteria,
                          int flags);
Mat findFundamentalMat(const std::vector<Point2f> &points1, const std::vector<Point2f> &points2, int method, double ransacReprojThreshold,
                       double confidence, std::vector<uint8_t> &mask);
void goodFeaturesToTrack(const Mat &image, std::vector<Point2f> &corners, int maxCorners, double qualityLevel, double minDistance,
                         const Mat &mask);
void cornerSubPix(const Mat &image, std::vector<Point2f> &corners, Size winSize, Size zeroZone, TermCriteria criteria);
void circle(Mat &img, Point2f center, int radius, const Scalar &color, int thickness);
void undistortPoints(const std::vector<Point2f> &src, std::vector<Point2f> &dst, const Mat &cameraMatrix, const Mat &distCoeffs,
                     const Mat &R, const Mat &P);
inline void initUndistortRectifyMap(const Mat &, const Mat &, const Mat &, const Mat &, Size, int, Mat &, Mat &) {} // image remap unused by the tracker
inline void remap(const Mat &src, Mat &dst, const Mat &, const Mat &, int, int, const Scalar &) { src.copyTo(dst); }

} // namespace cv
