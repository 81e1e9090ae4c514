// cvlite.h — the handful of OpenCV types the ORB front-end's class signatures mention, for containers WITHOUT OpenCV
// (this build container has none).  Same spelling as OpenCV 2.4/3.x so include/ORBextractor.h and include/ORBmatcher.h
// compile unchanged against either; define ORBHIP_USE_OPENCV to use the real headers instead.  Types only — no image
// processing lives here (that is all in liborbhip.so).
#pragma once
#include <cstdint>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_16U 2
#define CV_32F 5

namespace cv {

typedef unsigned char uchar;

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    Point_& operator*=(float s) { x = (T)(x * s); y = (T)(y * s); return *this; }
};
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;
struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
struct Rect { int x, y, width, height; Rect() : x(0), y(0), width(0), height(0) {} Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {} };

class KeyPoint {            // same field order and size (28 B) as cv::KeyPoint == orbhip_keypoint
public:
    Point2f pt; float size, angle, response; int octave, class_id;
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};

class Mat {                 // 8-bit / 16-bit / 32-bit single-channel, ref-counted or borrowed storage
public:
    int rows, cols, flags_type; size_t step; uchar* data;
    Mat() : rows(0), cols(0), flags_type(CV_8U), step(0), data(nullptr) {}
    Mat(int r, int c, int type) : Mat() { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t step_ = 0) : rows(r), cols(c), flags_type(type), step(step_ ? step_ : (size_t)c * elemSize1(type)), data((uchar*)ext) {}
    static size_t elemSize1(int type) { return type == CV_32F ? 4 : type == CV_16U ? 2 : 1; }
    void create(int r, int c, int type)
    {
        if (r == rows && c == cols && type == flags_type && data) return;     // like cv::Mat::create: a fitting buffer (also a view) is kept
        rows = r; cols = c; flags_type = type; step = (size_t)c * elemSize1(type);
        owner_ = std::shared_ptr<uchar>(new uchar[(size_t)r * step + 64], std::default_delete<uchar[]>()); data = owner_.get();
    }
    void release() { owner_.reset(); data = nullptr; rows = cols = 0; step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return flags_type; }
    size_t step1() const { return step / elemSize1(flags_type); }
    bool isContinuous() const { return step == (size_t)cols * elemSize1(flags_type); }
    // cv::Mat::zeros returns a MatExpr: assigning it to a Mat that already has the right size and type (e.g. a rowRange view)
    // fills that buffer IN PLACE — src/ORBextractor.cc:1037 relies on it to write into the caller's descriptor rows
    struct ZerosExpr { int rows, cols, type; };
    static ZerosExpr zeros(int r, int c, int type) { ZerosExpr e = {r, c, type}; return e; }
    Mat(const ZerosExpr& e) : Mat() { *this = e; }
    Mat& operator=(const ZerosExpr& e) { create(e.rows, e.cols, e.type); for (int y = 0; y < rows; y++) memset(data + (size_t)y * step, 0, (size_t)cols * elemSize1(flags_type)); return *this; }
    Mat clone() const { Mat m(rows, cols, flags_type); for (int y = 0; y < rows; y++) memcpy(m.data + y * m.step, data + y * step, cols * elemSize1(flags_type)); return m; }
    Mat(Size sz, int type) : Mat() { create(sz.height, sz.width, type); }
    Size size() const { return Size(cols, rows); }
    // views sharing the storage (ROI semantics of cv::Mat)
    Mat rowRange(int a, int b) const { Mat m = *this; m.rows = b - a; m.data = data + (size_t)a * step; return m; }
    Mat colRange(int a, int b) const { Mat m = *this; m.cols = b - a; m.data = data + (size_t)a * elemSize1(flags_type); return m; }
    Mat operator()(const Rect& r) const { return rowRange(r.y, r.y + r.height).colRange(r.x, r.x + r.width); }
    Mat row(int y) const { Mat m; m.rows = 1; m.cols = cols; m.flags_type = flags_type; m.step = step; m.data = data + (size_t)y * step; m.owner_ = owner_; return m; }
    template <typename T> T* ptr(int y = 0) { return (T*)(data + (size_t)y * step); }
    template <typename T> const T* ptr(int y = 0) const { return (const T*)(data + (size_t)y * step); }
    uchar* ptr(int y = 0) { return data + (size_t)y * step; }
    const uchar* ptr(int y = 0) const { return data + (size_t)y * step; }
    template <typename T> T& at(int y, int x) { return ((T*)(data + (size_t)y * step))[x]; }
    template <typename T> const T& at(int y, int x) const { return ((const T*)(data + (size_t)y * step))[x]; }
#ifdef CVLITE_ALGEBRA
    // ---- small CV_32F matrix algebra.  NOT needed by the drop-in classes: it exists so that the reference's own Frame.cc /
    // ORBmatcher.cc compile for the oracle/_ref builds (pose products, norms, the L1 patch correlation of stereo matching).
    template <typename T> T& at(int i) { return rows == 1 ? ((T*)data)[i] : *(T*)(data + (size_t)i * step); }
    template <typename T> const T& at(int i) const { return rows == 1 ? ((const T*)data)[i] : *(const T*)(data + (size_t)i * step); }
    static Mat ones(int r, int c, int type) { Mat m(r, c, type); for (int y = 0; y < r; y++) for (int x = 0; x < c; x++) { if (type == CV_32F) m.at<float>(y, x) = 1.0f; else m.at<uchar>(y, x) = 1; } return m; }
    static Mat eye(int r, int c, int type) { Mat m = Mat(zeros(r, c, type)); for (int i = 0; i < r && i < c; i++) { if (type == CV_32F) m.at<float>(i, i) = 1.0f; else m.at<uchar>(i, i) = 1; } return m; }
    Mat col(int x) const { return colRange(x, x + 1); }
    Mat t() const { Mat m(cols, rows, flags_type); for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) m.at<float>(x, y) = at<float>(y, x); return m; }
    // (no FMA contraction, whatever the flags of the translation unit: an OpenCV binary is not compiled with its caller's flags - see CVLITE_ALGEBRA below)
#if defined(__GNUC__) && !defined(__clang__)
    __attribute__((optimize("fp-contract=off")))
#endif
    double dot(const Mat& o) const { double s = 0; for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) s += (double)at<float>(y, x) * o.at<float>(y, x); return s; }
    void copyTo(Mat& m) const { m = clone(); }
    void convertTo(Mat& m, int rtype) const
    {
        Mat o(rows, cols, rtype);
        for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) {
            const float v = flags_type == CV_32F ? at<float>(y, x) : (float)at<uchar>(y, x);
            if (rtype == CV_32F) o.at<float>(y, x) = v; else o.at<uchar>(y, x) = (uchar)v;
        }
        m = o;
    }
    Mat reshape(int, int = 0) const;    // only reached for distorted cameras (cv::undistortPoints); defined in oracle/ref_shim
#endif
private:
    std::shared_ptr<uchar> owner_;
};

#ifdef CVLITE_ALGEBRA
enum { NORM_L1 = 2, NORM_L2 = 4 };
// The stand-in's arithmetic carries no FMA contraction whatever the translation unit's flags say: the OpenCV it stands for is a separate binary, not compiled
// with its caller's -march=native (oracle/Makefile ref_native_slam builds the reference's callers with their own flags around it: DESIGN.md H3)
#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC push_options
#pragma GCC optimize ("fp-contract=off")
#endif
#ifndef CVLITE_GEMM_SMALL
// A*B: every element the sum of its products in double, rounded to float - cv::gemm's generic kernel (GEMMSingleMul<float, double>); `A*B + C` then adds in float
inline Mat operator*(const Mat& a, const Mat& b)
{
    Mat m(a.rows, b.cols, CV_32F);
    for (int y = 0; y < a.rows; y++) for (int x = 0; x < b.cols; x++) { double s = 0; for (int k = 0; k < a.cols; k++) s += (double)a.at<float>(y, k) * b.at<float>(k, x); m.at<float>(y, x) = (float)s; }
    return m;
}
#else
// -DCVLITE_GEMM_SMALL: the OTHER way an OpenCV evaluates the reference's `Rcw*p3Dw+tcw`.  In OpenCV 2.4 / 3.x `A*B` is a MatExpr that absorbs a
// following `+ C` (MatOp_GEMM) and cv::gemm has a path for small shapes (inner length 2..4 equal to a side of the result, no transposition flags) that
// accumulates in FLOAT, left to right, and applies alpha / beta in double: d = (float)(t0*alpha + c*beta).  Other shapes: double accumulation and ONE rounding
// of alpha*s + beta*c.  Here: the product is an expression object, `+ Mat` evaluates it fused, a conversion evaluates it alone.  Test infrastructure
// (tests/cpp/test_gemm_probe.cc shows the drop-in's probe telling the two apart); compile with -ffp-contract=off.
#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC push_options
#pragma GCC optimize ("fp-contract=off")          // OpenCV's own binaries carry no FMA contraction (generic x86-64 builds): neither does this stand-in, whatever the flags
#endif
struct MatMul {
    Mat a, b;
    Mat eval(const Mat* c) const
    {
        const int len = a.cols;
        Mat m(a.rows, b.cols, CV_32F);
        const bool small = len >= 2 && len <= 4 && (len == b.cols || len == a.rows);
        for (int y = 0; y < a.rows; y++) for (int x = 0; x < b.cols; x++) {
            if (small) {
                float t0 = a.at<float>(y, 0) * b.at<float>(0, x);
                for (int k = 1; k < len; k++) t0 = t0 + a.at<float>(y, k) * b.at<float>(k, x);
                m.at<float>(y, x) = (float)((double)t0 * 1.0 + (c ? (double)c->at<float>(y, x) : 0.0) * (c ? 1.0 : 0.0));
            } else {
                double s = 0; for (int k = 0; k < len; k++) s += (double)a.at<float>(y, k) * b.at<float>(k, x);
                m.at<float>(y, x) = (float)(c ? s + (double)c->at<float>(y, x) : s);
            }
        }
        return m;
    }
    operator Mat() const { return eval(nullptr); }
};
inline MatMul operator*(const Mat& a, const Mat& b) { MatMul m = {a, b}; return m; }
inline MatMul operator*(const MatMul& p, const Mat& b) { MatMul m = {Mat(p), b}; return m; }
inline Mat operator+(const MatMul& p, const Mat& c) { return p.eval(&c); }
#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC pop_options
#endif
#endif
inline Mat operator*(float f, const Mat& a) { Mat m(a.rows, a.cols, CV_32F); for (int y = 0; y < a.rows; y++) for (int x = 0; x < a.cols; x++) m.at<float>(y, x) = f * a.at<float>(y, x); return m; }
inline Mat operator*(const Mat& a, float f) { return f * a; }
inline Mat operator*(double f, const Mat& a) { return (float)f * a; }
inline Mat operator/(const Mat& a, float f) { Mat m(a.rows, a.cols, CV_32F); for (int y = 0; y < a.rows; y++) for (int x = 0; x < a.cols; x++) m.at<float>(y, x) = a.at<float>(y, x) / f; return m; }
inline Mat operator-(const Mat& a, const Mat& b) { Mat m(a.rows, a.cols, CV_32F); for (int y = 0; y < a.rows; y++) for (int x = 0; x < a.cols; x++) m.at<float>(y, x) = a.at<float>(y, x) - b.at<float>(y, x); return m; }
inline Mat operator+(const Mat& a, const Mat& b) { Mat m(a.rows, a.cols, CV_32F); for (int y = 0; y < a.rows; y++) for (int x = 0; x < a.cols; x++) m.at<float>(y, x) = a.at<float>(y, x) + b.at<float>(y, x); return m; }
inline Mat operator-(const Mat& a) { return -1.0f * a; }
inline double norm(const Mat& a) { return std::sqrt(a.dot(a)); }
inline double norm(const Mat& a, const Mat& b, int type)
{
    double s = 0;
    for (int y = 0; y < a.rows; y++) for (int x = 0; x < a.cols; x++) { const double d = (double)a.at<float>(y, x) - b.at<float>(y, x); s += type == NORM_L1 ? std::fabs(d) : d * d; }
    return type == NORM_L1 ? s : std::sqrt(s);
}
template <typename T> class Mat_ : public Mat {           // (cv::Mat_<float>(3,1) << x, y, z)
public:
    Mat_(int r, int c) : Mat(r, c, CV_32F) {}
    struct Init { Mat m; int i; Init& operator,(T v) { m.at<T>(i / m.cols, i % m.cols) = v; i++; return *this; } operator Mat() const { return m; } };
    Init operator<<(T v) { Init in = {*this, 0}; in, v; return in; }
};
#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC pop_options
#endif
#endif

class _InputArray {
public:
    _InputArray() : m_(nullptr) {}
    _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)) {}
    Mat getMat() const { return m_ ? *m_ : Mat(); }
    bool empty() const { return !m_ || m_->empty(); }
protected:
    Mat* m_;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray() {}
    _OutputArray(Mat& m) : _InputArray(m) {}
    void create(int rows, int cols, int type) const { if (m_) m_->create(rows, cols, type); }
    void release() const { if (m_) m_->release(); }
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

}  // namespace cv
