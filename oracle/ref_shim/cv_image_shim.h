// Stand-ins for the OpenCV functions src/ORBextractor.cc calls, used ONLY to compile the reference's own ORBextractor.cc into
// oracle/_ref/liborbextractor_ref.so.  Types come from include/cvlite/cvlite.h; the four image primitives (resize, GaussianBlur,
// FAST, fastAtan2) are NOT OpenCV here: they forward to the oracle's restatements (oracle/orb_oracle.cpp), so this build pins
// everything the reference file itself does — constructor tables, pyramid sequencing, the per-cell FAST loop with its threshold
// fallback, DistributeOctTree / DivideNode, IC_Angle, the steered BRIEF with the file's own pattern table, the final assembly —
// and leaves exactly the OpenCV primitives "unpinned" (DESIGN.md §3).  Test infrastructure, not product code.
#pragma once
#include <algorithm>
#include <cmath>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include "cvlite/cvlite.h"

#define CV_PI 3.1415926535897932384626433832795
#ifndef CV_8UC1
#define CV_8UC1 0
#endif

extern "C" {   // oracle/orb_oracle.cpp
void orb_oracle_resize(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh);
void orb_oracle_blur(const uint8_t* src, int w, int ht, int stride, uint8_t* dst, int round_mode);
int orb_oracle_fast(const uint8_t* img, int w, int ht, int stride, int threshold, int nms, int* xys, int cap);
float orb_oracle_fastatan2(float y, float x);
}

// cvRound & co: round-half-even like the SSE2 cvtsd2si / cvtss2si OpenCV compiles to
static inline int cvRound(double v) { return (int)lrint(v); }
static inline int cvRound(float v) { return (int)lrintf(v); }
static inline int cvRound(int v) { return v; }
static inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
static inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }

namespace cv {

using std::vector;
enum { BORDER_REFLECT_101 = 4, BORDER_ISOLATED = 16, INTER_LINEAR = 1 };
typedef Size Size2i;

static inline float fastAtan2(float y, float x) { return orb_oracle_fastatan2(y, x); }

static inline void FAST(InputArray image, std::vector<KeyPoint>& keypoints, int threshold, bool nonmaxSuppression)
{
    const Mat m = image.getMat();
    keypoints.clear();
    if (m.cols < 7 || m.rows < 7) return;
    std::vector<int> xys((size_t)m.cols * m.rows * 3 + 3);
    const int n = orb_oracle_fast(m.data, m.cols, m.rows, (int)m.step, threshold, nonmaxSuppression ? 1 : 0, xys.data(), m.cols * m.rows);
    for (int i = 0; i < n; i++) keypoints.push_back(KeyPoint((float)xys[3 * i], (float)xys[3 * i + 1], 7.f, -1, (float)xys[3 * i + 2]));
}

static inline void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sx, double sy, int borderType)
{
    (void)ksize; (void)sx; (void)sy; (void)borderType;          // the reference's only call: 7x7, sigma 2, BORDER_REFLECT_101 (ORBextractor.cc:1086)
    const Mat s = src.getMat();
    std::vector<uint8_t> out((size_t)s.cols * s.rows);
    // ORB_REF_BLUR_ROUND_MODE selects which real-world cv::GaussianBlur this stands for: 0 = generic C++ (default), 1 = x86 SSE2 build
    const char* brm = getenv("ORB_REF_BLUR_ROUND_MODE");
    orb_oracle_blur(s.data, s.cols, s.rows, (int)s.step, out.data(), brm && atoi(brm) ? 1 : 0);
    dst.create(s.rows, s.cols, s.type());
    Mat d = dst.getMat();
    for (int y = 0; y < s.rows; y++) memcpy(d.ptr(y), &out[(size_t)y * s.cols], s.cols);
}

static inline void resize(InputArray src, OutputArray dst, Size dsize, double fx, double fy, int interpolation)
{
    (void)fx; (void)fy; (void)interpolation;                     // INTER_LINEAR to an explicit size (ORBextractor.cc:1120)
    const Mat s = src.getMat();
    std::vector<uint8_t> out((size_t)dsize.width * dsize.height);
    orb_oracle_resize(s.data, s.cols, s.rows, (int)s.step, out.data(), dsize.width, dsize.height);
    dst.create(dsize.height, dsize.width, s.type());            // the reference passes a view of the right size: written in place
    Mat d = dst.getMat();
    for (int y = 0; y < dsize.height; y++) memcpy(d.ptr(y), &out[(size_t)y * dsize.width], dsize.width);
}

static inline int reflect101(int p, int len) { if (len == 1) return 0; while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * (len - 1) - p; } return p; }
static inline void copyMakeBorder(InputArray src, OutputArray dst, int top, int bottom, int left, int right, int borderType)
{
    (void)borderType;                                            // BORDER_REFLECT_101 (+ BORDER_ISOLATED), ORBextractor.cc:1122-1128
    const Mat s = src.getMat();
    Mat tmp(s.rows + top + bottom, s.cols + left + right, s.type());
    for (int y = 0; y < tmp.rows; y++) {
        const uint8_t* row = s.ptr(reflect101(y - top, s.rows));
        uint8_t* o = tmp.ptr(y);
        for (int x = 0; x < tmp.cols; x++) o[x] = row[reflect101(x - left, s.cols)];
    }
    dst.create(tmp.rows, tmp.cols, s.type());
    Mat d = dst.getMat();
    for (int y = 0; y < tmp.rows; y++) memcpy(d.ptr(y), tmp.ptr(y), tmp.cols);
}

#ifdef CVLITE_ALGEBRA
// Distorted cameras (Frame::UndistortKeyPoints / ComputeImageBounds, Frame.cc:404-464, take this branch when mDistCoef[0] != 0):
// cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) forwards to the oracle's restatement.  cvlite has no channels, so
// reshape(2) / reshape(1) return the same N x 2 float matrix and the points are read as its rows.
extern "C" void orb_oracle_undistort_points(const float* K4, const float* D5, const float* in, int n, float* out);
static inline void undistortPoints(InputArray src, OutputArray dst, InputArray cameraMatrix, InputArray distCoeffs, InputArray, InputArray)
{
    const Mat s = src.getMat(), K = cameraMatrix.getMat(), D = distCoeffs.getMat();
    if (s.cols != 2 || s.type() != CV_32F || K.type() != CV_32F || D.type() != CV_32F) { fprintf(stderr, "undistortPoints shim: N x 2 CV_32F points, CV_32F K and D expected\n"); abort(); }
    const float K4[4] = {K.at<float>(0, 0), K.at<float>(1, 1), K.at<float>(0, 2), K.at<float>(1, 2)};
    float D5[5] = {0, 0, 0, 0, 0};
    const int nd = D.rows * D.cols;
    for (int i = 0; i < nd && i < 5; i++) D5[i] = D.rows == 1 ? D.at<float>(0, i) : D.at<float>(i, 0);
    std::vector<float> in((size_t)s.rows * 2), out((size_t)s.rows * 2);
    for (int i = 0; i < s.rows; i++) { in[2 * i] = s.at<float>(i, 0); in[2 * i + 1] = s.at<float>(i, 1); }
    orb_oracle_undistort_points(K4, D5, in.data(), s.rows, out.data());
    dst.create(s.rows, 2, CV_32F);
    Mat d = dst.getMat();
    for (int i = 0; i < s.rows; i++) { d.at<float>(i, 0) = out[2 * i]; d.at<float>(i, 1) = out[2 * i + 1]; }
}
inline Mat Mat::reshape(int, int) const { return *this; }
#endif

struct KeyPointsFilter {                                         // only used by the dead ComputeKeyPointsOld (ORBextractor.cc:1006,1024)
    static void retainBest(std::vector<KeyPoint>& k, int n)
    {
        if (n >= 0 && (int)k.size() > n) {
            std::stable_sort(k.begin(), k.end(), [](const KeyPoint& a, const KeyPoint& b) { return a.response > b.response; });
            k.resize(n);
        }
    }
};

}  // namespace cv
