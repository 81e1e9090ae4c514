// The drop-in matcher's probe (include/orbhip_gemm_probe.h) against three differently behaving cv::Mat stand-ins: include/cvlite as it is
// (cv::gemm's generic kernel: mode 0), with -DCVLITE_GEMM_SMALL (OpenCV's small-matrix path with the fused addition: mode 1), and with -DPROBE_ODD_ALGEBRA
// (a product nobody restates - long double accumulation, the sum rounded TWICE: the probe must refuse both forms: mode 2).  Prints "gemm mode <m>".
#define CVLITE_ALGEBRA
#include "cvlite/cvlite.h"
#ifdef PROBE_ODD_ALGEBRA
namespace odd {
struct Mat : cv::Mat { Mat(int r, int c, int t) : cv::Mat(r, c, t) {} Mat(const cv::Mat& m) : cv::Mat(m) {} };
inline Mat operator*(const Mat& a, const Mat& b)
{
    cv::Mat m(a.rows, b.cols, CV_32F);
    for (int y = 0; y < a.rows; y++) for (int x = 0; x < b.cols; x++) {
        float s = 0; for (int k = a.cols - 1; k >= 0; k--) s += a.at<float>(y, k) * b.at<float>(k, x);      // float, summed BACKWARDS
        m.at<float>(y, x) = s;
    }
    return Mat(m);
}
inline Mat operator+(const Mat& a, const Mat& b) { return Mat(static_cast<const cv::Mat&>(a) + static_cast<const cv::Mat&>(b)); }
}
typedef odd::Mat ProbedMat;
#else
typedef cv::Mat ProbedMat;
#endif
#include "orbhip_gemm_probe.h"
#include <cstdio>
int main() { printf("gemm mode %d\n", orbhip_probe_gemm_mode<ProbedMat>(CV_32F)); return 0; }
