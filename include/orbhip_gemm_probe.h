// orbhip_gemm_probe.h — how does the cv::Mat this translation unit is compiled against round `R*x + t` (3x3 * 3x1 + 3x1, CV_32F)?
//
// The five projection-guided members of ORB_SLAM2::ORBmatcher evaluate `cv::Mat p3Dc = Rcw*p3Dw+tcw;` per map point (ORBmatcher.cc:320, 855, 1009,
// 1166-1167, 1246-1247, 1361, 1495).  The drop-in (orb_slam2_amd/cpp/ORBmatcher.cc) moves that statement onto the device, so it has to know which
// arithmetic the linked OpenCV performs for it - a property of the OpenCV BUILD, not of the reference:
//   mode 0   cv::gemm's generic kernel: each row's three products summed in double, rounded to float; the addition of t in float
//            (include/cvlite; OpenCV builds that do not fuse the addition into the product);
//   mode 1   the small-matrix path of cv::gemm (OpenCV 2.4 / 3.x matmul.cpp, inner length 2..4, flags 0), taken because `A*B + C` is ONE MatExpr:
//            t0 = a0*b0 + a1*b1 + a2*b2 in float, d = (float)(t0*alpha + c*beta) in double;
//   mode 2   neither: the device must not guess - the caller evaluates its own cv::Mat expression per point and hands over camera-frame coordinates.
// orbhip_probe_gemm_mode<cv::Mat>() evaluates 1024 pseudo-random triples with the linked cv::Mat and with both flat forms and returns the mode that
// reproduces every bit of every result (0 if both do: they differ on about one triple in three).  Header-only so that tests/cpp/test_gemm_probe.cc
// can instantiate it against differently behaving cv::Mat stand-ins.
#ifndef ORBHIP_GEMM_PROBE_H
#define ORBHIP_GEMM_PROBE_H
#include <cstdlib>
#include <cstring>

// the flat forms must round every operation once, whatever flags this file is compiled with (the reference's CMakeLists asks for -O3 -march=native,
// under which gcc contracts a*b + c into an fma)
#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC push_options
#pragma GCC optimize ("fp-contract=off")
#elif defined(__clang__)
#pragma clang fp contract(off)
#endif
static inline void orbhip_flat_gemm(int mode, const float* R, const float* x, const float* t, float* out)
{
    for (int r = 0; r < 3; r++) {
        if (mode == 1) {
            const float t0 = R[3 * r] * x[0] + R[3 * r + 1] * x[1] + R[3 * r + 2] * x[2];
            out[r] = (float)((double)t0 * 1.0 + (double)t[r] * 1.0);
        } else {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)R[3 * r + k] * (double)x[k];
            out[r] = (float)s + t[r];
        }
    }
}
#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC pop_options
#endif

template <class Mat> static inline int orbhip_probe_gemm_mode(int type_32f)
{
    if (const char* e = getenv("ORBHIP_GEMM_MODE")) { const int m = atoi(e); if (m >= 0 && m <= 2) return m; }
    bool ok[2] = {true, true};
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&](float scale) { s = s * 6364136223846793005ull + 1442695040888963407ull; return scale * ((float)((s >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f); };
    for (int it = 0; it < 1024 && (ok[0] || ok[1]); it++) {
        Mat R(3, 3, type_32f), x(3, 1, type_32f), t(3, 1, type_32f);
        float fR[9], fx[3], ft[3];
        for (int i = 0; i < 9; i++) R.template at<float>(i / 3, i % 3) = fR[i] = rnd(1.0f);
        for (int i = 0; i < 3; i++) { x.template at<float>(i, 0) = fx[i] = rnd(it & 1 ? 40.0f : 3.0f); t.template at<float>(i, 0) = ft[i] = rnd(it & 2 ? 5.0f : 0.05f); }
        const Mat y = R * x + t;                              // the members' own expression form
        for (int m = 0; m < 2; m++) {
            float f[3]; orbhip_flat_gemm(m, fR, fx, ft, f);
            for (int i = 0; i < 3; i++) if (memcmp(&f[i], &y.template at<float>(i, 0), 4)) ok[m] = false;
        }
    }
    return ok[0] ? 0 : ok[1] ? 1 : 2;
}
#endif
