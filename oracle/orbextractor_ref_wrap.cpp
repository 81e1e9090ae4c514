// C API around the REFERENCE's own src/ORBextractor.cc (compiled where it lies under /root/reference by oracle/Makefile into
// oracle/_ref/liborbextractor_ref.so).  The OpenCV image primitives it calls are the oracle's restatements
// (oracle/ref_shim/cv_image_shim.h); everything else — the whole extractor logic — is the reference's code.
//
// std::list node addresses: DistributeOctTree sorts (size, ExtractorNode*) pairs (ORBextractor.cc:684), i.e. equal sizes are
// ordered by heap address.  Here the list nodes come from a bump arena that never reuses memory, so a later push_front has a
// higher address: under that allocator the reference's order IS the canonical tie-break H1 (later-created node first).
// Test infrastructure only.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include "ORBextractor.h"

// -DORBEXTRACTOR_REF_STOCK_ALLOCATOR (`make -C oracle ref_stock`): no arena - the list nodes come from glibc's malloc, as in a maintainer's own binary.
// tests/test_reference_stock_allocator.py measures what H1 costs a drop-in vs stock-binary A/B.
#ifndef ORBEXTRACTOR_REF_STOCK_ALLOCATOR
namespace {
const size_t kNodeBytes = sizeof(std::_List_node<ORB_SLAM2::ExtractorNode>);
const size_t kArenaBytes = (size_t)256 << 20;
char* g_arena = nullptr; size_t g_used = 0;
inline bool in_arena(void* p) { return g_arena && (char*)p >= g_arena && (char*)p < g_arena + kArenaBytes; }
}
void* operator new(size_t n)
{
    if (n == kNodeBytes) {
        if (!g_arena) g_arena = (char*)malloc(kArenaBytes);
        const size_t a = (n + 15) & ~(size_t)15;
        if (g_arena && g_used + a <= kArenaBytes) { void* p = g_arena + g_used; g_used += a; return p; }
    }
    void* p = malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void operator delete(void* p) noexcept { if (p && !in_arena(p)) free(p); }
void operator delete(void* p, size_t) noexcept { if (p && !in_arena(p)) free(p); }
#else
namespace { size_t g_used = 0; }
#endif

struct RefExtractor : ORB_SLAM2::ORBextractor {
    RefExtractor(int n, float s, int l, int i, int m) : ORB_SLAM2::ORBextractor(n, s, l, i, m) {}
    const std::vector<int>& featuresPerLevel() const { return mnFeaturesPerLevel; }
    const std::vector<int>& uMax() const { return umax; }
};

extern "C" {

void* orbextractor_ref_new(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
{
    return new RefExtractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
}
void orbextractor_ref_delete(void* h) { delete (RefExtractor*)h; }
void orbextractor_ref_params(void* h, int* featuresPerLevel, float* scaleFactors, float* invScaleFactors, float* sigma2, float* invSigma2, int* umax16)
{
    RefExtractor* e = (RefExtractor*)h;
    const int L = e->GetLevels();
    std::vector<float> a = e->GetScaleFactors(), b = e->GetInverseScaleFactors(), c = e->GetScaleSigmaSquares(), d = e->GetInverseScaleSigmaSquares();
    for (int i = 0; i < L; i++) { featuresPerLevel[i] = e->featuresPerLevel()[i]; scaleFactors[i] = a[i]; invScaleFactors[i] = b[i]; sigma2[i] = c[i]; invSigma2[i] = d[i]; }
    for (int i = 0; i < 16; i++) umax16[i] = e->uMax()[i];
}
// ORBextractor::operator() (ORBextractor.cc:1043-1105)
int orbextractor_ref_extract(void* h, const uint8_t* img, int w, int ht, int stride, void* kps, uint8_t* desc, int cap)
{
    g_used = 0;                                   // list nodes of the previous call are gone (lNodes is local to DistributeOctTree)
    RefExtractor* e = (RefExtractor*)h;
    cv::Mat im(ht, w, CV_8UC1, (void*)img, (size_t)stride);
    std::vector<cv::KeyPoint> k; cv::Mat d;
    (*e)(im, cv::Mat(), k, d);
    const int n = (int)k.size(), m = n < cap ? n : cap;
    static_assert(sizeof(cv::KeyPoint) == 28, "KeyPoint layout");
    if (m > 0) { memcpy(kps, &k[0], (size_t)m * 28); for (int i = 0; i < m; i++) memcpy(desc + (size_t)i * 32, d.ptr(i), 32); }
    return n;
}
int orbextractor_ref_level(void* h, int level, uint8_t* dst, int* w, int* ht)
{
    RefExtractor* e = (RefExtractor*)h;
    if (level < 0 || level >= (int)e->mvImagePyramid.size() || e->mvImagePyramid[level].empty()) return 0;
    const cv::Mat& m = e->mvImagePyramid[level];
    *w = m.cols; *ht = m.rows;
    if (dst) for (int y = 0; y < m.rows; y++) memcpy(dst + (size_t)y * m.cols, m.ptr(y), m.cols);
    return 1;
}

}  // extern "C"
