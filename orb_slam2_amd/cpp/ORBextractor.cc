// ORBextractor.cc — host side of the drop-in ORB_SLAM2::ORBextractor: forwards to the C ABI (include/orbhip.h).
// Replaces src/ORBextractor.cc of the reference; constructor bookkeeping follows ORBextractor.cc:410-446 so the accessors
// return the same tables even before the first image has been seen.
#include "ORBextractor.h"
#include "orbhip.h"

#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>

namespace ORB_SLAM2
{

static_assert(sizeof(cv::KeyPoint) == sizeof(orbhip_keypoint), "cv::KeyPoint and orbhip_keypoint must share one 28-byte layout");

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST),
      mbHasCamera(false), mnRawCols(0), mnRawRows(0), mnLastN(0), mnStageCap(0), mbPairResults(false),
      mpCtx(NULL), mnCtxW(0), mnCtxH(0), mnCtxBatch(0), mnCtxDevice(0), mnDevice(0),
#if defined(__SSE2__) || defined(_M_X64)
      mnBlurRounding(1),          // an x86-64 OpenCV (<= 3.3) runs the SSE2 column filter: round-half-even on 4-column groups
#else
      mnBlurRounding(0),
#endif
#if defined(__FMA__)
      mnFpContract(1),            // this translation unit is compiled with FMA code generation (the reference's -march=native): gcc would have fused the rotation
#else
      mnFpContract(0),
#endif
      mnSettings(0), mnStamp(0), mvTicketSizes(4, 0), mbDownloadPyramid(false), mnPendingTickets(0), mbFrameState(false), mfScaleFactorArg(_scaleFactor),
      mnBoundFrame(0), mbBound(false), mbStereoColumns(false), mnOwnerThread(0)
{
    memset(mCamera, 0, sizeof mCamera);
    mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) { mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor); mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i]; }
    mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) { mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i]; mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i]; }
    mvImagePyramid.resize(nlevels); mvImagePyramid.mpOwner = this;
    mnFeaturesPerLevel.resize(nlevels);
    if (const char* dev = getenv("ORBHIP_DEVICE")) mnDevice = atoi(dev);
    if (const char* br = getenv("ORBHIP_BLUR_ROUNDING")) mnBlurRounding = atoi(br) ? 1 : 0;      // deployment override of the host default (see SetBlurRounding)
    if (const char* fc = getenv("ORBHIP_FP_CONTRACT")) mnFpContract = atoi(fc) ? 1 : 0;
}

ORBextractor::~ORBextractor()
{
    if (mpCtx) orbhip_destroy(mpCtx);
    for (size_t i = 0; i < mvCtxCache.size(); i++) orbhip_destroy(mvCtxCache[i].ctx);
}

void ORBextractor::Fail(const char* where) const
{
    throw ORBhipError(std::string(where) + ": " + orbhip_last_error());
}

void ORBextractor::RequireFrameState(const char* where) const
{
    if (!mbFrameState)
        throw ORBhipError(std::string(where) + ": only valid after a single-image call (operator(), ExtractColor, ExtractRectified); after Submit / Collect the "
                          "context holds a batch — use the results Collect returned");
}

void ORBextractor::SetBlurRounding(int mode)
{
    mnBlurRounding = mode ? 1 : 0; mnSettings++;
    if (mpCtx && orbhip_set_blur_rounding(mpCtx, mnBlurRounding) != ORBHIP_OK) Fail("ORBextractor::SetBlurRounding");
}

void ORBextractor::SetFpContract(int mode)
{
    mnFpContract = mode ? 1 : 0; mnSettings++;
    if (mpCtx && orbhip_set_fp_contract(mpCtx, mnFpContract) != ORBHIP_OK) Fail("ORBextractor::SetFpContract");
}

void ORBimagePyramid::Refresh()
{
    std::lock_guard<std::mutex> lock(mMutex);
    if (!mbStale || !mpOwner) return;
    mpOwner->FetchPyramid(mvLevels);
    mbStale = false;
}

void ORBextractor::FetchPyramid(std::vector<cv::Mat>& levels)
{
    if (!mpCtx) return;
    RequireFrameState("ORBextractor::mvImagePyramid");
    std::vector<uint8_t*> dst(nlevels); std::vector<int> stride(nlevels);
    for (int l = 0; l < nlevels; l++) {
        int w = 0, h = 0; orbhip_level_size(mpCtx, l, &w, &h);
        levels[l].create(h, w, CV_8UC1);
        dst[l] = levels[l].data; stride[l] = (int)levels[l].step;
    }
    if (orbhip_pyramid_fetch_all(mpCtx, 0, &dst[0], &stride[0]) != ORBHIP_OK) Fail("ORBextractor::mvImagePyramid");
}

void ORBextractor::ApplySettings(orbhip_ctx* ctx)
{
    if (orbhip_set_blur_rounding(ctx, mnBlurRounding) != ORBHIP_OK || orbhip_set_fp_contract(ctx, mnFpContract) != ORBHIP_OK) Fail("ORBextractor");
    orbhip_camera cam; memcpy(&cam, mCamera, sizeof cam);
    if (orbhip_set_camera(ctx, mbHasCamera ? &cam : NULL) != ORBHIP_OK) Fail("ORBextractor");
}

void ORBextractor::EnsureContext(int width, int height, int maxBatch)
{
    if (mpCtx && width == mnCtxW && height == mnCtxH && maxBatch <= mnCtxBatch && mnCtxDevice == mnDevice) return;      // (SetDevice takes effect here)
    if (mpCtx && mnPendingTickets > 0)
        throw ORBhipError("ORBextractor: another device context is needed (image size or batch changed) while Submit()ed batches are still in flight: Collect them first");
    // the current context steps aside (its size may come back: a rig of two resolutions, a caller that crops) ...
    if (mpCtx) {
        // ... unless nobody could ever take it back: the same size with a batch that only grew (every later call needs at least the new batch), or a
        // context that carries rectification maps (such contexts are never looked up below)
        const bool reusable = !(width == mnCtxW && height == mnCtxH && mnCtxDevice == mnDevice) && mvMapX.empty();
        if (reusable) {
            CtxSlot keep = {mpCtx, mnCtxW, mnCtxH, mnCtxBatch, mnCtxDevice, mnSettings, ++mnStamp};
            mvCtxCache.push_back(keep);
        } else orbhip_destroy(mpCtx);
        mpCtx = NULL;
        if (mvCtxCache.size() > 3) {                                  // four contexts alive at most: the least recently used one goes
            size_t lru = 0;
            for (size_t i = 1; i < mvCtxCache.size(); i++) if (mvCtxCache[i].stamp < mvCtxCache[lru].stamp) lru = i;
            orbhip_destroy(mvCtxCache[lru].ctx); mvCtxCache.erase(mvCtxCache.begin() + lru);
        }
    }
    mbFrameState = false; mbBound = false; mbStereoColumns = false; mbPairResults = false; mnLastN = 0;      // whatever follows belongs to another context
    // ... and one that was laid out for this size ON THIS DEVICE comes back (rectification maps belong to one size: such contexts are never shared)
    for (size_t i = 0; i < mvCtxCache.size() && mvMapX.empty(); i++) {
        const CtxSlot c = mvCtxCache[i];
        if (c.w != width || c.h != height || c.batch < maxBatch || c.device != mnDevice) continue;
        mvCtxCache.erase(mvCtxCache.begin() + i);
        mpCtx = c.ctx; mnCtxW = c.w; mnCtxH = c.h; mnCtxBatch = c.batch; mnCtxDevice = c.device;
        if (c.settings != mnSettings) ApplySettings(mpCtx);
        orbhip_get_scale_tables(mpCtx, &mvScaleFactor[0], &mvInvScaleFactor[0], &mvLevelSigma2[0], &mvInvLevelSigma2[0], &mnFeaturesPerLevel[0]);
        return;
    }
    orbhip_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.nfeatures = nfeatures; cfg.scale_factor = mfScaleFactorArg; cfg.nlevels = nlevels; cfg.ini_th_fast = iniThFAST; cfg.min_th_fast = minThFAST;
    cfg.width = width; cfg.height = height; cfg.max_batch = maxBatch; cfg.device = mnDevice; cfg.stream = NULL; cfg.blur_round_mode = mnBlurRounding; cfg.num_streams = 1;
    // the reference cannot fail here; a missing GPU or an unsupported geometry must not silently produce empty frames
    if (orbhip_create(&mpCtx, &cfg) != ORBHIP_OK) { mpCtx = NULL; Fail("ORBextractor"); }
    mnCtxW = width; mnCtxH = height; mnCtxBatch = maxBatch; mnCtxDevice = mnDevice;
    if (orbhip_set_fp_contract(mpCtx, mnFpContract) != ORBHIP_OK) Fail("ORBextractor");
    if (mbHasCamera) {
        orbhip_camera cam; memcpy(&cam, mCamera, sizeof cam);
        if (orbhip_set_camera(mpCtx, &cam) != ORBHIP_OK) Fail("ORBextractor");
    }
    if (!mvMapX.empty() && (size_t)width * height == mvMapX.size() &&
        orbhip_set_rectification(mpCtx, &mvMapX[0], &mvMapY[0], mnRawCols, mnRawRows) != ORBHIP_OK) Fail("ORBextractor");
    // the device tables are the authority (bit-identical to the constructor's by construction; checked in tests)
    orbhip_get_scale_tables(mpCtx, &mvScaleFactor[0], &mvInvScaleFactor[0], &mvLevelSigma2[0], &mvInvLevelSigma2[0], &mnFeaturesPerLevel[0]);
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors)
{
    if (_image.empty()) return;                                  // ORBextractor.cc:1046-1047
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    EnsureContext(image.cols, image.rows);

    ReserveStage(1);
    int n = 0;
    if (orbhip_extract(mpCtx, image.data, (int)image.step, reinterpret_cast<orbhip_keypoint*>(&mvKpStage[0]), &mvDescStage[0], mnStageCap, &n) != ORBHIP_OK) Fail("ORBextractor");
    Deliver(n, 0, _keypoints, _descriptors);
}

void ORBextractor::ReserveStage(int slots)
{
    mnStageCap = orbhip_keypoint_capacity(mpCtx);
    const size_t need = (size_t)slots * mnStageCap;
    if (mvKpStage.size() < need * sizeof(orbhip_keypoint)) { mvKpStage.resize(need * sizeof(orbhip_keypoint)); mvDescStage.resize(need * 32); }
}

void ORBextractor::ExtractStereo(cv::InputArray _imLeft, cv::InputArray _imRight, std::vector<cv::KeyPoint>& keysLeft, cv::OutputArray descLeft,
                                 std::vector<cv::KeyPoint>& keysRight, cv::OutputArray descRight, float mbf, float mb)
{
    if (_imLeft.empty() || _imRight.empty()) {                   // the reference's two operator() calls return silently on an empty image (ORBextractor.cc:1046-1047)
        mbPairResults = false;
        if (!_imLeft.empty()) (*this)(_imLeft, cv::Mat(), keysLeft, descLeft);
        return;
    }
    mbPairResults = false;                                       // (set again at the very end: a throw below must not leave the pair before's columns behind)
    cv::Mat L = _imLeft.getMat(), R = _imRight.getMat();
    assert(L.type() == CV_8UC1 && R.type() == CV_8UC1 && L.cols == R.cols && L.rows == R.rows);
    if (L.step != R.step) throw ORBhipError("ORBextractor::ExtractStereo: the two images must share their row step");
    EnsureContext(L.cols, L.rows, 2);
    ReserveStage(2);
    mvPairURight.resize(mnStageCap); mvPairDepth.resize(mnStageCap);
    int n[2] = {0, 0};
    if (orbhip_extract_stereo(mpCtx, L.data, R.data, (int)L.step, reinterpret_cast<orbhip_keypoint*>(&mvKpStage[0]), &mvDescStage[0], mnStageCap, n, mbf, mb,
                              &mvPairURight[0], &mvPairDepth[0]) != ORBHIP_OK) Fail("ORBextractor::ExtractStereo");
    Deliver(n[1], 1, keysRight, descRight);
    Deliver(n[0], 0, keysLeft, descLeft);                        // last: the extractor's frame state is the LEFT image's
    mbPairResults = true; mbStereoColumns = true;
}

void ORBextractor::ExtractColor(const unsigned char* data, int step, int cols, int rows, int channels, bool bRGB,
                                std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors)
{
    if (!data || cols <= 0 || rows <= 0) return;
    EnsureContext(cols, rows);
    ReserveStage(1);
    int n = 0;
    const uint8_t* imgs[1] = {data};
    if (orbhip_extract_batch_color(mpCtx, 1, imgs, step, channels, bRGB ? 1 : 0, reinterpret_cast<orbhip_keypoint*>(&mvKpStage[0]), &mvDescStage[0], mnStageCap, &n) != ORBHIP_OK) Fail("ORBextractor");
    Deliver(n, 0, _keypoints, _descriptors);
}

static_assert(sizeof(orbhip_camera) == 9 * sizeof(float), "orbhip_camera is nine floats");

void ORBextractor::SetCamera(const cv::Mat& K, const cv::Mat& distCoef)
{
    {   // the same camera again (a caller that attaches it per frame): nothing to do
        float c[9] = {K.at<float>(0, 0), K.at<float>(1, 1), K.at<float>(0, 2), K.at<float>(1, 2), 0, 0, 0, 0, 0};
        const int ndc = distCoef.rows * distCoef.cols;
        for (int i = 0; i < 5; i++) c[4 + i] = i < ndc ? (distCoef.rows == 1 ? distCoef.at<float>(0, i) : distCoef.at<float>(i, 0)) : 0.0f;
        if (mbHasCamera && !memcmp(c, mCamera, sizeof c)) return;
    }
    mnSettings++;
    mCamera[0] = K.at<float>(0, 0); mCamera[1] = K.at<float>(1, 1); mCamera[2] = K.at<float>(0, 2); mCamera[3] = K.at<float>(1, 2);
    const int nd = distCoef.rows * distCoef.cols;                 // 4x1, or 5x1 when Camera.k3 != 0 (Tracking.cc:70-82)
    for (int i = 0; i < 5; i++) mCamera[4 + i] = i < nd ? (distCoef.rows == 1 ? distCoef.at<float>(0, i) : distCoef.at<float>(i, 0)) : 0.0f;
    mbHasCamera = true;
    if (mpCtx) {
        orbhip_camera cam; memcpy(&cam, mCamera, sizeof cam);
        if (orbhip_set_camera(mpCtx, &cam) != ORBHIP_OK) Fail("ORBextractor");
    }
}

void ORBextractor::UndistortKeyPoints(std::vector<cv::KeyPoint>& mvKeysUn)
{
    if (mpCtx) RequireFrameState("ORBextractor::UndistortKeyPoints");
    if (mnLastN == 0 || !mpCtx) { mvKeysUn.clear(); return; }
    if (!(mbHasCamera && mCamera[4] != 0.0f)) {                                       // if(mDistCoef.at<float>(0)==0.0) mvKeysUn=mvKeys (Frame.cc:406-410): no device trip
        const cv::KeyPoint* src = reinterpret_cast<const cv::KeyPoint*>(&mvKpStage[0]);
        mvKeysUn.assign(src, src + mnLastN);
        return;
    }
    mvKeysUn.resize(mnLastN);
    if (orbhip_fetch_undistorted(mpCtx, 1, reinterpret_cast<orbhip_keypoint*>(&mvKeysUn[0]), mnLastN) != ORBHIP_OK) Fail("ORBextractor::UndistortKeyPoints");
}

void ORBextractor::ComputeImageBounds(int cols, int rows, float& mnMinX, float& mnMaxX, float& mnMinY, float& mnMaxY)
{
    orbhip_bounds b = {0.0f, 0.0f, (float)cols, (float)rows};     // Frame.cc:455-463
    if (mbHasCamera) {
        orbhip_camera cam; memcpy(&cam, mCamera, sizeof cam);
        if (orbhip_image_bounds(mnDevice, &cam, cols, rows, &b) != ORBHIP_OK) Fail("ORBextractor::ComputeImageBounds");
    }
    mnMinX = b.min_x; mnMaxX = b.max_x; mnMinY = b.min_y; mnMaxY = b.max_y;
}

void ORBextractor::ComputeStereoFromRGBD(const cv::Mat& imDepth, float depthFactor, float mbf, int N, std::vector<float>& mvuRight, std::vector<float>& mvDepth)
{
    mvuRight = std::vector<float>(N, -1);                        // Frame.cc:645-646
    mvDepth = std::vector<float>(N, -1);
    if (N == 0 || !mpCtx) return;
    RequireFrameState("ORBextractor::ComputeStereoFromRGBD");
    assert(imDepth.type() == CV_32F || imDepth.type() == CV_16U);
    const void* maps[1] = {imDepth.data};
    if (orbhip_compute_stereo_from_rgbd(mpCtx, 1, maps, (int)imDepth.step, imDepth.type() == CV_32F ? 0 : 1, depthFactor, mbf, &mvuRight[0], &mvDepth[0], N) != ORBHIP_OK) Fail("ORBextractor::ComputeStereoFromRGBD");
    mbStereoColumns = true;
}

void ORBextractor::SetStereoColumns(const std::vector<float>& mvuRight)
{
    if (!mpCtx || !mbFrameState || (int)mvuRight.size() != mnLastN) return;                 // not this extractor's frame any more: the searches take the host path (HoldsFrame / HoldsStereoColumns)
    if (orbhip_set_stereo_columns(mpCtx, 0, mvuRight.empty() ? NULL : &mvuRight[0], mnLastN) != ORBHIP_OK) Fail("ORBextractor::SetStereoColumns");
    mbStereoColumns = true;
}

void ORBextractor::SetRectification(const cv::Mat& M1, const cv::Mat& M2, int rawCols, int rawRows)
{
    assert(M1.type() == CV_32F && M2.type() == CV_32F && M1.rows == M2.rows && M1.cols == M2.cols);
    mvMapX.resize((size_t)M1.rows * M1.cols); mvMapY.resize(mvMapX.size());
    for (int y = 0; y < M1.rows; y++) {
        memcpy(&mvMapX[(size_t)y * M1.cols], M1.ptr<float>(y), sizeof(float) * M1.cols);
        memcpy(&mvMapY[(size_t)y * M1.cols], M2.ptr<float>(y), sizeof(float) * M1.cols);
    }
    mnRawCols = rawCols; mnRawRows = rawRows;
    if (mpCtx) { orbhip_destroy(mpCtx); mpCtx = NULL; }            // the context of the rectified size is re-created with the maps
    for (size_t i = 0; i < mvCtxCache.size(); i++) orbhip_destroy(mvCtxCache[i].ctx);
    mvCtxCache.clear();
    EnsureContext(M1.cols, M1.rows);
}

void ORBextractor::ExtractRectified(const cv::Mat& raw, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors)
{
    if (raw.empty() || !mpCtx || mvMapX.empty()) return;
    assert(raw.type() == CV_8UC1 && raw.cols == mnRawCols && raw.rows == mnRawRows);
    ReserveStage(1);
    int n = 0;
    const uint8_t* imgs[1] = {raw.data};
    if (orbhip_extract_batch_rectify(mpCtx, 1, imgs, (int)raw.step, reinterpret_cast<orbhip_keypoint*>(&mvKpStage[0]), &mvDescStage[0], mnStageCap, &n) != ORBHIP_OK) Fail("ORBextractor");
    Deliver(n, 0, _keypoints, _descriptors);
}

void ORBextractor::Deliver(int n, int slot, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors)
{
    if (slot == 0) { mnLastN = n; mbFrameState = true; mbBound = false; mbStereoColumns = false; mbPairResults = false; }
    // one copy of exactly n records each (the staging block belongs to the object: nothing is allocated or value-initialised per call)
    const cv::KeyPoint* src = reinterpret_cast<const cv::KeyPoint*>(&mvKpStage[(size_t)slot * mnStageCap * sizeof(orbhip_keypoint)]);
    _keypoints.assign(src, src + n);
    if (n == 0) _descriptors.release();                          // ORBextractor.cc:1064-1065
    else {
        _descriptors.create(n, 32, CV_8U);                       // :1068
        cv::Mat d = _descriptors.getMat();
        const unsigned char* ds = &mvDescStage[(size_t)slot * mnStageCap * 32];
        if ((size_t)d.step == 32) memcpy(d.data, ds, (size_t)n * 32);
        else for (int i = 0; i < n; i++) memcpy(d.ptr(i), ds + (size_t)i * 32, 32);
    }
    if (slot != 0) return;
    {   // the planes stay in HBM until somebody indexes mvImagePyramid (only the reference's own ComputeStereoMatches does)
        std::lock_guard<std::mutex> lock(mvImagePyramid.mMutex);
        mvImagePyramid.mbStale = true;
    }
    if (mbDownloadPyramid) mvImagePyramid.Refresh();
}

int ORBextractor::Submit(const std::vector<cv::Mat>& images, int maxBatch)
{
    if (images.empty()) throw ORBhipError("ORBextractor::Submit: no images");
    const int n = (int)images.size();
    for (int i = 0; i < n; i++)
        if (images[i].empty() || images[i].type() != CV_8UC1 || images[i].cols != images[0].cols || images[i].rows != images[0].rows || images[i].step != images[0].step)
            throw ORBhipError("ORBextractor::Submit: images must be non-empty CV_8UC1 of one size and row step");
    EnsureContext(images[0].cols, images[0].rows, std::max(std::max(maxBatch, n), mnCtxBatch));
    std::vector<const uint8_t*> ptrs(n);
    for (int i = 0; i < n; i++) ptrs[i] = images[i].data;
    int ticket = -1;
    if (orbhip_submit(mpCtx, n, &ptrs[0], (int)images[0].step, &ticket) != ORBHIP_OK) Fail("ORBextractor::Submit");
    mvTicketSizes[ticket & 3] = n; mnPendingTickets++; mbFrameState = false; mbPairResults = false;
    return ticket;
}

void ORBextractor::Collect(int ticket, std::vector<std::vector<cv::KeyPoint> >& keypoints, std::vector<cv::Mat>& descriptors)
{
    if (!mpCtx) throw ORBhipError("ORBextractor::Collect: nothing was submitted");
    const int n = mvTicketSizes[ticket & 3], cap = orbhip_keypoint_capacity(mpCtx);
    std::vector<orbhip_keypoint> kps((size_t)n * cap); std::vector<unsigned char> desc((size_t)n * cap * 32); std::vector<int> cnt(n, 0);
    const orbhip_status st = orbhip_collect(mpCtx, ticket, &kps[0], &desc[0], cap, &cnt[0]);
    // every status but INVALID (not the oldest ticket: it stays collectable) retired the ticket on the C side, a device failure included
    if (st != ORBHIP_ERR_INVALID && mnPendingTickets > 0) mnPendingTickets--;
    if (st != ORBHIP_OK) Fail("ORBextractor::Collect");
    keypoints.resize(n); descriptors.resize(n);
    for (int i = 0; i < n; i++) {
        keypoints[i].resize(cnt[i]);
        if (cnt[i] == 0) { descriptors[i].release(); continue; }
        memcpy(static_cast<void*>(&keypoints[i][0]), &kps[(size_t)i * cap], (size_t)cnt[i] * sizeof(orbhip_keypoint));
        descriptors[i].create(cnt[i], 32, CV_8U);
        for (int r = 0; r < cnt[i]; r++) memcpy(descriptors[i].ptr(r), &desc[((size_t)i * cap + r) * 32], 32);
    }
    mnLastN = 0; mbFrameState = false; mbBound = false; mbStereoColumns = false; mbPairResults = false;     // the context's planes / key points are some batch's, not one image's
    std::lock_guard<std::mutex> lock(mvImagePyramid.mMutex);
    mvImagePyramid.mbStale = true;
}

void ORBextractor::ComputeStereoMatches(ORBextractor& right, float mbf, float mb, int N, std::vector<float>& mvuRight, std::vector<float>& mvDepth)
{
    if (mbPairResults && N == mnLastN) {                         // ExtractStereo already ran the matcher behind the extraction
        mvuRight.assign(mvPairURight.begin(), mvPairURight.begin() + N); mvDepth.assign(mvPairDepth.begin(), mvPairDepth.begin() + N);
        return;
    }
    mvuRight = std::vector<float>(N, -1.0f);                     // Frame.cc:468-469
    mvDepth = std::vector<float>(N, -1.0f);
    if (N == 0 || !mpCtx || !right.mpCtx) return;
    RequireFrameState("ORBextractor::ComputeStereoMatches"); right.RequireFrameState("ORBextractor::ComputeStereoMatches (right)");
    if (orbhip_compute_stereo_matches(mpCtx, right.mpCtx, 1, mbf, mb, &mvuRight[0], &mvDepth[0], N) != ORBHIP_OK) Fail("ORBextractor::ComputeStereoMatches");
    mbStereoColumns = true;
}

} // namespace ORB_SLAM2
