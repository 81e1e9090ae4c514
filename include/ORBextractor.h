// ORBextractor.h — drop-in replacement for raulmur/ORB_SLAM2 include/ORBextractor.h (ORBextractor.h:45-111 there):
// same namespace, class name, enum, constructor, operator() and accessors, same public mvImagePyramid member, so
// Frame.cc / Tracking.cc compile and call it unchanged (Frame::ExtractORB, Frame.cc:247-253; Tracking.cc:119-125).
// The work happens on an MI355X through the C ABI of include/orbhip.h; there is no CPU path behind this class.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <list>
#include <vector>
#ifdef ORBHIP_USE_OPENCV
#include <opencv/cv.h>
#else
#include "cvlite/cvlite.h"
#endif

struct orbhip_ctx;

namespace ORB_SLAM2
{

class ORBextractor
{
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();

    // Compute the ORB features and descriptors on an image (mask is ignored, as in the reference).
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors);

    // The same for an interleaved 8-bit colour frame (3 or 4 channels): Tracking::GrabImage* (Tracking.cc:172-198, 217-229,
    // 248-260) converts with cv::cvtColor on the CPU before building the Frame; here the conversion runs on the GPU and
    // mvImagePyramid[0] is the gray frame.  bRGB is Tracking::mbRGB (Camera.RGB, Tracking.cc:82): true = R first.
    void ExtractColor(const unsigned char* data, int step, int cols, int rows, int channels, bool bRGB,
                      std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // mvImagePyramid[l] is filled after every call (read by Frame::ComputeStereoMatches, Frame.cc:473,563-580).
    // Monocular / RGB-D pipelines never read it: SetPyramidDownload(false) skips the 8 device-to-host copies.
    std::vector<cv::Mat> mvImagePyramid;
    void SetPyramidDownload(bool on) { mbDownloadPyramid = on; }
    // Frame::ComputeStereoMatches (Frame.cc:466-640) on the GPU.  `this` is the left extractor, `right` the right one; both
    // must have processed their image of the current frame (Frame.cc:78-81).  N = number of left keypoints (Frame::N).
    // Fills mvuRight / mvDepth exactly like the reference; reads keypoints, descriptors and pyramids that are still in HBM.
    void ComputeStereoMatches(ORBextractor& right, float mbf, float mb, int N, std::vector<float>& mvuRight, std::vector<float>& mvDepth);
    // The device context of the last image (NULL before the first call): ORBVocabulary::ComputeBoW reads the descriptors there.
    orbhip_ctx* Context() { return mpCtx; }
    // Distorted cameras (TUM1-3.yaml).  SetCamera takes Frame's mK / mDistCoef (CV_32F; 4 or 5 coefficients, Tracking.cc:60-82);
    // UndistortKeyPoints then is the body of Frame::UndistortKeyPoints (Frame.cc:404-434) for the image this extractor processed
    // last — mvKeysUn was computed on the device behind the descriptors — and ComputeImageBounds the body of
    // Frame::ComputeImageBounds (Frame.cc:436-464).
    void SetCamera(const cv::Mat& K, const cv::Mat& distCoef);
    void UndistortKeyPoints(std::vector<cv::KeyPoint>& mvKeysUn);
    void ComputeImageBounds(int cols, int rows, float& mnMinX, float& mnMaxX, float& mnMinY, float& mnMaxY);
    // RGB-D sensors: the body of Frame::ComputeStereoFromRGBD (Frame.cc:643-665) for the image this extractor processed last (key points
    // and, with SetCamera, mvKeysUn are still on the device).  imDepth: CV_32F as the reference's Frame receives it (depthFactor 1), or the
    // raw CV_16U map with mDepthMapFactor — the convertTo of Tracking::GrabImageRGBD (Tracking.cc:226-227) is then applied per key point.
    void ComputeStereoFromRGBD(const cv::Mat& imDepth, float depthFactor, float mbf, int N, std::vector<float>& mvuRight, std::vector<float>& mvDepth);
    // Raw (unrectified) stereo input: the cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR) the EuRoC example runs before TrackStereo
    // (Examples/Stereo/stereo_euroc.cc:136-137) moves onto the device.  M1 / M2: the CV_32FC1 maps of initUndistortRectifyMap
    // (stereo_euroc.cc:97-98).  ExtractRectified(raw, ..) == operator()(remap(raw), ..); mvImagePyramid[0] is the rectified image.
    void SetRectification(const cv::Mat& M1, const cv::Mat& M2, int rawCols, int rawRows);
    void ExtractRectified(const cv::Mat& raw, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors);
    // HIP device this extractor runs on (default 0); takes effect at the next (re)creation of the device context.
    void SetDevice(int device) { mnDevice = device; }

protected:
    void EnsureContext(int width, int height);
    void Deliver(int n, const std::vector<unsigned char>& desc, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors);

    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;

    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor;
    std::vector<float> mvInvScaleFactor;
    std::vector<float> mvLevelSigma2;
    std::vector<float> mvInvLevelSigma2;

    float mCamera[9]; bool mbHasCamera;                         // fx, fy, cx, cy, k1, k2, p1, p2, k3
    std::vector<float> mvMapX, mvMapY; int mnRawCols, mnRawRows; // rectification maps (applied when the context is (re)created)
    int mnLastN;                                                 // key points of the last image
    orbhip_ctx* mpCtx;          // device context for the current image size (created lazily, re-created on a size change)
    int mnCtxW, mnCtxH, mnDevice;
    bool mbDownloadPyramid;
    float mfScaleFactorArg;
};

} // namespace ORB_SLAM2

#endif
