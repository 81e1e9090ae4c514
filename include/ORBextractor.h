// ORBextractor.h — drop-in replacement for raulmur/ORB_SLAM2 include/ORBextractor.h (ORBextractor.h:45-111 there):
// same namespace, class name, enum, constructor, operator() and accessors, same public mvImagePyramid member, so
// Frame.cc / Tracking.cc compile and call it unchanged (Frame::ExtractORB, Frame.cc:247-253; Tracking.cc:119-125).
// The work happens on an MI355X through the C ABI of include/orbhip.h; there is no CPU path behind this class.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <atomic>
#include <list>
#include <mutex>
#include <stdexcept>
#include <vector>
#ifdef ORBHIP_USE_OPENCV
#include <opencv/cv.h>
#else
#include "cvlite/cvlite.h"
#endif

struct orbhip_ctx;

namespace ORB_SLAM2
{

// The reference's operator() cannot fail (ORBextractor.h:59-61 returns void); a GPU call can.  Every failure of the device library
// (no usable GPU, a HIP error, an image geometry outside the supported envelope) is thrown as this exception with the library's
// message — never swallowed (an empty key point list would silently derail tracking), never abort(): an application that wants
// to survive a transient device fault can catch it around Frame construction; one that does not terminates with the message.
class ORBhipError : public std::runtime_error
{
public:
    explicit ORBhipError(const std::string& what) : std::runtime_error(what) {}
};

class ORBextractor;
// What callers see as `std::vector<cv::Mat> mvImagePyramid` (ORBextractor.h:85).  The only readers in the reference are
// Frame::ComputeStereoMatches' `mvImagePyramid[level]` expressions (Frame.cc:473, 563, 575, 580); monocular and RGB-D pipelines never
// touch it.  The planes live in HBM: the first operator[] after an extraction fetches all levels of that image in one batched copy
// with a single synchronisation, later accesses are free.  SetPyramidDownload(true) restores the eager download after every call.
class ORBimagePyramid
{
public:
    ORBimagePyramid() : mpOwner(NULL), mbStale(false) {}
    cv::Mat& operator[](size_t level) { Refresh(); return mvLevels[level]; }
    const cv::Mat& operator[](size_t level) const { const_cast<ORBimagePyramid*>(this)->Refresh(); return mvLevels[level]; }
    size_t size() const { return mvLevels.size(); }
    bool empty() const { return mvLevels.empty(); }
    void resize(size_t n) { mvLevels.resize(n); }
    // forks that walk the pyramid (`for (auto& im : mvImagePyramid)`, `.at(l)`, `.front()`, a copy into a std::vector<cv::Mat>): the same lazy fetch first
    typedef std::vector<cv::Mat>::iterator iterator; typedef std::vector<cv::Mat>::const_iterator const_iterator;
    iterator begin() { Refresh(); return mvLevels.begin(); }
    iterator end() { Refresh(); return mvLevels.end(); }
    const_iterator begin() const { const_cast<ORBimagePyramid*>(this)->Refresh(); return mvLevels.begin(); }
    const_iterator end() const { const_cast<ORBimagePyramid*>(this)->Refresh(); return mvLevels.end(); }
    cv::Mat& at(size_t level) { Refresh(); return mvLevels.at(level); }
    const cv::Mat& at(size_t level) const { const_cast<ORBimagePyramid*>(this)->Refresh(); return mvLevels.at(level); }
    cv::Mat& front() { Refresh(); return mvLevels.front(); }
    cv::Mat& back() { Refresh(); return mvLevels.back(); }
    operator const std::vector<cv::Mat>&() const { const_cast<ORBimagePyramid*>(this)->Refresh(); return mvLevels; }
private:
    friend class ORBextractor;
    void Refresh();
    ORBextractor* mpOwner; bool mbStale; std::vector<cv::Mat> mvLevels; std::mutex mMutex;
};

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

    // mvImagePyramid[l] of the last image (read by Frame::ComputeStereoMatches, Frame.cc:473,563-580): fetched from the device the first
    // time it is indexed after a call (see ORBimagePyramid); SetPyramidDownload(true) downloads eagerly after every call instead.
    ORBimagePyramid mvImagePyramid;
    void SetPyramidDownload(bool eager) { mbDownloadPyramid = eager; }
    // cv::GaussianBlur's last rounding step: 0 = OpenCV's generic C++ path, (sum + 2^15) >> 16; 1 = the SSE2 column filter
    // every x86-64 build of OpenCV <= 3.3 runs (round-half-even on whole 4-column groups).  The default is the host's own: 1 when this
    // class is compiled for x86-64, 0 elsewhere — i.e. what the OpenCV this class stands in for would have computed on this machine.
    void SetBlurRounding(int mode);
    int GetBlurRounding() const { return mnBlurRounding; }
    // The pattern rotation of computeOrbDescriptor (ORBextractor.cc:118-120): 1 = the fused multiply-adds gcc emits for the
    // reference's own flags (-O3 -march=native on an FMA-capable host), 0 = two roundings per expression.  The default is what THIS class's
    // compiler flags say the reference's translation unit would have been built like: 1 when compiled with FMA code generation enabled
    // (__FMA__, e.g. by the reference's -march=native), 0 otherwise.  About one descriptor bit per few hundred frames depends on it.
    void SetFpContract(int mode);
    int GetFpContract() const { return mnFpContract; }

    // Asynchronous form for callers that own more than one camera or keep a queue of frames (the reference has none: Frame::ExtractORB is
    // synchronous, Frame.cc:247-253).  Submit() uploads and enqueues a batch of equally sized CV_8UC1 images and returns a ticket; up to
    // three batches may be in flight; Collect() blocks for the oldest ticket and delivers per image what operator() delivers.  Pageable
    // images are consumed when Submit returns; pinned ones are read by DMA until Collect.  maxBatch is fixed by the first Submit.
    // The per-frame follow-ups below (UndistortKeyPoints, mvImagePyramid, ComputeStereoMatches, ComputeStereoFromRGBD) read the state of
    // the last SINGLE-image call (operator(), ExtractColor, ExtractRectified).  After a Submit or a Collect that state belongs to some
    // batch, not to one image: those members then throw ORBhipError until the next single-image call — batched callers use what Collect
    // hands back.  A Submit that would have to re-create the context (a larger batch, another image size) while tickets are still in
    // flight throws as well instead of destroying them.
    int Submit(const std::vector<cv::Mat>& images, int maxBatch = 0);
    void Collect(int ticket, std::vector<std::vector<cv::KeyPoint> >& keypoints, std::vector<cv::Mat>& descriptors);
    // Frame::ComputeStereoMatches (Frame.cc:466-640) on the GPU.  `this` is the left extractor, `right` the right one; both
    // must have processed their image of the current frame (Frame.cc:78-81).  N = number of left keypoints (Frame::N).
    // Fills mvuRight / mvDepth exactly like the reference; reads keypoints, descriptors and pyramids that are still in HBM.
    void ComputeStereoMatches(ORBextractor& right, float mbf, float mb, int N, std::vector<float>& mvuRight, std::vector<float>& mvDepth);
    // The stereo pair as ONE call: what Frame::Frame(imLeft, imRight, ...) does with two std::threads, two extractors and ComputeStereoMatches
    // (Frame.cc:78-90), on THIS extractor alone - both images are uploaded together and run through one launch chain on one device context, the stereo
    // matcher is queued behind it, one synchronisation.  mb = mbf / fx (the reference reads the member before assigning it, Frame.cc:89, 113: DESIGN.md
    // H7).  The ComputeStereoMatches() that follows (on this object, any `right`) hands out the columns computed here without another device call;
    // UndistortKeyPoints / BindFrame / the resident-frame searches see the LEFT image.  integration/apply_dropin.py --stereo-one-call emits the call.
    void ExtractStereo(cv::InputArray imLeft, cv::InputArray imRight, std::vector<cv::KeyPoint>& keysLeft, cv::OutputArray descLeft,
                       std::vector<cv::KeyPoint>& keysRight, cv::OutputArray descRight, float mbf, float mb);
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
    // HIP device this extractor runs on (default: ORBHIP_DEVICE, else 0); SetDevice takes effect at the next call: the context of another device is
    // never reused (the ones kept for recently seen image sizes remember their device).
    void SetDevice(int device) { mnDevice = device; }
    int Device() const { return mnDevice; }
    // Which Frame the features in HBM belong to.  The forwarded Frame::UndistortKeyPoints (every Frame constructor calls it right after
    // ExtractORB, Frame.cc:84, 136, 194) names the frame with its mnId; the next image replaces the state.  A matcher that is handed that
    // Frame (Tracking's SearchByProjection calls on mCurrentFrame) then searches the resident key points / descriptors / mvuRight through the
    // *_frame entry points of orbhip.h — only its queries cross PCIe — and any other Frame goes through host buffers as before.
    // The frame-resident forms are for the thread that built the frame (Tracking: the constructor, Frame::ComputeBoW and the Frame searches all run there,
    // Tracking.cc:167-238, 867-928, 1143-1193, 1344-1520).  BindFrame records it; a caller on ANY OTHER thread is told "not held" without a look at the
    // extractor's state - which the owning thread may be rewriting with its next image at that moment - and takes the host path on the frame's own copies.
    void BindFrame(unsigned long frameId) { mnBoundFrame = frameId; mbBound = true; mnOwnerThread.store(ThisThread(), std::memory_order_release); }
    bool HoldsFrame(unsigned long frameId, int N) const
    { return mnOwnerThread.load(std::memory_order_acquire) == ThisThread() && mpCtx && mbFrameState && mbBound && mnBoundFrame == frameId && N == mnLastN; }
    static unsigned long ThisThread() { static std::atomic<unsigned long> next(0); static thread_local unsigned long mine = ++next; return mine; }     // never 0, never reused
    // mvuRight the caller computed itself (the reference's Frame::ComputeStereoFromRGBD loop, Frame.cc:643-665) for the frame this extractor still holds:
    // N floats go to the device for the resident searches instead of the depth map.  integration/apply_dropin.py appends the call to the reference's loop.
    void SetStereoColumns(const std::vector<float>& mvuRight);
    bool HoldsStereoColumns() const { return mbStereoColumns; }      // mvuRight of that frame is on the device too (ComputeStereoMatches / ComputeStereoFromRGBD ran)

protected:
    friend class ORBimagePyramid;
    void EnsureContext(int width, int height, int maxBatch = 1);
    void FetchPyramid(std::vector<cv::Mat>& levels);
    void Fail(const char* where) const;
    void Deliver(int n, int slot, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors);
    void ReserveStage(int slots);

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
    // what the library hands back lands here (sized once per context: no allocation, no value-initialisation per call), [slot][capacity]; slot 0 also
    // serves UndistortKeyPoints of an undistorted camera (mvKeysUn = mvKeys) until the next call
    std::vector<unsigned char> mvKpStage, mvDescStage; int mnStageCap;
    std::vector<float> mvPairURight, mvPairDepth; bool mbPairResults;      // mvuRight / mvDepth of the last ExtractStereo, until ComputeStereoMatches picks them up
    orbhip_ctx* mpCtx;          // device context for the current image size (created lazily at the first image of that size)
    int mnCtxW, mnCtxH, mnCtxBatch, mnCtxDevice, mnDevice, mnBlurRounding, mnFpContract;      // mnCtxDevice: where mpCtx lives; mnDevice: where the next one goes
    // The reference takes any image size per call (it re-allocates its pyramid every time, ORBextractor.cc:1043-1056, 1107-1132); a device context is
    // laid out for ONE size.  Contexts of the sizes seen recently are therefore kept (up to four, least recently used goes first): a caller that
    // alternates between sizes switches contexts in microseconds instead of paying a context creation (tens of milliseconds) per change.
    struct CtxSlot { orbhip_ctx* ctx; int w, h, batch, device; unsigned settings, stamp; };
    std::vector<CtxSlot> mvCtxCache; unsigned mnSettings, mnStamp;          // mnSettings counts SetBlurRounding / SetFpContract / SetCamera calls
    void ApplySettings(orbhip_ctx* ctx);
    std::vector<int> mvTicketSizes;                              // images per ticket in flight (Submit / Collect), by ticket mod 4
    bool mbDownloadPyramid;
    int mnPendingTickets;                                        // Submit()ed, not yet Collect()ed
    bool mbFrameState;                                           // the context's current state is that of one single-image call
    void RequireFrameState(const char* where) const;
    float mfScaleFactorArg;
    unsigned long mnBoundFrame; bool mbBound, mbStereoColumns;
    std::atomic<unsigned long> mnOwnerThread;                    // ThisThread() of the last BindFrame
};

} // namespace ORB_SLAM2

#endif
