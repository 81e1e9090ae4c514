// ORBmatcher.h — drop-in replacement for raulmur/ORB_SLAM2 include/ORBmatcher.h (the class declared at ORBmatcher.h:37-102 there).
//
// The class keeps every public signature of the reference so that Tracking, LocalMapping, LoopClosing, MapPoint and Frame
// compile unchanged.  ALL twelve members are implemented in this repository (orb_slam2_amd/cpp/ORBmatcher.cc, through the C ABI include/orbhip.h):
// each is a gather of what the member reads, one device call (Hamming distances, window / vocabulary-node searches with their order-dependent
// bookkeeping, and the per-point projection algebra of the five pose-guided members) and the member's write-back / map surgery.
// integration/apply_dropin.py installs both files into a checkout; the comment on each declaration names the entry point.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <set>
#include <utility>
#include <vector>

#ifdef ORBHIP_USE_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include "MapPoint.h"
#include "KeyFrame.h"
#include "Frame.h"
#else
#include "cvlite/cvlite.h"
namespace ORB_SLAM2 { class Frame; class KeyFrame; class MapPoint; }
#endif

namespace ORB_SLAM2
{

class ORBmatcher
{
public:
    // thresholds of the reference (ORBmatcher.cc:37-39): 50, 100, 30
    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

    ORBmatcher(float nnratio = 0.6, bool checkOri = true);

    // 256-bit Hamming distance of two descriptor rows                                   -> orbhip_descriptor_distance
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);

    // ---- runs on the GPU in this drop-in --------------------------------------------------------------------------------
    // Map initialisation, monocular only (Tracking.cc:599-600): F1's level-0 key points against the windows around vbPrevMatched
    // in F2, rotation-consistency histogram                                               -> orbhip_search_for_initialization_bounds
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                                int windowSize = 10);

    // ---- the map-dependent members (compiled where the tree's MapPoint / KeyFrame / Frame types exist) ------------------
    // local map points projected into the frame (Tracking::SearchLocalPoints)             -> orbhip_search_by_projection_frame / _bounds, mode 0
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3);
    // last frame's map points projected with the motion model (TrackWithMotionModel)      -> orbhip_project_search_frame / _bounds, LAST_FRAME
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);
    // relocalisation: key-frame map points projected into the frame                       -> orbhip_project_search_frame / _bounds, FRAME_KF
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                           const int ORBdist);
    // loop closing: points seen by the loop key frame projected with a similarity         -> orbhip_project_search_bounds, KF_SIM3
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched,
                           int th);

    // bag-of-words guided matching, features of the same vocabulary node only             -> orbhip_search_by_bow, mode 0 / 1
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);

    // new map points: unmatched features along epipolar lines (LocalMapping)              -> orbhip_search_for_triangulation
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                               const bool bOnlyStereo);

    // Sim3-guided search in both directions (LoopClosing::ComputeSim3)                    -> orbhip_project_best_in_window_batch, SIM3 x 2 slots
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
                     const cv::Mat& t12, const float th);

    // duplicate map points merged into a key frame, Euclidean / similarity pose           -> orbhip_project_best_in_window_batch, FUSE / FUSE_SIM3
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0);
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint);

    // Frame::isInFrustum (Frame.cc:269-325; Tracking::SearchLocalPoints calls it for every local map point of every frame) without its five cv::Mat
    // temporaries: the same statements on flat floats, the map point visited once.  integration/apply_dropin.py --flat-frustum makes the member a one-line
    // forward to this (Rcw / tcw / Ow = the frame's private mRcw / mtcw / mOw, which only the member itself can name).
    static bool IsInFrustum(Frame &F, const cv::Mat &Rcw, const cv::Mat &tcw, const cv::Mat &Ow, MapPoint* pMP, float viewingCosLimit);

    // The readers behind `friend class ORBmatcher;` in include/MapPoint.h (integration/apply_dropin.py adds that line): a map point's position, viewing
    // direction, scale-invariance range, mfMaxDistance and descriptor read in place under the point's own mutexes.  Defined in ORBmatcher.cc; declared here
    // because friendship reaches the class's members (this nested type is one), not the file's free functions.  No data member: the class layout is the reference's.
    struct Access;

protected:
    // helpers of the reference's own bodies (ORBmatcher.cc:131-157, 1601-1642); the GPU entry points carry their own versions
    float RadiusByViewingCos(const float& viewCos);
    bool CheckDistEpipolarLine(const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const cv::Mat& F12, const KeyFrame* pKF);
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);

    float mfNNratio;            // best / second-best ratio
    bool mbCheckOrientation;    // rotation-consistency check on / off
};

} // namespace ORB_SLAM2

#endif
