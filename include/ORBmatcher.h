// ORBmatcher.h — drop-in replacement for raulmur/ORB_SLAM2 include/ORBmatcher.h (the class declared at ORBmatcher.h:37-102 there).
//
// The class keeps every public signature of the reference so that Tracking, LocalMapping, LoopClosing, MapPoint and Frame
// compile unchanged.  What lives where:
//   * implemented in this repository (orb_slam2_amd/cpp/ORBmatcher.cc, through the C ABI include/orbhip.h):
//       the constructor, DescriptorDistance and SearchForInitialization — the part of the matcher that needs no map;
//   * the nine map-dependent searches keep the reference's own bodies for their pose algebra and map bookkeeping; their candidate
//     loops — the data-parallel part — are entry points of liborbhip.so on flat queries; INTEGRATION.md §2 shows the few lines that
//     hand each loop over and integration/apply_dropin.py applies them to the reference's ORBmatcher.cc (checked against the unmodified
//     reference by tests/test_reference_dropin.py).  The comment on each declaration names the entry point.
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

    // ---- bodies stay the reference's; the search loop of each maps to the entry point named -----------------------------
    // local map points projected into the frame (Tracking::SearchLocalPoints)             -> orbhip_search_by_projection, mode 0
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3);
    // last frame's map points projected with the motion model (TrackWithMotionModel)      -> orbhip_search_by_projection, mode 1
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);
    // relocalisation: key-frame map points projected into the frame                       -> orbhip_search_by_projection, mode 1
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                           const int ORBdist);
    // loop closing: points seen by the loop key frame projected with a similarity         -> orbhip_search_by_projection, mode 1
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched,
                           int th);

    // bag-of-words guided matching, features of the same vocabulary node only             -> orbhip_search_by_bow, mode 0 / 1
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);

    // new map points: unmatched features along epipolar lines (LocalMapping)              -> orbhip_search_for_triangulation
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                               const bool bOnlyStereo);

    // Sim3-guided search in both directions (LoopClosing::ComputeSim3)                    -> orbhip_search_best_in_window
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
                     const cv::Mat& t12, const float th);

    // duplicate map points merged into a key frame, Euclidean / similarity pose           -> orbhip_search_best_in_window
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0);
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint);

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
