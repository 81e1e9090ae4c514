// ORBmatcher.h — drop-in replacement for raulmur/ORB_SLAM2 include/ORBmatcher.h (ORBmatcher.h:37-102 there).
// Every public signature of the reference class is kept so Tracking / LocalMapping / LoopClosing / MapPoint / Frame
// compile unchanged.  This repository implements (orb_slam2_amd/cpp/ORBmatcher.cc) the part that is on the hot path and
// needs no map: the constructor, DescriptorDistance and SearchForInitialization, all through include/orbhip.h.
// The nine map-dependent searches (SearchByProjection x4, SearchByBoW x2, SearchForTriangulation, SearchBySim3, Fuse x2)
// stay the reference's own bodies (they only reach the GPU through DescriptorDistance); INTEGRATION.md shows the patch.
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
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);

    // Computes the Hamming distance between two ORB descriptors
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);

    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3);
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist);
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th);
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);

    // Matching for the Map Initialization (only used in the monocular case) — runs on the GPU
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10);

    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs, const bool bOnlyStereo);
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12, const float th);
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0);
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint);

public:
    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

protected:
    bool CheckDistEpipolarLine(const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const cv::Mat& F12, const KeyFrame* pKF);
    float RadiusByViewingCos(const float& viewCos);
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);

    float mfNNratio;
    bool mbCheckOrientation;
};

} // namespace ORB_SLAM2

#endif
