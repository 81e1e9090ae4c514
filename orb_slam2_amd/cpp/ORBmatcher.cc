// ORBmatcher.cc — host side of the drop-in ORB_SLAM2::ORBmatcher for the map-free part of the hot path:
// constructor, constants, DescriptorDistance (ORBmatcher.cc:1647-1663 of the reference) and SearchForInitialization
// (ORBmatcher.cc:405-520), forwarded to the C ABI (include/orbhip.h).  The Frame members it reads are exactly the ones the
// reference reads: mvKeysUn, mDescriptors, mnMinX/mnMaxX/mnMinY/mnMaxY (Frame.h:120-190).
#include "ORBmatcher.h"
#include "orbhip.h"
#ifndef ORBHIP_USE_OPENCV
#include "Frame.h"          // the caller's Frame (tests/cpp/Frame.h stands in for the reference's include/Frame.h here)
#endif

#include <cstdio>
#include <cstdlib>
#include <string>

namespace ORB_SLAM2
{

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b)
{
    return orbhip_descriptor_distance(a.ptr<unsigned char>(), b.ptr<unsigned char>());
}

int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize)
{
    const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
    vnMatches12 = std::vector<int>(n1, -1);
    if (n1 == 0) return 0;
    // the grid of GetFeaturesInArea spans the (undistorted) image bounds (Frame.cc:101-102, 327-346, 436-464)
    const orbhip_bounds bounds = {Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
    int nmatches = 0;
    static const int device = getenv("ORBHIP_DEVICE") ? atoi(getenv("ORBHIP_DEVICE")) : 0;      // read once per process
    const orbhip_status st = orbhip_search_for_initialization_bounds(
        device, reinterpret_cast<const orbhip_keypoint*>(&F1.mvKeysUn[0]), F1.mDescriptors.ptr<unsigned char>(), n1,
        n2 ? reinterpret_cast<const orbhip_keypoint*>(&F2.mvKeysUn[0]) : NULL, n2 ? F2.mDescriptors.ptr<unsigned char>() : NULL, n2,
        &bounds, reinterpret_cast<float*>(&vbPrevMatched[0]), &vnMatches12[0], windowSize, mfNNratio, mbCheckOrientation ? 1 : 0, &nmatches);
    if (st != ORBHIP_OK) throw ORBhipError(std::string("ORBmatcher::SearchForInitialization: ") + orbhip_last_error());   // see ORBextractor.h: never swallowed, never abort()
    return nmatches;
}

} // namespace ORB_SLAM2
