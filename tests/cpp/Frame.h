// Minimal stand-in for the reference's include/Frame.h: only the members ORBextractor / ORBmatcher touch, with the
// reference's names (Frame.h:52-58, 120-190; Frame.cc:61-117, 247-253).  Test scaffolding, not product.
#pragma once
#include <vector>
#include "ORBextractor.h"

namespace ORB_SLAM2
{
class Frame
{
public:
    Frame() : N(0), mpORBextractorLeft(NULL) {}
    Frame(const cv::Mat& imGray, ORBextractor* extractor) : mpORBextractorLeft(extractor)
    {
        ExtractORB(0, imGray);
        N = (int)mvKeys.size();
        mvKeysUn = mvKeys;                                   // UndistortKeyPoints with k1 == 0 (Frame.cc:406-410)
        mnMinX = 0.0f; mnMaxX = (float)imGray.cols; mnMinY = 0.0f; mnMaxY = (float)imGray.rows;   // ComputeImageBounds (Frame.cc:455-463)
    }
    // distorted monocular camera: Frame.cc:174-225 with UndistortKeyPoints / ComputeImageBounds forwarded to the extractor
    Frame(const cv::Mat& imGray, ORBextractor* extractor, const cv::Mat& K, const cv::Mat& distCoef) : mpORBextractorLeft(extractor)
    {
        extractor->SetCamera(K, distCoef);
        ExtractORB(0, imGray);
        N = (int)mvKeys.size();
        extractor->UndistortKeyPoints(mvKeysUn);
        extractor->ComputeImageBounds(imGray.cols, imGray.rows, mnMinX, mnMaxX, mnMinY, mnMaxY);
    }
    void ExtractORB(int flag, const cv::Mat& im) { (void)flag; (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors); }   // Frame.cc:247-253

    int N;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    cv::Mat mDescriptors;
    ORBextractor* mpORBextractorLeft;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
};
}
