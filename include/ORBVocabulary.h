// ORBVocabulary.h — drop-in for ORB_SLAM2's include/ORBVocabulary.h (:31-32, a typedef of
// DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB>) on top of liborbhip.so.
//
// Only the members ORB_SLAM2 calls are provided, with the reference's signatures:
//   loadFromTextFile(filename)                               System.cc:68
//   transform(features, BowVector&, FeatureVector&, levelsup) Frame.cc:400, KeyFrame.cc:50 (levelsup = 4)
//   score(BowVector, BowVector)                               KeyFrameDatabase.cc:133,249, LoopClosing.cc:134
//   size(), empty()
// plus ComputeBoW(extractor, ...) which transforms the descriptors of the extractor's last frame where they lie in HBM
// (Frame::ComputeBoW, Frame.cc:395-402, without the host round trip).
//
// DBoW2::BowVector / DBoW2::FeatureVector are the reference's own types (std::map<WordId, WordValue> and
// std::map<NodeId, std::vector<unsigned int>>, Thirdparty/DBoW2/DBoW2/BowVector.h:56-57, FeatureVector.h:23-24).  Where the
// reference tree is on the include path, define ORBHIP_USE_DBOW2_TYPES and its headers are used; otherwise the two
// typedef-equivalent definitions below stand in (this repo is built without the reference's sources).
#ifndef ORBVOCABULARY_H
#define ORBVOCABULARY_H

#include <map>
#include <string>
#include <vector>
#if defined(ORBHIP_USE_OPENCV)
#include <opencv2/core/core.hpp>
#else
#include "cvlite/cvlite.h"
#endif
#if defined(ORBHIP_USE_DBOW2_TYPES)
#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
// Inside the reference tree this header replaces one that includes Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h, and the reference's own
// headers (Frame.h, KeyFrame.h, Map.h, KeyFrameDatabase.h ...) silently depend on what that file drags in: a handful of standard headers
// and the using-directive at TemplatedVocabulary.h:36.  Reproduced here so that they compile unchanged.
#include <algorithm>
#include <cassert>
#include <fstream>
#include <limits>
#include <list>
#include <numeric>
#include <set>
using namespace std;
#else
namespace DBoW2 {
typedef unsigned int WordId;
typedef double WordValue;
typedef unsigned int NodeId;
class BowVector : public std::map<WordId, WordValue> {};
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {};
}  // namespace DBoW2
#endif

struct orbhip_voc;

namespace ORB_SLAM2
{

class ORBextractor;

class ORBVocabulary
{
public:
    ORBVocabulary();
    ~ORBVocabulary();

    // Loads the vocabulary from a text file (k L scoring weighting / parent isLeaf 32 bytes weight per node)
    bool loadFromTextFile(const std::string &filename);

    // Number of words; whether the vocabulary is empty
    unsigned int size() const;
    bool empty() const { return size() == 0; }

    // Transforms a set of descriptors (1x32 CV_8U each, Converter::toDescriptorVector) into a bow vector and a feature vector
    void transform(const std::vector<cv::Mat>& features, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup) const;
    // The same on an n x 32 descriptor matrix (Frame::mDescriptors) without splitting it into rows first
    void transform(const cv::Mat& descriptors, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup) const;
    // The same on the descriptors of the extractor's last frame, read where they lie in device memory
    void ComputeBoW(ORBextractor& extractor, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup = 4) const;

    // Score of two bow vectors with the scoring object the file names (L1_NORM for ORBvoc.txt)
    double score(const DBoW2::BowVector &a, const DBoW2::BowVector &b) const;

    // HIP device the vocabulary lives on (default 0, or ORBHIP_DEVICE); takes effect at the next loadFromTextFile
    void SetDevice(int device) { mnDevice = device; }

private:
    ORBVocabulary(const ORBVocabulary&);
    ORBVocabulary& operator=(const ORBVocabulary&);
    void Deliver(int n, std::vector<unsigned int>& bowId, std::vector<double>& bowVal, int nbow, std::vector<unsigned int>& fvNode,
                 std::vector<int>& fvOff, std::vector<unsigned int>& fvFeat, int nfv, DBoW2::BowVector &v, DBoW2::FeatureVector &fv) const;

    orbhip_voc* mpVoc;
    int mnDevice;
};

} //namespace ORB_SLAM

#endif // ORBVOCABULARY_H
