// ORBVocabulary.cc — ORB_SLAM2::ORBVocabulary on liborbhip.so (see include/ORBVocabulary.h).  The reference's class is the
// DBoW2 template instantiated for ORB (include/ORBVocabulary.h:31-32); this file forwards the members ORB_SLAM2 uses to the
// C ABI and rebuilds the two std::map results from the flat, already key-ordered arrays (hinted inserts: linear time).
#include "ORBVocabulary.h"
#include "ORBextractor.h"
#include "orbhip.h"
#include <string>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace ORB_SLAM2
{

ORBVocabulary::ORBVocabulary() : mpVoc(NULL), mnDevice(0)
{
    if (const char* dev = getenv("ORBHIP_DEVICE")) mnDevice = atoi(dev);
}

ORBVocabulary::~ORBVocabulary() { if (mpVoc) orbhip_voc_destroy(mpVoc); }

bool ORBVocabulary::loadFromTextFile(const std::string &filename)
{
    if (mpVoc) { orbhip_voc_destroy(mpVoc); mpVoc = NULL; }
    if (orbhip_voc_load_text(&mpVoc, filename.c_str(), mnDevice) != ORBHIP_OK) {
        fprintf(stderr, "ORBVocabulary: %s\n", orbhip_last_error());      // the reference prints to cerr and returns false (:1360-1361)
        mpVoc = NULL;
        return false;
    }
    return true;
}

unsigned int ORBVocabulary::size() const
{
    int nwords = 0;
    if (mpVoc) orbhip_voc_info(mpVoc, NULL, NULL, NULL, NULL, NULL, &nwords);
    return (unsigned int)nwords;
}

void ORBVocabulary::Deliver(int n, std::vector<unsigned int>& bowId, std::vector<double>& bowVal, int nbow, std::vector<unsigned int>& fvNode,
                            std::vector<int>& fvOff, std::vector<unsigned int>& fvFeat, int nfv, DBoW2::BowVector &v, DBoW2::FeatureVector &fv) const
{
    (void)n;
    for (int i = 0; i < nbow; i++) v.insert(v.end(), std::make_pair(bowId[i], bowVal[i]));
    for (int j = 0; j < nfv; j++)
        fv.insert(fv.end(), std::make_pair(fvNode[j], std::vector<unsigned int>(fvFeat.begin() + fvOff[j], fvFeat.begin() + fvOff[j + 1])));
}

void ORBVocabulary::transform(const cv::Mat& descriptors, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup) const
{
    v.clear(); fv.clear();                                           // TemplatedVocabulary.h:1130-1131
    const int n = descriptors.rows;
    if (!mpVoc || n == 0) return;
    std::vector<unsigned char> packed;
    const unsigned char* d = descriptors.data;
    if ((size_t)descriptors.step != 32) { packed.resize((size_t)n * 32); for (int i = 0; i < n; i++) memcpy(&packed[(size_t)i * 32], descriptors.ptr(i), 32); d = &packed[0]; }
    std::vector<unsigned int> bowId(n), fvNode(n), fvFeat(n); std::vector<double> bowVal(n); std::vector<int> fvOff(n + 1);
    int nbow = 0, nfv = 0;
    if (orbhip_voc_transform(mpVoc, d, n, levelsup, &bowId[0], &bowVal[0], &nbow, &fvNode[0], &fvOff[0], &fvFeat[0], &nfv) != ORBHIP_OK) throw ORBhipError(std::string("ORBVocabulary::transform: ") + orbhip_last_error());
    Deliver(n, bowId, bowVal, nbow, fvNode, fvOff, fvFeat, nfv, v, fv);
}

void ORBVocabulary::transform(const std::vector<cv::Mat>& features, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup) const
{
    const int n = (int)features.size();
    cv::Mat all;
    if (n > 0) { all.create(n, 32, CV_8U); for (int i = 0; i < n; i++) memcpy(all.ptr(i), features[i].data, 32); }
    transform(all, v, fv, levelsup);
}

void ORBVocabulary::ComputeBoW(ORBextractor& extractor, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup) const
{
    v.clear(); fv.clear();
    orbhip_ctx* ctx = extractor.Context();
    if (!mpVoc || !ctx) return;
    const int n = orbhip_keypoint_capacity(ctx);
    std::vector<unsigned int> bowId(n), fvNode(n), fvFeat(n); std::vector<double> bowVal(n); std::vector<int> fvOff(n + 1);
    int nbow = 0, nfv = 0;
    if (orbhip_compute_bow(ctx, mpVoc, 1, levelsup) != ORBHIP_OK ||
        orbhip_fetch_bow(ctx, mpVoc, 0, &bowId[0], &bowVal[0], &nbow, &fvNode[0], &fvOff[0], &fvFeat[0], &nfv) != ORBHIP_OK) throw ORBhipError(std::string("ORBVocabulary::ComputeBoW: ") + orbhip_last_error());
    Deliver(n, bowId, bowVal, nbow, fvNode, fvOff, fvFeat, nfv, v, fv);
}

double ORBVocabulary::score(const DBoW2::BowVector &a, const DBoW2::BowVector &b) const
{
    if (!mpVoc) return 0.0;
    std::vector<unsigned int> ia, ib; std::vector<double> va, vb;
    ia.reserve(a.size()); va.reserve(a.size()); ib.reserve(b.size()); vb.reserve(b.size());
    for (DBoW2::BowVector::const_iterator it = a.begin(); it != a.end(); ++it) { ia.push_back(it->first); va.push_back(it->second); }
    for (DBoW2::BowVector::const_iterator it = b.begin(); it != b.end(); ++it) { ib.push_back(it->first); vb.push_back(it->second); }
    return orbhip_voc_score(mpVoc, ia.empty() ? NULL : &ia[0], va.empty() ? NULL : &va[0], (int)ia.size(), ib.empty() ? NULL : &ib[0], vb.empty() ? NULL : &vb[0], (int)ib.size());
}

} //namespace ORB_SLAM
