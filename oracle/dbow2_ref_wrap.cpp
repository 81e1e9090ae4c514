// C API around the REFERENCE's own DBoW2 (Thirdparty/DBoW2/DBoW2/{TemplatedVocabulary.h,FORB.cpp,BowVector.cpp,
// FeatureVector.cpp,ScoringObject.cpp} + DUtils, compiled where they lie under /root/reference by oracle/Makefile into
// oracle/_ref/libdbow2_ref.so).  Used by tests/ and tests/golden/make_golden_bow.py to pin the BoW restatement
// (oracle/orb_oracle.cpp, bow_* functions) and the HIP path against the real reference code.  Test infrastructure only.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "DBoW2/FORB.h"
#include "DBoW2/TemplatedVocabulary.h"

typedef DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB> ORBVocabularyBase;   // include/ORBVocabulary.h:31-32
struct ORBVocabulary : ORBVocabularyBase { using ORBVocabularyBase::transform; };   // the per-feature overload is protected

static std::vector<cv::Mat> to_mats(const uint8_t* desc, int n)
{
    std::vector<cv::Mat> v(n);
    for (int i = 0; i < n; i++) { v[i].create(1, 32, CV_8U); memcpy(v[i].data, desc + (size_t)i * 32, 32); }   // Converter::toDescriptorVector
    return v;
}

extern "C" {

void* dbow2_ref_new() { return new ORBVocabulary(); }
void dbow2_ref_delete(void* h) { delete (ORBVocabulary*)h; }
int dbow2_ref_load_text(void* h, const char* path) { return ((ORBVocabulary*)h)->loadFromTextFile(path) ? 1 : 0; }   // System.cc:68
void dbow2_ref_save_text(void* h, const char* path) { ((ORBVocabulary*)h)->saveToTextFile(path); }
// TemplatedVocabulary::create (k-means++ on the training descriptors; DUtils::Random seeded by the caller)
void dbow2_ref_create(void* h, const uint8_t* desc, const int* counts, int nimages, int k, int L, int weighting, int scoring, int seed)
{
    DUtils::Random::SeedRandOnce(seed);
    std::vector<std::vector<cv::Mat> > feats(nimages);
    size_t off = 0;
    for (int i = 0; i < nimages; i++) { feats[i] = to_mats(desc + off * 32, counts[i]); off += counts[i]; }
    ((ORBVocabulary*)h)->create(feats, k, L, (DBoW2::WeightingType)weighting, (DBoW2::ScoringType)scoring);
}
int dbow2_ref_size(void* h) { return (int)((ORBVocabulary*)h)->size(); }
// per-feature transform(feature, word_id, weight, &nid, levelsup)  (TemplatedVocabulary.h:1218-1262)
void dbow2_ref_transform_features(void* h, const uint8_t* desc, int n, int levelsup, uint32_t* word, double* weight, uint32_t* node)
{
    std::vector<cv::Mat> f = to_mats(desc, n);
    for (int i = 0; i < n; i++) { DBoW2::WordId w; DBoW2::WordValue v; DBoW2::NodeId nid = 0; ((ORBVocabulary*)h)->transform(f[i], w, v, &nid, levelsup); word[i] = w; weight[i] = v; node[i] = nid; }
}
// transform(features, BowVector, FeatureVector, levelsup)  (TemplatedVocabulary.h:1127-1194), flattened in map order.
// Returns the BowVector size; *nfv = FeatureVector size; fv_off has *nfv + 1 entries into fv_feat.
int dbow2_ref_transform(void* h, const uint8_t* desc, int n, int levelsup, uint32_t* bow_id, double* bow_val, int* nfv,
                        uint32_t* fv_node, int* fv_off, uint32_t* fv_feat)
{
    std::vector<cv::Mat> f = to_mats(desc, n);
    DBoW2::BowVector bv; DBoW2::FeatureVector fv;
    ((ORBVocabulary*)h)->transform(f, bv, fv, levelsup);
    int m = 0;
    for (DBoW2::BowVector::const_iterator it = bv.begin(); it != bv.end(); ++it, ++m) { bow_id[m] = it->first; bow_val[m] = it->second; }
    int q = 0, o = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++q) {
        fv_node[q] = it->first; fv_off[q] = o;
        for (size_t j = 0; j < it->second.size(); j++) fv_feat[o++] = it->second[j];
    }
    fv_off[q] = o; *nfv = q;
    return m;
}
// score(v1, v2) through the vocabulary's scoring object (KeyFrameDatabase.cc, LoopClosing.cc call sites)
double dbow2_ref_score(void* h, const uint32_t* id1, const double* val1, int n1, const uint32_t* id2, const double* val2, int n2)
{
    DBoW2::BowVector a, b;
    for (int i = 0; i < n1; i++) a.insert(a.end(), std::make_pair(id1[i], val1[i]));
    for (int i = 0; i < n2; i++) b.insert(b.end(), std::make_pair(id2[i], val2[i]));
    return ((ORBVocabulary*)h)->score(a, b);
}
int dbow2_ref_distance(const uint8_t* a, const uint8_t* b)      // FORB::distance (FORB.cpp:81-101)
{
    cv::Mat ma(1, 32, CV_8U), mb(1, 32, CV_8U); memcpy(ma.data, a, 32); memcpy(mb.data, b, 32);
    return DBoW2::FORB::distance(ma, mb);
}

}  // extern "C"
