// bow_oracle.cpp — CPU restatement of the DBoW2 vocabulary path ORB_SLAM2 runs on every frame it needs a bag of words for
// (Frame::ComputeBoW, Frame.cc:395-402; KeyFrame::ComputeBoW): TemplatedVocabulary::loadFromTextFile, ::transform and the
// scoring objects, on flat arrays instead of std::map / cv::Mat.  SURVEY.md §8(f)-3.
//
// TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// PARITY PINNED for this row: unlike the OpenCV-dependent extractor, the reference's own DBoW2 sources compile here
// (oracle/_ref/libdbow2_ref.so, `make -C oracle ref`), and tests/test_bow.py checks this restatement against them on
// vocabularies trained by the reference's own TemplatedVocabulary::create.
//
// One declared canonicalisation (H6): loadFromTextFile loops `while(!f.eof())` (TemplatedVocabulary.h:1378), so a file that
// ends with a newline (saveToTextFile writes one, :1429-1450) yields one extra iteration on an empty line whose `>>`
// extractions all fail and leave `pid` / `nIsLeaf` UNINITIALISED — the reference then appends a phantom node whose parent,
// leaf flag and descriptor are whatever the stack and heap held.  Here (and in the HIP loader) blank lines are ignored.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace bow {

enum Weighting { TF_IDF = 0, TF = 1, IDF = 2, BINARY = 3 };                                   // BowVector.h:36-42
enum Scoring { L1_NORM = 0, L2_NORM = 1, CHI_SQUARE = 2, KL = 3, BHATTACHARYYA = 4, DOT_PRODUCT = 5 };   // BowVector.h:45-53

struct Node { int parent = 0; std::vector<int> children; uint8_t desc[32]; double weight = 0; int word_id = -1; bool leaf_flag = false; };

struct Vocabulary {
    int k = 0, L = 0, scoring = 0, weighting = 0;
    std::vector<Node> nodes;
    std::vector<int> words;           // node id of word w
};

// FORB::distance (FORB.cpp:81-101): SWAR popcount over 8 int32 words
static int distance(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y; memcpy(&x, a + 4 * i, 4); memcpy(&y, b + 4 * i, 4);
        uint32_t v = x ^ y;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

// TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1338-1425).  Header `k L scoring weighting`, then one node per
// line `parent isLeaf b0 .. b31 weight`; node ids are line numbers (root = 0 is implicit), word ids count the leaves in order.
static bool load_text(Vocabulary& V, const char* path)
{
    std::ifstream f(path);
    if (!f.is_open()) return false;
    V.nodes.clear(); V.words.clear();
    std::string s;
    std::getline(f, s);
    std::stringstream ss; ss << s;
    int n1 = -1, n2 = -1; V.k = -1; V.L = -1;
    ss >> V.k; ss >> V.L; ss >> n1; ss >> n2;
    if (V.k < 0 || V.k > 20 || V.L < 1 || V.L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) return false;    // :1358-1362
    V.scoring = n1; V.weighting = n2;
    V.nodes.resize(1);
    while (std::getline(f, s)) {
        if (s.find_first_not_of(" \t\r\n") == std::string::npos) continue;       // H6
        std::stringstream sn; sn << s;
        const int nid = (int)V.nodes.size();
        V.nodes.resize(nid + 1);
        int pid = 0, leaf = 0;
        sn >> pid;
        if (pid < 0 || pid >= nid) return false;                                   // the reference would index out of bounds
        V.nodes[nid].parent = pid; V.nodes[pid].children.push_back(nid);
        sn >> leaf;
        for (int i = 0; i < 32; i++) { int b = 0; sn >> b; V.nodes[nid].desc[i] = (uint8_t)b; }      // FORB::fromString (FORB.cpp:120-135)
        sn >> V.nodes[nid].weight;
        if (leaf > 0) { V.nodes[nid].word_id = (int)V.words.size(); V.nodes[nid].leaf_flag = true; V.words.push_back(nid); }
    }
    return true;
}

// transform(feature, word_id, weight, nid, levelsup) (TemplatedVocabulary.h:1218-1262); isLeaf() == children.empty()
static void transform_feature(const Vocabulary& V, const uint8_t* feat, int levelsup, uint32_t& word, double& weight, uint32_t& nid)
{
    const int nid_level = V.L - levelsup;
    nid = 0;                                          // root if nid_level <= 0 (and otherwise overwritten on the way down)
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const std::vector<int>& ch = V.nodes[final_id].children;
        final_id = ch[0];
        int best = distance(feat, V.nodes[final_id].desc);
        for (size_t c = 1; c < ch.size(); c++) {
            const int d = distance(feat, V.nodes[ch[c]].desc);
            if (d < best) { best = d; final_id = ch[c]; }                          // strict: the first minimum wins
        }
        if (current_level == nid_level) nid = (uint32_t)final_id;
    } while (!V.nodes[final_id].children.empty());
    word = (uint32_t)V.nodes[final_id].word_id; weight = V.nodes[final_id].weight;
}

struct Bow { std::vector<uint32_t> id; std::vector<double> val; };
struct FeatVec { std::vector<uint32_t> node; std::vector<int> off; std::vector<uint32_t> feat; };

// transform(features, BowVector, FeatureVector, levelsup) (TemplatedVocabulary.h:1127-1194) + BowVector::addWeight /
// addIfNotExist / normalize (BowVector.cpp:34-94) + FeatureVector::addFeature (FeatureVector.cpp:29-43); std::map order = ascending key
static void transform(const Vocabulary& V, const uint8_t* desc, int n, int levelsup, Bow& bv, FeatVec& fv)
{
    bv.id.clear(); bv.val.clear(); fv.node.clear(); fv.off.clear(); fv.feat.clear();
    fv.off.push_back(0);
    if (V.words.empty() || V.nodes.size() <= 1) return;
    struct E { uint32_t key; uint32_t idx; double w; };
    std::vector<E> bw, nf;
    for (int i = 0; i < n; i++) {
        uint32_t w, nd; double wt;
        transform_feature(V, desc + (size_t)i * 32, levelsup, w, wt, nd);
        if (wt > 0) { bw.push_back({w, (uint32_t)i, wt}); nf.push_back({nd, (uint32_t)i, wt}); }      // "not stopped"
    }
    auto by_key = [](const E& a, const E& b) { return a.key != b.key ? a.key < b.key : a.idx < b.idx; };
    std::sort(bw.begin(), bw.end(), by_key); std::sort(nf.begin(), nf.end(), by_key);
    const bool accumulate = V.weighting == TF || V.weighting == TF_IDF;
    for (size_t i = 0; i < bw.size();) {
        size_t j = i; double v = bw[i].w;
        for (j = i + 1; j < bw.size() && bw[j].key == bw[i].key; j++) if (accumulate) v += bw[j].w;   // insert, then `+= v` in feature order
        bv.id.push_back(bw[i].key); bv.val.push_back(v);
        i = j;
    }
    const bool must = V.scoring != DOT_PRODUCT;                                   // ScoringObject.h:73-90
    const bool l2 = V.scoring == L2_NORM;
    if (accumulate && !bv.id.empty() && !must) { const double nd = (double)bv.id.size(); for (double& v : bv.val) v /= nd; }
    if (must) {
        double norm = 0.0;
        if (!l2) for (double v : bv.val) norm += std::fabs(v);
        else { for (double v : bv.val) norm += v * v; norm = std::sqrt(norm); }
        if (norm > 0.0) for (double& v : bv.val) v /= norm;
    }
    for (size_t i = 0; i < nf.size();) {
        size_t j = i;
        fv.node.push_back(nf[i].key);
        for (; j < nf.size() && nf[j].key == nf[i].key; j++) fv.feat.push_back(nf[j].idx);
        fv.off.push_back((int)fv.feat.size());
        i = j;
    }
}

static const double LOG_EPS = std::log(2.220446049250313e-16);                    // ScoringObject.cpp:18: log(DBL_EPSILON)

// ScoringObject.cpp:24-313 on two ascending (id, value) arrays; lower_bound on a std::map == advancing to the first id >= key
static double score(int scoring, const uint32_t* i1, const double* v1, int n1, const uint32_t* i2, const double* v2, int n2)
{
    int a = 0, b = 0; double s = 0;
    auto seek = [](const uint32_t* ids, int n, int from, uint32_t key) { return (int)(std::lower_bound(ids + from, ids + n, key) - ids); };
    while (a < n1 && b < n2) {
        const double vi = v1[a], wi = v2[b];
        if (i1[a] == i2[b]) {
            switch (scoring) {
            case L1_NORM: s += std::fabs(vi - wi) - std::fabs(vi) - std::fabs(wi); break;
            case L2_NORM: case DOT_PRODUCT: s += vi * wi; break;
            case CHI_SQUARE: if (vi + wi != 0.0) s += vi * wi / (vi + wi); break;
            case KL: if (vi != 0 && wi != 0) s += vi * std::log(vi / wi); break;
            case BHATTACHARYYA: s += std::sqrt(vi * wi); break;
            }
            a++; b++;
        } else if (i1[a] < i2[b]) {
            if (scoring == KL) { s += vi * (std::log(vi) - LOG_EPS); a++; }
            else a = seek(i1, n1, a, i2[b]);
        } else b = seek(i2, n2, b, i1[a]);
    }
    switch (scoring) {
    case L1_NORM: return -s / 2.0;
    case L2_NORM: return s >= 1 ? 1.0 : 1.0 - std::sqrt(1.0 - s);
    case CHI_SQUARE: return 2. * s;
    case KL: for (; a < n1; a++) if (v1[a] != 0) s += v1[a] * (std::log(v1[a]) - LOG_EPS); return s;
    default: return s;
    }
}

// ORBmatcher::ComputeThreeMaxima (ORBmatcher.cc:1601-1642) on bin sizes
static void three_maxima(const int* size, int L, int& ind1, int& ind2, int& ind3)
{
    int max1 = 0, max2 = 0, max3 = 0; ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = size[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

// ORBmatcher::SearchByBoW on flat data.  mode 0 = (KeyFrame*, Frame&, vpMapPointMatches) ORBmatcher.cc:159-288,
// mode 1 = (KeyFrame*, KeyFrame*, vpMatches12) :522-655.  Side 1 = the key frame whose map points are handed over
// (valid1[i] = "vpMapPoints1[i] exists and is not bad"), side 2 = the frame / second key frame (valid2 only in mode 1).
// match12[i1] = index on side 2 (the caller turns it into vpMapPointMatches[match12[i1]] = vpMapPointsKF[i1] resp.
// vpMatches12[i1] = vpMapPoints2[match12[i1]]); returns nmatches.
static int search_by_bow(int mode, const uint8_t* d1, const float* ang1, const uint8_t* valid1, int n1,
                         const uint32_t* fn1, const int* fo1, const uint32_t* ff1, int nf1,
                         const uint8_t* d2, const float* ang2, const uint8_t* valid2, int n2,
                         const uint32_t* fn2, const int* fo2, const uint32_t* ff2, int nf2,
                         float nnratio, bool check_ori, int* match12)
{
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    std::vector<uint8_t> matched2(n2, 0);
    std::vector<int> bin_of(n1, -1);
    int hist[HISTO_LENGTH] = {0};
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0, a = 0, b = 0;
    while (a < nf1 && b < nf2) {
        if (fn1[a] == fn2[b]) {
            for (int i1 = fo1[a]; i1 < fo1[a + 1]; i1++) {
                const unsigned idx1 = ff1[i1];
                if (!valid1[idx1]) continue;                                       // !pMP || pMP->isBad()
                int best1 = 256, bestIdx = -1, best2 = 256;
                for (int i2 = fo2[b]; i2 < fo2[b + 1]; i2++) {
                    const unsigned idx2 = ff2[i2];
                    if (matched2[idx2]) continue;                                  // vpMapPointMatches[realIdxF] / vbMatched2[idx2]
                    if (mode == 1 && !valid2[idx2]) continue;
                    const int dist = distance(d1 + (size_t)idx1 * 32, d2 + (size_t)idx2 * 32);
                    if (dist < best1) { best2 = best1; best1 = dist; bestIdx = (int)idx2; }
                    else if (dist < best2) best2 = dist;
                }
                const bool close = mode == 0 ? best1 <= TH_LOW : best1 < TH_LOW;   // :221 vs :588
                if (close && (float)best1 < nnratio * (float)best2) {
                    match12[idx1] = bestIdx; matched2[bestIdx] = 1;
                    if (check_ori) {
                        float rot = ang1[idx1] - ang2[bestIdx];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        bin_of[idx1] = bin; hist[bin]++;
                    }
                    nmatches++;
                }
            }
            a++; b++;
        } else if (fn1[a] < fn2[b]) a = (int)(std::lower_bound(fn1 + a, fn1 + nf1, fn2[b]) - fn1);
        else b = (int)(std::lower_bound(fn2 + b, fn2 + nf2, fn1[a]) - fn2);
    }
    if (check_ori) {
        int ind1, ind2, ind3;
        three_maxima(hist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < n1; i++)
            if (bin_of[i] >= 0 && bin_of[i] != ind1 && bin_of[i] != ind2 && bin_of[i] != ind3) { match12[i] = -1; nmatches--; }
    }
    return nmatches;
}

// ORBmatcher::SearchForTriangulation (ORBmatcher.cc:657-823) + CheckDistEpipolarLine (:140-157) on flat data.  Side 1 / 2 = the
// two key frames: has_mp = the feature already has a map point (skipped), stereo = mvuRight >= 0.  kp = (x, y, angle, octave) of
// mvKeysUn.  (ex, ey) = the epipole the caller computed from the two poses (:663-669), F12 row-major.  Note: the reference never
// sets vbMatched2 (:729 reads it, nothing writes it), so a feature of side 2 may be handed to several features of side 1.
static int search_for_triangulation(const uint8_t* d1, const float* kp1 /*n1 x 4*/, const uint8_t* has1, const uint8_t* st1, int n1,
                                    const uint32_t* fn1, const int* fo1, const uint32_t* ff1, int nf1,
                                    const uint8_t* d2, const float* kp2 /*n2 x 4*/, const uint8_t* has2, const uint8_t* st2, int n2,
                                    const uint32_t* fn2, const int* fo2, const uint32_t* ff2, int nf2,
                                    const float* F12, float ex, float ey, const float* scale2, const float* sigma2_2, bool only_stereo, bool check_ori, int* match12)
{
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    std::vector<int> bin_of(n1, -1);
    int hist[HISTO_LENGTH] = {0};
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0, a = 0, b = 0;
    while (a < nf1 && b < nf2) {
        if (fn1[a] == fn2[b]) {
            for (int i1 = fo1[a]; i1 < fo1[a + 1]; i1++) {
                const unsigned idx1 = ff1[i1];
                if (has1[idx1]) continue;                                          // "If there is already a MapPoint skip"
                const bool bStereo1 = st1[idx1] != 0;
                if (only_stereo && !bStereo1) continue;
                const float x1 = kp1[4 * idx1], y1 = kp1[4 * idx1 + 1];
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int i2 = fo2[b]; i2 < fo2[b + 1]; i2++) {
                    const unsigned idx2 = ff2[i2];
                    if (has2[idx2]) continue;                                      // vbMatched2 is never set by the reference
                    const bool bStereo2 = st2[idx2] != 0;
                    if (only_stereo && !bStereo2) continue;
                    const int dist = distance(d1 + (size_t)idx1 * 32, d2 + (size_t)idx2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const float x2 = kp2[4 * idx2], y2 = kp2[4 * idx2 + 1]; const int oct2 = (int)kp2[4 * idx2 + 3];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - x2, distey = ey - y2;
                        if (distex * distex + distey * distey < 100 * scale2[oct2]) continue;
                    }
                    const float la = x1 * F12[0] + y1 * F12[3] + F12[6];            // l = x1' F12 = [a b c]
                    const float lb = x1 * F12[1] + y1 * F12[4] + F12[7];
                    const float lc = x1 * F12[2] + y1 * F12[5] + F12[8];
                    const float num = la * x2 + lb * y2 + lc;
                    const float den = la * la + lb * lb;
                    if (den == 0) continue;
                    const float dsqr = num * num / den;
                    if (dsqr < 3.84 * sigma2_2[oct2]) { bestIdx2 = (int)idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    match12[idx1] = bestIdx2; nmatches++;
                    if (check_ori) {
                        float rot = kp1[4 * idx1 + 2] - kp2[4 * bestIdx2 + 2];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        bin_of[idx1] = bin; hist[bin]++;
                    }
                }
            }
            a++; b++;
        } else if (fn1[a] < fn2[b]) a = (int)(std::lower_bound(fn1 + a, fn1 + nf1, fn2[b]) - fn1);
        else b = (int)(std::lower_bound(fn2 + b, fn2 + nf2, fn1[a]) - fn2);
    }
    if (check_ori) {
        int ind1, ind2, ind3;
        three_maxima(hist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < n1; i++)
            if (bin_of[i] >= 0 && bin_of[i] != ind1 && bin_of[i] != ind2 && bin_of[i] != ind3) { match12[i] = -1; nmatches--; }
    }
    return nmatches;
}

}  // namespace bow

extern "C" {

void* orb_oracle_voc_load(const char* path)
{
    bow::Vocabulary* V = new bow::Vocabulary();
    if (!bow::load_text(*V, path)) { delete V; return nullptr; }
    return V;
}
void orb_oracle_voc_free(void* h) { delete (bow::Vocabulary*)h; }
void orb_oracle_voc_info(void* h, int* out6)
{
    const bow::Vocabulary* V = (const bow::Vocabulary*)h;
    out6[0] = V->k; out6[1] = V->L; out6[2] = V->scoring; out6[3] = V->weighting; out6[4] = (int)V->nodes.size(); out6[5] = (int)V->words.size();
}
void orb_oracle_voc_transform_features(void* h, const uint8_t* desc, int n, int levelsup, uint32_t* word, double* weight, uint32_t* node)
{
    const bow::Vocabulary* V = (const bow::Vocabulary*)h;
    for (int i = 0; i < n; i++) bow::transform_feature(*V, desc + (size_t)i * 32, levelsup, word[i], weight[i], node[i]);
}
int orb_oracle_voc_transform(void* h, const uint8_t* desc, int n, int levelsup, uint32_t* bow_id, double* bow_val, int* nfv,
                             uint32_t* fv_node, int* fv_off, uint32_t* fv_feat)
{
    bow::Bow bv; bow::FeatVec fv;
    bow::transform(*(const bow::Vocabulary*)h, desc, n, levelsup, bv, fv);
    for (size_t i = 0; i < bv.id.size(); i++) { bow_id[i] = bv.id[i]; bow_val[i] = bv.val[i]; }
    for (size_t i = 0; i < fv.node.size(); i++) fv_node[i] = fv.node[i];
    for (size_t i = 0; i < fv.off.size(); i++) fv_off[i] = fv.off[i];
    for (size_t i = 0; i < fv.feat.size(); i++) fv_feat[i] = fv.feat[i];
    *nfv = (int)fv.node.size();
    return (int)bv.id.size();
}
double orb_oracle_voc_score(int scoring, const uint32_t* i1, const double* v1, int n1, const uint32_t* i2, const double* v2, int n2)
{
    return bow::score(scoring, i1, v1, n1, i2, v2, n2);
}
int orb_oracle_forb_distance(const uint8_t* a, const uint8_t* b) { return bow::distance(a, b); }
int orb_oracle_search_for_triangulation(const uint8_t* d1, const float* kp1, const uint8_t* has1, const uint8_t* st1, int n1,
                                        const uint32_t* fn1, const int* fo1, const uint32_t* ff1, int nf1,
                                        const uint8_t* d2, const float* kp2, const uint8_t* has2, const uint8_t* st2, int n2,
                                        const uint32_t* fn2, const int* fo2, const uint32_t* ff2, int nf2,
                                        const float* F12, float ex, float ey, const float* scale2, const float* sigma2_2, int only_stereo, int check_ori, int* match12)
{
    return bow::search_for_triangulation(d1, kp1, has1, st1, n1, fn1, fo1, ff1, nf1, d2, kp2, has2, st2, n2, fn2, fo2, ff2, nf2, F12, ex, ey, scale2, sigma2_2,
                                         only_stereo != 0, check_ori != 0, match12);
}
int orb_oracle_search_by_bow(int mode, const uint8_t* d1, const float* ang1, const uint8_t* valid1, int n1,
                             const uint32_t* fn1, const int* fo1, const uint32_t* ff1, int nf1,
                             const uint8_t* d2, const float* ang2, const uint8_t* valid2, int n2,
                             const uint32_t* fn2, const int* fo2, const uint32_t* ff2, int nf2,
                             float nnratio, int check_ori, int* match12)
{
    return bow::search_by_bow(mode, d1, ang1, valid1, n1, fn1, fo1, ff1, nf1, d2, ang2, valid2, n2, fn2, fo2, ff2, nf2, nnratio, check_ori != 0, match12);
}

}  // extern "C"
