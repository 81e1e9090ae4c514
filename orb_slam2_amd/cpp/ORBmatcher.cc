// ORBmatcher.cc — the drop-in ORB_SLAM2::ORBmatcher: ALL twelve public members of the reference's class (include/ORBmatcher.h:37-102 there) as this
// repository's own code, each one a gather of what the member reads, ONE call of the C ABI (include/orbhip.h), and the member's write-back / map surgery.
//
// What runs where:
//   * on the device (liborbhip.so): every Hamming distance, every window / vocabulary-node search with its order-dependent bookkeeping and rotation
//     histogram, and - new in round 6 - the per-point PROJECTION of the five pose-guided members: world point -> camera -> depth / bounds / distance /
//     viewing-angle gates -> MapPoint::PredictScale -> radius (orbhip_project_search_*, orbhip_project_best_in_window_*), bit for bit what the
//     reference's cv::Mat statements compute (ORBmatcher.cc:316-362, 850-892, 1004-1051, 1154-1191, 1234-1271, 1353-1395, 1490-1528);
//   * here: the MapPoint / KeyFrame / Frame reads (one flat record per map point), the once-per-call pose algebra - kept as the reference's own cv::Mat
//     expressions so that whatever OpenCV the tree links rounds them its way - and the map surgery (Replace / AddObservation / AddMapPoint, vpMatched, ...).
//
// The gathers read MapPoint's position, normal, distance range and descriptor under the point's own mutexes WITHOUT the clones of GetWorldPos() /
// GetNormal() / GetDescriptor(), and mfMaxDistance itself (MapPoint::PredictScale's numerator has no accessor): integration/apply_dropin.py adds ONE line,
// `friend class ORBmatcher;`, to include/MapPoint.h (and defines ORBHIP_MAPPOINT_FRIEND there).  Without it this file does not compile - on purpose.
//
// How `Rcw*p3Dw+tcw` rounds depends on the OpenCV build: the first projection of a process PROBES the linked cv::Mat with 1024 random
// triples and picks the device form that reproduces all of them (generic cv::gemm kernel / OpenCV's small-matrix path); if neither does, the transform
// stays on the host - the member's own cv::Mat expression per point - and the device starts from the camera-frame point (gemm_mode 2).
// ORBHIP_GEMM_MODE=0|1|2 overrides the probe.
//
// Without ORBHIP_USE_OPENCV (this repository's own C++ tests, which have no map types) only the map-free members are compiled: the constructor,
// DescriptorDistance, SearchForInitialization.
#include "ORBmatcher.h"
#include "orbhip.h"
#ifndef ORBHIP_USE_OPENCV
#include "Frame.h"          // the caller's Frame (tests/cpp/Frame.h stands in for the reference's include/Frame.h here)
#else
#include "ORBmatcherBatch.h"
#include "ORBextractor.h"
#endif
#include "ORBextractor.h"   // ORBhipError
#include "orbhip_gemm_probe.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

using namespace std;

namespace ORB_SLAM2
{

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

static int orbhip_default_device() { static const int d = getenv("ORBHIP_DEVICE") ? atoi(getenv("ORBHIP_DEVICE")) : 0; return d; }      // read once per process
// failures are thrown like the extractor's (include/ORBextractor.h): never swallowed, never abort()
static void orbhip_check(orbhip_status st, const char* who = "ORBmatcher") { if(st!=ORBHIP_OK) throw ORBhipError(std::string(who) + ": " + orbhip_last_error()); }

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b)
{
    return orbhip_descriptor_distance(a.ptr<unsigned char>(), b.ptr<unsigned char>());
}

// ORBmatcher.cc:405-520.  The Frame members read are the ones the reference reads: mvKeysUn, mDescriptors, the static image bounds (Frame.h:120-190).
int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize)
{
    const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
    vnMatches12 = std::vector<int>(n1, -1);
    if (n1 == 0) return 0;
    const orbhip_bounds bounds = {Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};      // the grid of GetFeaturesInArea spans them (Frame.cc:327-346)
    int nmatches = 0;
#ifdef ORBHIP_USE_OPENCV
    const int device = F1.mpORBextractorLeft ? F1.mpORBextractorLeft->Device() : orbhip_default_device();
#else
    const int device = orbhip_default_device();
#endif
    orbhip_check(orbhip_search_for_initialization_bounds(
        device, reinterpret_cast<const orbhip_keypoint*>(&F1.mvKeysUn[0]), F1.mDescriptors.ptr<unsigned char>(), n1,
        n2 ? reinterpret_cast<const orbhip_keypoint*>(&F2.mvKeysUn[0]) : NULL, n2 ? F2.mDescriptors.ptr<unsigned char>() : NULL, n2,
        &bounds, reinterpret_cast<float*>(&vbPrevMatched[0]), &vnMatches12[0], windowSize, mfNNratio, mbCheckOrientation ? 1 : 0, &nmatches), "ORBmatcher::SearchForInitialization");
    return nmatches;
}

#if defined(ORBHIP_USE_OPENCV) && !defined(ORBHIP_MATCHER_MAP_FREE_ONLY)      // (MAP_FREE_ONLY: a tree that takes the extractor and the two map-free members only)
#ifndef ORBHIP_MAPPOINT_FRIEND
#error "include/MapPoint.h needs `friend class ORBmatcher;` (integration/apply_dropin.py adds it and defines ORBHIP_MAPPOINT_FRIEND): the matcher reads a map point's position, descriptor and mfMaxDistance in place"
#endif

// ================================================================================================ protected helpers of the reference's class
// Declared by the header (ORBmatcher.h:85-91 of the reference); the device carries its own versions, these serve subclasses and keep the class complete.
float ORBmatcher::RadiusByViewingCos(const float &viewCos) { return viewCos>0.998 ? 2.5f : 4.0f; }          // ORBmatcher.cc:131-137

bool ORBmatcher::CheckDistEpipolarLine(const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const cv::Mat &F12, const KeyFrame* pKF2)
{
    // l = x1' F12 = [a b c], squared distance of kp2 to it against the 3.84 chi-square bound at kp2's level (ORBmatcher.cc:140-157)
    float l[3];
    for(int c=0;c<3;c++) l[c] = kp1.pt.x*F12.at<float>(0,c)+kp1.pt.y*F12.at<float>(1,c)+F12.at<float>(2,c);
    const float num = l[0]*kp2.pt.x+l[1]*kp2.pt.y+l[2];
    const float den = l[0]*l[0]+l[1]*l[1];
    if(den==0) return false;
    return num*num/den < 3.84*pKF2->mvLevelSigma2[kp2.octave];
}

void ORBmatcher::ComputeThreeMaxima(vector<int>* histo, const int L, int &ind1, int &ind2, int &ind3)
{
    // the three fullest bins; the second / third are dropped when under a tenth of the first (ORBmatcher.cc:1601-1642)
    int m[3] = {0,0,0}; int ix[3] = {-1,-1,-1};
    for(int i=0;i<L;i++)
    {
        const int s = (int)histo[i].size();
        int k = 0;
        while(k<3 && s<=m[k]) k++;
        if(k==3) continue;
        for(int j=2;j>k;j--) { m[j]=m[j-1]; ix[j]=ix[j-1]; }
        m[k]=s; ix[k]=i;
    }
    if(m[1]<0.1f*(float)m[0]) { ix[1]=-1; ix[2]=-1; }
    else if(m[2]<0.1f*(float)m[0]) ix[2]=-1;
    ind1=ix[0]; ind2=ix[1]; ind3=ix[2];
}

// ================================================================================================ gathers
// What a member reads of a map point, read once: position / viewing direction / scale-invariance range under mMutexPos, descriptor under mMutexFeatures
// - the locks MapPoint's own accessors take (MapPoint.cc:80-92, 355-383, 330-334), without their cv::Mat clones.  ORBmatcher::Access is the nested
// type the one-line friend declaration in include/MapPoint.h reaches.
struct ORBmatcher::Access
{
    // Everything a member reads of one map point in ONE visit: mMutexFeatures then mMutexPos, the order MapPoint::isBad() takes them (MapPoint.cc:170-175) -
    // two lock operations per point where the reference's isBad() + GetWorldPos() + GetNormal() + Get{Min,Max}DistanceInvariance() + GetDescriptor() +
    // Observations() take eight, and no cv::Mat clone.  Returns false for a bad point (nothing else is read then).
    // pKFs / nKFs / inKFs (optional): bit t of *inKFs = IsInKeyFrame(pKFs[t]) (MapPoint.cc:254-258: mObservations under mMutexFeatures, which is held here
    // anyway) - Fuse's own filter (ORBmatcher.cc:848) for up to 64 key frames in the same visit instead of a lock per point and key frame
    static inline bool Read(MapPoint* pMP, orbhip_map_point &o, unsigned char* d32, bool bRange, bool bSkipBad, int &nObs,
                            KeyFrame* const* pKFs = NULL, int nKFs = 0, unsigned long long* inKFs = NULL)
    {
        unique_lock<mutex> lock1(pMP->mMutexFeatures);
        unique_lock<mutex> lock2(pMP->mMutexPos);
        if(bSkipBad && pMP->mbBad) return false;
        if(inKFs)
        {
            unsigned long long m = 0;
            for(int t=0; t<nKFs; t++) if(pMP->mObservations.count(pKFs[t])) m |= 1ull<<t;
            *inKFs = m;
            if(nKFs==1 && m) return true;          // a single Fuse skips the point: nothing else is read
        }
        nObs = pMP->nObs;
        memcpy(d32, pMP->mDescriptor.ptr<unsigned char>(), 32);
        const float* w = pMP->mWorldPos.ptr<float>();             // 3 x 1 CV_32F made by copyTo / clone: continuous
        const size_t ws = pMP->mWorldPos.step/sizeof(float);
        o.x = w[0]; o.y = w[ws]; o.z = w[2*ws];
        if(bRange)
        {
            const float* nv = pMP->mNormalVector.ptr<float>(); const size_t ns = pMP->mNormalVector.step/sizeof(float);
            o.nx = nv[0]; o.ny = nv[ns]; o.nz = nv[2*ns];
            o.min_dist = 0.8f*pMP->mfMinDistance;          // GetMinDistanceInvariance(), MapPoint.cc:373-377
            o.max_dist = 1.2f*pMP->mfMaxDistance;          // GetMaxDistanceInvariance(), MapPoint.cc:379-383
            o.scale_dist = pMP->mfMaxDistance;             // PredictScale's numerator, MapPoint.cc:390, 407
        }
        return true;
    }
    static inline void Descriptor(MapPoint* pMP, unsigned char* d32)
    {
        unique_lock<mutex> lock(pMP->mMutexFeatures);
        memcpy(d32, pMP->mDescriptor.ptr<unsigned char>(), 32);
    }
    // GetWorldPos() + GetNormal() + Get{Min,Max}DistanceInvariance() + PredictScale's numerator: one lock (Frame::isInFrustum's reads)
    static inline void Position(MapPoint* pMP, orbhip_map_point &o)
    {
        unique_lock<mutex> lock(pMP->mMutexPos);
        const float* w = pMP->mWorldPos.ptr<float>(); const size_t ws = pMP->mWorldPos.step/sizeof(float);
        const float* nv = pMP->mNormalVector.ptr<float>(); const size_t ns = pMP->mNormalVector.step/sizeof(float);
        o.x = w[0]; o.y = w[ws]; o.z = w[2*ws];
        o.nx = nv[0]; o.ny = nv[ns]; o.nz = nv[2*ns];
        o.min_dist = 0.8f*pMP->mfMinDistance; o.max_dist = 1.2f*pMP->mfMaxDistance; o.scale_dist = pMP->mfMaxDistance;
    }
    // isBad() + Observations() + GetDescriptor() of a point whose projection the caller already holds (Tracking::SearchLocalPoints): one lock.
    // (SetBadFlag writes mbBad under BOTH mutexes, MapPoint.cc:137-153: holding one of them excludes it.)
    static inline bool Tracked(MapPoint* pMP, unsigned char* d32, int &nObs)
    {
        unique_lock<mutex> lock(pMP->mMutexFeatures);
        if(pMP->mbBad) return false;
        nObs = pMP->nObs;
        memcpy(d32, pMP->mDescriptor.ptr<unsigned char>(), 32);
        return true;
    }
};
static inline void orbhip_read_descriptor(MapPoint* pMP, unsigned char* d32) { ORBmatcher::Access::Descriptor(pMP, d32); }
struct OrbhipPoints
{
    std::vector<orbhip_map_point> pts; std::vector<unsigned char> desc; std::vector<MapPoint*> owner; std::vector<int> index;
    size_t n;
    OrbhipPoints() : n(0) {}
    // room for `cap` points up front: add() writes in place, the vectors are cut to size by done()
    void reserve(size_t cap) { pts.resize(cap); desc.resize(32*cap); owner.resize(cap); index.resize(cap); n = 0; }
    size_t size() const { return n; }
    // level >= 0: given by the caller; -1: MapPoint::PredictScale on the device.  blocks < 0: Observations() > 0 of the point itself.  false: the point is bad (bSkipBad)
    // pKFs / nKFs / inKFs: see Access::Read; with ONE key frame a point that is in it is not added (returns false, *inKFs = 1)
    bool add(MapPoint* pMP, int idx, bool bRange, bool bSkipBad, int level, int blocks, float angle, KeyFrame* const* pKFs = NULL, int nKFs = 0, unsigned long long* inKFs = NULL)
    {
        if(n==pts.size()) { const size_t cap = std::max<size_t>(2*n, 64); pts.resize(cap); desc.resize(32*cap); owner.resize(cap); index.resize(cap); }
        orbhip_map_point &o = pts[n];
        int nObs = 0;
        if(!ORBmatcher::Access::Read(pMP, o, &desc[32*n], bRange, bSkipBad, nObs, pKFs, nKFs, inKFs)) return false;
        if(inKFs && nKFs==1 && *inKFs) return false;
        o.cam_x = o.cam_y = o.cam_z = 0.f;
        if(!bRange) { o.nx = o.ny = o.nz = o.min_dist = o.max_dist = o.scale_dist = 0.f; }
        o.level = level; o.blocks = blocks<0 ? (nObs>0) : blocks; o.angle = angle;
        owner[n] = pMP; index[n] = idx;
        n++;
        return true;
    }
    void done() { pts.resize(std::max<size_t>(n,1)); desc.resize(32*std::max<size_t>(n,1)); owner.resize(n); index.resize(n); }
};

// ---- how the linked cv::Mat rounds `R*x+t` (header of this file; include/orbhip_gemm_probe.h): probed once per process
int orbhip_gemm_mode() { static const int mode = orbhip_probe_gemm_mode<cv::Mat>(CV_32F); return mode; }        // (not static: the test scaffolding reports it)

// ---- MapPoint::PredictScale as a table of distance-ratio thresholds (include/orbhip.h): the expression below is MapPoint.cc:393 / :410 verbatim - `ratio`
// and the log scale factor are floats and this file, like MapPoint.cc, sees <cmath> under `using namespace std`, so log / ceil resolve to the same overloads
static int orbhip_level_of(float ratio, void* user)
{
    const std::pair<float,int> &L = *static_cast<const std::pair<float,int>*>(user);
    const float mfLogScaleFactor = L.first; const int mnScaleLevels = L.second;
    int nScale = ceil(log(ratio)/mfLogScaleFactor);
    if(nScale<0)
        nScale = 0;
    else if(nScale>=mnScaleLevels)
        nScale = mnScaleLevels-1;
    return nScale;
}
static void orbhip_level_table(float logScaleFactor, int nLevels, float* level_ratio)
{
    static std::mutex mtx; static std::map<std::pair<unsigned,int>, std::vector<float> > cache;
    unsigned bits; memcpy(&bits, &logScaleFactor, 4);
    unique_lock<mutex> lock(mtx);
    std::vector<float> &v = cache[std::make_pair(bits, nLevels)];
    if(v.empty())
    {
        v.resize(ORBHIP_MAX_PROJ_LEVELS);
        std::pair<float,int> spec(logScaleFactor, nLevels);
        orbhip_check(orbhip_predict_scale_table(orbhip_level_of, &spec, nLevels, &v[0]), "ORBmatcher (PredictScale table)");
    }
    memcpy(level_ratio, &v[0], sizeof(float)*ORBHIP_MAX_PROJ_LEVELS);
}

// the per-call part of a projection: pose, intrinsics, bounds, level tables
static void orbhip_set_pose(orbhip_projection &P, const cv::Mat &R, const cv::Mat &t)
{
    for(int r=0;r<3;r++) { for(int c=0;c<3;c++) P.R[3*r+c] = R.at<float>(r,c); P.t[r] = t.at<float>(r); }
}
static void orbhip_set_levels(orbhip_projection &P, const std::vector<float> &vScaleFactors, int nLevels, float logScaleFactor)
{
    if(nLevels<1 || nLevels>ORBHIP_MAX_PROJ_LEVELS || (int)vScaleFactors.size()<nLevels) throw ORBhipError("ORBmatcher: 1 to 16 pyramid levels are supported");
    P.nlevels = nLevels;
    for(int i=0;i<nLevels;i++) P.scale_factors[i] = vScaleFactors[i];
    orbhip_level_table(logScaleFactor, nLevels, P.level_ratio);
}
static orbhip_projection orbhip_projection_of(int kind, const cv::Mat &R, const cv::Mat &t, const cv::Mat &Ow, float fx, float fy, float cx, float cy, float bf,
                                              float minX, float minY, float maxX, float maxY, float th)
{
    orbhip_projection P; memset(&P, 0, sizeof P);
    P.kind = kind; P.gemm_mode = orbhip_gemm_mode();
    orbhip_set_pose(P, R, t);
    if(!Ow.empty()) for(int i=0;i<3;i++) P.Ow[i] = Ow.at<float>(i);
    P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.bf = bf;
    P.min_x = minX; P.min_y = minY; P.max_x = maxX; P.max_y = maxY; P.th = th;
    return P;
}
// gemm_mode 2: the member's own cv::Mat expression per point (the linked OpenCV rounds it its way), the device starts from the camera-frame point
static void orbhip_host_transform(OrbhipPoints &G, const cv::Mat &R, const cv::Mat &t, const cv::Mat* R2 = NULL, const cv::Mat* t2 = NULL)
{
    cv::Mat p3Dw(3,1,CV_32F);
    for(size_t k=0;k<G.pts.size();k++)
    {
        orbhip_map_point &o = G.pts[k];
        p3Dw.at<float>(0) = o.x; p3Dw.at<float>(1) = o.y; p3Dw.at<float>(2) = o.z;
        cv::Mat p3Dc = R*p3Dw+t;
        if(R2) { cv::Mat q = (*R2)*p3Dc+(*t2); p3Dc = q; }
        o.cam_x = p3Dc.at<float>(0); o.cam_y = p3Dc.at<float>(1); o.cam_z = p3Dc.at<float>(2);
    }
}
static orbhip_bounds orbhip_kf_bounds(KeyFrame* pKF) { orbhip_bounds b = {(float)pKF->mnMinX, (float)pKF->mnMinY, (float)pKF->mnMaxX, (float)pKF->mnMaxY}; return b; }
static const orbhip_bounds orbhip_frame_bounds() { orbhip_bounds b = {Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY}; return b; }

// A Frame whose features its extractor still holds in HBM (Tracking's calls on mCurrentFrame) is searched there: only queries / points travel.
static bool orbhip_resident(Frame &F, bool bUseRight, bool &right)
{
    ORBextractor* ex = F.mpORBextractorLeft;
    bool resident = ex && ex->HoldsFrame(F.mnId, F.N);
    right = bUseRight;
    if(resident && bUseRight && !ex->HoldsStereoColumns())
    {
        // no mvuRight on the device: a monocular frame (all -1: the right-coordinate test never fires) is searched there without it, a frame whose columns
        // exist on the host only (a ComputeStereoFromRGBD that did not hand them over) is searched through its host copies
        right = false;
        for(int i=0; i<F.N; i++)
            if(F.mvuRight[i]>0) { resident = false; break; }
    }
    return resident;
}

// ================================================================================================ Frame::isInFrustum   Frame.cc:269-325
// The member's statements in their order on flat floats (installed by apply_dropin.py --flat-frustum).  `mRcw*P+mtcw` is evaluated in the rounding the probe
// found in the linked cv::Mat (or by that cv::Mat itself, mode 2); cv::norm / Mat::dot accumulate in double like OpenCV's; MapPoint::PredictScale is the
// expression of MapPoint.cc:407-421 (orbhip_level_of above).  Everything else is the member's own float arithmetic under this file's compiler flags.
bool ORBmatcher::IsInFrustum(Frame &F, const cv::Mat &Rcw, const cv::Mat &tcw, const cv::Mat &Ow, MapPoint* pMP, float viewingCosLimit)
{
    pMP->mbTrackInView = false;
    orbhip_map_point o;
    Access::Position(pMP, o);
    const float P[3] = {o.x, o.y, o.z};
    float Pc[3];
    const int gm = orbhip_gemm_mode();
    if(gm==2)
    {
        cv::Mat Pm(3,1,CV_32F); Pm.at<float>(0) = P[0]; Pm.at<float>(1) = P[1]; Pm.at<float>(2) = P[2];
        const cv::Mat Pcm = Rcw*Pm+tcw;
        Pc[0] = Pcm.at<float>(0); Pc[1] = Pcm.at<float>(1); Pc[2] = Pcm.at<float>(2);
    }
    else
    {
        float R[9], t[3];
        for(int r=0;r<3;r++) { for(int c=0;c<3;c++) R[3*r+c] = Rcw.at<float>(r,c); t[r] = tcw.at<float>(r); }
        orbhip_flat_gemm(gm, R, P, t, Pc);
    }
    const float &PcX = Pc[0];
    const float &PcY = Pc[1];
    const float &PcZ = Pc[2];
    if(PcZ<0.0f)                                     // positive depth
        return false;
    const float invz = 1.0f/PcZ;
    const float u=F.fx*PcX*invz+F.cx;
    const float v=F.fy*PcY*invz+F.cy;
    if(u<Frame::mnMinX || u>Frame::mnMaxX)          // inside the image
        return false;
    if(v<Frame::mnMinY || v>Frame::mnMaxY)
        return false;
    const float maxDistance = o.max_dist;            // inside the scale-invariance range of the point
    const float minDistance = o.min_dist;
    const float PO[3] = {P[0]-Ow.at<float>(0), P[1]-Ow.at<float>(1), P[2]-Ow.at<float>(2)};
    double ss = 0; for(int k=0;k<3;k++) ss += (double)PO[k]*(double)PO[k];
    const float dist = std::sqrt(ss);               // cv::norm(PO)
    if(dist<minDistance || dist>maxDistance)
        return false;
    double dp = 0; { const float Pn[3] = {o.nx, o.ny, o.nz}; for(int k=0;k<3;k++) dp += (double)PO[k]*(double)Pn[k]; }
    const float viewCos = dp/dist;                   // PO.dot(Pn)/dist
    if(viewCos<viewingCosLimit)
        return false;
    std::pair<float,int> spec(F.mfLogScaleFactor, F.mnScaleLevels);
    const int nPredictedLevel = orbhip_level_of(o.scale_dist/dist, &spec);      // pMP->PredictScale(dist,this)
    pMP->mbTrackInView = true;                       // what Tracking::SearchLocalPoints / SearchByProjection read afterwards
    pMP->mTrackProjX = u;
    pMP->mTrackProjXR = u - F.mbf*invz;
    pMP->mTrackProjY = v;
    pMP->mnTrackScaleLevel= nPredictedLevel;
    pMP->mTrackViewCos = viewCos;
    return true;
}

// ================================================================================================ SearchByProjection(Frame, local map points)   ORBmatcher.cc:45-129
// The points were projected by Frame::isInFrustum (Tracking::SearchLocalPoints): nothing to project here; flat queries, same-level ratio rule on the device.
int ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th)
{
    int nmatches=0;
    const bool bFactor = th!=1.0;
    const size_t cap = std::max<size_t>(vpMapPoints.size(), 1);
    std::vector<orbhip_proj_query> q(cap); std::vector<unsigned char> qd(32*cap); std::vector<MapPoint*> owner(cap);
    size_t nq = 0;
    for(size_t iMP=0; iMP<vpMapPoints.size(); iMP++)
    {
        MapPoint* pMP = vpMapPoints[iMP];
        if(!pMP->mbTrackInView) continue;
        int nObs = 0;
        if(!Access::Tracked(pMP, &qd[32*nq], nObs)) continue;
        const int nPredictedLevel = pMP->mnTrackScaleLevel;
        float r = RadiusByViewingCos(pMP->mTrackViewCos);       // the window depends on the viewing direction
        if(bFactor) r*=th;
        const orbhip_proj_query e = { pMP->mTrackProjX, pMP->mTrackProjY, r*F.mvScaleFactors[nPredictedLevel], pMP->mTrackProjXR,
                                      nPredictedLevel-1, nPredictedLevel, nObs>0, 0.f };
        q[nq] = e; owner[nq] = pMP; nq++;
    }
    q.resize(nq);
    if(q.empty() || F.N==0) return 0;
    std::vector<unsigned char> blocked(F.N); std::vector<int> fq(F.N);
    for(int i=0;i<F.N;i++) blocked[i] = F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations()>0;
    bool right;
    if(orbhip_resident(F, true, right))
        orbhip_check(orbhip_search_by_projection_frame(F.mpORBextractorLeft->Context(), 0, F.N, right, &blocked[0], &q[0], &qd[0], (int)q.size(), 0, mfNNratio, TH_HIGH, 0, &fq[0], &nmatches));
    else
    {
        const orbhip_bounds bounds = orbhip_frame_bounds();
        orbhip_check(orbhip_search_by_projection_bounds(F.mpORBextractorLeft ? F.mpORBextractorLeft->Device() : orbhip_default_device(), (const orbhip_keypoint*)&F.mvKeysUn[0],
                                                        F.mDescriptors.ptr<unsigned char>(), &F.mvuRight[0], &blocked[0], F.N, &bounds, &q[0], &qd[0], (int)q.size(), 0, mfNNratio, TH_HIGH, 0, &fq[0], &nmatches));
    }
    for(int i=0;i<F.N;i++) if(fq[i]>=0) F.mvpMapPoints[i]=owner[fq[i]];
    return nmatches;
}

// the two Frame-searching projection members share their call: resident frame or host arrays, projection on the device
static int orbhip_project_into_frame(Frame &F, bool bUseRight, const std::vector<unsigned char> &blocked, orbhip_projection &P, OrbhipPoints &G, float nnratio, int thHigh, bool bCheckOri)
{
    int nmatches = 0;
    if(G.size()==0 || F.N==0) return 0;
    std::vector<int> fq(F.N);
    bool right;
    if(orbhip_resident(F, bUseRight, right))
        orbhip_check(orbhip_project_search_frame(F.mpORBextractorLeft->Context(), 0, F.N, right, &blocked[0], &P, &G.pts[0], &G.desc[0], (int)G.size(), nnratio, thHigh, bCheckOri, &fq[0], &nmatches, NULL));
    else
    {
        const orbhip_bounds bounds = orbhip_frame_bounds();
        orbhip_check(orbhip_project_search_bounds(F.mpORBextractorLeft ? F.mpORBextractorLeft->Device() : orbhip_default_device(), (const orbhip_keypoint*)&F.mvKeysUn[0], F.mDescriptors.ptr<unsigned char>(),
                                                  bUseRight ? &F.mvuRight[0] : NULL, &blocked[0], F.N, &bounds, &P, &G.pts[0], &G.desc[0], (int)G.size(), nnratio, thHigh, bCheckOri, &fq[0], &nmatches, NULL));
    }
    for(int i=0;i<F.N;i++)
    {
        if(fq[i]>=0) F.mvpMapPoints[i]=G.owner[fq[i]];
        else if(fq[i]==-2) F.mvpMapPoints[i]=static_cast<MapPoint*>(NULL);      // claimed, then removed by the rotation check (ORBmatcher.cc:1452-1466, 1581-1596)
    }
    return nmatches;
}

// ================================================================================================ SearchByProjection(Current, Last)   ORBmatcher.cc:1328-1470
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
{
    // once per call: the poses as the reference takes them apart (:1339-1349)
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0,3).colRange(0,3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0,3).col(3);
    const cv::Mat twc = -Rcw.t()*tcw;
    const cv::Mat Rlw = LastFrame.mTcw.rowRange(0,3).colRange(0,3);
    const cv::Mat tlw = LastFrame.mTcw.rowRange(0,3).col(3);
    const cv::Mat tlc = Rlw*twc+tlw;
    const bool bForward = tlc.at<float>(2)>CurrentFrame.mb && !bMono;
    const bool bBackward = -tlc.at<float>(2)>CurrentFrame.mb && !bMono;

    OrbhipPoints G; G.reserve(LastFrame.N);
    for(int i=0; i<LastFrame.N; i++)
    {
        MapPoint* pMP = LastFrame.mvpMapPoints[i];
        if(!pMP || LastFrame.mvbOutlier[i]) continue;
        G.add(pMP, i, false, false, LastFrame.mvKeys[i].octave, -1, LastFrame.mvKeysUn[i].angle);          // (no isBad() test in this member)
    }
    G.done();
    orbhip_projection P = orbhip_projection_of(ORBHIP_PROJ_LAST_FRAME, Rcw, tcw, cv::Mat(), CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy, CurrentFrame.mbf,
                                               Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY, th);
    P.forward = bForward; P.backward = bBackward;
    orbhip_set_levels(P, CurrentFrame.mvScaleFactors, CurrentFrame.mnScaleLevels, CurrentFrame.mfLogScaleFactor);
    if(P.gemm_mode==2) orbhip_host_transform(G, Rcw, tcw);
    std::vector<unsigned char> blocked(CurrentFrame.N);
    for(int i=0;i<CurrentFrame.N;i++) blocked[i] = CurrentFrame.mvpMapPoints[i] && CurrentFrame.mvpMapPoints[i]->Observations()>0;
    return orbhip_project_into_frame(CurrentFrame, true, blocked, P, G, mfNNratio, TH_HIGH, mbCheckOrientation);
}

// ================================================================================================ SearchByProjection(Current, KF, found)   ORBmatcher.cc:1472-1599
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th, const int ORBdist)
{
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0,3).colRange(0,3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0,3).col(3);
    const cv::Mat Ow = -Rcw.t()*tcw;

    const vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
    OrbhipPoints G; G.reserve(vpMPs.size());
    for(size_t i=0, iend=vpMPs.size(); i<iend; i++)
    {
        MapPoint* pMP = vpMPs[i];
        if(!pMP || sAlreadyFound.count(pMP)) continue;
        G.add(pMP, (int)i, true, true, -1, 1, pKF->mvKeysUn[i].angle);
    }
    G.done();
    orbhip_projection P = orbhip_projection_of(ORBHIP_PROJ_FRAME_KF, Rcw, tcw, Ow, CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy, CurrentFrame.mbf,
                                               Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY, th);
    orbhip_set_levels(P, CurrentFrame.mvScaleFactors, CurrentFrame.mnScaleLevels, CurrentFrame.mfLogScaleFactor);
    if(P.gemm_mode==2) orbhip_host_transform(G, Rcw, tcw);
    std::vector<unsigned char> blocked(CurrentFrame.N);
    for(int i=0;i<CurrentFrame.N;i++) blocked[i] = CurrentFrame.mvpMapPoints[i]!=NULL;          // any map point blocks (:1536-1537)
    return orbhip_project_into_frame(CurrentFrame, false, blocked, P, G, mfNNratio, ORBdist, mbCheckOrientation);
}

// ---- Scw -> Rcw | tcw | Ow as the two Sim3 members take it apart (ORBmatcher.cc:298-303, 982-987)
static void orbhip_decompose_sim3(const cv::Mat &Scw, cv::Mat &Rcw, cv::Mat &tcw, cv::Mat &Ow)
{
    cv::Mat sRcw = Scw.rowRange(0,3).colRange(0,3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    Rcw = sRcw/scw;
    tcw = Scw.rowRange(0,3).col(3)/scw;
    Ow = -Rcw.t()*tcw;
}

// ================================================================================================ SearchByProjection(KF, Scw, points, matched)   ORBmatcher.cc:290-403
int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th)
{
    cv::Mat Rcw, tcw, Ow;
    orbhip_decompose_sim3(Scw, Rcw, tcw, Ow);
    set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint*>(NULL));

    OrbhipPoints G; G.reserve(vpPoints.size());
    for(int iMP=0, iendMP=vpPoints.size(); iMP<iendMP; iMP++)
    {
        MapPoint* pMP = vpPoints[iMP];
        if(spAlreadyFound.count(pMP)) continue;
        G.add(pMP, iMP, true, true, -1, 1, 0.f);
    }
    G.done();
    int nmatches=0;
    if(G.size()==0 || pKF->N==0) return 0;
    const orbhip_bounds bounds = orbhip_kf_bounds(pKF);
    orbhip_projection P = orbhip_projection_of(ORBHIP_PROJ_KF_SIM3, Rcw, tcw, Ow, pKF->fx, pKF->fy, pKF->cx, pKF->cy, 0.f, bounds.min_x, bounds.min_y, bounds.max_x, bounds.max_y, (float)th);
    orbhip_set_levels(P, pKF->mvScaleFactors, pKF->mnScaleLevels, pKF->mfLogScaleFactor);
    if(P.gemm_mode==2) orbhip_host_transform(G, Rcw, tcw);
    std::vector<unsigned char> blocked(pKF->N); std::vector<int> fq(pKF->N);
    for(int i=0;i<pKF->N;i++) blocked[i] = vpMatched[i]!=NULL;
    orbhip_check(orbhip_project_search_bounds(orbhip_default_device(), (const orbhip_keypoint*)&pKF->mvKeysUn[0], pKF->mDescriptors.ptr<unsigned char>(), NULL, &blocked[0], pKF->N, &bounds,
                                              &P, &G.pts[0], &G.desc[0], (int)G.size(), mfNNratio, TH_LOW, 0, &fq[0], &nmatches, NULL));
    for(int i=0;i<pKF->N;i++) if(fq[i]>=0) vpMatched[i]=G.owner[fq[i]];
    return nmatches;
}

// ================================================================================================ Fuse(KF, points, th)   ORBmatcher.cc:825-972
// collect (the member's filters + gather) -> search (device: projection, gates, window, chi-square) -> apply (the member's surgery, in the original order).
// FuseBatch runs collect for every target, ONE search over all targets, then apply target by target.
struct OrbhipFuseJob
{
    KeyFrame* pKF; orbhip_projection P; OrbhipPoints G; std::vector<int> bi, bd;
    const OrbhipPoints* S;      // the points bi / bd answer for: &G, or FuseBatch's one shared set (orbhip_project_best_in_window_shared: the device skips the ones this target holds)
    OrbhipFuseJob() : pKF(NULL), S(NULL) {}
    const OrbhipPoints& pts() const { return S ? *S : G; }
};
// `shared`: the points read ONCE for every target of a FuseBatch (shared->index[k] = position in vpMapPoints; bad points are absent) - a target then copies
// the records of the points it does not hold instead of visiting every map point again; NULL: read here (a single Fuse)
// `sharedIn` (with `shared`): bit `t` of sharedIn[k] = shared point k is in target t (read in the same visit as the point), t = this job's number
static void orbhip_fuse_collect(KeyFrame* pKF, const vector<MapPoint*> &vpMapPoints, const float th, OrbhipFuseJob &job, const OrbhipPoints* shared = NULL,
                                const std::vector<unsigned long long>* sharedIn = NULL, int t = 0)
{
    job.pKF = pKF;
    cv::Mat Rcw = pKF->GetRotation();
    cv::Mat tcw = pKF->GetTranslation();
    cv::Mat Ow = pKF->GetCameraCenter();
    const orbhip_bounds b = orbhip_kf_bounds(pKF);
    job.P = orbhip_projection_of(ORBHIP_PROJ_FUSE, Rcw, tcw, Ow, pKF->fx, pKF->fy, pKF->cx, pKF->cy, pKF->mbf, b.min_x, b.min_y, b.max_x, b.max_y, th);
    orbhip_set_levels(job.P, pKF->mvScaleFactors, pKF->mnScaleLevels, pKF->mfLogScaleFactor);
    const int nMPs = vpMapPoints.size();
    job.G.reserve(shared ? shared->size() : nMPs);
    if(shared)
    {
        OrbhipPoints &G = job.G;
        for(size_t k=0; k<shared->size(); k++)
        {
            MapPoint* pMP = shared->owner[k];
            if(sharedIn ? (((*sharedIn)[k]>>t)&1ull)!=0 : pMP->IsInKeyFrame(pKF)) continue;
            G.pts[G.n] = shared->pts[k]; memcpy(&G.desc[32*G.n], &shared->desc[32*k], 32); G.owner[G.n] = pMP; G.index[G.n] = shared->index[k];
            G.n++;
        }
    }
    else
        for(int i=0; i<nMPs; i++)
        {
            MapPoint* pMP = vpMapPoints[i];
            if(!pMP) continue;
            unsigned long long in = 0;
            job.G.add(pMP, i, true, true, -1, 0, 0.f, &pKF, 1, &in);      // (:848-849: a bad point or one that is in the key frame already is not added)
        }
    job.G.done();
    if(job.P.gemm_mode==2) orbhip_host_transform(job.G, Rcw, tcw);
    job.bi.assign(job.G.size(), -1); job.bd.assign(job.G.size(), 256);
}
static orbhip_project_best_slot orbhip_fuse_slot(OrbhipFuseJob &job)
{
    KeyFrame* pKF = job.pKF;
    orbhip_project_best_slot S; memset(&S, 0, sizeof S);
    S.n = pKF->N;
    if(pKF->N>0) { S.kps = (const orbhip_keypoint*)&pKF->mvKeysUn[0]; S.desc = pKF->mDescriptors.ptr<unsigned char>(); S.u_right = &pKF->mvuRight[0]; }
    S.bounds = orbhip_kf_bounds(pKF); S.inv_level_sigma2 = &pKF->mvInvLevelSigma2[0]; S.nlevels = (int)pKF->mvInvLevelSigma2.size();
    S.proj = &job.P; S.np = (int)job.G.size();
    if(S.np>0) { S.points = &job.G.pts[0]; S.point_desc = &job.G.desc[0]; S.best_idx = &job.bi[0]; S.best_dist = &job.bd[0]; }
    return S;
}
// survivors: the points that absorbed another one in the surgery of EARLIER targets of the same FuseBatch (MapPoint::Replace recomputes the survivor's
// descriptor, MapPoint.cc:177-215) - NULL for a single Fuse, whose points were read just before its search: nothing to re-check
static int orbhip_fuse_apply(OrbhipFuseJob &job, std::set<MapPoint*> *survivors, int slot = -1)
{
    KeyFrame* pKF = job.pKF;
    const OrbhipPoints &Q = job.pts();      // (a shared set also lists the points this target held when they were read: the device answered 256 for them, the filters below skip them)
    int nFused=0;
    // Survivors whose descriptor is not the collected one any more are searched again, all of them in one call, before this target's surgery starts.
    // (Inside one target no collected point's descriptor changes: a survivor is either a point already handled or a point of this key frame, which the
    // filter below skips.)  Only survivors are looked at - nobody else's descriptor can have changed.
    if(survivors && !survivors->empty() && pKF->N>0)
    {
        OrbhipFuseJob again; again.pKF = pKF; again.P = job.P; std::vector<size_t> which;
        for(size_t k=0; k<Q.size(); k++)
        {
            MapPoint* pMP = Q.owner[k];
            if(!survivors->count(pMP) || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
            unsigned char dNow[32]; orbhip_read_descriptor(pMP, dNow);
            if(!memcmp(dNow, &Q.desc[32*k], 32)) continue;
            which.push_back(k); again.G.pts.push_back(Q.pts[k]); again.G.desc.insert(again.G.desc.end(), dNow, dNow+32); again.G.n++;
        }
        if(!which.empty())
        {
            again.bi.assign(which.size(), -1); again.bd.assign(which.size(), 256);
            // this target's key frame and grid are still on the device when the points travelled as one shared set (slot = the job's number): only the changed
            // points go up; otherwise (or when another call of this thread has used the scratch since) the whole search
            if(slot<0 || orbhip_project_best_in_window_held(orbhip_default_device(), slot, &again.P, &again.G.pts[0], &again.G.desc[0], (int)which.size(), 1, &again.bi[0], &again.bd[0])!=ORBHIP_OK)
            {
                orbhip_project_best_slot S = orbhip_fuse_slot(again);
                orbhip_check(orbhip_project_best_in_window_batch(orbhip_default_device(), 1, &S, 1));
            }
            for(size_t a=0; a<which.size(); a++) { job.bi[which[a]] = again.bi[a]; job.bd[which[a]] = again.bd[a]; }
        }
    }
    for(size_t k=0; k<Q.size(); k++)
    {
        const int bestDist = job.bd[k], bestIdx = job.bi[k];
        if(bestDist>ORBmatcher::TH_LOW) continue;          // (:945; first, it needs no lock: either test alone skips the point and neither has a side effect)
        MapPoint* pMP = Q.owner[k];
        if(pMP->isBad() || pMP->IsInKeyFrame(pKF))        // the member's own filter (:848-849), as of NOW: the surgery of earlier points / targets may have changed it
            continue;
        // a map point already there: the one with fewer observations is replaced by the other; otherwise a new measurement (:951-968)
        MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
        if(pMPinKF)
        {
            if(!pMPinKF->isBad())
            {
                if(pMPinKF->Observations()>pMP->Observations()) { pMP->Replace(pMPinKF); if(survivors) survivors->insert(pMPinKF); }
                else { pMPinKF->Replace(pMP); if(survivors) survivors->insert(pMP); }
            }
        }
        else
        {
            pMP->AddObservation(pKF,bestIdx);
            pKF->AddMapPoint(pMP,bestIdx);
        }
        nFused++;
    }
    return nFused;
}
int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint *> &vpMapPoints, const float th)
{
    OrbhipFuseJob job;
    orbhip_fuse_collect(pKF, vpMapPoints, th, job);
    if(job.G.size()>0 && pKF->N>0)
    {
        orbhip_project_best_slot S = orbhip_fuse_slot(job);
        orbhip_check(orbhip_project_best_in_window_batch(orbhip_default_device(), 1, &S, 1));
    }
    return orbhip_fuse_apply(job, NULL);
}
// LocalMapping::SearchInNeighbors (LocalMapping.cc:483-514): `for every target key frame: matcher.Fuse(pKFi, vpMapPointMatches)` as ONE device pass
int FuseBatch(const std::vector<KeyFrame*> &vpTargetKFs, const std::vector<MapPoint*> &vpMapPoints, const float th)
{
    std::vector<OrbhipFuseJob> jobs(vpTargetKFs.size());
    std::vector<orbhip_project_best_slot> slots(vpTargetKFs.size());
    // every target is offered the same points: each is visited once (position, range, descriptor as they are before the loop - what an earlier target's
    // surgery changes afterwards is re-checked per target by orbhip_fuse_apply), a target keeps the ones it does not hold
    OrbhipPoints shared; shared.reserve(vpMapPoints.size());
    // ... together with "is the point in target t" for every target (up to 64 per visit; more targets: asked per target as before)
    const bool bMasks = vpTargetKFs.size()>=2 && vpTargetKFs.size()<=64;
    std::vector<unsigned long long> sharedIn; if(bMasks) sharedIn.reserve(vpMapPoints.size());
    for(size_t i=0; i<vpMapPoints.size(); i++)
    {
        if(!vpMapPoints[i]) continue;
        unsigned long long in = 0;
        if(shared.add(vpMapPoints[i], (int)i, true, true, -1, 0, 0.f, bMasks ? &vpTargetKFs[0] : NULL, bMasks ? (int)vpTargetKFs.size() : 0, bMasks ? &in : NULL) && bMasks)
            sharedIn.push_back(in);
    }
    shared.done();
    // With the masks the points travel ONCE for all targets (orbhip_project_best_in_window_shared: the device leaves out, per target, the points whose bit is
    // set) - no per-target copy of the records either.  Not when the caller's own transform rides in the records (H11 mode 2: it differs per target).
    const bool bOneUpload = bMasks && shared.size()>0 && orbhip_gemm_mode()!=2;
    if(bOneUpload)
    {
        for(size_t t=0; t<vpTargetKFs.size(); t++)
        {
            OrbhipFuseJob &job = jobs[t]; KeyFrame* pKF = vpTargetKFs[t];
            job.pKF = pKF; job.S = &shared;
            cv::Mat Rcw = pKF->GetRotation();
            cv::Mat tcw = pKF->GetTranslation();
            cv::Mat Ow = pKF->GetCameraCenter();
            const orbhip_bounds b = orbhip_kf_bounds(pKF);
            job.P = orbhip_projection_of(ORBHIP_PROJ_FUSE, Rcw, tcw, Ow, pKF->fx, pKF->fy, pKF->cx, pKF->cy, pKF->mbf, b.min_x, b.min_y, b.max_x, b.max_y, th);
            orbhip_set_levels(job.P, pKF->mvScaleFactors, pKF->mnScaleLevels, pKF->mfLogScaleFactor);
            job.bi.assign(shared.size(), -1); job.bd.assign(shared.size(), 256);
        }
        for(size_t t=0; t<vpTargetKFs.size(); t++)
        {
            slots[t] = orbhip_fuse_slot(jobs[t]);
            slots[t].points = &shared.pts[0]; slots[t].point_desc = &shared.desc[0]; slots[t].np = (int)shared.size();
            slots[t].best_idx = &jobs[t].bi[0]; slots[t].best_dist = &jobs[t].bd[0];
        }
        orbhip_check(orbhip_project_best_in_window_shared(orbhip_default_device(), (int)slots.size(), &slots[0], (const uint64_t*)&sharedIn[0], 1));
    }
    else
    {
        for(size_t t=0; t<vpTargetKFs.size(); t++)
            orbhip_fuse_collect(vpTargetKFs[t], vpMapPoints, th, jobs[t], &shared, bMasks ? &sharedIn : NULL, (int)t);
        for(size_t t=0; t<vpTargetKFs.size(); t++) slots[t] = orbhip_fuse_slot(jobs[t]);      // (after every job exists: the slots point into them)
        if(!slots.empty())
            orbhip_check(orbhip_project_best_in_window_batch(orbhip_default_device(), (int)slots.size(), &slots[0], 1));
    }
    int nFused=0;
    std::set<MapPoint*> survivors;
    for(size_t t=0; t<jobs.size(); t++)
        nFused += orbhip_fuse_apply(jobs[t], &survivors, bOneUpload ? (int)t : -1);
    return nFused;
}

// ================================================================================================ Fuse(KF, Scw, points, th, replace)   ORBmatcher.cc:974-1100
int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint *> &vpPoints, float th, vector<MapPoint *> &vpReplacePoint)
{
    cv::Mat Rcw, tcw, Ow;
    orbhip_decompose_sim3(Scw, Rcw, tcw, Ow);
    const set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();

    OrbhipFuseJob job; job.pKF = pKF;
    const orbhip_bounds b = orbhip_kf_bounds(pKF);
    job.P = orbhip_projection_of(ORBHIP_PROJ_FUSE_SIM3, Rcw, tcw, Ow, pKF->fx, pKF->fy, pKF->cx, pKF->cy, 0.f, b.min_x, b.min_y, b.max_x, b.max_y, th);
    orbhip_set_levels(job.P, pKF->mvScaleFactors, pKF->mnScaleLevels, pKF->mfLogScaleFactor);
    const int nPoints = vpPoints.size();
    job.G.reserve(nPoints);
    for(int iMP=0; iMP<nPoints; iMP++)
    {
        MapPoint* pMP = vpPoints[iMP];
        if(spAlreadyFound.count(pMP)) continue;
        job.G.add(pMP, iMP, true, true, -1, 0, 0.f);
    }
    job.G.done();
    if(job.P.gemm_mode==2) orbhip_host_transform(job.G, Rcw, tcw);
    job.bi.assign(job.G.size(), -1); job.bd.assign(job.G.size(), 256);
    if(job.G.size()>0 && pKF->N>0)
    {
        orbhip_project_best_slot S = orbhip_fuse_slot(job);
        S.u_right = NULL;                                      // no chi-square gate in this overload: mvuRight is not read
        orbhip_check(orbhip_project_best_in_window_batch(orbhip_default_device(), 1, &S, 0));
    }
    int nFused=0;
    for(size_t k=0; k<job.G.size(); k++)
    {
        if(job.bd[k]>TH_LOW) continue;
        MapPoint* pMP = job.G.owner[k];
        const int bestIdx = job.bi[k], iMP = job.G.index[k];
        MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);          // a point already there is noted for replacement by the caller, else a new measurement (:1082-1095)
        if(pMPinKF)
        {
            if(!pMPinKF->isBad())
                vpReplacePoint[iMP] = pMPinKF;
        }
        else
        {
            pMP->AddObservation(pKF,bestIdx);
            pKF->AddMapPoint(pMP,bestIdx);
        }
        nFused++;
    }
    return nFused;
}

// ================================================================================================ SearchBySim3   ORBmatcher.cc:1102-1326
int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12, const float th)
{
    // once per call (:1110-1122): both cameras from the world, the similarity in both directions
    cv::Mat R1w = pKF1->GetRotation();
    cv::Mat t1w = pKF1->GetTranslation();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat sR12 = s12*R12;
    cv::Mat sR21 = (1.0/s12)*R12.t();
    cv::Mat t21 = -sR21*t12;

    const vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const int N1 = vpMapPoints1.size();
    const vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N2 = vpMapPoints2.size();

    vector<bool> vbAlreadyMatched1(N1,false), vbAlreadyMatched2(N2,false);
    for(int i=0; i<N1; i++)
    {
        MapPoint* pMP = vpMatches12[i];
        if(!pMP) continue;
        vbAlreadyMatched1[i]=true;
        const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
        if(idx2>=0 && idx2<N2) vbAlreadyMatched2[idx2]=true;
    }

    // the two directions are two slots of ONE device pass: points of key frame 1 searched in key frame 2 and the reverse
    struct Side { KeyFrame* target; OrbhipPoints G; orbhip_projection P; std::vector<int> bi, bd; } side[2];
    for(int d=0; d<2; d++)
    {
        Side &S = side[d];
        KeyFrame* from = d==0 ? pKF1 : pKF2; S.target = d==0 ? pKF2 : pKF1;
        const vector<MapPoint*> &vp = d==0 ? vpMapPoints1 : vpMapPoints2;
        const vector<bool> &vbAlready = d==0 ? vbAlreadyMatched1 : vbAlreadyMatched2;
        const cv::Mat &Rw = d==0 ? R1w : R2w, &tw = d==0 ? t1w : t2w, &sR = d==0 ? sR21 : sR12; const cv::Mat &ts = d==0 ? t21 : t12;
        (void)from;
        S.G.reserve(vp.size());
        for(int i=0, n=(int)vp.size(); i<n; i++)
        {
            MapPoint* pMP = vp[i];
            if(!pMP || vbAlready[i]) continue;
            S.G.add(pMP, i, true, true, -1, 0, 0.f);
        }
        S.G.done();
        const orbhip_bounds b = orbhip_kf_bounds(S.target);
        // the intrinsics are key frame 1's in BOTH directions, as in the reference (:1105-1108)
        S.P = orbhip_projection_of(ORBHIP_PROJ_SIM3, Rw, tw, cv::Mat(), pKF1->fx, pKF1->fy, pKF1->cx, pKF1->cy, 0.f, b.min_x, b.min_y, b.max_x, b.max_y, th);
        for(int r=0;r<3;r++) { for(int c=0;c<3;c++) S.P.R2[3*r+c] = sR.at<float>(r,c); S.P.t2[r] = ts.at<float>(r); }
        orbhip_set_levels(S.P, S.target->mvScaleFactors, S.target->mnScaleLevels, S.target->mfLogScaleFactor);
        if(S.P.gemm_mode==2) orbhip_host_transform(S.G, Rw, tw, &sR, &ts);
        S.bi.assign(S.G.size(), -1); S.bd.assign(S.G.size(), 256);
    }
    orbhip_project_best_slot slots[2];
    for(int d=0; d<2; d++)
    {
        Side &S = side[d]; KeyFrame* kf = S.target;
        orbhip_project_best_slot &B = slots[d]; memset(&B, 0, sizeof B);
        B.n = kf->N;
        if(kf->N>0) { B.kps = (const orbhip_keypoint*)&kf->mvKeysUn[0]; B.desc = kf->mDescriptors.ptr<unsigned char>(); }
        B.bounds = orbhip_kf_bounds(kf); B.proj = &S.P; B.np = (int)S.G.size();
        if(B.np>0) { B.points = &S.G.pts[0]; B.point_desc = &S.G.desc[0]; B.best_idx = &S.bi[0]; B.best_dist = &S.bd[0]; }
    }
    orbhip_check(orbhip_project_best_in_window_batch(orbhip_default_device(), 2, slots, 0));

    vector<int> vnMatch1(N1,-1), vnMatch2(N2,-1);
    for(size_t k=0;k<side[0].G.size();k++) if(side[0].bd[k]<=TH_HIGH) vnMatch1[side[0].G.index[k]] = side[0].bi[k];
    for(size_t k=0;k<side[1].G.size();k++) if(side[1].bd[k]<=TH_HIGH) vnMatch2[side[1].G.index[k]] = side[1].bi[k];

    // mutual agreement (:1309-1323)
    int nFound = 0;
    for(int i1=0; i1<N1; i1++)
    {
        const int idx2 = vnMatch1[i1];
        if(idx2>=0 && vnMatch2[idx2]==i1)
        {
            vpMatches12[i1] = vpMapPoints2[idx2];
            nFound++;
        }
    }
    return nFound;
}

// ================================================================================================ bag-of-words guided matching
// DBoW2::FeatureVector (std::map node id -> feature indices) flattened in map order
static void orbhip_flatten(const DBoW2::FeatureVector& fv, std::vector<unsigned int>& node, std::vector<int>& off, std::vector<unsigned int>& feat)
{
    node.clear(); off.assign(1, 0); feat.clear();
    for(DBoW2::FeatureVector::const_iterator it=fv.begin(); it!=fv.end(); ++it)
    {
        node.push_back(it->first); feat.insert(feat.end(), it->second.begin(), it->second.end()); off.push_back((int)feat.size());
    }
    if(node.empty()) node.push_back(0);
    if(feat.empty()) feat.push_back(0);
}
struct OrbhipBowSide
{
    std::vector<unsigned char> valid; std::vector<float> ang; std::vector<unsigned int> node, feat; std::vector<int> off; orbhip_bow_side s;
    // valid[i] = the feature carries a good map point (NULL pts: every feature takes part)
    void fill(const cv::Mat &desc, const std::vector<cv::KeyPoint> &keys, int n, const DBoW2::FeatureVector &fv, const std::vector<MapPoint*>* pts)
    {
        ang.resize(std::max(n,1));
        for(int i=0;i<n;i++) ang[i] = keys[i].angle;
        if(pts) { valid.resize(std::max(n,1)); for(int i=0;i<n;i++) { MapPoint* p = (*pts)[i]; valid[i] = p && !p->isBad(); } }
        orbhip_flatten(fv, node, off, feat);
        const orbhip_bow_side t = { desc.ptr<unsigned char>(), &ang[0], pts ? &valid[0] : NULL, n, &node[0], &off[0], &feat[0], (int)fv.size() };
        s = t;
    }
};
static orbhip_status orbhip_bow_search(int mode, const OrbhipBowSide &a, const OrbhipBowSide &b, float nnratio, bool checkOri, int* m12, int* nmatches)
{
    return orbhip_search_by_bow(orbhip_default_device(), mode, a.s.desc, a.s.angle, a.s.valid, a.s.n, a.s.fv_node, a.s.fv_off, a.s.fv_feat, a.s.nfv,
                                b.s.desc, b.s.angle, b.s.valid, b.s.n, b.s.fv_node, b.s.fv_off, b.s.fv_feat, b.s.nfv, nnratio, checkOri, m12, nmatches);
}

// SearchByBoW(KF, Frame)   ORBmatcher.cc:159-288  (TrackReferenceKeyFrame, Relocalization)
int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches)
{
    const vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = vector<MapPoint*>(F.N,static_cast<MapPoint*>(NULL));
    const int n1 = (int)vpMapPointsKF.size(), n2 = F.N;
    if(n1==0 || n2==0 || pKF->mFeatVec.empty() || F.mFeatVec.empty()) return 0;
    OrbhipBowSide s1, s2;
    s1.fill(pKF->mDescriptors, pKF->mvKeysUn, n1, pKF->mFeatVec, &vpMapPointsKF);
    s2.fill(F.mDescriptors, F.mvKeys, n2, F.mFeatVec, NULL);                   // the frame's angles are mvKeys' (:250), the key frame's mvKeysUn's
    std::vector<int> m12(n1, -1); int nmatches=0;
    orbhip_check(orbhip_bow_search(0, s1, s2, mfNNratio, mbCheckOrientation, &m12[0], &nmatches));
    for(int i=0;i<n1;i++) if(m12[i]>=0) vpMapPointMatches[m12[i]] = vpMapPointsKF[i];
    return nmatches;
}
// SearchByBoW(KF, KF)   ORBmatcher.cc:522-655  (LoopClosing::ComputeSim3)
int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint *> &vpMatches12)
{
    const vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = vector<MapPoint*>(vpMapPoints1.size(),static_cast<MapPoint*>(NULL));
    const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
    if(n1==0 || n2==0 || pKF1->mFeatVec.empty() || pKF2->mFeatVec.empty()) return 0;
    OrbhipBowSide s1, s2;
    s1.fill(pKF1->mDescriptors, pKF1->mvKeysUn, n1, pKF1->mFeatVec, &vpMapPoints1);
    s2.fill(pKF2->mDescriptors, pKF2->mvKeysUn, n2, pKF2->mFeatVec, &vpMapPoints2);
    std::vector<int> m12(n1, -1); int nmatches=0;
    orbhip_check(orbhip_bow_search(1, s1, s2, mfNNratio, mbCheckOrientation, &m12[0], &nmatches));
    for(int i=0;i<n1;i++) if(m12[i]>=0) vpMatches12[i] = vpMapPoints2[m12[i]];
    return nmatches;
}
// Tracking::Relocalization (Tracking.cc:1357-1380): `for every candidate: matcher.SearchByBoW(pKF, mCurrentFrame, vvpMapPointMatches[i])` as ONE device pass.
// vpKFs[i] == NULL or bad: skipped (vnMatches[i] = 0, vvpMapPointMatches[i] all NULL).
void SearchByBoWBatch(float nnratio, bool checkOri, const std::vector<KeyFrame*> &vpKFs, Frame &F, std::vector<std::vector<MapPoint*> > &vvpMapPointMatches, std::vector<int> &vnMatches)
{
    const size_t nKFs = vpKFs.size();
    vvpMapPointMatches.assign(nKFs, std::vector<MapPoint*>(F.N, static_cast<MapPoint*>(NULL)));
    vnMatches.assign(nKFs, 0);
    if(nKFs==0 || F.N==0 || F.mFeatVec.empty()) return;
    struct Side { OrbhipBowSide b; std::vector<MapPoint*> pts; std::vector<int> m12; };
    std::vector<Side> sides(nKFs); OrbhipBowSide fr;
    fr.fill(F.mDescriptors, F.mvKeys, F.N, F.mFeatVec, NULL);
    std::vector<orbhip_bow_pair> pairs; std::vector<size_t> which;
    for(size_t k=0;k<nKFs;k++)
    {
        KeyFrame* pKF = vpKFs[k];
        if(!pKF || pKF->isBad() || pKF->mFeatVec.empty()) continue;
        Side &S = sides[k];
        S.pts = pKF->GetMapPointMatches();
        const int n1 = (int)S.pts.size();
        if(n1==0) continue;
        S.m12.assign(n1, -1);
        S.b.fill(pKF->mDescriptors, pKF->mvKeysUn, n1, pKF->mFeatVec, &S.pts);
        which.push_back(k);
    }
    for(size_t j=0;j<which.size();j++) { Side &S = sides[which[j]]; const orbhip_bow_pair p = { &S.b.s, &fr.s, &S.m12[0], 0 }; pairs.push_back(p); }
    if(pairs.empty()) return;
    orbhip_check(orbhip_search_by_bow_batch(orbhip_default_device(), 0, (int)pairs.size(), &pairs[0], nnratio, checkOri));
    for(size_t j=0;j<which.size();j++)
    {
        const size_t k = which[j]; Side &S = sides[k];
        for(size_t i=0;i<S.m12.size();i++) if(S.m12[i]>=0) vvpMapPointMatches[k][S.m12[i]] = S.pts[i];
        vnMatches[k] = pairs[j].nmatches;
    }
}

// ================================================================================================ SearchForTriangulation   ORBmatcher.cc:657-823
struct OrbhipTriSide
{
    std::vector<float> kp; std::vector<unsigned char> has, st; std::vector<unsigned int> node, feat; std::vector<int> off; orbhip_tri_side s;
    void fill(KeyFrame* pKF)
    {
        const int n = pKF->N; kp.resize(4*std::max(n,1)); has.resize(std::max(n,1)); st.resize(std::max(n,1));
        const vector<MapPoint*> vp = pKF->GetMapPointMatches();                // one lock instead of one GetMapPoint() per feature
        for(int i=0;i<n;i++)
        {
            const cv::KeyPoint &k = pKF->mvKeysUn[i];
            kp[4*i]=k.pt.x; kp[4*i+1]=k.pt.y; kp[4*i+2]=k.angle; kp[4*i+3]=(float)k.octave;
            has[i] = vp[i]!=NULL; st[i] = pKF->mvuRight[i]>=0;
        }
        orbhip_flatten(pKF->mFeatVec, node, off, feat);
        const orbhip_tri_side t = { pKF->mDescriptors.ptr<unsigned char>(), &kp[0], &has[0], &st[0], n, &node[0], &off[0], &feat[0], (int)pKF->mFeatVec.size(),
                                    &pKF->mvScaleFactors[0], &pKF->mvLevelSigma2[0], (int)pKF->mvScaleFactors.size() };
        s = t;
    }
};
// the epipole of camera 1 in image 2 (:663-670), once per pair: the reference's cv::Mat statements
static void orbhip_epipole(const cv::Mat &Cw, KeyFrame* pKF2, float &ex, float &ey)
{
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat C2 = R2w*Cw+t2w;
    const float invz = 1.0f/C2.at<float>(2);
    ex = pKF2->fx*C2.at<float>(0)*invz+pKF2->cx;
    ey = pKF2->fy*C2.at<float>(1)*invz+pKF2->cy;
}
int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo)
{
    vector<int> vMatches12(pKF1->N,-1);
    int nmatches=0;
    if(pKF1->N>0 && pKF2->N>0 && !pKF1->mFeatVec.empty() && !pKF2->mFeatVec.empty())
    {
        OrbhipTriSide s1, s2; s1.fill(pKF1); s2.fill(pKF2);
        float F12flat[9], ex, ey;
        for(int r=0;r<3;r++) for(int c=0;c<3;c++) F12flat[3*r+c] = F12.at<float>(r,c);
        orbhip_epipole(pKF1->GetCameraCenter(), pKF2, ex, ey);
        const orbhip_tri_side &a = s1.s, &b = s2.s;
        orbhip_check(orbhip_search_for_triangulation(orbhip_default_device(), a.desc, a.kp, a.has_mp, a.stereo, a.n, a.fv_node, a.fv_off, a.fv_feat, a.nfv,
                                                     b.desc, b.kp, b.has_mp, b.stereo, b.n, b.fv_node, b.fv_off, b.fv_feat, b.nfv,
                                                     F12flat, ex, ey, b.scale_factors, b.level_sigma2, b.nlevels, bOnlyStereo, mbCheckOrientation, &vMatches12[0], &nmatches));
    }
    vMatchedPairs.clear();
    vMatchedPairs.reserve(nmatches);
    for(size_t i=0, iend=vMatches12.size(); i<iend; i++)
        if(vMatches12[i]>=0) vMatchedPairs.push_back(make_pair(i,vMatches12[i]));
    return nmatches;
}
// LocalMapping::CreateNewMapPoints (LocalMapping.cc:237-268): `for every neighbour: matcher.SearchForTriangulation(mpCurrentKeyFrame, pKF2, F12, vMatchedIndices, false)`
// as ONE device pass.  vvMatches12[i][idx1] = feature of neighbour i, searched with key frame 1's map points as they are NOW; the reference's loop gives key
// frame 1 new map points between neighbours, and its search (no orientation check, LocalMapping.cc:215; vbMatched2 is never written, ORBmatcher.cc:677, 725)
// treats every feature of key frame 1 by itself: TriangulationPairs(pKF1, vvMatches12[i], vMatchedPairs), called when neighbour i's turn comes, drops the
// features that have received a map point in the meantime and returns exactly the reference's vMatchedPairs.
void SearchForTriangulationBatch(KeyFrame* pKF1, const std::vector<KeyFrame*> &vpKF2, const std::vector<cv::Mat> &vF12, const bool bOnlyStereo, std::vector<std::vector<int> > &vvMatches12)
{
    const size_t nn = vpKF2.size();
    const int n1 = pKF1->N;
    vvMatches12.assign(nn, std::vector<int>(n1, -1));
    if(nn==0 || n1==0 || pKF1->mFeatVec.empty()) return;
    OrbhipTriSide s1; s1.fill(pKF1);
    std::vector<OrbhipTriSide> s2(nn); std::vector<orbhip_tri_pair> pairs(nn);
    const cv::Mat Cw = pKF1->GetCameraCenter();
    for(size_t i=0;i<nn;i++)
    {
        s2[i].fill(vpKF2[i]);
        orbhip_tri_pair &P = pairs[i]; memset(&P, 0, sizeof P);
        P.kf2 = &s2[i].s; P.match12 = &vvMatches12[i][0];
        for(int r=0;r<3;r++) for(int c=0;c<3;c++) P.F12[3*r+c] = vF12[i].at<float>(r,c);
        orbhip_epipole(Cw, vpKF2[i], P.ex, P.ey);
    }
    orbhip_check(orbhip_search_for_triangulation_batch(orbhip_default_device(), &s1.s, (int)nn, &pairs[0], bOnlyStereo, 0));
}
int TriangulationPairs(KeyFrame* pKF1, const std::vector<int> &vMatches12, std::vector<std::pair<size_t,size_t> > &vMatchedPairs)
{
    vMatchedPairs.clear();
    for(size_t i=0, iend=vMatches12.size(); i<iend; i++)
    {
        if(vMatches12[i]<0 || pKF1->GetMapPoint(i))          // "If there is already a MapPoint skip" (ORBmatcher.cc:698-700), as of now
            continue;
        vMatchedPairs.push_back(make_pair(i,vMatches12[i]));
    }
    return (int)vMatchedPairs.size();
}
#endif  // ORBHIP_USE_OPENCV && !ORBHIP_MATCHER_MAP_FREE_ONLY

} // namespace ORB_SLAM2
