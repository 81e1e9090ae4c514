// ORBmatcherBatch.h — the back end's matcher LOOPS as single device passes (optional; the members of ORBmatcher keep their per-call forms).
//
// The reference calls three ORBmatcher members in loops over key frames:
//   Tracking::Relocalization           for every candidate:  matcher.SearchByBoW(pKF, mCurrentFrame, vvpMapPointMatches[i])          Tracking.cc:1357-1380
//   LocalMapping::CreateNewMapPoints   for every neighbour:  matcher.SearchForTriangulation(mpCurrentKeyFrame, pKF2, F12, pairs, false) LocalMapping.cc:237-268
//   LocalMapping::SearchInNeighbors    for every target:     matcher.Fuse(pKFi, vpMapPointMatches)                                   LocalMapping.cc:483-514
// On the device one such call is launch latency, not work.  The functions below run a whole loop as ONE upload, ONE launch set and ONE download
// (orbhip_search_by_bow_batch, orbhip_search_for_triangulation_batch, orbhip_search_best_in_window_batch of include/orbhip.h) and return, per key frame,
// exactly what the member returns when it is called in the reference's order.  They are defined in orb_slam2_amd/cpp/ORBmatcher.cc (the file
// integration/apply_dropin.py installs as src/ORBmatcher.cc); INTEGRATION.md section 2-3h shows the three loops rewritten.  Failures throw ORBhipError like the members'.
#ifndef ORBMATCHERBATCH_H
#define ORBMATCHERBATCH_H

#include <utility>
#include <vector>

namespace ORB_SLAM2
{

class Frame;
class KeyFrame;
class MapPoint;

// vvpMapPointMatches[i] / vnMatches[i] = what ORBmatcher(nnratio, checkOri).SearchByBoW(vpKFs[i], F, vvpMapPointMatches[i]) fills / returns; a NULL or bad
// key frame is skipped (no matches).  F.ComputeBoW() must have run (Tracking.cc:1344).
void SearchByBoWBatch(float nnratio, bool checkOri, const std::vector<KeyFrame*> &vpKFs, Frame &F, std::vector<std::vector<MapPoint*> > &vvpMapPointMatches, std::vector<int> &vnMatches);

// All neighbours of pKF1 searched at once (vF12[i] = ComputeF12(pKF1, vpKF2[i]), LocalMapping.cc:262), WITHOUT the orientation check - how LocalMapping
// constructs its matcher (LocalMapping.cc:215).  When neighbour i's turn comes in the caller's loop, TriangulationPairs(pKF1, vvMatches12[i], vMatchedPairs)
// gives the vMatchedPairs ORBmatcher::SearchForTriangulation would give at that moment: it drops the features of pKF1 that have received a map point from
// an earlier neighbour in the meantime (the search treats every feature of key frame 1 by itself: ORBmatcher.cc:677, 725 never mark key frame 2's features).
void SearchForTriangulationBatch(KeyFrame* pKF1, const std::vector<KeyFrame*> &vpKF2, const std::vector<cv::Mat> &vF12, const bool bOnlyStereo, std::vector<std::vector<int> > &vvMatches12);
int TriangulationPairs(KeyFrame* pKF1, const std::vector<int> &vMatches12, std::vector<std::pair<size_t,size_t> > &vMatchedPairs);

// = sum over the targets, in order, of ORBmatcher().Fuse(vpTargetKFs[t], vpMapPoints, th): every target's window searches in one device pass, then the
// reference's own map surgery target by target; a point whose state an earlier target's surgery changed (MapPoint::Replace) is re-checked / searched again.
// The points travel once for all targets (orbhip_project_best_in_window_shared) and a re-check searches the target's slot where that call left it on the
// device (orbhip_project_best_in_window_held); more than 64 targets, or a cv::Mat whose R*x+t rounding the probe does not know (H11 mode 2): one copy per target.
int FuseBatch(const std::vector<KeyFrame*> &vpTargetKFs, const std::vector<MapPoint*> &vpMapPoints, const float th = 3.0);

} // namespace ORB_SLAM2

#endif
