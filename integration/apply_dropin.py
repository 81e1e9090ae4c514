#!/usr/bin/env python3
"""apply_dropin.py — the binding of INTEGRATION.md §2 as ONE command a maintainer of raulmur/ORB_SLAM2 runs on a checkout.

    python integration/apply_dropin.py /path/to/ORB_SLAM2 /path/to/out      # writes the replaced / edited tree pieces under out/
    python integration/apply_dropin.py --patch /path/to/ORB_SLAM2 > dropin.patch   # the same as a unified diff (patch -p1)
    ... --stereo-one-call        optional: the stereo Frame constructor extracts both images in ONE call on one device context (ORBextractor::ExtractStereo)
    ... --device-rgbd            optional: Frame::ComputeStereoFromRGBD samples the depth map on the device (default: the reference's loop + N floats uploaded)
    ... --resident-bow           optional (with this repository's ORBVocabulary class in place, step 3e): Frame::ComputeBoW reads the descriptors in HBM
    ... --flat-frustum           optional: Frame::isInFrustum forwards to ORBmatcher::IsInFrustum (flat floats, the map point visited once, no cv::Mat temporaries)
    ... --skip-host-grid         optional: Frame::AssignFeaturesToGrid does nothing - with this repository's ORBmatcher.cc nobody reads the host's 64 x 48 grid

What it produces (nothing else of the checkout changes; Tracking.cc, LocalMapping.cc, LoopClosing.cc, KeyFrame*.cc, MapPoint.cc compile as they are):
  REPLACED by this repository's files (an installer's copy, no text surgery):
    out/include/ORBextractor.h, out/src/ORBextractor.cc     the drop-in extractor class (include/ORBextractor.h, orb_slam2_amd/cpp/ORBextractor.cc)
    out/include/ORBmatcher.h,   out/src/ORBmatcher.cc       the drop-in matcher class: all twelve members (orb_slam2_amd/cpp/ORBmatcher.cc)
    out/include/ORBmatcherBatch.h                           the back end's matcher loops as single device passes (optional use)
    out/include/orbhip_gemm_probe.h                         how the linked cv::Mat rounds `R*x+t` (ORBmatcher.cc asks once per process)
    out/include/orbhip.h                                    the C ABI of liborbhip.so (link with -lorbhip)
  EDITED (located by the reference's own statements; the script fails loudly - never skips silently - where a checkout differs from upstream):
    out/src/Frame.cc         the bodies of ComputeStereoMatches, UndistortKeyPoints, ComputeImageBounds (and optionally ComputeStereoFromRGBD / ComputeBoW /
                             the stereo constructor's two extractor threads) become one-line forwards to the extractor that just processed the frame (§2-3b, 3d')
    out/include/MapPoint.h   ONE line: `friend class ORBmatcher;` (+ a #define saying so): the matcher's gathers read a map point's position, normal,
                             descriptor and mfMaxDistance in place, under the point's own mutexes (§2-3c)

Until round 5 src/ORBmatcher.cc was the reference's file with each member's search loop cut out by regular expressions; it is now a file of this repository
like the extractor's (a fork that reformatted a signature loses nothing).  The test builds of this repository consume exactly this script for the two edited
files (oracle/Makefile, `apply_dropin.py --files <Frame.cc> <out> <MapPoint.h> <out>`) and compile orb_slam2_amd/cpp/ORBmatcher.cc as it lies, so what
tests/test_reference_dropin.py checks against the unmodified reference is what a maintainer installs.  No reference source is kept in this repository."""
import difflib
import os
import re
import shutil
import sys

FORWARDS = {
    # the only reader of mvImagePyramid becomes a device call on both extractors' resident results
    r"void\s+Frame::ComputeStereoMatches\s*\(\s*\)":
        "{ mpORBextractorLeft->ComputeStereoMatches(*mpORBextractorRight, mbf, mb, N, mvuRight, mvDepth); }",
    # §2-3d': mvKeysUn / the image bounds / the depth lookup come from the extractor that just processed this frame
    r"void\s+Frame::UndistortKeyPoints\s*\(\s*\)":
        "{ mpORBextractorLeft->UndistortKeyPoints(mvKeysUn); mpORBextractorLeft->BindFrame(mnId); }",
    r"void\s+Frame::ComputeImageBounds\s*\(\s*const\s+cv::Mat\s*&\s*imLeft\s*\)":
        "{ mpORBextractorLeft->ComputeImageBounds(imLeft.cols, imLeft.rows, mnMinX, mnMaxX, mnMinY, mnMaxY); }",
}


# Frame::ComputeStereoFromRGBD (Frame.cc:643-665) keeps the reference's own loop - N samples of a depth map that is in host memory cost less there than the map's
# trip to the device - and hands its result to the frame the extractor still holds (the resident searches' right-coordinate test reads mvuRight in HBM).
# --device-rgbd replaces the loop by ORBextractor::ComputeStereoFromRGBD instead (depth map sampled on the device: the form for batches of frames).
RGBD_SIG = r"void\s+Frame::ComputeStereoFromRGBD\s*\(\s*const\s+cv::Mat\s*&\s*imDepth\s*\)"
RGBD_APPEND = "    if(mpORBextractorLeft) mpORBextractorLeft->SetStereoColumns(mvuRight);\n"
RGBD_DEVICE_BODY = "{ mpORBextractorLeft->ComputeStereoFromRGBD(imDepth, 1.0f, mbf, N, mvuRight, mvDepth); }"


def body_span(src, signature):
    m = re.search(signature, src)
    if not m:
        raise SystemExit(f"signature not found: {signature}")
    i = src.index("{", m.end())
    depth, j = 0, i
    while True:
        c = src[j]
        depth += c == "{"
        depth -= c == "}"
        if depth == 0:
            return i, j
        j += 1


def append_to_body(src, signature, statement):
    i, j = body_span(src, signature)
    return src[:j] + statement + src[j:]


def replace_body(src, signature, body):
    m = re.search(signature, src)
    if not m:
        raise SystemExit(f"signature not found: {signature}")
    i = src.index("{", m.end())
    depth, j = 0, i
    while True:
        c = src[j]
        depth += c == "{"
        depth -= c == "}"
        j += 1
        if depth == 0:
            break
    return src[:i] + body + src[j:]


# src/ORBmatcher.cc is this repository's own file (orb_slam2_amd/cpp/ORBmatcher.cc: all twelve members, the projection of the five
# pose-guided ones on the device).  Its gathers read a map point's position / normal / scale range / descriptor in place (no cv::Mat clones) and
# mfMaxDistance, which MapPoint::PredictScale uses and no accessor returns: ONE line in include/MapPoint.h lets them.
MAPPOINT_CLASS = r"class\s+MapPoint\s*\{\s*public\s*:"
MAPPOINT_FRIEND = """class MapPoint
{
    friend class ORBmatcher;      // orbhip drop-in: src/ORBmatcher.cc reads mWorldPos / mNormalVector / mDescriptor / mfMaxDistance under the point's own mutexes
public:"""


def patch_mappoint_header(src):
    src, n = re.subn(MAPPOINT_CLASS, MAPPOINT_FRIEND, src, count=1)
    if n != 1:
        raise SystemExit("`class MapPoint { public:` not found in include/MapPoint.h")
    m = re.search(r"#define\s+MAPPOINT_H\s*\n", src)
    if not m:
        raise SystemExit("include guard MAPPOINT_H not found in include/MapPoint.h")
    return src[:m.end()] + "#define ORBHIP_MAPPOINT_FRIEND 1     // orbhip drop-in: ORBmatcher is a friend of MapPoint (see below)\n" + src[m.end():]


# INTEGRATION.md §2-3g (optional, --stereo-one-call): the stereo constructor's two extractor threads (Frame.cc:78-81) become ONE call on the left
# extractor - both images through one device context, the stereo matcher queued behind the extraction; the forwarded ComputeStereoMatches() that
# follows (step 3b) hands out the columns computed there.  mb = mbf / fx: the constructor assigns the member only at its end (Frame.cc:113).
STEREO_THREADS = (r"thread\s+threadLeft\s*\(\s*&Frame::ExtractORB\s*,\s*this\s*,\s*0\s*,\s*imLeft\s*\)\s*;\s*thread\s+threadRight\s*\(\s*&Frame::ExtractORB\s*,\s*this\s*,\s*1\s*,\s*imRight\s*\)\s*;"
                  r"\s*threadLeft\.join\(\)\s*;\s*threadRight\.join\(\)\s*;")
STEREO_ONE_CALL = "mpORBextractorLeft->ExtractStereo(imLeft, imRight, mvKeys, mDescriptors, mvKeysRight, mDescriptorsRight, mbf, mbf/K.at<float>(0,0));"


# INTEGRATION.md §2-3e (optional, --resident-bow; needs this repository's ORBVocabulary class): Frame::ComputeBoW (Frame.cc:395-402) transforms the descriptors
# where the extraction left them while the frame is still its extractor's last one (TrackReferenceKeyFrame / Relocalization call it right after the
# constructor); any other frame goes through the reference's own statement
RESIDENT_BOW_SIG = r"void\s+Frame::ComputeBoW\s*\(\s*\)"
RESIDENT_BOW_BODY = """{
    if(mBowVec.empty())
    {
        if(mpORBextractorLeft && mpORBextractorLeft->HoldsFrame(mnId, N))
            mpORBvocabulary->ComputeBoW(*mpORBextractorLeft, mBowVec, mFeatVec, 4);
        else
        {
            vector<cv::Mat> vCurrentDesc = Converter::toDescriptorVector(mDescriptors);
            mpORBvocabulary->transform(vCurrentDesc,mBowVec,mFeatVec,4);
        }
    }
}"""


# optional, --skip-host-grid: Frame::AssignFeaturesToGrid (Frame.cc:230-245) fills the 64 x 48 vectors that only Frame::GetFeaturesInArea / KeyFrame::GetFeaturesInArea
# read - and those two are called from src/ORBmatcher.cc alone, which is now this repository's file and searches the grid the DEVICE builds from the same key
# points (k_match_grid).  With the flag the host grid stays empty: no 3072 vector reservations per Frame, nothing to copy in Frame(const Frame&) and KeyFrame().
# Leave it out if your fork reads mGrid / GetFeaturesInArea elsewhere.
GRID_SIG = r"void\s+Frame::AssignFeaturesToGrid\s*\(\s*\)"
GRID_BODY = "{ /* orbhip drop-in (--skip-host-grid): the feature grid is built on the device by the matcher entry points; nothing on the host reads mGrid */ }"


# optional, --flat-frustum: Frame::isInFrustum (Frame.cc:269-325), called by Tracking::SearchLocalPoints for every local map point of every frame, builds five
# cv::Mat temporaries per point; ORBmatcher::IsInFrustum (this repository's ORBmatcher.cc) evaluates the same statements on flat floats, the point visited once
FRUSTUM_SIG = r"bool\s+Frame::isInFrustum\s*\(\s*MapPoint\s*\*\s*pMP\s*,\s*float\s+viewingCosLimit\s*\)"
FRUSTUM_BODY = "{ return ORBmatcher::IsInFrustum(*this, mRcw, mtcw, mOw, pMP, viewingCosLimit); }"


def patch_frame(src, stereo_one_call=False, resident_bow=False, device_rgbd=False, skip_host_grid=False, flat_frustum=False):
    for sig, body in FORWARDS.items():
        src = replace_body(src, sig, body)
    src = replace_body(src, RGBD_SIG, RGBD_DEVICE_BODY) if device_rgbd else append_to_body(src, RGBD_SIG, RGBD_APPEND)
    if skip_host_grid:
        src = replace_body(src, GRID_SIG, GRID_BODY)
    if flat_frustum:
        src = replace_body(src, FRUSTUM_SIG, FRUSTUM_BODY)
    if resident_bow:
        src = replace_body(src, RESIDENT_BOW_SIG, RESIDENT_BOW_BODY)
    if stereo_one_call:
        src, n = re.subn(STEREO_THREADS, STEREO_ONE_CALL, src)
        if n != 1:
            raise SystemExit("the stereo constructor's two extractor threads (Frame.cc:78-81) were not found")
    return src


def main():
    argv = sys.argv[1:]
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if argv and argv[0] == "--files":                       # the form the test builds use: explicit input / output files, Frame.cc [+ MapPoint.h]
        one_call = "--stereo-one-call" in argv
        rbow = "--resident-bow" in argv
        drgbd = "--device-rgbd" in argv
        nogrid = "--skip-host-grid" in argv
        flat = "--flat-frustum" in argv
        argv = [a for a in argv[1:] if a not in ("--stereo-one-call", "--resident-bow", "--device-rgbd", "--skip-host-grid", "--flat-frustum")]
        open(argv[1], "w").write(patch_frame(open(argv[0]).read(), stereo_one_call=one_call, resident_bow=rbow, device_rgbd=drgbd, skip_host_grid=nogrid, flat_frustum=flat))
        if len(argv) > 3:
            os.makedirs(os.path.dirname(argv[3]) or ".", exist_ok=True)
            open(argv[3], "w").write(patch_mappoint_header(open(argv[2]).read()))
        return
    one_call = "--stereo-one-call" in argv
    rbow = "--resident-bow" in argv
    drgbd = "--device-rgbd" in argv
    nogrid = "--skip-host-grid" in argv
    flat = "--flat-frustum" in argv
    argv = [a for a in argv if a not in ("--stereo-one-call", "--resident-bow", "--device-rgbd", "--skip-host-grid", "--flat-frustum")]
    emit_patch = bool(argv) and argv[0] == "--patch"
    if emit_patch:
        argv = argv[1:]
    if len(argv) < (1 if emit_patch else 2):
        raise SystemExit(__doc__)
    ref = argv[0]
    edited = {"src/Frame.cc": patch_frame(open(os.path.join(ref, "src/Frame.cc")).read(), stereo_one_call=one_call, resident_bow=rbow, device_rgbd=drgbd, skip_host_grid=nogrid, flat_frustum=flat),
              "include/MapPoint.h": patch_mappoint_header(open(os.path.join(ref, "include/MapPoint.h")).read())}
    copies = {"include/ORBextractor.h": "include/ORBextractor.h", "src/ORBextractor.cc": "orb_slam2_amd/cpp/ORBextractor.cc", "include/orbhip.h": "include/orbhip.h",
              "include/ORBmatcher.h": "include/ORBmatcher.h", "src/ORBmatcher.cc": "orb_slam2_amd/cpp/ORBmatcher.cc", "include/ORBmatcherBatch.h": "include/ORBmatcherBatch.h",
              "include/orbhip_gemm_probe.h": "include/orbhip_gemm_probe.h"}
    if emit_patch:
        for rel, new in edited.items():
            old = open(os.path.join(ref, rel)).read()
            sys.stdout.writelines(difflib.unified_diff(old.splitlines(True), new.splitlines(True), "a/" + rel, "b/" + rel))
        for rel, mine in copies.items():
            old = open(os.path.join(ref, rel)).read() if os.path.exists(os.path.join(ref, rel)) else ""
            sys.stdout.writelines(difflib.unified_diff(old.splitlines(True), open(os.path.join(here, mine)).read().splitlines(True), "a/" + rel, "b/" + rel))
        return
    out = argv[1]
    for rel, new in edited.items():
        os.makedirs(os.path.dirname(os.path.join(out, rel)), exist_ok=True)
        open(os.path.join(out, rel), "w").write(new)
    for rel, mine in copies.items():
        os.makedirs(os.path.dirname(os.path.join(out, rel)), exist_ok=True)
        shutil.copyfile(os.path.join(here, mine), os.path.join(out, rel))
    print("wrote", ", ".join(sorted(list(edited) + list(copies))), "under", out)


if __name__ == "__main__":
    main()
