"""integration/apply_dropin.py as a maintainer runs it: the unified diff it prints applies with `patch -p1` to a checkout of the reference, gives
exactly the files its directory mode writes, and the patched src/Frame.cc / src/ORBmatcher.cc compile (syntax only: against the OpenCV type
stand-in the oracle builds use, with the drop-in headers in place of the reference's).  Needs /root/reference (a scratch copy of the few files
involved is made; nothing is written there) and patch(1); skipped elsewhere."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SCRIPT = os.path.join(ROOT, "integration", "apply_dropin.py")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")) or shutil.which("patch") is None, reason="reference sources not mounted or patch(1) missing")
def test_patch_applies_and_patched_sources_compile(tmp_path):
    co = tmp_path / "ORB_SLAM2"
    for sub in ("src", "include"):
        (co / sub).mkdir(parents=True)
    for rel in ("src/Frame.cc", "src/ORBmatcher.cc", "src/ORBextractor.cc", "include/ORBextractor.h"):
        shutil.copyfile(os.path.join(REF, rel), co / rel)
    diff = subprocess.run([sys.executable, SCRIPT, "--patch", REF], capture_output=True, text=True, check=True).stdout
    assert diff.count("\n--- a/") + diff.startswith("--- a/") >= 5 and "+++ b/include/orbhip.h" in diff
    r = subprocess.run(["patch", "-p1", "-d", str(co)], input=diff, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = tmp_path / "out"
    subprocess.run([sys.executable, SCRIPT, REF, str(out)], capture_output=True, text=True, check=True)
    for rel in ("src/Frame.cc", "src/ORBmatcher.cc", "src/ORBextractor.cc", "include/ORBextractor.h", "include/orbhip.h", "include/ORBmatcherBatch.h"):
        assert (co / rel).read_text() == (out / rel).read_text(), rel + ": patch -p1 and the directory mode disagree"
    patched = (co / "src/ORBmatcher.cc").read_text()
    assert "abort();" not in patched and "orbhip_search_by_projection_frame" in patched and patched.count("orbhip_check(orbhip_") >= 10
    assert "BindFrame(mnId)" in (co / "src/Frame.cc").read_text()
    # the patched translation units compile against the drop-in headers (the flags of oracle/Makefile's dropin_full rule, -fsyntax-only)
    ora = os.path.join(ROOT, "oracle")
    flags = ["-std=c++14", "-fsyntax-only", "-w", "-DCVLITE_ALGEBRA", "-DORBHIP_USE_OPENCV", "-DORBSLAM_DROPIN_BUILD", "-DORBHIP_USE_DBOW2_TYPES",
             "-include", os.path.join(ora, "ref_shim/dropin/ORBVocabulary.h"), "-include", os.path.join(ora, "ref_shim/dropin/ORBextractor.h"),
             "-I" + os.path.join(ora, "ref_shim/dropin"), "-I" + os.path.join(ora, "ref_shim"), "-I" + os.path.join(REF, "include"), "-I" + REF,
             "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(REF, "Thirdparty/DBoW2")]
    for rel in ("src/ORBmatcher.cc", "src/Frame.cc"):
        r = subprocess.run(["g++"] + flags + [str(co / rel)], capture_output=True, text=True)
        assert r.returncode == 0, rel + ":\n" + r.stderr[-3000:]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference sources not mounted")
def test_optional_steps(tmp_path):
    """--stereo-one-call replaces exactly the constructor's two extractor threads with ONE ExtractStereo call (and fails loudly on a checkout that has
    none); the batch forms of the back end's loops are emitted with their header; both patched files still compile."""
    plain, one = tmp_path / "plain", tmp_path / "one"
    subprocess.run([sys.executable, SCRIPT, REF, str(plain)], capture_output=True, text=True, check=True)
    subprocess.run([sys.executable, SCRIPT, "--stereo-one-call", "--resident-bow", REF, str(one)], capture_output=True, text=True, check=True)
    a, b = (plain / "src/Frame.cc").read_text(), (one / "src/Frame.cc").read_text()
    # --resident-bow: Frame::ComputeBoW reads the descriptors where the extraction left them, the reference's statement kept for every other frame
    assert "HoldsFrame" not in a and b.count("mpORBvocabulary->ComputeBoW(*mpORBextractorLeft, mBowVec, mFeatVec, 4);") == 1
    assert b.count("mpORBvocabulary->transform(vCurrentDesc,mBowVec,mFeatVec,4);") == 1
    assert "thread threadLeft(&Frame::ExtractORB,this,0,imLeft);" in a and "threadLeft" not in b
    assert b.count("mpORBextractorLeft->ExtractStereo(imLeft, imRight, mvKeys, mDescriptors, mvKeysRight, mDescriptorsRight, mbf, mbf/K.at<float>(0,0));") == 1
    # RGB-D: by default the reference's own depth loop stays and one line hands mvuRight to the resident frame; --device-rgbd replaces the loop by the device sampling
    assert a.count("if(mpORBextractorLeft) mpORBextractorLeft->SetStereoColumns(mvuRight);") == 1 and "const float d = imDepth.at<float>(v,u);" in a
    dev = tmp_path / "dev"
    subprocess.run([sys.executable, SCRIPT, "--device-rgbd", REF, str(dev)], capture_output=True, text=True, check=True)
    c = (dev / "src/Frame.cc").read_text()
    assert "SetStereoColumns" not in c and c.count("mpORBextractorLeft->ComputeStereoFromRGBD(imDepth, 1.0f, mbf, N, mvuRight, mvDepth);") == 1 and "imDepth.at<float>(v,u)" not in c
    assert (plain / "src/ORBmatcher.cc").read_text() == (one / "src/ORBmatcher.cc").read_text()
    m = (one / "src/ORBmatcher.cc").read_text()
    for name in ("void SearchByBoWBatch(", "void SearchForTriangulationBatch(", "int TriangulationPairs(", "int FuseBatch(", "static int orbhip_fuse_apply(", "static void orbhip_fuse_collect("):
        assert m.count(name) == 1, name
    assert (one / "include/ORBmatcherBatch.h").read_text() == open(os.path.join(ROOT, "include", "ORBmatcherBatch.h")).read()
    # a Frame.cc without the two threads (already patched, or a fork that extracts differently) is refused, not silently left alone
    twice = tmp_path / "twice"
    (twice / "src").mkdir(parents=True); (twice / "include").mkdir()
    shutil.copyfile(one / "src/Frame.cc", twice / "src/Frame.cc")
    r = subprocess.run([sys.executable, SCRIPT, "--files", "--stereo-one-call", str(twice / "src/Frame.cc"), str(tmp_path / "x.cc")], capture_output=True, text=True)
    assert r.returncode != 0 and "not found" in (r.stderr + r.stdout)
    ora = os.path.join(ROOT, "oracle")
    flags = ["-std=c++14", "-fsyntax-only", "-w", "-DCVLITE_ALGEBRA", "-DORBHIP_USE_OPENCV", "-DORBSLAM_DROPIN_BUILD", "-DORBHIP_USE_DBOW2_TYPES",
             "-include", os.path.join(ora, "ref_shim/dropin/ORBVocabulary.h"), "-include", os.path.join(ora, "ref_shim/dropin/ORBextractor.h"),
             "-I" + os.path.join(ora, "ref_shim/dropin"), "-I" + os.path.join(ora, "ref_shim"), "-I" + os.path.join(REF, "include"), "-I" + REF,
             "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(REF, "Thirdparty/DBoW2")]
    for rel in ("src/ORBmatcher.cc", "src/Frame.cc"):
        r = subprocess.run(["g++"] + flags + [str(one / rel)], capture_output=True, text=True)
        assert r.returncode == 0, rel + ":\n" + r.stderr[-3000:]
