"""Monocular and RGB-D sequences with relocalisation through the binding (the members the stereo loop of tests/test_dropin_loop.py never drives).

oracle/orbslam_ref_wrap.cpp::orbslam_ref_sequence_loop restates the rest of Tracking's matcher sequences around the reference's own Frame.cc / ORBmatcher.cc:
  monocular   the 2 x nFeatures initialisation extractor (Tracking.cc:124-125, 257-260); MonocularInitialization's SearchForInitialization until >= 100 matches
              (:563-635); CreateInitialMapMonocular's bags of words (:637-720); then, alternating, TrackReferenceKeyFrame (:757-799: Frame::ComputeBoW +
              SearchByBoW(KF, Frame)) and TrackWithMotionModel (bMono = true, th = 15); SearchLocalPoints after both; new key frames with their ComputeBoW
  RGB-D       Frame(imGray, imDepth, ...) (Frame.cc:119-172) with TUM1's distorted camera: UndistortKeyPoints' cv::undistortPoints branch, the grid over the
              undistorted bounds, ComputeStereoFromRGBD; TrackWithMotionModel + SearchLocalPoints (th = 3) + CreateNewKeyFrame's depth points
  every 5th frame is "lost": Relocalization's sequence (:1341-1502) - ComputeBoW, SearchByBoW(pKF, Frame) against the last five key frames, the best candidate's
              matches adopted, then SearchByProjection(Frame, pKF, sFound, 10, 100) and (.., 3, 64) (ORBmatcher.cc:1472-1599)
The all-reference build (liborbslam_ref.so) and the drop-in builds must agree frame by frame in key points, mvKeysUn, descriptors, depth columns, the map point
of every feature after each matcher, every counter and the hash of every bag of words.  Here: a small shape on the CPU emulation; `-m gpu`: BASELINE.json
configs[0]'s shape (640x480, 1000 features) on the MI355X."""
import os

import numpy as np
import pytest

from orb_slam2_amd import synth
from conftest import gpu_session

HERE = os.path.dirname(os.path.abspath(__file__))
VOC = os.path.join(HERE, "golden", "voc_k6_L3_ref.txt")
TUM1_DIST = (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)                                                     # Examples/Monocular/TUM1.yaml
SMALL = dict(w=400, h=300, n=500, fx=231.5, fy=231.5, cx=200.0, cy=150.0, bf=25.5, th_depth=35.0)
TUM1 = dict(w=640, h=480, n=1000, fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, bf=40.0, th_depth=40.0)     # configs[0]: TUM fr1, 1000 features


@pytest.fixture(scope="module", params=["steps1-3", "all-steps", pytest.param("steps1-3-gpu", marks=pytest.mark.gpu), pytest.param("all-steps-gpu", marks=pytest.mark.gpu)])
def builds(request):
    from oracle import orbslam_ref as S
    if request.param.endswith("-gpu"):
        if not (S.build() and S.build_dropin_gpu()):
            pytest.fail("oracle/_ref/liborbslam_dropin_gpu.so did not travel with the repository (build it with `make -C oracle dropin_gpu` where /root/reference is mounted)")
        return S, S.dropin_gpu_lib(full=request.param.startswith("all-steps"))
    if gpu_session(request.config):
        pytest.skip("a -m gpu session maps liborbhip.so only: the CPU-emulation builds of the binding are not loaded beside it")
    request.getfixturevalue("emu_lib")
    if not (S.build() and S.build_dropin()):
        pytest.skip("reference sources not mounted")
    return S, (S.dropin_full_lib() if request.param == "all-steps" else S.dropin_lib())


def run_and_compare(S, D, sensor, cfg, nframes, voc_path, dist=None, kf_every=3, lost_every=5, seed=3):
    L, R, T, P, depth = synth.stereo_sequence(cfg["w"], cfg["h"], nframes, cfg["fx"], cfg["bf"], seed=seed, return_depth=True)
    args = (sensor, L, depth, T, P, cfg["n"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["bf"], cfg["th_depth"], voc_path)
    ref = S.sequence_loop(*args, dist=dist, kf_every=kf_every, lost_every=lost_every)
    got = S.sequence_loop(*args, dist=dist, kf_every=kf_every, lost_every=lost_every, library=D)
    for k, (a, b) in enumerate(zip(ref, got)):
        assert a.same(b), f"{sensor} frame {k} differs: " + ", ".join(f"{f} {getattr(a, f)} vs {getattr(b, f)}" for f in a.FIELDS + ("n_extra", "bow_hash") if getattr(a, f) != getattr(b, f))
    modes = [f.used_wide for f in ref]
    if sensor == "mono":                       # the sequence exercises what it claims to
        assert 12 in modes and ref[modes.index(12)].n_motion >= 100 and ref[modes.index(12)].N > 1.8 * cfg["n"]           # initialised, on 2 x nFeatures frames
        assert sum(m == 2 for m in modes) >= 2 and all(f.n_motion >= 15 and f.bow_hash for f in ref if f.used_wide == 2)   # TrackReferenceKeyFrame
    else:
        assert all((f.depth > 0).sum() > cfg["n"] // 2 for f in ref)
    assert sum(m == 4 for m in modes) >= 1 and all(f.n_motion >= 15 and f.n_extra % 10000 > 0 for f in ref if f.used_wide == 4)  # Relocalization found more by projection
    assert sum(f.n_local for f in ref) > 50 and sum(f.n_new_points > 0 for f in ref) >= 3
    return ref, got


def _voc(tmp_path_factory):
    p = tmp_path_factory.mktemp("voc") / "voc_no_final_newline.txt"               # (the reference's loader must not see the file's final newline)
    p.write_text(open(VOC).read().rstrip("\n"))
    return str(p)


@pytest.mark.parametrize("sensor", ["mono", "rgbd"])
def test_sequences_small(builds, request, tmp_path_factory, sensor):
    if "gpu" in request.node.name:
        pytest.skip("the GPU runs use configs[0]'s shape")
    S, D = builds
    run_and_compare(S, D, sensor, SMALL, nframes=12, voc_path=_voc(tmp_path_factory), dist=TUM1_DIST if sensor == "rgbd" else None)


@pytest.mark.gpu
@pytest.mark.parametrize("sensor", ["mono", "rgbd"])
def test_sequences_tum_shape(builds, request, tmp_path_factory, sensor):
    """BASELINE.json configs[0]'s shape (TUM fr1: 640x480, 1000 features, TUM1.yaml's camera - its distortion for the RGB-D run), 24 frames on the MI355X"""
    if "gpu" not in request.node.name.split("[")[1]:
        pytest.skip("full shapes run on the GPU builds")
    S, D = builds
    ref, got = run_and_compare(S, D, sensor, TUM1, nframes=24, voc_path=_voc(tmp_path_factory), dist=TUM1_DIST if sensor == "rgbd" else None, kf_every=4)
    ms_ref, ms_got = np.median([f.ms for f in ref[2:]]), np.median([f.ms for f in got[2:]])
    print(f"\n[sequence] {request.node.name}: {sensor} 640x480 / 1000: reference {ms_ref:.2f} ms/frame, drop-in {ms_got:.3f} ms/frame, {ms_ref / ms_got:.0f}x, 24 frames bit-exact")
