"""The drop-in as ORB_SLAM2 drives it: Tracking's per-frame sequence on a stereo stream, frame after frame, through the binding.

oracle/orbslam_ref_wrap.cpp::orbslam_ref_tracking_loop restates what Tracking::GrabImageStereo -> Track() does with every stereo pair
(Tracking.cc:167-204, 267-503) around the reference's own code: the stereo Frame constructor with its two extractor threads
(Frame.cc:61-117), ComputeStereoMatches (Frame.cc:466-640), StereoInitialization's map points (Tracking.cc:509-561),
TrackWithMotionModel's SearchByProjection(Current, Last, 7, false) under a constant-velocity pose (Tracking.cc:867-928,
ORBmatcher.cc:1328-1470), SearchLocalPoints' Frame::isInFrustum + SearchByProjection(Frame, MapPoints, 1) (Tracking.cc:1143-1193,
Frame.cc:269-325, ORBmatcher.cc:45-129), CreateNewKeyFrame's stereo points (Tracking.cc:1063-1133) and mLastFrame = Frame(mCurrentFrame).
The optimiser is not part of this build: the pose it would return is the sequence's ground truth.

The same loop runs in the all-reference build (liborbslam_ref.so) and in the builds whose extractor / stereo matcher / projection matchers
are this repository's (liborbslam_dropin*.so; the *_gpu ones on the real liborbhip.so).  Every frame must agree in everything a later frame
could depend on: key points, mvKeysUn, descriptors, mvuRight / mvDepth, the map point of every feature after each of the two matchers, and
the counters (return values, points in the frustum, points created).  BASELINE.json configs[2] (EuRoC 752x480 / 1200) and configs[1]
(KITTI 1241x376 / 2000) at their own shapes run on the GPU; a small shape runs here on the CPU emulation of the kernels."""
import numpy as np
import pytest

from orb_slam2_amd import synth
from conftest import gpu_session

EUROC = dict(w=752, h=480, n=1200, fx=435.2047, fy=435.2047, cx=367.4517, cy=252.2008, bf=47.9064, th_depth=35.0)       # Examples/Stereo/EuRoC.yaml
KITTI = dict(w=1241, h=376, n=2000, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448, th_depth=35.0)       # Examples/Stereo/KITTI00-02.yaml
SMALL = dict(w=400, h=300, n=500, fx=231.5, fy=231.5, cx=200.0, cy=150.0, bf=25.5, th_depth=35.0)


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


def run_and_compare(S, D, cfg, nframes, seed, kf_every):
    L, R, T, P = synth.stereo_sequence(cfg["w"], cfg["h"], nframes, cfg["fx"], cfg["bf"], seed=seed)
    args = (L, R, T, P, cfg["n"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["bf"], cfg["th_depth"])
    ref = S.tracking_loop(*args, kf_every=kf_every)
    got = S.tracking_loop(*args, kf_every=kf_every, library=D)
    for k, (a, b) in enumerate(zip(ref, got)):
        assert a.same(b), f"frame {k} differs: " + ", ".join(f"{f} {getattr(a, f)} vs {getattr(b, f)}" for f in a.FIELDS if getattr(a, f) != getattr(b, f))
    # the sequence exercises what it claims to: stereo depth, both matchers, new points
    assert all((f.depth > 0).sum() > cfg["n"] // 5 for f in ref)
    assert all(f.n_motion > 50 and f.n_local > 5 and not f.used_wide for f in ref[1:])
    assert sum(f.n_new_points > 0 for f in ref) >= 1 + (nframes - 1) // kf_every
    return ref, got


def test_front_end_loop_small(builds, request):
    if "gpu" in request.node.name:
        pytest.skip("the GPU runs use the BASELINE shapes")
    S, D = builds
    run_and_compare(S, D, SMALL, nframes=7, seed=3, kf_every=3)


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg", [("euroc", EUROC), ("kitti", KITTI)])
def test_front_end_loop_baseline_shapes(builds, request, name, cfg):
    """>= 20 stereo frames at 752x480 / 1200 and 1241x376 / 2000, on the MI355X, through the reference's Frame.cc / ORBmatcher.cc"""
    if "gpu" not in request.node.name.split("[")[1]:
        pytest.skip("full shapes run on the GPU builds")
    S, D = builds
    if "steps1-3" in request.node.name and name == "kitti":
        pytest.skip("one shape is enough for the steps 1-3 build")
    ref, got = run_and_compare(S, D, cfg, nframes=24, seed=5, kf_every=5)
    ms_ref, ms_got = np.median([f.ms for f in ref[1:]]), np.median([f.ms for f in got[1:]])
    part = lambda fs: " + ".join(f"{np.median([getattr(f, k) for f in fs[1:]]):.3f}" for k in ("ms_ctor", "ms_motion", "ms_local"))
    print(f"\n[dropin_loop] {request.node.name}: {cfg['w']}x{cfg['h']} / {cfg['n']}: reference {ms_ref:.2f} ms/frame ({part(ref)}), drop-in {ms_got:.3f} ms/frame ({part(got)}: "
          f"Frame constructor + motion-model search + local-map search), {ms_ref / ms_got:.0f}x, bit-exact over 24 frames")
