"""H3 for the matcher's algebra, measured: the reference's Frame.cc / ORBmatcher.cc built with the reference's OWN flags (CMakeLists.txt:11-14: -O3 -march=native,
C++ => -ffp-contract=fast - gcc fuses the scalar projection statements, `fx*xc*invz+cx` ..., into FMAs; `make -C oracle ref_native_slam`) against the canonical
build (-ffp-contract=off: one rounding per operation, the form the drop-in's device code reproduces).  The cv::Mat stand-in keeps its own arithmetic
uncontracted in both, like a separately built OpenCV.  Every pose-guided member under the general poses of tests/test_projection_poses.py; reported: how many
entries of the members' outputs differ.  Asserted: the builds agree on (almost) every match - a contraction moves a projected pixel by an ulp, which matters only
where that ulp crosses a cell / radius / level boundary.  CPU only (-march=native is the build host's)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import gpu_session  # noqa: E402


def test_native_flags_build_of_the_matcher_against_the_canonical_build(request, capsys):
    if gpu_session(request.config):
        pytest.skip("CPU only")
    from oracle import orbslam_ref as S
    if not (S.build() and S.build_native()):
        pytest.skip("reference sources not mounted")
    import test_projection_poses as T
    checked, differs = T._run(S, S.native_lib(), strict=False)
    total, bad = sum(v for k, v in checked.items() if k != "frustum"), sum(v for k, v in differs.items() if k != "frustum")
    with capsys.disabled():
        print("\nnative-flags build (-O3 -march=native, FMA contraction) against the canonical build: entries of the members' outputs that differ: "
              + ", ".join(f"{k} {differs.get(k, 0)}" for k in sorted(checked)) + "  (of " + ", ".join(f"{k} {checked[k]}" for k in sorted(checked)) + " matches)")
    assert total > 5000
    assert bad <= total // 200, (bad, total)        # (0 on this host; a boundary crossing per few thousand matches would still be the same exposure)


def test_native_flags_build_of_the_dropin_against_the_native_flags_reference(request, capsys):
    """What a maintainer's own CMake build of the installed drop-in is (the reference's flags: -O3 -march=native) against the reference built the same way:
    the repository's ORBmatcher.cc / ORBextractor.cc and the patched Frame.cc are contracted by the compiler like the reference's files, the device (here: the
    emulation) rounds every operation of the projection once.  ORBmatcher::IsInFrustum is host code written in the reference's expression shapes: its outputs must
    be identical under these flags too.  The members' matches: reported; asserted equal up to the boundary crossings H3 / H11 allow (0 on this host)."""
    if gpu_session(request.config):
        pytest.skip("CPU only")
    from oracle import orbslam_ref as S
    if not (S.build() and S.build_native() and S.build_dropin() and S.build_dropin_native()):
        pytest.skip("reference sources not mounted")
    import test_projection_poses as T
    checked, differs = T._run(S, S.dropin_native_lib(), strict=False, base=S.native_lib())
    total, bad = sum(v for k, v in checked.items() if k != "frustum"), sum(v for k, v in differs.items() if k != "frustum")
    with capsys.disabled():
        print("\nnative-flags drop-in against the native-flags reference: entries of the members' outputs that differ: "
              + ", ".join(f"{k} {differs.get(k, 0)}" for k in sorted(checked)) + "  (of " + ", ".join(f"{k} {checked[k]}" for k in sorted(checked)) + " matches)")
    assert total > 5000
    assert bad <= total // 200, (bad, total)
    assert differs.get("frustum", 0) == 0, "ORBmatcher::IsInFrustum under the reference's own flags"


def test_native_flags_front_end_loop(request):
    """Tracking's per-frame sequence (tests/test_dropin_loop.py) with BOTH sides built with the reference's own flags: the native reference's extractor fuses the
    pattern rotation (H3: the drop-in class then starts with fp_contract = 1, its own translation unit being compiled with FMA code generation), its matcher
    members contract their scalar statements; every frame must agree in key points, descriptors, mvuRight / mvDepth, the map point of every feature after each
    matcher and the counters."""
    if gpu_session(request.config):
        pytest.skip("CPU only")
    from oracle import orbslam_ref as S
    if not (S.build() and S.build_native() and S.build_dropin() and S.build_dropin_native()):
        pytest.skip("reference sources not mounted")
    from orb_slam2_amd import synth
    import test_dropin_loop as TL
    cfg = TL.SMALL
    L, R, T, P = synth.stereo_sequence(cfg["w"], cfg["h"], 7, cfg["fx"], cfg["bf"], seed=3)
    args = (L, R, T, P, cfg["n"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["bf"], cfg["th_depth"])
    ref = S.tracking_loop(*args, kf_every=3, library=S.native_lib())
    got = S.tracking_loop(*args, kf_every=3, library=S.dropin_native_lib())
    for k, (a, b) in enumerate(zip(ref, got)):
        assert a.same(b), f"frame {k} differs: " + ", ".join(f"{f} {getattr(a, f)} vs {getattr(b, f)}" for f in a.FIELDS if getattr(a, f) != getattr(b, f))
    assert all(f.n_motion > 50 and f.n_local > 5 for f in ref[1:])


@pytest.mark.parametrize("sensor", ["mono", "rgbd"])
def test_native_flags_sequences(request, tmp_path_factory, sensor):
    """... and the monocular / RGB-D sequences with relocalisation (tests/test_sequences.py), both sides built with the reference's own flags: initialisation on
    2 x nFeatures frames, TrackReferenceKeyFrame with its bags of words, the distorted RGB-D constructor, Relocalization's two projection searches."""
    if gpu_session(request.config):
        pytest.skip("CPU only")
    from oracle import orbslam_ref as S
    if not (S.build() and S.build_native() and S.build_dropin() and S.build_dropin_native()):
        pytest.skip("reference sources not mounted")
    from orb_slam2_amd import synth
    import test_sequences as TS
    cfg = TS.SMALL
    L, R, T, P, depth = synth.stereo_sequence(cfg["w"], cfg["h"], 12, cfg["fx"], cfg["bf"], seed=3, return_depth=True)
    args = (sensor, L, depth, T, P, cfg["n"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["bf"], cfg["th_depth"], TS._voc(tmp_path_factory))
    kw = dict(dist=TS.TUM1_DIST if sensor == "rgbd" else None, kf_every=3, lost_every=5)
    ref = S.sequence_loop(*args, library=S.native_lib(), **kw)
    got = S.sequence_loop(*args, library=S.dropin_native_lib(), **kw)
    for k, (a, b) in enumerate(zip(ref, got)):
        assert a.same(b), f"{sensor} frame {k} differs: " + ", ".join(f"{f} {getattr(a, f)} vs {getattr(b, f)}" for f in a.FIELDS + ("n_extra", "bow_hash") if getattr(a, f) != getattr(b, f))
    assert sum(f.used_wide == 4 for f in ref) >= 1 and sum(f.n_local for f in ref) > 50
