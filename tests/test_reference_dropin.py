"""The drop-in classes under the reference's OWN callers (the binding of INTEGRATION.md §2, steps 1-3, built and run).

oracle/_ref/liborbslam_dropin.so = the reference's src/Frame.cc and src/ORBmatcher.cc, compiled where they lie, against this repository's
include/ORBextractor.h + orb_slam2_amd/cpp/ORBextractor.cc in place of the reference's header and src/ORBextractor.cc, with
ORBmatcher::DescriptorDistance and ORBmatcher::SearchForInitialization taken from orb_slam2_amd/cpp/ORBmatcher.cc (the reference's two
bodies are renamed out of the way in its translation unit), linked to the CPU emulation build of the HIP kernels.  In that build the
reference's Frame constructors run
    Frame::ExtractORB -> (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors)          (Frame.cc:247-253)
on the drop-in class, everything behind (UndistortKeyPoints, ComputeStereoFromRGBD, AssignFeaturesToGrid) consumes its output, and
SearchForInitialization on two such Frames goes through orbhip_search_for_initialization_bounds with Frame's static image bounds.
Every frame and every match list must equal what the all-reference build (liborbslam_ref.so) makes.
A second build, liborbslam_dropin_full.so ("all-steps"), additionally applies the optional steps 3b / 3d' of INTEGRATION.md: the bodies of
Frame::ComputeStereoMatches, UndistortKeyPoints, ComputeImageBounds and ComputeStereoFromRGBD forward to the extractor (device stereo
matching without the pyramid download, mvKeysUn and the depth columns from the device) — same requirement.
Monocular, distorted monocular, RGB-D and stereo constructors (the stereo one extracts the two images on two std::threads — two device
contexts used concurrently — and its ComputeStereoMatches reads the drop-in class's mvImagePyramid).  Skipped where /root/reference is not
mounted."""
import numpy as np
import pytest

from orb_slam2_amd import synth
from conftest import gpu_session


@pytest.fixture(scope="module", params=["steps1-3", "all-steps", pytest.param("steps1-3-gpu", marks=pytest.mark.gpu), pytest.param("all-steps-gpu", marks=pytest.mark.gpu)])
def builds(request):
    """The binding on the CPU emulation of the kernels (here) and — the same reference sources, the same drop-in classes, linked to the real
    liborbhip.so (oracle/_ref/liborbslam_dropin*_gpu.so, built here, shipped with the repository snapshot) — on the MI355X box."""
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


def _same(a, b):
    return (a.N == b.N and a.keys.tobytes() == b.keys.tobytes() and a.keys_un.tobytes() == b.keys_un.tobytes() and np.array_equal(a.desc, b.desc)
            and a.u_right.tobytes() == b.u_right.tobytes() and a.depth.tobytes() == b.depth.tobytes())


@pytest.mark.parametrize("w,h,n,levels,scale,blur", [(480, 360, 700, 8, 1.2, 0), (333, 250, 300, 6, 1.3, 0), (480, 360, 700, 8, 1.2, 1)])
def test_reference_frame_on_dropin_extractor(builds, request, monkeypatch, w, h, n, levels, scale, blur):
    S, D = builds
    if "all-steps" in request.node.name and (w != 480 or blur):
        pytest.skip("one geometry is enough for the forwarding members")
    # blur = 1: the rounding of cv::GaussianBlur in an x86-64 (SSE2) OpenCV build — the drop-in class's own default on such hosts — on both sides
    monkeypatch.setenv("ORB_REF_BLUR_ROUND_MODE", str(blur))
    seq = synth.sequence(w, h, 2, seed=w + n)
    tum1 = (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)
    cam = dict(fx=517.3 * w / 640, fy=516.5 * h / 480, cx=318.6 * w / 640, cy=255.3 * h / 480)
    rng = np.random.default_rng(1)
    depth = (1.0 + rng.random((h, w))).astype(np.float32)
    depth[rng.random((h, w)) < 0.2] = 0.0
    variants = (dict(), dict(depth=depth, dist=tum1, **cam)) if "steps1-3" in request.node.name else (dict(dist=tum1, **cam), dict(depth=depth, **cam))
    for kw in variants:
        S.RefFrame._geometry = None
        S.RefFrame._geometry_other.clear()
        R = [S.RefFrame(im, nfeatures=n, nlevels=levels, scale=scale, **kw) for im in seq]
        F = [S.RefFrame(im, nfeatures=n, nlevels=levels, scale=scale, library=D, **kw) for im in seq]
        assert all(_same(a, b) for a, b in zip(R, F)) and R[0].N > 100
        assert S.RefFrame.bounds().tobytes() == S.RefFrame.bounds(D).tobytes()
        if "depth" in kw:
            assert (R[0].depth > 0).sum() > 50
        # all-reference matcher on reference frames  vs  drop-in matcher (C ABI, kernels) on frames whose features came from the drop-in extractor
        n_r, m_r, p_r = S.search_for_initialization(R[0], R[1], window=80, nnratio=0.9, check_ori=True)
        n_f, m_f, p_f = S.search_for_initialization(F[0], F[1], window=80, nnratio=0.9, check_ori=True)
        assert n_r == n_f and np.array_equal(m_r, m_f) and p_r.tobytes() == p_f.tobytes() and n_r > 20
        for _ in range(50):
            x, y, r = np.float32(rng.uniform(0, w)), np.float32(rng.uniform(0, h)), np.float32(rng.uniform(2, 90))
            if "all-steps" in request.node.name:
                # built with --skip-host-grid: the host's 64 x 48 grid has no reader left (src/ORBmatcher.cc is the drop-in's and searches the device's grid), it stays empty
                assert len(F[1].features_in_area(x, y, r, 0, 2)) == 0
            else:
                assert np.array_equal(R[1].features_in_area(x, y, r, 0, 2), F[1].features_in_area(x, y, r, 0, 2))
        for f in R + F:
            f.close()
    S.RefFrame._geometry = None


def test_reference_stereo_frame_on_dropin_extractors(builds):
    """Frame::Frame(imLeft, imRight, ...) (Frame.cc:62-115): threadLeft / threadRight call the two drop-in extractor objects concurrently, then
    the reference's own ComputeStereoMatches walks both classes' mvImagePyramid members."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_parity_stereo import stereo_pair
    S, D = builds
    fx, bf = 718.856, 386.1448
    for (w, h, n, seed, disp) in ((400, 300, 500, 8, 7),):
        L, R = stereo_pair(w, h, seed, disp)
        S.RefFrame._geometry = None
        S.RefFrame._geometry_other.clear()
        a = S.RefFrame(L, R, nfeatures=n, fx=fx, fy=fx, cx=w / 2, cy=h / 2, bf=bf)
        b = S.RefFrame(L, R, nfeatures=n, fx=fx, fy=fx, cx=w / 2, cy=h / 2, bf=bf, library=D)
        assert _same(a, b) and (a.depth > 0).sum() > 100
        a.close(); b.close()
    S.RefFrame._geometry = None


def test_projection_matchers_through_the_binding(builds, request):
    """INTEGRATION.md §2-3c in the "all-steps" build: ORBmatcher::SearchByProjection(Frame, MapPoints, th) and (CurrentFrame, LastFrame, th, bMono)
    keep the reference's projection / bookkeeping code and send their search loops through orbhip_search_by_projection_bounds; the final
    Frame::mvpMapPoints and the return values must equal the unmodified reference's, on mono and stereo frames."""
    if "all-steps" not in request.node.name:
        pytest.skip("the steps 1-3 build keeps the reference's own search loops")
    S, D = builds
    w, h, n = 480, 360, 700
    seq = synth.sequence(w, h, 2, seed=41)
    for stereo in (False, True):
        S.RefFrame._geometry = None
        S.RefFrame._geometry_other.clear()
        kw = dict(nfeatures=n)
        if stereo:
            kw.update(fx=64.0, fy=64.0, cx=0.0, cy=0.0, bf=40.0)
        mk = lambda im, lib: S.RefFrame(im, np.roll(im, -9, axis=1) if stereo else None, library=lib, **kw)
        R, F = [mk(im, None) for im in seq], [mk(im, D) for im in seq]
        kl, dl, kc = R[0].keys_un, R[0].desc, R[1].keys_un
        rng = np.random.default_rng(5 + stereo)
        nq = len(kl)
        # local map points (Tracking::SearchLocalPoints)
        for th in (3.0,):
            px = (kl["x"] - 3.0 + rng.normal(0, 1, nq)).astype(np.float32); py = (kl["y"] - 1.0 + rng.normal(0, 1, nq)).astype(np.float32)
            pxr = (px - rng.uniform(2, 40, nq)).astype(np.float32); level = kl["octave"].astype(np.int32)
            vc = np.where(rng.random(nq) < 0.5, 0.9995, 0.9).astype(np.float32)
            inview = (rng.random(nq) < 0.85).astype(np.uint8); bad = (rng.random(nq) < 0.05).astype(np.uint8); nobs = (rng.random(nq) < 0.9).astype(np.int32)
            state = rng.choice([0, 0, 0, 1, 2], len(kc)).astype(np.uint8)
            n_r, fq_r = S.search_by_projection_points(R[1], px, py, pxr, level, vc, inview, bad, nobs, dl, state, th=th, nnratio=0.8)
            n_f, fq_f = S.search_by_projection_points(F[1], px, py, pxr, level, vc, inview, bad, nobs, dl, state, th=th, nnratio=0.8)
            assert n_r == n_f and np.array_equal(fq_r, fq_f) and n_r > 100
        # last frame's points under the motion model (TrackWithMotionModel)
        fx = np.float32(64.0 if stereo else 1.0)
        for th, ori in ((15.0, True), (7.0, False)):
            has = (rng.random(nq) < 0.85).astype(np.uint8); outl = (rng.random(nq) < 0.1).astype(np.uint8)
            X = (kl["x"] - 3.0 + rng.normal(0, 1.5, nq)).astype(np.float32); Y = (kl["y"] - 1.0 + rng.normal(0, 1.5, nq)).astype(np.float32)
            state = rng.choice([0, 0, 0, 1, 2], len(kc)).astype(np.uint8)
            args = (has, X / fx, Y / fx, np.ones(nq, np.float32), dl)
            n_r, fq_r = S.search_by_projection_last(R[1], R[0], *args, outlier=outl, cur_state=state, th=th, mono=not stereo, nnratio=0.9, check_ori=ori)
            n_f, fq_f = S.search_by_projection_last(F[1], F[0], *args, outlier=outl, cur_state=state, th=th, mono=not stereo, nnratio=0.9, check_ori=ori)
            assert n_r == n_f and np.array_equal(fq_r, fq_f) and n_r > 100
            assert not ori or (fq_r == -2).sum() > 0                  # features claimed and NULLed by the rotation check are part of the comparison
        for f in R + F:
            f.close()
    S.RefFrame._geometry = None


def test_back_end_matchers_through_the_binding(builds, request):
    """The other two SearchByProjection overloads (loop closing: KeyFrame + Sim3, ORBmatcher.cc:290-403; relocalisation: Frame + KeyFrame,
    :1472-1599) with their search loops sent through orbhip_search_by_projection_bounds, and ORBmatcher::Fuse(pKF, vpMapPoints, th) (:825-972)
    around orbhip_search_best_in_window_bounds, both SearchByBoW overloads (:159-288, :522-655) through orbhip_search_by_bow and
    SearchForTriangulation (:657-823) through orbhip_search_for_triangulation, SearchBySim3 (:1102-1326) through two
    orbhip_search_best_in_window_bounds calls, in the "all-steps" build."""
    if "all-steps" not in request.node.name:
        pytest.skip("the steps 1-3 build keeps the reference's own search loops")
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_reference_matchers import _world
    S, D = builds
    w, h, n = 480, 360, 700
    seq = synth.sequence(w, h, 2, seed=41)
    S.RefFrame._geometry = None
    S.RefFrame._geometry_other.clear()
    R, F = [S.RefFrame(im, nfeatures=n) for im in seq], [S.RefFrame(im, nfeatures=n, library=D) for im in seq]
    kl, dl, kc = R[0].keys_un, R[0].desc, R[1].keys_un
    rng = np.random.default_rng(12)
    X, Y, level = _world(rng, kl, w, h)
    nq = len(kl)
    bad = (rng.random(nq) < 0.05).astype(np.uint8)
    ms = (rng.random(len(kc)) < 0.2).astype(np.uint8)
    n_r, fq_r = S.search_by_projection_kf(R[1], ms, X, Y, np.ones(nq, np.float32), level, bad, dl, th=10)
    n_f, fq_f = S.search_by_projection_kf(F[1], ms, X, Y, np.ones(nq, np.float32), level, bad, dl, th=10)
    assert n_r == n_f and np.array_equal(fq_r, fq_f) and n_r > 100
    for th, orb_dist, ori in ((10.0, 100, True), (3.0, 64, False)):
        has = (rng.random(nq) < 0.8).astype(np.uint8); found = (rng.random(nq) < 0.1).astype(np.uint8)
        cs = rng.choice([0, 0, 0, 1, 2], len(kc)).astype(np.uint8)
        a = (has, X, Y, np.ones(nq, np.float32), level, bad, found, dl, cs)
        n_r, fq_r = S.search_by_projection_reloc(R[1], R[0], *a, th=th, orb_dist=orb_dist, nnratio=0.9, check_ori=ori)
        n_f, fq_f = S.search_by_projection_reloc(F[1], F[0], *a, th=th, orb_dist=orb_dist, nnratio=0.9, check_ori=ori)
        assert n_r == n_f and np.array_equal(fq_r, fq_f) and n_r > 100
    # Fuse (LocalMapping::SearchInNeighbors): two passes around one orbhip_search_best_in_window_bounds call, the reference's surgery block moved
    for th in (3.0, 7.0):
        nobs = rng.integers(0, 4, nq).astype(np.int32)
        state = rng.choice([0, 0, 1, 2], len(kc)).astype(np.uint8)
        a = (state, X, Y, np.ones(nq, np.float32), level, nobs, bad, dl)
        n_r, b_r = S.fuse(R[1], *a, th=th)
        n_f, b_f = S.fuse(F[1], *a, th=th)
        assert n_r == n_f and np.array_equal(b_r, b_f) and n_r > 50
    # SearchByBoW, both overloads: the FeatureVector maps flattened, one orbhip_search_by_bow call, the reference's point vectors written back
    import os
    from oracle import orb_oracle as O
    ov = O.OracleVocabulary(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "voc_k6_L3_ref.txt"))
    for mode, levelsup, ratio, ori in ((0, 2, 0.7, True), (1, 3, 0.9, True), (0, 4, 0.75, False)):
        fv1, fv2 = ov.transform(R[0].desc, levelsup)[2:], ov.transform(R[1].desc, levelsup)[2:]
        has1 = (rng.random(R[0].N) < 0.75).astype(np.uint8); bad1 = (rng.random(R[0].N) < 0.07).astype(np.uint8)
        has2 = (rng.random(R[1].N) < 0.85).astype(np.uint8); bad2 = (rng.random(R[1].N) < 0.07).astype(np.uint8)
        n_r, m_r = S.search_by_bow(mode, R[0], has1, bad1, fv1, R[1], has2, bad2, fv2, nnratio=ratio, check_ori=ori)
        n_f, m_f = S.search_by_bow(mode, F[0], has1, bad1, fv1, F[1], has2, bad2, fv2, nnratio=ratio, check_ori=ori)
        assert n_r == n_f and np.array_equal(m_r, m_f) and n_r > 30
    # SearchForTriangulation: the reference's epipole code stays, the search fills its vMatches12 through orbhip_search_for_triangulation
    Fm = np.array([[0, -1e-3, 1.0 / 300], [1e-3, 0, -3.0 / 300], [-1.0 / 300, 3.0 / 300, 0]], np.float32) + rng.normal(0, 1e-5, (3, 3)).astype(np.float32)
    for levelsup, only, ori, t2w in ((2, False, True, (0.3, 0.1, 1.0)), (1, False, False, (2.0, 1.0, 4.0))):
        fv1, fv2 = ov.transform(R[0].desc, levelsup)[2:], ov.transform(R[1].desc, levelsup)[2:]
        has1 = (rng.random(R[0].N) < 0.3).astype(np.uint8); has2 = (rng.random(R[1].N) < 0.3).astype(np.uint8)
        n_r, m_r = S.search_for_triangulation(R[0], has1, fv1, R[1], has2, fv2, Fm, np.array(t2w, np.float32), only_stereo=only, check_ori=ori)
        n_f, m_f = S.search_for_triangulation(F[0], has1, fv1, F[1], has2, fv2, Fm, np.array(t2w, np.float32), only_stereo=only, check_ori=ori)
        assert n_r == n_f and np.array_equal(m_r, m_f) and n_r > 30
    # Fuse, Sim3 overload (loop closing): same two-pass shape, no chi-square gate, vpReplacePoint filled by the moved surgery block
    st = rng.choice([0, 0, 1], len(kc)).astype(np.uint8)
    n_r, b_r = S.fuse_sim3(R[1], st, X, Y, np.ones(nq, np.float32), level, bad, dl, th=4.0)
    n_f, b_f = S.fuse_sim3(F[1], st, X, Y, np.ones(nq, np.float32), level, bad, dl, th=4.0)
    assert n_r == n_f and np.array_equal(b_r, b_f) and n_r > 50
    # SearchBySim3: both directions collected by the reference's projection code, searched by two calls in front of its mutual check
    X2 = (kc["x"] + 3.0 + rng.normal(0, 1.2, len(kc))).astype(np.float32); Y2 = (kc["y"] + 1.0 + rng.normal(0, 1.2, len(kc))).astype(np.float32)
    lev2 = np.clip(kc["octave"] + rng.integers(0, 2, len(kc)), 0, 7).astype(np.int32)
    hs1 = (rng.random(nq) < 0.8).astype(np.uint8); hs2 = (rng.random(len(kc)) < 0.8).astype(np.uint8)
    a = (hs1, X, Y, np.ones(nq, np.float32), level, dl)
    b = (hs2, X2, Y2, np.ones(len(kc), np.float32), lev2, R[1].desc)
    n_r, m_r = S.search_by_sim3(R[0], *a, R[1], *b, th=7.5)
    n_f, m_f = S.search_by_sim3(F[0], *a, F[1], *b, th=7.5)
    assert n_r == n_f and np.array_equal(m_r, m_f) and n_r > 100
    kw = dict(nfeatures=n, fx=64.0, fy=64.0, cx=0.0, cy=0.0, bf=40.0)                  # stereo key frame: the chi-square gate's stereo branch
    S.RefFrame._geometry = None
    S.RefFrame._geometry_other.clear()
    ks_r = S.RefFrame(seq[1], np.roll(seq[1], -9, axis=1), **kw); ks_f = S.RefFrame(seq[1], np.roll(seq[1], -9, axis=1), library=D, **kw)
    a = (rng.choice([0, 0, 1], len(kc)).astype(np.uint8), X / np.float32(64), Y / np.float32(64), np.ones(nq, np.float32), level, rng.integers(0, 4, nq).astype(np.int32), bad, dl)
    n_r, b_r = S.fuse(ks_r, *a, th=3.0)
    n_f, b_f = S.fuse(ks_f, *a, th=3.0)
    assert n_r == n_f and np.array_equal(b_r, b_f) and n_r > 50 and int((ks_r.u_right >= 0).sum()) > 100
    ks_r.close(); ks_f.close()
    for f in R + F:
        f.close()
    S.RefFrame._geometry = None


def test_compute_bow_through_the_binding(builds, tmp_path):
    """Frame::ComputeBoW (Frame.cc:395-402) on frames built by the reference's constructor: the reference's DBoW2 vocabulary in the all-reference
    and the steps 1-3 builds, this repository's ORBVocabulary class (DBoW2's own BowVector / FeatureVector types) in the
    all-steps build — same maps, and equal to the oracle's transform.  (The reference's loader must not see the file's final newline, H6.)"""
    import os
    from oracle import orb_oracle as O
    S, D = builds
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "voc_k6_L3_ref.txt")
    voc = tmp_path / "voc_no_final_newline.txt"
    voc.write_text(open(golden).read().rstrip("\n"))
    w, h, n = 480, 360, 700
    img = synth.frame(w, h, seed=17)
    S.RefFrame._geometry = None
    S.RefFrame._geometry_other.clear()
    a, b = S.RefFrame(img, nfeatures=n), S.RefFrame(img, nfeatures=n, library=D)
    ra, rb = a.compute_bow(voc), b.compute_bow(voc)
    assert all(x.tobytes() == y.tobytes() for x, y in zip(ra, rb)) and len(ra[0]) > 50
    want = O.OracleVocabulary(golden).transform(a.desc, 4)
    assert all(x.tobytes() == y.astype(x.dtype).tobytes() for x, y in zip(ra, want))
    a.close(); b.close()
    S.RefFrame._geometry = None
