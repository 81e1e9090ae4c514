"""The oracle's matcher / stereo / feature-grid restatements against the REFERENCE's own src/Frame.cc and src/ORBmatcher.cc.

oracle/_ref/liborbslam_ref.so = those two files + src/ORBextractor.cc compiled where they lie under /root/reference
(`make -C oracle ref`): the Frame constructors (mono and stereo, with the reference's two extractor threads),
AssignFeaturesToGrid / GetFeaturesInArea, ComputeStereoMatches and every ORBmatcher member are the reference's code; the
OpenCV image primitives are the oracle's restatements, MapPoint / KeyFrame accessors (and the KeyFrame-from-Frame constructor)
are plain member copies in the wrapper
(oracle/orbslam_ref_wrap.cpp).  Cameras: zero distortion; for the projection searches identity pose and fx = fy = 1,
cx = cy = 0, so that a map point at (X, Y, 1) projects to (X, Y) exactly and the reference's own projection code feeds its
search loop with the same numbers the flat queries carry.  Skipped where neither /root/reference nor a prebuilt library exists."""
import os
import sys

import numpy as np
import pytest

import orb_slam2_amd
from orb_slam2_amd import synth

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_parity_stereo import stereo_pair  # noqa: E402

FX, BF = np.float32(718.856), np.float32(386.1448)             # KITTI00-02.yaml


@pytest.fixture(scope="module")
def ref():
    from oracle import orbslam_ref as S
    if not S.build():
        pytest.skip("reference sources not mounted (oracle/_ref/liborbslam_ref.so absent)")
    return S


@pytest.fixture(scope="module")
def pair_oracle(oracle):
    w, h, n = 480, 360, 700
    seq = synth.sequence(w, h, 2, seed=41)
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    return w, h, n, seq, ora.params()["scale_factors"], [ora.extract(im) for im in seq]


@pytest.fixture
def pair(ref, pair_oracle):
    """Frames are rebuilt per test: image bounds, grid cell sizes and intrinsics are STATIC members of the reference's Frame
    (Frame.cc:40-43, 204-221), set by the first frame after a geometry change — other tests construct other cameras."""
    w, h, n, seq, sf, K = pair_oracle
    ref.RefFrame._geometry = None
    F = [ref.RefFrame(im, nfeatures=n) for im in seq]
    yield w, h, n, seq, sf, K, F
    for f in F:
        f.close()


def test_frame_constructor_and_grid(ref, oracle, pair):
    """Frame::Frame (mono) runs the reference extractor; mvKeysUn == mvKeys without distortion; GetFeaturesInArea on the 64x48 grid."""
    w, h, n, seq, sf, K, F = pair
    for f in range(2):
        assert F[f].keys.tobytes() == K[f][0].tobytes() and F[f].keys_un.tobytes() == K[f][0].tobytes() and np.array_equal(F[f].desc, K[f][1])
        assert np.all(F[f].u_right == -1) and np.all(F[f].depth == -1)
    rng = np.random.default_rng(1)
    for _ in range(400):
        x, y = np.float32(rng.uniform(-40, w + 40)), np.float32(rng.uniform(-40, h + 40))
        r, mn, mx = np.float32(rng.uniform(1, 150)), int(rng.integers(-1, 4)), int(rng.integers(-1, 8))
        assert np.array_equal(F[1].features_in_area(x, y, r, mn, mx), oracle.features_in_area(K[1][0], w, h, x, y, r, mn, mx))
    assert ref.descriptor_distance(K[0][1][0], K[1][1][0]) == oracle.hamming(K[0][1][0], K[1][1][0])


@pytest.mark.parametrize("window,ratio,ori", [(100, 0.9, True), (50, 0.8, True), (100, 0.9, False), (15, 0.7, True), (300, 0.95, True)])
def test_search_for_initialization(ref, oracle, pair, window, ratio, ori):
    w, h, n, seq, sf, K, F = pair
    n_r, m_r, p_r = ref.search_for_initialization(F[0], F[1], window=window, nnratio=ratio, check_ori=ori)
    n_o, m_o, p_o = oracle.search_for_initialization(K[0][0], K[0][1], K[1][0], K[1][1], w, h, window=window, nnratio=ratio, check_ori=ori)
    assert n_r == n_o and np.array_equal(m_r, m_o) and p_r.tobytes() == p_o.tobytes()
    assert n_o > 40


@pytest.mark.parametrize("w,h,n,seed,disp", [(480, 360, 700, 5, 12), (640, 480, 1000, 6, 25), (1241, 376, 2000, 7, 40), (400, 300, 500, 8, 3), (640, 480, 800, 9, 60)])
def test_compute_stereo_matches(ref, oracle, w, h, n, seed, disp):
    """Frame::Frame (stereo): two extractor threads + ComputeStereoMatches.  The reference reads `mb` before it assigns it
    (Frame.cc:89 vs :113); the wrapper pre-seeds the member with mbf / fx."""
    L, R = stereo_pair(w, h, seed, disp)
    F = ref.RefFrame(L, R, nfeatures=n, fx=float(FX), fy=float(FX), cx=607.1928, cy=185.2157, bf=float(BF))
    eL, eR = oracle.OracleExtractor(n, 1.2, 8, 20, 7), oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    kl, _ = eL.extract(L)
    eR.extract(R)
    uo, do = oracle.stereo_matches(eL, eR, BF, BF / FX)
    assert F.keys.tobytes() == kl.tobytes()
    assert F.u_right.tobytes() == uo.tobytes() and F.depth.tobytes() == do.tobytes()
    assert int((uo >= 0).sum()) > n // 4
    F.close()


def _cur_state(rng, n):
    return rng.choice([0, 0, 0, 1, 2], n).astype(np.uint8)      # none / map point without observations / with observations


@pytest.mark.parametrize("th,stereo", [(1.0, False), (3.0, False), (5.0, True)])
def test_search_by_projection_local_map(ref, oracle, pair, th, stereo):
    """ORBmatcher(0.8).SearchByProjection(Frame&, vector<MapPoint*>&, th) (ORBmatcher.cc:45-129)."""
    w, h, n, seq, sf, K, F = pair
    (kl, dl), (kc, dc) = K
    cur, u_right = F[1], None
    if stereo:
        rgt = np.roll(seq[1], -9, axis=1)
        cur = ref.RefFrame(seq[1], rgt, nfeatures=n, fx=64.0, fy=64.0, cx=0.0, cy=0.0, bf=40.0)    # disparity search range = fx px
        assert cur.keys.tobytes() == kc.tobytes()
        u_right = cur.u_right
    rng = np.random.default_rng(int(th * 10))
    nq = len(kl)
    px = (kl["x"] - 3.0 + rng.normal(0, 1, nq)).astype(np.float32); py = (kl["y"] - 1.0 + rng.normal(0, 1, nq)).astype(np.float32)
    pxr = (px - rng.uniform(2, 40, nq)).astype(np.float32); level = kl["octave"].astype(np.int32)
    vc = np.where(rng.random(nq) < 0.5, 0.9995, 0.9).astype(np.float32)
    inview = (rng.random(nq) < 0.85).astype(np.uint8); bad = (rng.random(nq) < 0.05).astype(np.uint8); nobs = (rng.random(nq) < 0.9).astype(np.int32)
    state = _cur_state(rng, len(kc))
    n_r, fq_r = ref.search_by_projection_points(cur, px, py, pxr, level, vc, inview, bad, nobs, dl, state, th=th, nnratio=0.8)
    keep = np.nonzero((inview == 1) & (bad == 0))[0]                              # :55-59
    r = np.where(vc > np.float32(0.998), np.float32(2.5), np.float32(4.0)).astype(np.float32)   # RadiusByViewingCos :131-137
    if th != 1.0:
        r = (r * np.float32(th)).astype(np.float32)
    q = np.zeros(len(keep), oracle.PROJ_QUERY_DTYPE)
    q["x"], q["y"], q["radius"], q["ur"] = px[keep], py[keep], (r * sf[level])[keep].astype(np.float32), pxr[keep]
    q["min_level"], q["max_level"], q["blocks"] = level[keep] - 1, level[keep], nobs[keep] > 0
    n_o, fq_o = oracle.search_by_projection(kc, dc, w, h, q, dl[keep], 0, nnratio=0.8, th_high=100, u_right=u_right, blocked=(state == 2).astype(np.uint8))
    assert n_r == n_o and np.array_equal(fq_r, np.where(fq_o >= 0, keep[np.maximum(fq_o, 0)], -1))
    assert n_o > 100
    if stereo:
        assert int((u_right > 0).sum()) > 100
        cur.close()


@pytest.mark.parametrize("th,ori,stereo", [(7.0, True, False), (15.0, True, False), (15.0, False, False), (15.0, True, True)])
def test_search_by_projection_last_frame(ref, oracle, pair, th, ori, stereo):
    """ORBmatcher(0.9, ori).SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORBmatcher.cc:1328-1470), identity poses."""
    w, h, n, seq, sf, K, F = pair
    (kl, dl), (kc, dc) = K
    cur, u_right, bf, fx = F[1], None, np.float32(40.0), np.float32(1.0)
    if stereo:
        rgt = np.roll(seq[1], -9, axis=1)
        fx = np.float32(64.0)                                # a power of two: fx * (u / fx) == u exactly; disparity search range = fx px
        cur = ref.RefFrame(seq[1], rgt, nfeatures=n, fx=float(fx), fy=float(fx), cx=0.0, cy=0.0, bf=float(bf))
        u_right = cur.u_right
        assert int((u_right > 0).sum()) > 100
    rng = np.random.default_rng(int(th) + ori)
    nq = len(kl)
    has = (rng.random(nq) < 0.85).astype(np.uint8); outl = (rng.random(nq) < 0.1).astype(np.uint8)
    X = (kl["x"] - 3.0 + rng.normal(0, 1.5, nq)).astype(np.float32); Y = (kl["y"] - 1.0 + rng.normal(0, 1.5, nq)).astype(np.float32)
    X[:5] = -4.0; X[5:8] = w + 2.0; Y[8:10] = h + 1.0                                   # outside the image bounds: skipped (:1365-1368)
    state = _cur_state(rng, len(kc))
    n_r, fq_r = ref.search_by_projection_last(cur, F[0], has, X / fx, Y / fx, np.ones(nq, np.float32), dl, outlier=outl, cur_state=state, th=th, mono=not stereo,
                                              nnratio=0.9, check_ori=ori)
    keep = np.nonzero((has == 1) & (outl == 0) & ~((X < 0) | (X > w) | (Y < 0) | (Y > h)))[0]
    q = np.zeros(len(keep), oracle.PROJ_QUERY_DTYPE)
    oc = kl["octave"][keep]
    q["x"], q["y"], q["radius"], q["ur"] = X[keep], Y[keep], (np.float32(th) * sf[oc]).astype(np.float32), X[keep] - bf     # ur = u - mbf*invzc (:1404)
    q["min_level"], q["max_level"], q["blocks"], q["angle"] = oc - 1, oc + 1, 1, kl["angle"][keep]
    n_o, fq_o = oracle.search_by_projection(kc, dc, w, h, q, dl[keep], 1, nnratio=0.9, th_high=100, check_ori=ori, u_right=u_right, blocked=(state == 2).astype(np.uint8))
    # -2 (claimed, then removed by the rotation check -> NULL) is observable in the reference only where the feature held a point before the call
    want = np.where(fq_o >= 0, keep[np.maximum(fq_o, 0)], np.where((fq_o == -2) & (state != 0), -2, -1))
    assert n_r == n_o and np.array_equal(fq_r, want)
    assert n_o > 100 and (not ori or (fq_o == -2).sum() > 0)
    if stereo:
        cur.close()


@pytest.mark.parametrize("mode,levelsup,ratio,ori", [(0, 2, 0.7, True), (0, 1, 0.75, False), (1, 2, 0.8, True), (1, 3, 0.9, True), (0, 4, 0.7, True)])
def test_search_by_bow(ref, oracle, pair, mode, levelsup, ratio, ori):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ..) (ORBmatcher.cc:159-288) and (KeyFrame*, KeyFrame*, ..) (:522-655) through real
    KeyFrame / Frame objects whose mFeatVec comes from the golden vocabulary; some map points missing, some bad."""
    w, h, n, seq, sf, K, F = pair
    (k1, d1), (k2, d2) = K
    ov = oracle.OracleVocabulary(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "voc_k6_L3_ref.txt"))
    fv1, fv2 = ov.transform(d1, levelsup)[2:], ov.transform(d2, levelsup)[2:]
    rng = np.random.default_rng(mode * 10 + levelsup)
    has1 = (rng.random(len(k1)) < 0.75).astype(np.uint8); bad1 = (rng.random(len(k1)) < 0.07).astype(np.uint8)
    has2 = (rng.random(len(k2)) < 0.85).astype(np.uint8); bad2 = (rng.random(len(k2)) < 0.07).astype(np.uint8)
    n_r, m_r = ref.search_by_bow(mode, F[0], has1, bad1, fv1, F[1], has2, bad2, fv2, nnratio=ratio, check_ori=ori)
    v1 = (has1 & (1 - bad1)).astype(np.uint8)
    v2 = (has2 & (1 - bad2)).astype(np.uint8) if mode == 1 else None
    n_o, m_o = oracle.search_by_bow(mode, d1, k1["angle"], v1, fv1, d2, k2["angle"], v2, fv2, nnratio=ratio, check_ori=ori)
    assert n_r == n_o and np.array_equal(m_r, m_o)
    assert n_o > 30


@pytest.mark.parametrize("fx,cx,cy,stereo,only,ori,levelsup,t2w", [(1.0, 0.0, 0.0, False, False, True, 2, (0.3, 0.1, 1.0)), (64.0, 240.0, 180.0, False, False, True, 1, (2.0, 1.0, 4.0)),
                                                                    (64.0, 200.0, 150.0, True, False, True, 2, (0.5, 0.2, 1.0)), (64.0, 200.0, 150.0, True, True, False, 3, (0.5, 0.2, 1.0))])
def test_search_for_triangulation(ref, oracle, pair_oracle, fx, cx, cy, stereo, only, ori, levelsup, t2w):
    """ORBmatcher::SearchForTriangulation (ORBmatcher.cc:657-823) through real KeyFrame objects: key frame 1 at the origin, key
    frame 2 = [I | t2w], so the epipole the reference derives is (fx t.x / t.z + cx, fy t.y / t.z + cy)."""
    w, h, n, seq, sf, K = pair_oracle
    (k1, d1), (k2, d2) = K
    ref.RefFrame._geometry = None
    if stereo:
        F1 = ref.RefFrame(seq[0], np.roll(seq[0], -9, axis=1), nfeatures=n, fx=fx, fy=fx, cx=cx, cy=cy, bf=40.0)
        F2 = ref.RefFrame(seq[1], np.roll(seq[1], -7, axis=1), nfeatures=n, fx=fx, fy=fx, cx=cx, cy=cy, bf=40.0)
    else:
        F1, F2 = ref.RefFrame(seq[0], nfeatures=n, fx=fx, fy=fx, cx=cx, cy=cy), ref.RefFrame(seq[1], nfeatures=n, fx=fx, fy=fx, cx=cx, cy=cy)
    par = oracle.OracleExtractor(n, 1.2, 8, 20, 7).params()
    ov = oracle.OracleVocabulary(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "voc_k6_L3_ref.txt"))
    fv1, fv2 = ov.transform(d1, levelsup)[2:], ov.transform(d2, levelsup)[2:]
    rng = np.random.default_rng(int(fx) + levelsup)
    has1 = (rng.random(len(k1)) < 0.3).astype(np.uint8); has2 = (rng.random(len(k2)) < 0.3).astype(np.uint8)
    F = np.array([[0, -1e-3, 1.0 / 300], [1e-3, 0, -3.0 / 300], [-1.0 / 300, 3.0 / 300, 0]], np.float32) + rng.normal(0, 1e-5, (3, 3)).astype(np.float32)
    t = np.array(t2w, np.float32)
    n_r, m_r = ref.search_for_triangulation(F1, has1, fv1, F2, has2, fv2, F, t, only_stereo=only, check_ori=ori)
    invz = np.float32(1.0) / t[2]
    ex, ey = np.float32(fx) * t[0] * invz + np.float32(cx), np.float32(fx) * t[1] * invz + np.float32(cy)      # :667-669
    n_o, m_o = oracle.search_for_triangulation(d1, k1, has1, F1.u_right >= 0, fv1, d2, k2, has2, F2.u_right >= 0, fv2, F, ex, ey, par["scale_factors"], par["sigma2"],
                                               only_stereo=only, check_ori=ori)
    assert n_r == n_o and np.array_equal(m_r, m_o)
    assert n_o > 30
    F1.close(); F2.close()


def _world(rng, kl, w, h):
    nq = len(kl)
    X = (kl["x"] - 3.0 + rng.normal(0, 1.2, nq)).astype(np.float32); Y = (kl["y"] - 1.0 + rng.normal(0, 1.2, nq)).astype(np.float32)
    X[:4] = -2.0; X[4:7] = w + 1.0; Y[7:9] = h                                   # outside the image: skipped by IsInImage / the bounds test
    level = np.clip(kl["octave"] + rng.integers(0, 2, nq), 0, 7).astype(np.int32)
    return X, Y, level


@pytest.mark.parametrize("th,stereo", [(3.0, False), (5.0, False), (3.0, True)])
def test_fuse(ref, oracle, pair, th, stereo):
    """ORBmatcher::Fuse(pKF, vpMapPoints, th) (ORBmatcher.cc:825-972) through a real KeyFrame at the origin: the feature each map
    point is attached to / merged at equals the oracle's best-in-window search (chi-square gate, mono and stereo branches)."""
    w, h, n, seq, sf, K, F = pair
    (kl, dl), (kc, dc) = K
    par = oracle.OracleExtractor(n, 1.2, 8, 20, 7).params()
    rng = np.random.default_rng(int(th) + stereo)
    X, Y, level = _world(rng, kl, w, h)
    nq = len(kl)
    bad = (rng.random(nq) < 0.05).astype(np.uint8); nobs = rng.integers(0, 4, nq).astype(np.int32)
    kf, fx = F[1], np.float32(1.0)
    if stereo:
        fx = np.float32(64.0)
        kf = ref.RefFrame(seq[1], np.roll(seq[1], -9, axis=1), nfeatures=n, fx=64.0, fy=64.0, cx=0.0, cy=0.0, bf=40.0)
        assert int((kf.u_right >= 0).sum()) > 100
    state = rng.choice([0, 0, 1], len(kc)).astype(np.uint8)                       # features that already carry a (good) map point
    n_r, b_r = ref.fuse(kf, state, X / fx, Y / fx, np.ones(nq, np.float32), level, nobs, bad, dl, th=th)
    keep = np.nonzero((bad == 0) & (X >= 0) & (X < w) & (Y >= 0) & (Y < h))[0]
    q = np.zeros(len(keep), oracle.BEST_QUERY_DTYPE)
    q["x"], q["y"], q["radius"], q["ur"], q["level"] = X[keep], Y[keep], (np.float32(th) * sf[level[keep]]).astype(np.float32), X[keep] - np.float32(40.0), level[keep]
    bi, bd = oracle.search_best_in_window(kc, dc, w, h, par["inv_sigma2"], q, dl[keep], True, u_right=kf.u_right)
    b_o = np.full(nq, -1, np.int32)
    ok = bd <= 50
    b_o[keep[ok]] = bi[ok]
    assert n_r == int(ok.sum()) and np.array_equal(b_r, b_o) and n_r > 50
    if stereo:
        kf.close()


@pytest.mark.parametrize("th", [4.0, 8.0])
def test_fuse_sim3(ref, oracle, pair, th):
    """ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (ORBmatcher.cc:974-1100, loop closing) with Scw = identity: the key point each
    candidate point is attached to (or whose map point it should replace) = the oracle's best-in-window search without the chi-square gate."""
    w, h, n, seq, sf, K, F = pair
    (kl, dl), (kc, dc) = K
    par = oracle.OracleExtractor(n, 1.2, 8, 20, 7).params()
    rng = np.random.default_rng(int(th) + 40)
    X, Y, level = _world(rng, kl, w, h)
    nq = len(kl)
    bad = (rng.random(nq) < 0.05).astype(np.uint8)
    state = rng.choice([0, 0, 1], len(kc)).astype(np.uint8)
    n_r, b_r = ref.fuse_sim3(F[1], state, X, Y, np.ones(nq, np.float32), level, bad, dl, th=th)
    keep = np.nonzero((bad == 0) & (X >= 0) & (X < w) & (Y >= 0) & (Y < h))[0]
    q = np.zeros(len(keep), oracle.BEST_QUERY_DTYPE)
    q["x"], q["y"], q["radius"], q["level"] = X[keep], Y[keep], (np.float32(th) * sf[level[keep]]).astype(np.float32), level[keep]
    bi, bd = oracle.search_best_in_window(kc, dc, w, h, par["inv_sigma2"], q, dl[keep], False)
    b_o = np.full(nq, -1, np.int32)
    ok = bd <= 50
    b_o[keep[ok]] = bi[ok]
    assert n_r == int(ok.sum()) and np.array_equal(b_r, b_o) and n_r > 50


def test_search_by_projection_keyframe_sim3(ref, oracle, pair):
    """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:290-403, loop closing) = the flat mode-1 search
    with levels [L-1, L], TH_LOW, no orientation check, vpMatched as the blocked set."""
    w, h, n, seq, sf, K, F = pair
    (kl, dl), (kc, dc) = K
    rng = np.random.default_rng(4)
    X, Y, level = _world(rng, kl, w, h)
    nq = len(kl)
    bad = (rng.random(nq) < 0.05).astype(np.uint8)
    ms = (rng.random(len(kc)) < 0.2).astype(np.uint8)
    n_r, fq_r = ref.search_by_projection_kf(F[1], ms, X, Y, np.ones(nq, np.float32), level, bad, dl, th=10)
    keep = np.nonzero((bad == 0) & (X >= 0) & (X < w) & (Y >= 0) & (Y < h))[0]
    q = np.zeros(len(keep), oracle.PROJ_QUERY_DTYPE)
    q["x"], q["y"], q["radius"] = X[keep], Y[keep], (np.float32(10) * sf[level[keep]]).astype(np.float32)
    q["min_level"], q["max_level"], q["blocks"] = level[keep] - 1, level[keep], 1
    n_o, fq_o = oracle.search_by_projection(kc, dc, w, h, q, dl[keep], 1, nnratio=0.75, th_high=50, check_ori=False, blocked=ms)
    assert n_r == n_o and np.array_equal(fq_r, np.where(fq_o >= 0, keep[np.maximum(fq_o, 0)], -1)) and n_o > 100


@pytest.mark.parametrize("th,orb_dist,ori", [(10.0, 100, True), (3.0, 64, True), (10.0, 100, False)])
def test_search_by_projection_relocalisation(ref, oracle, pair, th, orb_dist, ori):
    """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1472-1599) = the flat mode-1
    search with levels [L-1, L+1], th_high = ORBdist, every feature that already has a map point blocked."""
    w, h, n, seq, sf, K, F = pair
    (kl, dl), (kc, dc) = K
    rng = np.random.default_rng(int(th) + orb_dist)
    X, Y, level = _world(rng, kl, w, h)
    nq = len(kl)
    bad = (rng.random(nq) < 0.05).astype(np.uint8)
    has = (rng.random(nq) < 0.8).astype(np.uint8); found = (rng.random(nq) < 0.1).astype(np.uint8)
    cs = _cur_state(rng, len(kc))
    n_r, fq_r = ref.search_by_projection_reloc(F[1], F[0], has, X, Y, np.ones(nq, np.float32), level, bad, found, dl, cs, th=th, orb_dist=orb_dist, nnratio=0.9, check_ori=ori)
    keep = np.nonzero((has == 1) & (bad == 0) & (found == 0) & ~((X < 0) | (X > w) | (Y < 0) | (Y > h)))[0]
    q = np.zeros(len(keep), oracle.PROJ_QUERY_DTYPE)
    q["x"], q["y"], q["radius"] = X[keep], Y[keep], (np.float32(th) * sf[level[keep]]).astype(np.float32)
    q["min_level"], q["max_level"], q["blocks"], q["angle"] = level[keep] - 1, level[keep] + 1, 1, kl["angle"][keep]
    n_o, fq_o = oracle.search_by_projection(kc, dc, w, h, q, dl[keep], 1, nnratio=0.9, th_high=orb_dist, check_ori=ori, blocked=(cs != 0).astype(np.uint8))
    assert n_r == n_o and np.array_equal(fq_r, np.where(fq_o >= 0, keep[np.maximum(fq_o, 0)], -1)) and n_o > 100


def test_search_by_sim3(ref, oracle, pair):
    """ORBmatcher::SearchBySim3 (ORBmatcher.cc:1102-1326) with the identity similarity: two best-in-window passes + the mutual check."""
    w, h, n, seq, sf, K, F = pair
    (kl, dl), (kc, dc) = K
    par = oracle.OracleExtractor(n, 1.2, 8, 20, 7).params()
    rng = np.random.default_rng(6)
    X1, Y1, lev1 = _world(rng, kl, w, h)
    X2 = (kc["x"] + 3.0 + rng.normal(0, 1.2, len(kc))).astype(np.float32); Y2 = (kc["y"] + 1.0 + rng.normal(0, 1.2, len(kc))).astype(np.float32)
    lev2 = np.clip(kc["octave"] + rng.integers(0, 2, len(kc)), 0, 7).astype(np.int32)
    has1 = (rng.random(len(kl)) < 0.8).astype(np.uint8); has2 = (rng.random(len(kc)) < 0.8).astype(np.uint8)
    n_r, m_r = ref.search_by_sim3(F[0], has1, X1, Y1, np.ones(len(kl), np.float32), lev1, dl, F[1], has2, X2, Y2, np.ones(len(kc), np.float32), lev2, dc, th=7.5)

    def one_pass(hasA, XA, YA, levA, dA, kB, dB):
        keep = np.nonzero((hasA == 1) & (XA >= 0) & (XA < w) & (YA >= 0) & (YA < h))[0]
        q = np.zeros(len(keep), oracle.BEST_QUERY_DTYPE)
        q["x"], q["y"], q["radius"], q["level"] = XA[keep], YA[keep], (np.float32(7.5) * sf[levA[keep]]).astype(np.float32), levA[keep]
        bi, bd = oracle.search_best_in_window(kB, dB, w, h, par["inv_sigma2"], q, dA[keep], False)
        out = np.full(len(hasA), -1, np.int32)
        ok = (bd <= 100) & (bi >= 0)
        out[keep[ok]] = bi[ok]
        return out

    v1, v2 = one_pass(has1, X1, Y1, lev1, dl, kc, dc), one_pass(has2, X2, Y2, lev2, dc, kl, dl)
    m_o = np.full(len(kl), -1, np.int32)
    for i1 in range(len(kl)):
        if v1[i1] >= 0 and v2[v1[i1]] == i1:
            m_o[i1] = v1[i1]
    assert n_r == int((m_o >= 0).sum()) and np.array_equal(m_r, m_o) and n_r > 100


def test_distorted_camera_frames(ref, oracle, pair_oracle, emu_lib):
    """Frame::Frame (mono) with TUM1's distortion: the reference's own UndistortKeyPoints / ComputeImageBounds / AssignFeaturesToGrid
    / SearchForInitialization (cv::undistortPoints = the oracle's restatement) against the oracle's flat restatement with bounds —
    and the HIP sources (emulation build) against the reference directly."""
    w, h, n, seq, sf, K = pair_oracle
    cam = (517.306408 * w / 640, 516.469215 * h / 480, 318.643040 * w / 640, 255.313989 * h / 480, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314)
    for dist in (cam[4:], cam[4:8]):                                          # 5 and 4 coefficient forms (Tracking.cc:70-82)
        c = cam[:4] + tuple(dist)
        ref.RefFrame._geometry = None
        F = [ref.RefFrame(im, nfeatures=n, fx=c[0], fy=c[1], cx=c[2], cy=c[3], dist=dist) for im in seq]
        bounds = ref.RefFrame.bounds()
        assert bounds.tobytes() == oracle.image_bounds(c, w, h).tobytes()
        U = [oracle.undistort_keypoints(c, k) for k, _ in K]
        for f in range(2):
            assert F[f].keys.tobytes() == K[f][0].tobytes() and F[f].keys_un.tobytes() == U[f].tobytes()
        rng = np.random.default_rng(2)
        with oracle.image_bounds_set(bounds):
            for _ in range(300):
                x, y = np.float32(rng.uniform(-40, w + 40)), np.float32(rng.uniform(-40, h + 40))
                r, mn, mx = np.float32(rng.uniform(1, 150)), int(rng.integers(-1, 4)), int(rng.integers(-1, 8))
                assert np.array_equal(F[1].features_in_area(x, y, r, mn, mx), oracle.features_in_area(U[1], w, h, x, y, r, mn, mx))
            n_o, m_o, p_o = oracle.search_for_initialization(U[0], K[0][1], U[1], K[1][1], w, h, window=80, nnratio=0.9)
        n_r, m_r, p_r = ref.search_for_initialization(F[0], F[1], window=80, nnratio=0.9, check_ori=True)
        assert n_r == n_o and np.array_equal(m_r, m_o) and p_r.tobytes() == p_o.tobytes() and n_o > 40
        m = orb_slam2_amd.ORBmatcher(0.9, True, library=emu_lib)
        n_g, m_g, p_g = m.SearchForInitialization(U[0], K[0][1], U[1], K[1][1], w, h, windowSize=80, bounds=bounds)
        assert n_g == n_r and np.array_equal(m_g, m_r) and p_g.tobytes() == p_r.tobytes()
        assert orb_slam2_amd.image_bounds(c, w, h, library=emu_lib).tobytes() == bounds.tobytes()
        un_g = orb_slam2_amd.undistort_points(c, np.stack([K[0][0]["x"], K[0][0]["y"]], axis=1), library=emu_lib)
        assert un_g[:, 0].tobytes() == F[0].keys_un["x"].tobytes() and un_g[:, 1].tobytes() == F[0].keys_un["y"].tobytes()
        for f in F:
            f.close()
    ref.RefFrame._geometry = None


@pytest.mark.parametrize("dist", [None, (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)])
def test_rgbd_frame(ref, oracle, pair_oracle, emu_lib, dist):
    """Frame::Frame(imGray, imDepth, ...) (Frame.cc:117-172): the reference's ComputeStereoFromRGBD on a CV_32F depth map against the
    oracle's restatement and the HIP sources (emulation build)."""
    w, h, n, seq, sf, K = pair_oracle
    rng = np.random.default_rng(8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    depth = (np.float32(1.5) + np.float32(0.8) * np.sin(xx / np.float32(53.0)) * np.cos(yy / np.float32(29.0))).astype(np.float32)
    depth[rng.random((h, w)) < 0.2] = 0.0
    depth[rng.random((h, w)) < 0.02] = -1.0
    cam = (517.306408 * w / 640, 516.469215 * h / 480, 318.643040 * w / 640, 255.313989 * h / 480)
    ref.RefFrame._geometry = None
    F = ref.RefFrame(seq[0], nfeatures=n, fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3], bf=40.0, dist=dist, depth=depth)
    un = K[0][0] if dist is None else oracle.undistort_keypoints(cam + tuple(dist), K[0][0])
    assert F.keys.tobytes() == K[0][0].tobytes() and F.keys_un.tobytes() == un.tobytes()
    u_o, z_o = oracle.stereo_from_rgbd(K[0][0], un, depth, 1.0, 40.0)
    assert F.u_right.tobytes() == u_o.tobytes() and F.depth.tobytes() == z_o.tobytes() and (z_o > 0).sum() > 300
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, library=emu_lib)
    if dist is not None:
        ex.set_camera(cam + tuple(dist))
    ex.extract_batch([seq[0]])
    u_g, z_g = ex.ComputeStereoFromRGBD([depth], 1.0, 40.0)
    assert u_g[0, :F.N].tobytes() == F.u_right.tobytes() and z_g[0, :F.N].tobytes() == F.depth.tobytes()
    ex.close(); F.close()
    ref.RefFrame._geometry = None


def test_product_equals_reference_matcher_and_stereo(ref, pair, emu_lib):
    """HIP kernel sources (emulation build) against the reference's code directly: frame-to-frame matching and stereo matching."""
    w, h, n, seq, sf, K, F = pair
    m = orb_slam2_amd.ORBmatcher(0.9, True, library=emu_lib)
    n_g, m_g, p_g = m.SearchForInitialization(K[0][0], K[0][1], K[1][0], K[1][1], w, h, windowSize=100)
    n_r, m_r, p_r = ref.search_for_initialization(F[0], F[1], window=100, nnratio=0.9, check_ori=True)
    assert n_g == n_r and np.array_equal(m_g, m_r) and p_g.tobytes() == p_r.tobytes()
    L, R = stereo_pair(400, 300, 8, 7)
    Fs = ref.RefFrame(L, R, nfeatures=500, fx=float(FX), fy=float(FX), cx=607.1928, cy=185.2157, bf=float(BF))
    xl = orb_slam2_amd.ORBextractor(500, 1.2, 8, 20, 7, 400, 300, library=emu_lib)
    xr = orb_slam2_amd.ORBextractor(500, 1.2, 8, 20, 7, 400, 300, library=emu_lib)
    xl.extract_batch([L]); xr.extract_batch([R])
    u, d = xl.ComputeStereoMatches(xr, float(BF), float(BF / FX), nimg=1)
    assert u[0, :Fs.N].tobytes() == Fs.u_right.tobytes() and d[0, :Fs.N].tobytes() == Fs.depth.tobytes()
    xl.close(); xr.close(); Fs.close()


def test_keyframe_side_searches_with_a_distorted_camera(ref, oracle, pair_oracle, emu_lib):
    """a KeyFrame keeps the image bounds as ints (truncated from the Frame's floats, KeyFrame.h:185-188) but the Frame's grid, so
    with a distorted camera KeyFrame::GetFeaturesInArea lays a window's cell range out from truncated minima.  That cannot change which key
    points a window returns (cells are assigned by rounding: half a cell of margin against < 1 px of truncation) — checked here: Fuse (both
    overloads) and the loop-closing SearchByProjection through real KeyFrame objects of a TUM1-distorted frame equal the oracle and the HIP
    sources (emulation build) run with the Frame's float bounds."""
    w, h, n, seq, sf, K = pair_oracle
    cam = (np.float32(517.306408 * w / 640), np.float32(516.469215 * h / 480), np.float32(318.643040 * w / 640), np.float32(255.313989 * h / 480))
    dist = (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)
    ref.RefFrame._geometry = None
    F = ref.RefFrame(seq[1], nfeatures=n, fx=float(cam[0]), fy=float(cam[1]), cx=float(cam[2]), cy=float(cam[3]), dist=dist)
    b = ref.RefFrame.bounds()
    assert b[0] != np.floor(b[0]) and b[1] != np.floor(b[1])                      # truncation changes the minima
    kun, dc = F.keys_un, F.desc
    par = oracle.OracleExtractor(n, 1.2, 8, 20, 7).params()
    rng = np.random.default_rng(77)
    nq = 3000
    src = rng.integers(0, len(kun), nq)
    ut = (kun["x"][src] + rng.normal(0, 6, nq)).astype(np.float32); vt = (kun["y"][src] + rng.normal(0, 6, nq)).astype(np.float32)
    X = ((ut - cam[2]) / cam[0]).astype(np.float32); Y = ((vt - cam[3]) / cam[1]).astype(np.float32)
    u = (cam[0] * X + cam[2]).astype(np.float32); v = (cam[1] * Y + cam[3]).astype(np.float32)         # what the reference's projection computes (z = 1)
    level = kun["octave"][src].astype(np.int32)
    qd = dc[src].copy()
    for i in range(nq):
        for bit in rng.integers(0, 256, int(rng.integers(0, 30))):
            qd[i, bit >> 3] ^= 1 << (bit & 7)
    bad = np.zeros(nq, np.uint8)
    ib = b.astype(np.int32)                                                        # KeyFrame::IsInImage compares with the int bounds (KeyFrame.cc:610-613)
    keep = np.nonzero((u >= ib[0]) & (u < ib[2]) & (v >= ib[1]) & (v < ib[3]))[0]
    kb = tuple(b)
    for th in (3.0, 6.0):
        q = np.zeros(len(keep), oracle.BEST_QUERY_DTYPE)
        q["x"], q["y"], q["radius"], q["ur"], q["level"] = u[keep], v[keep], (np.float32(th) * sf[level[keep]]).astype(np.float32), u[keep] - np.float32(40.0), level[keep]
        # Fuse (pose overload, chi-square gate)
        state = rng.choice([0, 0, 1], len(kun)).astype(np.uint8)
        n_r, b_r = ref.fuse(F, state, X, Y, np.ones(nq, np.float32), level, np.ones(nq, np.int32), bad, qd, th=th)
        with oracle.image_bounds_set(b):
            bi, bd = oracle.search_best_in_window(kun, dc, w, h, par["inv_sigma2"], q, qd[keep], True, u_right=F.u_right)
        b_o = np.full(nq, -1, np.int32); ok = bd <= 50; b_o[keep[ok]] = bi[ok]
        assert n_r == int(ok.sum()) and np.array_equal(b_r, b_o) and n_r > 200
        bi_g, bd_g = orb_slam2_amd.search_best_in_window(kun, dc, w, h, par["inv_sigma2"], q, qd[keep], True, u_right=F.u_right, bounds=kb, library=emu_lib)
        assert np.array_equal(bi_g, bi) and np.array_equal(bd_g, bd)
        # Fuse (Sim3 overload, no gate)
        n_r, b_r = ref.fuse_sim3(F, state, X, Y, np.ones(nq, np.float32), level, bad, qd, th=th)
        with oracle.image_bounds_set(b):
            bi, bd = oracle.search_best_in_window(kun, dc, w, h, par["inv_sigma2"], q, qd[keep], False)
        b_o = np.full(nq, -1, np.int32); ok = bd <= 50; b_o[keep[ok]] = bi[ok]
        assert n_r == int(ok.sum()) and np.array_equal(b_r, b_o)
        bi_g, bd_g = orb_slam2_amd.search_best_in_window(kun, dc, w, h, par["inv_sigma2"], q, qd[keep], False, bounds=kb, library=emu_lib)
        assert np.array_equal(bi_g, bi) and np.array_equal(bd_g, bd)
    # loop-closing SearchByProjection(pKF, Scw, ...): the cell-walking kernel
    ms = (rng.random(len(kun)) < 0.2).astype(np.uint8)
    n_r, fq_r = ref.search_by_projection_kf(F, ms, X, Y, np.ones(nq, np.float32), level, bad, qd, th=10)
    pq = np.zeros(len(keep), oracle.PROJ_QUERY_DTYPE)
    pq["x"], pq["y"], pq["radius"] = u[keep], v[keep], (np.float32(10) * sf[level[keep]]).astype(np.float32)
    pq["min_level"], pq["max_level"], pq["blocks"] = level[keep] - 1, level[keep], 1
    with oracle.image_bounds_set(b):
        n_o, fq_o = oracle.search_by_projection(kun, dc, w, h, pq, qd[keep], 1, nnratio=0.75, th_high=50, check_ori=False, blocked=ms)
    assert n_r == n_o and np.array_equal(fq_r, np.where(fq_o >= 0, keep[np.maximum(fq_o, 0)], -1)) and n_o > 100
    n_g, fq_g = orb_slam2_amd.search_by_projection(kun, dc, w, h, pq, qd[keep], 1, nnratio=0.75, th_high=50, check_ori=False, blocked=ms, bounds=kb, library=emu_lib)
    assert n_g == n_o and np.array_equal(fq_g, fq_o)
    F.close()
    ref.RefFrame._geometry = None
