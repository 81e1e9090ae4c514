"""Search loops of the projection-guided matchers on flat data: ORBmatcher::SearchByProjection(Frame&,
vector<MapPoint*>&, th) (ORBmatcher.cc:45-129, mode 0) and SearchByProjection(Current, Last, th, bMono) (:1328-1470, mode 1).
The queries play the map points the caller would have projected; assignments must equal the oracle's restatement."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import orb_slam2_amd  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402


@pytest.fixture(scope="module")
def scene(oracle):
    w, h, n = 480, 360, 700
    seq = synth.sequence(w, h, 2, seed=41)
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    last, cur = ora.extract(seq[0]), ora.extract(seq[1])
    return w, h, ora.params()["scale_factors"], last, cur


def _queries(oracle, scene, mode, th, rng, level_rule):
    w, h, sf, (kl, dl), (kc, dc) = scene
    q = np.zeros(len(kl), oracle.PROJ_QUERY_DTYPE)
    # last-frame keypoints "projected" with the true image motion (3, 1) px plus a little pose error
    q["x"] = kl["x"] - 3.0 + rng.normal(0, 1.0, len(kl)).astype(np.float32)
    q["y"] = kl["y"] - 1.0 + rng.normal(0, 1.0, len(kl)).astype(np.float32)
    oct_ = kl["octave"]
    q["radius"] = (np.float32(th) * sf[oct_]).astype(np.float32)
    q["ur"] = q["x"] - rng.uniform(2, 40, len(kl)).astype(np.float32)
    if level_rule == "local_map":                       # GetFeaturesInArea(x, y, r, level-1, level)      ORBmatcher.cc:69
        q["min_level"], q["max_level"] = oct_ - 1, oct_
    elif level_rule == "forward":                       # GetFeaturesInArea(u, v, radius, octave)          ORBmatcher.cc:1384
        q["min_level"], q["max_level"] = oct_, -1
    elif level_rule == "backward":                      # GetFeaturesInArea(u, v, radius, 0, octave)       ORBmatcher.cc:1386
        q["min_level"], q["max_level"] = 0, oct_
    else:                                               # GetFeaturesInArea(u, v, radius, octave-1, octave+1)
        q["min_level"], q["max_level"] = oct_ - 1, oct_ + 1
    q["blocks"] = rng.random(len(kl)) < 0.9            # most map points have observations
    q["angle"] = kl["angle"]
    keep = rng.random(len(kl)) < 0.85                  # some map points failed the caller's frustum test
    return q[keep], dl[keep]


@pytest.mark.parametrize("mode,rule,th,ratio,stereo", [(0, "local_map", 3.0, 0.8, False), (0, "local_map", 5.0, 0.8, True),
                                                     (1, "window", 7.0, 0.9, False), (1, "forward", 15.0, 0.9, True),
                                                     (1, "backward", 15.0, 0.9, False)])
def test_search_by_projection(backend, oracle, scene, select_tables, mode, rule, th, ratio, stereo):
    w, h, sf, (kl, dl), (kc, dc) = scene
    rng = np.random.default_rng(mode * 100 + int(th))
    q, qd = _queries(oracle, scene, mode, th, rng, rule)
    u_right = None
    if stereo:
        u_right = np.where(rng.random(len(kc)) < 0.6, kc["x"] - rng.uniform(1, 45, len(kc)), -1).astype(np.float32)
    blocked = (rng.random(len(kc)) < 0.15).astype(np.uint8)
    for check_ori in (True, False):
        n_o, f_o = oracle.search_by_projection(kc, dc, w, h, q, qd, mode, nnratio=ratio, th_high=100, check_ori=check_ori, u_right=u_right, blocked=blocked)
        n_g, f_g = orb_slam2_amd.search_by_projection(kc, dc, w, h, q, qd, mode, nnratio=ratio, th_high=100, check_ori=check_ori, u_right=u_right,
                                                      blocked=blocked, library=backend)
        assert n_g == n_o and np.array_equal(f_g, f_o)
        assert n_o > 50


def test_search_by_projection_edge_cases(backend, oracle, scene):
    w, h, sf, (kl, dl), (kc, dc) = scene
    rng = np.random.default_rng(3)
    q, qd = _queries(oracle, scene, 0, 4.0, rng, "local_map")
    # no queries / no features
    n_g, f_g = orb_slam2_amd.search_by_projection(kc, dc, w, h, q[:0], qd[:0], 0, library=backend)
    assert n_g == 0 and np.all(f_g == -1)
    n_g, f_g = orb_slam2_amd.search_by_projection(kc[:0], dc[:0], w, h, q, qd, 0, library=backend)
    assert n_g == 0 and len(f_g) == 0
    # every map point projects onto the same spot with non-blocking map points: the last query wins each overwritten feature,
    # and the reference still counts every assignment
    q2 = q[:60].copy()
    q2["x"], q2["y"], q2["radius"], q2["min_level"], q2["max_level"], q2["blocks"] = kc["x"][5], kc["y"][5], 40.0, 0, -1, 0
    for mode in (0, 1):
        n_o, f_o = oracle.search_by_projection(kc, dc, w, h, q2, qd[:60], mode, nnratio=0.8, check_ori=True)
        n_g, f_g = orb_slam2_amd.search_by_projection(kc, dc, w, h, q2, qd[:60], mode, nnratio=0.8, check_ori=True, library=backend)
        assert n_g == n_o and np.array_equal(f_g, f_o)
    # more than one staging round of queries (> 1024) and windows covering the whole image (lists beyond the LDS stage)
    nbig = 1060 if backend.endswith("_emu.so") else 1500        # the emulation runs this ~100x slower than the GPU
    big = np.concatenate([q] * 3)[:nbig].copy()
    big["radius"][:(120 if backend.endswith("_emu.so") else nbig)] = 600.0
    big["min_level"], big["max_level"] = 0, -1
    bigd = np.concatenate([qd] * 3)[:nbig]
    n_o, f_o = oracle.search_by_projection(kc, dc, w, h, big, bigd, 0, nnratio=0.8)
    n_g, f_g = orb_slam2_amd.search_by_projection(kc, dc, w, h, big, bigd, 0, nnratio=0.8, library=backend)
    assert n_g == n_o and np.array_equal(f_g, f_o)


@pytest.mark.parametrize("chi2,stereo,th", [(True, False, 3.0), (True, True, 3.0), (False, False, 7.5), (False, True, 12.0)])
def test_search_best_in_window(backend, oracle, scene, chi2, stereo, th):
    """Candidate loop of ORBmatcher::Fuse (chi-square gate, stereo and mono branches) and of SearchBySim3's passes (no gate)."""
    w, h, sf, (kl, dl), (kc, dc) = scene
    inv = (1.0 / (sf * sf)).astype(np.float32)
    rng = np.random.default_rng(int(th * 10) + chi2)
    nq = len(kl)
    q = np.zeros(nq, oracle.BEST_QUERY_DTYPE)
    q["x"] = kl["x"] - 3.0 + rng.normal(0, 1.2, nq).astype(np.float32)
    q["y"] = kl["y"] - 1.0 + rng.normal(0, 1.2, nq).astype(np.float32)
    q["level"] = np.clip(kl["octave"] + rng.integers(0, 2, nq), 0, 7)
    q["radius"] = (np.float32(th) * sf[q["level"]]).astype(np.float32)
    q["ur"] = q["x"] - np.float32(9.0)
    q["x"][:3] = -50.0; q["y"][3:5] = h + 80.0                                   # windows that miss the image: no candidate
    u_right = np.where(rng.random(len(kc)) < 0.6, kc["x"] - rng.uniform(8, 10, len(kc)), -1).astype(np.float32) if stereo else None
    bi_o, bd_o = oracle.search_best_in_window(kc, dc, w, h, inv, q, dl, chi2, u_right=u_right)
    bi_g, bd_g = orb_slam2_amd.search_best_in_window(kc, dc, w, h, inv, q, dl, chi2, u_right=u_right, library=backend)
    assert np.array_equal(bi_g, bi_o) and np.array_equal(bd_g, bd_o)
    assert int((bd_o <= 50).sum()) > 100 and np.all(bi_o[:5] == -1)
    e = orb_slam2_amd.search_best_in_window(kc, dc, w, h, inv, q[:0], dl[:0], chi2, library=backend)
    assert len(e[0]) == 0


def test_feature_grid_of_a_frame_too_large_for_lds(backend, oracle, scene, monkeypatch):
    """AssignFeaturesToGrid for a frame whose bucket table does not fit the LDS form of k_match_grid (12 KB + 6 bytes per key point, ~23 000 key
    points): the same table built in place in global memory.  On the GPU a real 30 000-key-point frame; on the CPU emulation the small scene, pushed
    through the large-frame form by the test hook (and once more through the LDS form: same answers)."""
    w, h, sf, (kl, dl), (kc, dc) = scene
    inv = (1.0 / (sf * sf)).astype(np.float32)
    rng = np.random.default_rng(77)
    if backend.endswith("_emu.so"):
        monkeypatch.setenv("ORBHIP_TEST_MATCH_GRID_LDS_MAX", "1024")
        K, D = kc, dc
    else:
        reps = 30000 // len(kc) + 1                                # the scene's key points, repeated with jitter: dense cells, ties in the cell order
        K = np.concatenate([kc] * reps)[:30000].copy(); D = np.concatenate([dc] * reps)[:30000].copy()
        K["x"] = np.clip(K["x"] + rng.normal(0, 6, len(K)), 0, w - 1).astype(np.float32); K["y"] = np.clip(K["y"] + rng.normal(0, 6, len(K)), 0, h - 1).astype(np.float32)
        D[rng.random(len(D)) < 0.5, 5] ^= 0x10
    nq = len(kl)
    q = np.zeros(nq, oracle.BEST_QUERY_DTYPE)
    q["x"] = kl["x"] - 3.0 + rng.normal(0, 1.2, nq).astype(np.float32); q["y"] = kl["y"] - 1.0 + rng.normal(0, 1.2, nq).astype(np.float32)
    q["level"] = np.clip(kl["octave"] + rng.integers(0, 2, nq), 0, 7); q["radius"] = (np.float32(7.5) * sf[q["level"]]).astype(np.float32); q["ur"] = q["x"] - np.float32(9.0)
    bi_o, bd_o = oracle.search_best_in_window(K, D, w, h, inv, q, dl, False)
    bi_g, bd_g = orb_slam2_amd.search_best_in_window(K, D, w, h, inv, q, dl, False, library=backend)
    assert np.array_equal(bi_g, bi_o) and np.array_equal(bd_g, bd_o) and int((bd_o <= 50).sum()) > 100
    if backend.endswith("_emu.so"):
        monkeypatch.delenv("ORBHIP_TEST_MATCH_GRID_LDS_MAX")
        bi_l, bd_l = orb_slam2_amd.search_best_in_window(K, D, w, h, inv, q, dl, False, library=backend)
        assert np.array_equal(bi_l, bi_o) and np.array_equal(bd_l, bd_o)
        # the other users of the table: the order-dependent projection search
        monkeypatch.setenv("ORBHIP_TEST_MATCH_GRID_LDS_MAX", "1024")
        pq, pqd = _queries(oracle, scene, 1, 7.0, rng, "window")
        n_o, f_o = oracle.search_by_projection(kc, dc, w, h, pq, pqd, 1, nnratio=0.9, th_high=100, check_ori=True)
        n_g, f_g = orb_slam2_amd.search_by_projection(kc, dc, w, h, pq, pqd, 1, nnratio=0.9, th_high=100, check_ori=True, library=backend)
        assert n_g == n_o and np.array_equal(f_g, f_o)


def test_searches_on_device_resident_frames(backend, oracle, select_tables):
    """orbhip_search_by_projection_frame / orbhip_search_best_in_window_frame: the frame's key points, descriptors and mvuRight stay on
    the device (stereo pair extracted and matched there); results equal the oracle's on the fetched copies."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_parity_stereo import stereo_pair
    w, h, n = 400, 300, 500
    L, R = stereo_pair(w, h, 8, 7)
    mbf, fx = np.float32(386.1448), np.float32(718.856)
    xl = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, max_batch=2, library=backend)
    xr = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, max_batch=2, library=backend)
    imgs_l = [L, synth.frame(w, h, seed=3)]
    ks, ds = xl.extract_batch(imgs_l); xr.extract_batch([R, R])
    with pytest.raises(orb_slam2_amd.OrbHipError):
        xl.search_by_projection(0, len(ks[0]), np.zeros(1, oracle.PROJ_QUERY_DTYPE), np.zeros((1, 32), np.uint8), 0, use_u_right=True)     # no mvuRight yet
    u, d = xl.ComputeStereoMatches(xr, float(mbf), float(mbf / fx), nimg=2)
    sf = xl.GetScaleFactors()
    inv = (1.0 / (sf * sf)).astype(np.float32)
    rng = np.random.default_rng(9)
    for f in (1, 0) if not backend.endswith("_emu.so") else (0,):        # (the emulation runs one frame)
        kc, dc = ks[f], ds[f]
        nk = len(kc)
        q = np.zeros(nk, oracle.PROJ_QUERY_DTYPE)
        q["x"] = kc["x"] + rng.normal(0, 1.5, nk).astype(np.float32); q["y"] = kc["y"] + rng.normal(0, 1.5, nk).astype(np.float32)
        q["radius"] = (np.float32(5.0) * sf[kc["octave"]]).astype(np.float32)
        q["ur"] = np.where(u[f, :nk] > 0, u[f, :nk] + rng.normal(0, 2.0, nk), q["x"] - 20).astype(np.float32)
        q["min_level"], q["max_level"] = kc["octave"] - 1, kc["octave"] + 1
        q["blocks"] = rng.random(nk) < 0.8
        q["angle"] = kc["angle"]
        qd = dc.copy()
        for i in range(nk):
            for b in rng.integers(0, 256, int(rng.integers(0, 25))):
                qd[i, b >> 3] ^= 1 << (b & 7)
        blocked = (rng.random(nk) < 0.1).astype(np.uint8)
        for mode in (0, 1):
            for use_ur in (False, True):
                n_o, f_o = oracle.search_by_projection(kc, dc, w, h, q, qd, mode, nnratio=0.9, th_high=100, check_ori=True, u_right=u[f, :nk] if use_ur else None, blocked=blocked)
                n_g, f_g = xl.search_by_projection(f, nk, q, qd, mode, nnratio=0.9, th_high=100, check_ori=True, use_u_right=use_ur, blocked=blocked)
                assert n_g == n_o and np.array_equal(f_g, f_o) and n_o > 50
        bq = np.zeros(nk, oracle.BEST_QUERY_DTYPE)
        bq["x"], bq["y"], bq["ur"], bq["level"] = q["x"], q["y"], q["ur"], np.clip(kc["octave"] + rng.integers(0, 2, nk), 0, 7)
        bq["radius"] = (np.float32(3.0) * sf[bq["level"]]).astype(np.float32)
        for use_ur in (False, True):
            bi_o, bd_o = oracle.search_best_in_window(kc, dc, w, h, inv, bq, qd, True, u_right=u[f, :nk] if use_ur else None)
            bi_g, bd_g = xl.search_best_in_window(f, nk, bq, qd, True, use_u_right=use_ur)
            assert np.array_equal(bi_g, bi_o) and np.array_equal(bd_g, bd_o)
    xl.extract_batch(imgs_l)                                     # a new extraction invalidates mvuRight
    with pytest.raises(orb_slam2_amd.OrbHipError):
        xl.search_best_in_window(0, len(ks[0]), bq[:1], qd[:1], True, use_u_right=True)
    xl.close(); xr.close()


def test_stereo_columns_handed_to_a_resident_frame(backend, oracle):
    """orbhip_set_stereo_columns: mvuRight the caller computed itself (one RGB-D frame at a time: the reference's own depth-map loop) reaches the frame that is
    still on the device; the resident searches with the right-coordinate test equal the oracle on the same columns; a second set replaces the first, a new
    extraction drops them, a wrong count is refused."""
    w, h, n = 400, 300, 500
    x = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, max_batch=1, library=backend)
    sf = x.GetScaleFactors()
    inv = (1.0 / (sf * sf)).astype(np.float32)
    rng = np.random.default_rng(33)
    for t in range(3):
        kc, dc = x(synth.frame(w, h, seed=40 + t))
        nk = len(kc)
        q = np.zeros(nk, oracle.PROJ_QUERY_DTYPE)
        q["x"] = kc["x"] + rng.normal(0, 1.5, nk).astype(np.float32); q["y"] = kc["y"] + rng.normal(0, 1.5, nk).astype(np.float32)
        q["radius"] = (np.float32(5.0) * sf[kc["octave"]]).astype(np.float32)
        q["min_level"], q["max_level"] = kc["octave"] - 1, kc["octave"] + 1
        q["blocks"] = rng.random(nk) < 0.8
        q["angle"] = kc["angle"]
        qd = dc.copy()
        for i in range(nk):
            for b in rng.integers(0, 256, int(rng.integers(0, 25))):
                qd[i, b >> 3] ^= 1 << (b & 7)
        if t == 0:
            with pytest.raises(orb_slam2_amd.OrbHipError):
                x.search_by_projection(0, nk, q[:1], qd[:1], 0, use_u_right=True)                      # none yet
            with pytest.raises(orb_slam2_amd.OrbHipError):
                x.set_stereo_columns(np.zeros(x.capacity + 1, np.float32))
        for rep in range(2):                                                                              # the second set replaces the first
            depth = np.where(rng.random(nk) < 0.7, rng.uniform(0.5, 8.0, nk), -1.0).astype(np.float32)
            u = np.where(depth > 0, kc["x"] - np.float32(40.0) / depth, -1.0).astype(np.float32)         # Frame.cc:660-662
            x.set_stereo_columns(u)
            q["ur"] = np.where(u > 0, u + rng.normal(0, 2.0, nk), q["x"] - 20).astype(np.float32)
            for mode in (0, 1):
                n_o, f_o = oracle.search_by_projection(kc, dc, w, h, q, qd, mode, nnratio=0.9, th_high=100, check_ori=True, u_right=u)
                n_g, f_g = x.search_by_projection(0, nk, q, qd, mode, nnratio=0.9, th_high=100, check_ori=True, use_u_right=True)
                assert n_g == n_o and np.array_equal(f_g, f_o) and n_o > 50, (t, rep, mode)
            bq = np.zeros(nk, oracle.BEST_QUERY_DTYPE)
            bq["x"], bq["y"], bq["ur"], bq["level"] = q["x"], q["y"], q["ur"], np.clip(kc["octave"] + rng.integers(0, 2, nk), 0, 7)
            bq["radius"] = (np.float32(3.0) * sf[bq["level"]]).astype(np.float32)
            bi_o, bd_o = oracle.search_best_in_window(kc, dc, w, h, inv, bq, qd, True, u_right=u)
            bi_g, bd_g = x.search_best_in_window(0, nk, bq, qd, True, use_u_right=True)
            assert np.array_equal(bi_g, bi_o) and np.array_equal(bd_g, bd_o)
    x(synth.frame(w, h, seed=50))
    with pytest.raises(orb_slam2_amd.OrbHipError):
        x.search_best_in_window(0, 10, bq[:1], qd[:1], True, use_u_right=True)                              # a new extraction drops the columns
    x.close()


def test_stereo_columns_of_one_frame_leave_the_others_without(backend, oracle):
    """A context that holds two frames: columns handed to frame 1 only.  A resident search on frame 0 with the right-coordinate test then sees mvuRight = -1
    everywhere (Frame.cc:468: no right coordinate) - not whatever the freshly allocated block held; frame 1 sees its own columns."""
    w, h, n = 400, 300, 400
    x = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, max_batch=2, library=backend)
    sf = x.GetScaleFactors()
    kb, db = x.extract_batch(np.stack([synth.frame(w, h, seed=61), synth.frame(w, h, seed=62)]))
    rng = np.random.default_rng(5)
    k1 = kb[1]
    u1 = np.where(rng.random(len(k1)) < 0.7, k1["x"] - np.float32(12.0), -1.0).astype(np.float32)
    x.set_stereo_columns(u1, frame=1)
    for fidx, u in ((0, None), (1, u1)):
        kc, dc = kb[fidx], db[fidx]
        nk = len(kc)
        uu = np.full(nk, -1.0, np.float32) if u is None else u
        q = np.zeros(nk, oracle.PROJ_QUERY_DTYPE)
        q["x"], q["y"] = kc["x"] + np.float32(0.5), kc["y"] - np.float32(0.5)
        q["radius"] = (np.float32(5.0) * sf[kc["octave"]]).astype(np.float32)
        q["min_level"], q["max_level"] = kc["octave"] - 1, kc["octave"] + 1
        q["blocks"] = 1; q["angle"] = kc["angle"]
        q["ur"] = np.where(uu > 0, uu + np.float32(1.0), q["x"] - 20).astype(np.float32)
        n_o, f_o = oracle.search_by_projection(kc, dc, w, h, q, dc, 0, nnratio=0.9, th_high=100, check_ori=True, u_right=uu)
        n_g, f_g = x.search_by_projection(fidx, nk, q, dc, 0, nnratio=0.9, th_high=100, check_ori=True, use_u_right=True)
        assert n_g == n_o and np.array_equal(f_g, f_o) and n_o > 50, fidx
    x.close()


def test_frame_epilogues_single_image_sequence(backend, oracle):
    """What ORB_SLAM2 does: one stereo pair at a time on two max_batch = 1 contexts, each pair followed by ComputeStereoMatches and two
    projection searches on the left frame.  From the second pair on the contexts build the right image's row table and the left image's 64x48
    grid behind their own extractions (frame epilogues, orbhip_api.hip); every pair's results must equal the oracle's all the same, also when
    the follow-ups change (a pair without any, a search without the stereo step)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_parity_stereo import stereo_pair
    w, h, n = 400, 300, 500
    mbf, fx = np.float32(386.1448), np.float32(718.856)
    xl = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, max_batch=1, library=backend)
    xr = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, max_batch=1, library=backend)
    eL, eR = oracle.OracleExtractor(n, 1.2, 8, 20, 7), oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    sf = xl.GetScaleFactors()
    rng = np.random.default_rng(21)
    for t, (seed, disp, follow) in enumerate([(8, 7, "all"), (9, 11, "all"), (10, 5, "none"), (11, 9, "search"), (12, 7, "all")]):
        L, R = stereo_pair(w, h, seed, disp)
        kc, dc = xl(L)
        xr(R)
        ko, do = eL.extract(L); eR.extract(R)
        assert kc.tobytes() == ko.tobytes() and np.array_equal(dc, do)
        nk = len(kc)
        if follow == "none":
            continue
        u = None
        if follow == "all":
            u, d = xl.ComputeStereoMatches(xr, float(mbf), float(mbf / fx))
            uo, dpo = oracle.stereo_matches(eL, eR, float(mbf), float(mbf / fx))
            assert u[0, :nk].tobytes() == uo.tobytes() and d[0, :nk].tobytes() == dpo.tobytes() and (uo > 0).sum() > 50
            u = u[0, :nk]
        q = np.zeros(nk, oracle.PROJ_QUERY_DTYPE)
        q["x"] = kc["x"] + rng.normal(0, 1.5, nk).astype(np.float32); q["y"] = kc["y"] + rng.normal(0, 1.5, nk).astype(np.float32)
        q["radius"] = (np.float32(5.0) * sf[kc["octave"]]).astype(np.float32)
        q["ur"] = (q["x"] - 20).astype(np.float32) if u is None else np.where(u > 0, u + rng.normal(0, 2.0, nk), q["x"] - 20).astype(np.float32)
        q["min_level"], q["max_level"] = kc["octave"] - 1, kc["octave"] + 1
        q["blocks"] = rng.random(nk) < 0.8
        q["angle"] = kc["angle"]
        qd = dc.copy()
        for i in range(nk):
            for b in rng.integers(0, 256, int(rng.integers(0, 25))):
                qd[i, b >> 3] ^= 1 << (b & 7)
        for mode in (1, 0):
            n_o, f_o = oracle.search_by_projection(kc, dc, w, h, q, qd, mode, nnratio=0.9, th_high=100, check_ori=True, u_right=u)
            n_g, f_g = xl.search_by_projection(0, nk, q, qd, mode, nnratio=0.9, th_high=100, check_ori=True, use_u_right=u is not None)
            assert n_g == n_o and np.array_equal(f_g, f_o) and n_o > 50, (t, mode)
    xl.close(); xr.close()


@pytest.mark.parametrize("mode,rule,th,ratio", [(0, "local_map", 3.0, 0.8), (1, "window", 7.0, 0.9)])
def test_search_by_projection_batch_of_camera_slots(backend, oracle, scene, select_tables, mode, rule, th, ratio):
    """orbhip_search_by_projection_batch (the multi-camera form of M2 / M3): slots with different frames, query sets, stereo
    gates and blocked sets in ONE pass; every slot must equal the oracle's answer for that slot alone (and thereby the per-slot entry point)."""
    w, h, sf, (kl, dl), (kc, dc) = scene
    rng = np.random.default_rng(mode * 10 + 3)
    slots, want = [], []
    for s in range(5):
        q, qd = _queries(oracle, scene, mode, th + s, rng, rule)
        if s == 2:                                                    # a slot searching the OTHER frame with fewer queries; one empty slot below
            q, qd = q[:200], qd[:200]
            kf, df = kl, dl
        else:
            kf, df = kc, dc
        ur = np.where(rng.random(len(kf)) < 0.6, kf["x"] - rng.uniform(1, 45, len(kf)), -1).astype(np.float32) if s % 2 else None
        bl = (rng.random(len(kf)) < 0.15).astype(np.uint8) if s != 3 else None
        slots.append((kf, df, q, qd, ur, bl))
        want.append(oracle.search_by_projection(kf, df, w, h, q, qd, mode, nnratio=ratio, th_high=100, check_ori=True, u_right=ur, blocked=bl))
    slots.append((kc[:0], dc[:0], slots[0][2][:0], slots[0][3][:0]))                     # no features, no queries
    want.append((0, np.zeros(0, np.int32)))
    got = orb_slam2_amd.search_by_projection_batch(slots, w, h, mode, nnratio=ratio, th_high=100, check_ori=True, library=backend)
    assert len(got) == len(want)
    for s, ((n_g, f_g), (n_o, f_o)) in enumerate(zip(got, want)):
        assert n_g == n_o and np.array_equal(f_g, f_o), f"slot {s}"
    assert sum(n for n, _ in want) > 500


def _fuzz_case(oracle, rng, library):
    """one random projection search: scene size, feature count, window, level rule, conflicts (many queries crowded onto few features), blocked features"""
    w, h, n = int(rng.integers(200, 900)), int(rng.integers(160, 500)), int(rng.integers(150, 1600))
    seq = synth.sequence(w, h, 2, seed=int(rng.integers(1 << 30)))
    ora = oracle.OracleExtractor(n, 1.2, int(rng.integers(3, 9)), 20, 7)
    (kl, dl), (kc, dc) = ora.extract(seq[0]), ora.extract(seq[1])
    if len(kl) < 20 or len(kc) < 20:
        return None
    sf = ora.params()["scale_factors"]
    mode = int(rng.integers(0, 2))
    rule = ["local_map", "window", "forward", "backward"][int(rng.integers(0, 4))]
    th = float(rng.choice([3.0, 7.0, 15.0, 40.0]))                     # 40: windows so wide that chains of queries looking at the same features get long
    scene = (w, h, sf, (kl, dl), (kc, dc))
    q, qd = _queries(oracle, scene, mode, th, rng, rule)
    if rng.random() < 0.5:                                             # crowd: repeat a block of queries (the same map point seen twice, near-duplicates)
        rep = rng.integers(0, len(q), int(len(q) * rng.uniform(0.2, 1.5)))
        q, qd = np.concatenate([q, q[rep]]), np.concatenate([qd, qd[rep]])
        perm = rng.permutation(len(q)); q, qd = q[perm], qd[perm]
    u_right = np.where(rng.random(len(kc)) < 0.6, kc["x"] - rng.uniform(1, 45, len(kc)), -1).astype(np.float32) if rng.random() < 0.4 else None
    blocked = (rng.random(len(kc)) < rng.uniform(0, 0.4)).astype(np.uint8)
    ratio, check_ori = float(rng.choice([0.6, 0.8, 0.9])), bool(rng.integers(0, 2))
    n_o, f_o = oracle.search_by_projection(kc, dc, w, h, q, qd, mode, nnratio=ratio, th_high=100, check_ori=check_ori, u_right=u_right, blocked=blocked)
    n_g, f_g = orb_slam2_amd.search_by_projection(kc, dc, w, h, q, qd, mode, nnratio=ratio, th_high=100, check_ori=check_ori, u_right=u_right,
                                                  blocked=blocked, library=library)
    return f"{w}x{h} n={len(kc)} nq={len(q)} mode={mode} {rule} th={th} ratio={ratio} ori={check_ori}: {n_o} matches", n_g == n_o and np.array_equal(f_g, f_o)


def test_search_by_projection_random_cases(backend, oracle, request, select_tables):
    """random scenes / windows / level rules / crowded query sets through the 256-queries-per-step selection (k_proj_select): 12 cases on the emulation,
    40 on the GPU; `python tests/test_parity_projection.py [ncases] [seed]` runs a longer sweep by hand"""
    rng = np.random.default_rng(77)
    done = 0
    for _ in range(40 if "gpu" in request.node.name else 12 if select_tables == "lds" else 6):
        r = _fuzz_case(oracle, rng, backend)
        if r is None:
            continue
        assert r[1], r[0]
        done += 1
    assert done >= (8 if select_tables == "lds" or "gpu" in request.node.name else 4)


if __name__ == "__main__":
    from oracle import orb_oracle as O
    ncases, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 200), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(ncases):
        r = _fuzz_case(O, rng, None)
        if r is None:
            continue
        print(f"case {c}: {r[0]} {'OK' if r[1] else 'MISMATCH'}")
        bad += not r[1]
    print(f"projection fuzz: {ncases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)
