"""Batched forms of the back end's matcher loops (include/orbhip.h: orbhip_search_by_bow_batch, orbhip_search_for_triangulation_batch,
orbhip_search_best_in_window_batch): one upload / launch set / download for a whole loop of the reference —

    Tracking::Relocalization      for every candidate key frame:  SearchByBoW(pKF, mCurrentFrame, ...)              Tracking.cc:1357-1380
    LoopClosing::ComputeSim3      for every candidate key frame:  SearchByBoW(mpCurrentKF, pKF, ...)                LoopClosing.cc:239-375
    LocalMapping::CreateNewMapPoints   for every neighbour:       SearchForTriangulation(mpCurrentKeyFrame, pKF2)   LocalMapping.cc:237-268
    LocalMapping::SearchInNeighbors    for every target:          Fuse(pKFi, vpMapPointMatches)                     LocalMapping.cc:483-514

Per pair / slot the answers must equal the per-call entry points (which tests/test_bow.py, test_parity_projection.py and the reference-built goldens pin),
bit for bit; the triangulation loop is additionally driven the way LocalMapping drives it — key frame 1 gains map points between neighbours — and must
equal the sequential per-call loop after the documented filter."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import orb_slam2_amd  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

VOC = os.path.join(ROOT, "tests", "golden", "voc_k6_L3_ref.txt")


@pytest.fixture(scope="module")
def frames(oracle):
    w, h, n = 480, 360, 700
    seq = synth.sequence(w, h, 5, seed=23)
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    par = ora.params()
    ov = oracle.OracleVocabulary(VOC)
    out = []
    for im in seq:
        k, d = ora.extract(im)
        out.append(dict(k=k, d=d, fv=[ov.transform(d, lu)[2:] for lu in (1, 2, 3)]))
    return w, h, par, out


def _bow_side(f, lu, valid):
    return dict(desc=f["d"], angle=f["k"]["angle"], valid=valid, fv=f["fv"][lu])


@pytest.mark.parametrize("mode,shared", [(0, "side2"), (1, "side1")])
def test_search_by_bow_batch(backend, frames, mode, shared):
    """Relocalization's loop (the frame shared on side 2) and ComputeSim3's (the current key frame shared on side 1); an empty key frame and a pair without
    common nodes in the middle of the batch."""
    w, h, par, F = frames
    rng = np.random.default_rng(5 + mode)
    cur = _bow_side(F[0], 1, (rng.random(len(F[0]["d"])) < 0.8).astype(np.uint8) if (mode == 1 or shared == "side1") else None)
    if shared == "side1":
        cur["valid"] = (rng.random(len(F[0]["d"])) < 0.8).astype(np.uint8)
    others = [_bow_side(F[i], 1, (rng.random(len(F[i]["d"])) < 0.8).astype(np.uint8)) for i in (1, 2, 3, 4)]
    empty = dict(desc=np.zeros((0, 32), np.uint8), angle=np.zeros(0, np.float32), valid=np.zeros(0, np.uint8), fv=(np.zeros(1, np.uint32), np.zeros(1, np.int32), np.zeros(1, np.uint32)))
    empty["fv"] = (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.uint32))
    other_lu = _bow_side(F[2], 2, others[1]["valid"])                      # FeatureVector of another level: node ids of levels 1 and 2 rarely coincide
    cands = [others[0], others[1], empty, others[2], other_lu, others[3]]
    pairs = [(c, cur) if shared == "side2" else (cur, c) for c in cands]
    for ori in (True, False):
        got = orb_slam2_amd.search_by_bow_batch(mode, pairs, nnratio=0.75, check_ori=ori, library=backend)
        total = 0
        for (a, b), (n_g, m_g) in zip(pairs, got):
            if len(a["desc"]) == 0 or len(b["desc"]) == 0:
                assert n_g == 0 and len(m_g) == len(a["desc"])
                continue
            n_o, m_o = orb_slam2_amd.search_by_bow(mode, a["desc"], a["angle"], a["valid"], a["fv"], b["desc"], b["angle"], b["valid"], b["fv"], nnratio=0.75, check_ori=ori, library=backend)
            assert n_g == n_o and np.array_equal(m_g, m_o)
            total += n_o
        assert total > 100
    assert orb_slam2_amd.search_by_bow_batch(mode, [], library=backend) == []


def _tri_side(f, par, has_mp, stereo, lu=1):
    return dict(desc=f["d"], kps=f["k"], has_mp=has_mp, stereo=stereo, fv=f["fv"][lu], scale_factors=par["scale_factors"], level_sigma2=par["scale_factors"] ** 2)


def test_search_for_triangulation_batch(backend, frames):
    """(1) every pair of the batch equals the per-call entry with the same has_mp; (2) LocalMapping's loop: key frame 1 gains a map point for (a rule standing
    for the triangulation checks) two thirds of each neighbour's matches before the next neighbour is searched — the sequential per-call loop equals the
    batch filtered by "the feature has a map point by now" (no orientation check, as LocalMapping.cc:215 constructs its matcher)."""
    w, h, par, F = frames
    rng = np.random.default_rng(9)
    n1 = len(F[0]["d"])
    has1 = (rng.random(n1) < 0.3).astype(np.uint8); st1 = (rng.random(n1) < 0.5).astype(np.uint8)
    kf1 = _tri_side(F[0], par, has1, st1)
    nbs = []
    for i in (1, 2, 3, 4):
        n2 = len(F[i]["d"])
        Fm = np.array([[0, -1e-3, 1.0 / 300], [1e-3, 0, -3.0 / 300], [-1.0 / 300, 3.0 / 300, 0]], np.float32) + rng.normal(0, 1e-5, (3, 3)).astype(np.float32)
        nbs.append(dict(kf=_tri_side(F[i], par, (rng.random(n2) < 0.3).astype(np.uint8), (rng.random(n2) < 0.5).astype(np.uint8)), F12=Fm, ex=float(600 + 10 * i), ey=float(180 - 5 * i)))

    def single(has_mp1, nb, only_stereo, ori):
        b = nb["kf"]
        return orb_slam2_amd.search_for_triangulation(kf1["desc"], kf1["kps"], has_mp1, kf1["stereo"], kf1["fv"], b["desc"], b["kps"], b["has_mp"], b["stereo"], b["fv"],
                                                      nb["F12"], nb["ex"], nb["ey"], b["scale_factors"], b["level_sigma2"], only_stereo=only_stereo, check_ori=ori, library=backend)
    for only_stereo, ori in ((False, True), (True, False), (False, False)):
        got = orb_slam2_amd.search_for_triangulation_batch(kf1, nbs, only_stereo=only_stereo, check_ori=ori, library=backend)
        tot = 0
        for nb, (n_g, m_g) in zip(nbs, got):
            n_o, m_o = single(has1, nb, only_stereo, ori)
            assert n_g == n_o and np.array_equal(m_g, m_o)
            tot += n_o
        assert tot > 60
    # LocalMapping's loop
    got = orb_slam2_amd.search_for_triangulation_batch(kf1, nbs, only_stereo=False, check_ori=False, library=backend)
    has_now = has1.copy()
    gained = 0
    for nb, (_, m_b) in zip(nbs, got):
        _, m_seq = single(has_now, nb, False, False)                       # what the reference's loop computes for this neighbour
        m_flt = np.where(has_now != 0, -1, m_b)                            # the batch's answer, filtered
        assert np.array_equal(m_flt, m_seq)
        for i1 in np.nonzero(m_seq >= 0)[0]:
            if (int(i1) * 7 + int(m_seq[i1])) % 3 != 0:                    # "triangulated": mpCurrentKeyFrame->AddMapPoint(pMP, idx1)
                has_now[i1] = 1; gained += 1
    assert gained > 40
    assert orb_slam2_amd.search_for_triangulation_batch(kf1, [], library=backend) == []


@pytest.mark.parametrize("chi2,same_bounds", [(True, True), (False, True), (True, False)])
def test_search_best_in_window_batch(backend, oracle, frames, chi2, same_bounds):
    """Fuse over several target key frames: stereo and mono targets, a target without key points, a slot without queries, equal and different image bounds."""
    w, h, par, F = frames
    inv = par["inv_sigma2"]
    sf = par["scale_factors"]
    rng = np.random.default_rng(31 + chi2)
    kl, dl = F[0]["k"], F[0]["d"]
    nq = len(kl)
    slots = []
    for j, i in enumerate((1, 2, 3, 4)):
        kc, dc = F[i]["k"], F[i]["d"]
        q = np.zeros(nq, orb_slam2_amd.BEST_QUERY_DTYPE)
        q["x"] = kl["x"] - 3.0 * i + rng.normal(0, 1.2, nq).astype(np.float32); q["y"] = kl["y"] - 1.0 * i + rng.normal(0, 1.2, nq).astype(np.float32)
        q["level"] = np.clip(kl["octave"] + rng.integers(0, 2, nq), 0, 7); q["radius"] = (np.float32(3.0 + j) * sf[q["level"]]).astype(np.float32); q["ur"] = q["x"] - np.float32(9.0)
        keep = rng.random(nq) < 0.8
        ur = np.where(rng.random(len(kc)) < 0.6, kc["x"] - rng.uniform(8, 10, len(kc)), -1).astype(np.float32) if j % 2 == 0 else None
        b = (0.0, 0.0, float(w), float(h)) if same_bounds or j % 2 == 0 else (-7.5, -4.0, w + 9.0, h + 3.5)
        slots.append(dict(kps=kc, desc=dc, u_right=ur, bounds=b, inv_level_sigma2=inv, queries=q[keep], qdesc=dl[keep]))
    slots.insert(2, dict(kps=kl[:0], desc=dl[:0], u_right=None, bounds=(0.0, 0.0, float(w), float(h)), inv_level_sigma2=inv, queries=slots[0]["queries"][:5], qdesc=slots[0]["qdesc"][:5]))
    slots.insert(4, dict(kps=F[2]["k"], desc=F[2]["d"], u_right=None, bounds=(0.0, 0.0, float(w), float(h)), inv_level_sigma2=inv, queries=slots[0]["queries"][:0], qdesc=slots[0]["qdesc"][:0]))
    got = orb_slam2_amd.search_best_in_window_batch(slots, chi2, library=backend)
    close = 0
    for sl, (bi_g, bd_g) in zip(slots, got):
        if len(sl["kps"]) == 0 or len(sl["queries"]) == 0:
            assert np.all(bi_g == -1) and np.all(bd_g == 256)
            continue
        bi_o, bd_o = orb_slam2_amd.search_best_in_window(sl["kps"], sl["desc"], w, h, inv, sl["queries"], sl["qdesc"], chi2, u_right=sl["u_right"], bounds=sl["bounds"], library=backend)
        assert np.array_equal(bi_g, bi_o) and np.array_equal(bd_g, bd_o)
        with oracle.image_bounds_set(sl["bounds"]):
            bi_c, bd_c = oracle.search_best_in_window(sl["kps"], sl["desc"], w, h, inv, sl["queries"], sl["qdesc"], chi2, u_right=sl["u_right"])
        assert np.array_equal(bi_g, bi_c) and np.array_equal(bd_g, bd_c)
        close += int((bd_o <= 50).sum())
    assert close > 200
    assert orb_slam2_amd.search_best_in_window_batch([], chi2, library=backend) == []


# ---- the same loops through the reference's own callers: per-call members (all-reference build) vs include/ORBmatcherBatch.h (the all-steps build) -----------
@pytest.fixture(scope="module", params=["all-steps", pytest.param("all-steps-gpu", marks=pytest.mark.gpu)])
def builds(request):
    from oracle import orbslam_ref as S
    from conftest import gpu_session
    if request.param.endswith("-gpu"):
        if not (S.build() and S.build_dropin_gpu()):
            pytest.fail("oracle/_ref/liborbslam_dropin_full_gpu.so did not travel with the repository (build it with `make -C oracle dropin_gpu` where /root/reference is mounted)")
        return S, S.dropin_gpu_lib(full=True)
    if gpu_session(request.config):
        pytest.skip("a -m gpu session maps liborbhip.so only")
    request.getfixturevalue("emu_lib")
    if not (S.build() and S.build_dropin()):
        pytest.skip("reference sources not mounted")
    return S, S.dropin_full_lib()


def test_local_mapping_loops_through_the_binding(builds, request, tmp_path):
    """LocalMapping::CreateNewMapPoints and SearchInNeighbors on a key frame with five neighbours: the all-reference build runs the reference's loops
    (SearchForTriangulation per neighbour with map points created in between; Fuse per target with REAL map surgery: Replace makes points bad and changes
    descriptors between targets), the all-steps build runs SearchForTriangulationBatch + TriangulationPairs and FuseBatch.  The matched pairs of every
    neighbour, the map point of every feature of every key frame afterwards and nFused must be equal."""
    S, D = builds
    gpu = "gpu" in request.node.name
    cfg = dict(w=1241, h=376, n=2000, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448, th_depth=35.0) if gpu else dict(w=400, h=300, n=500, fx=231.5, fy=231.5, cx=200.0, cy=150.0, bf=25.5, th_depth=35.0)
    nn = 5
    L, R, _, _ = synth.stereo_sequence(cfg["w"], cfg["h"], nn + 1, cfg["fx"], cfg["bf"], seed=9)
    voc = tmp_path / "voc_no_final_newline.txt"
    voc.write_text(open(VOC).read().rstrip("\n"))
    cam = dict(nfeatures=cfg["n"], fx=cfg["fx"], fy=cfg["fy"], cx=cfg["cx"], cy=cfg["cy"], bf=cfg["bf"], th_depth=cfg["th_depth"])
    rng = np.random.default_rng(2)
    F12 = np.stack([np.array([[0, -1e-3, 1.0 / 300], [1e-3, 0, -3.0 / 300], [-1.0 / 300, 3.0 / 300, 0]], np.float32) + rng.normal(0, 1e-5, (3, 3)).astype(np.float32) for _ in range(nn)])
    b = cfg["bf"] / cfg["fx"]
    t2w = np.stack([np.array([-0.25 * b * (i + 1), -0.125 * b * (i + 1), 0.0], np.float32) for i in range(nn)])       # the sequence's own camera motion
    res = {}
    for name, lib in (("ref", None), ("dropin", D)):
        S.RefFrame._geometry = None
        S.RefFrame._geometry_other.clear()
        frames = [S.RefFrame(L[i], R[i], library=lib, **cam) for i in range(nn + 1)]
        res[name] = S.local_mapping_loops(frames, F12, t2w, str(voc), point_depth=cfg["bf"] / 8.0)       # the far plane of the synthetic scene: disparity 8
        for f in frames:
            f.close()
    S.RefFrame._geometry = None
    (pr, ptr, nfr, msr), (pd, ptd, nfd, msd) = res["ref"], res["dropin"]
    for i in range(nn):
        assert np.array_equal(pr[i], pd[i]), f"matched pairs of neighbour {i} differ"
    assert np.array_equal(ptr, ptd) and nfr == nfd
    ids0 = set(int(v) for v in ptr[0] if v >= 0)
    shared = sum(len(ids0 & set(int(v) for v in ptr[i] if v >= 0)) for i in range(1, nn + 1))
    assert sum(len(p) for p in pr) > 100 and nfr > 50 and shared > 100 and int((ptr >= 1000000).sum()) == 0      # triangulation matches, fusions, points now seen by two key frames; no bad point left in a key frame
    print(f"\n[local_mapping] {cfg['w']}x{cfg['h']}, {nn} neighbours: SearchForTriangulation loop reference {msr[0]:.3f} ms / batch {msd[0]:.3f} ms, Fuse loop reference {msr[1]:.3f} ms / batch {msd[1]:.3f} ms, "
          f"{sum(len(p) for p in pr)} pairs, {nfr} fused, equal")
