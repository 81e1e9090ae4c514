"""The matcher entry points against results FROZEN FROM THE REFERENCE's own ORBmatcher members (tests/golden/matchers_ref.npz, written by
tests/golden/make_golden_matchers.py through real Frame / KeyFrame / MapPoint objects of the reference).  No oracle, no reference at run
time: on the GPU box this compares the HIP path with the reference's results directly."""
import os

import numpy as np

import orb_slam2_amd

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "matchers_ref.npz"))


def test_projection_matchers_and_fuse_equal_the_reference(backend):
    w, h = int(G["width"]), int(G["height"])
    kc, dc = G["cur_keys"], G["cur_desc"]
    # ORBmatcher(0.8).SearchByProjection(Frame, MapPoints, th = 3)
    n, fq = orb_slam2_amd.search_by_projection(kc, dc, w, h, G["local_queries"], G["local_qdesc"], 0, nnratio=0.8, th_high=100, blocked=G["local_blocked"], library=backend)
    assert n == int(G["local_n"]) and np.array_equal(fq, G["local_feature_query"]) and n > 100
    # ORBmatcher(0.9, true).SearchByProjection(CurrentFrame, LastFrame, th = 15, bMono = true)
    n, fq = orb_slam2_amd.search_by_projection(kc, dc, w, h, G["last_queries"], G["last_qdesc"], 1, nnratio=0.9, th_high=100, check_ori=True, blocked=G["last_blocked"],
                                               library=backend)
    assert n == int(G["last_n"]) and np.array_equal(np.where(fq >= 0, fq, -1), G["last_feature_query"]) and n > 100
    # ORBmatcher().Fuse(KeyFrame, MapPoints, th = 3): the key point per map point; the caller's threshold is TH_LOW
    bi, bd = orb_slam2_amd.search_best_in_window(kc, dc, w, h, G["inv_sigma2"], G["fuse_queries"], G["fuse_qdesc"], True, library=backend)
    best = np.where(bd <= 50, bi, -1)
    assert np.array_equal(best, G["fuse_best"]) and int((best >= 0).sum()) == int(G["fuse_n"]) > 100


def test_bow_matchers_and_triangulation_equal_the_reference(backend):
    k1, d1, k2, d2 = G["prev_keys"], G["prev_desc"], G["cur_keys"], G["cur_desc"]
    fv1, fv2 = (G["fv1_node"], G["fv1_off"], G["fv1_feat"]), (G["fv2_node"], G["fv2_off"], G["fv2_feat"])
    # ORBmatcher(0.8, true).SearchByBoW(KeyFrame, Frame, ..) and (KeyFrame, KeyFrame, ..)
    for mode in (0, 1):
        n, m12 = orb_slam2_amd.search_by_bow(mode, d1, k1["angle"], G["bow_valid1"], fv1, d2, k2["angle"], G["bow_valid2"] if mode == 1 else None, fv2, nnratio=0.8, check_ori=True,
                                             library=backend)
        assert n == int(G["bow%d_n" % mode]) and np.array_equal(m12, G["bow%d_match12" % mode]) and n > 100
    # ORBmatcher(0.6, true).SearchForTriangulation(KF1, KF2, F12, .., bOnlyStereo = false): monocular key frames (no stereo flags)
    none1, none2 = np.zeros(len(k1), np.uint8), np.zeros(len(k2), np.uint8)
    n, m12 = orb_slam2_amd.search_for_triangulation(d1, k1, G["tri_has1"], none1, fv1, d2, k2, G["tri_has2"], none2, fv2, G["tri_F12"], float(G["tri_ex"]), float(G["tri_ey"]),
                                                    G["scale_factors"], G["sigma2"], only_stereo=False, check_ori=True, library=backend)
    assert n == int(G["tri_n"]) and np.array_equal(m12, G["tri_match12"]) and n > 50


def test_loop_closing_and_relocalisation_members_equal_the_reference(backend):
    """Fuse(pKF, Scw, ..), SearchByProjection(pKF, Scw, ..), SearchByProjection(CurrentFrame, pKF, ..) and SearchBySim3 as the reference's own
    code ran them on real KeyFrame / MapPoint objects (identity pose / similarity), against the entry points their search loops map to."""
    w, h = int(G["width"]), int(G["height"])
    k1, d1, k2, d2 = G["prev_keys"], G["prev_desc"], G["cur_keys"], G["cur_desc"]
    # ORBmatcher().Fuse(pKF, Scw, vpPoints, th = 4, vpReplacePoint): best-in-window without the chi-square gate, caller's threshold TH_LOW
    bi, bd = orb_slam2_amd.search_best_in_window(k2, d2, w, h, G["inv_sigma2"], G["fuse3_queries"], G["fuse3_qdesc"], False, library=backend)
    best = np.where(bd <= 50, bi, -1)
    assert np.array_equal(best, G["fuse3_best"]) and int((best >= 0).sum()) == int(G["fuse3_n"]) > 100
    # ORBmatcher(0.75, true).SearchByProjection(pKF, Scw, vpPoints, vpMatched, th = 10): mode 1, levels [L-1, L], TH_LOW, no orientation check
    n, fq = orb_slam2_amd.search_by_projection(k2, d2, w, h, G["kfsim3_queries"], G["kfsim3_qdesc"], 1, nnratio=0.75, th_high=50, check_ori=False, blocked=G["kfsim3_blocked"], library=backend)
    assert n == int(G["kfsim3_n"]) and np.array_equal(np.where(fq >= 0, fq, -1), G["kfsim3_feature_query"]) and n > 100
    # ORBmatcher(0.9, true).SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th = 10, ORBdist = 100): mode 1, levels [L-1, L+1]
    n, fq = orb_slam2_amd.search_by_projection(k2, d2, w, h, G["reloc_queries"], G["reloc_qdesc"], 1, nnratio=0.9, th_high=100, check_ori=True, blocked=G["reloc_blocked"], library=backend)
    assert n == int(G["reloc_n"]) and np.array_equal(np.where(fq >= 0, fq, -1), G["reloc_feature_query"]) and n > 100
    # ORBmatcher(0.75, true).SearchBySim3: one best-in-window pass per direction (no gate, caller's threshold TH_HIGH), then the mutual check
    def one_pass(keep, q, dA, kB, dB, nA):
        bi, bd = orb_slam2_amd.search_best_in_window(kB, dB, w, h, G["inv_sigma2"], q, dA[keep], False, library=backend)
        out = np.full(nA, -1, np.int32)
        ok = (bd <= 100) & (bi >= 0)
        out[keep[ok]] = bi[ok]
        return out
    v1 = one_pass(G["sim3_keep1"], G["sim3_queries1"], d1, k2, d2, len(k1))
    v2 = one_pass(G["sim3_keep2"], G["sim3_queries2"], d2, k1, d1, len(k2))
    m12 = np.full(len(k1), -1, np.int32)
    for i1 in range(len(k1)):
        if v1[i1] >= 0 and v2[v1[i1]] == i1:
            m12[i1] = v1[i1]                                                # ORBmatcher.cc:1306-1322
    assert np.array_equal(m12, G["sim3_match12"]) and int((m12 >= 0).sum()) == int(G["sim3_n"]) > 100
