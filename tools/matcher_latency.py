#!/usr/bin/env python3
"""Secondary measurement: wall-clock latency of the host-pointer matcher entry points (one call = upload, kernels, download) on
KITTI-sized inputs: 2000 features per frame, ~1500 queries.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orb_slam2_amd  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402


def timed(fn, reps=30):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return round((time.perf_counter() - t0) / reps * 1e3, 3)


def main():
    W, H, N = 1241, 376, 2000
    seq = synth.sequence(W, H, 2, seed=5)
    ex = orb_slam2_amd.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=2)
    ks, ds = ex.extract_batch(seq)
    sf = ex.GetScaleFactors()
    (k1, d1), (k2, d2) = (ks[0], ds[0]), (ks[1], ds[1])
    rng = np.random.default_rng(1)
    keep = rng.random(len(k1)) < 0.75
    q = np.zeros(int(keep.sum()), orb_slam2_amd.PROJ_QUERY_DTYPE)
    q["x"], q["y"] = k1["x"][keep] - 3, k1["y"][keep] - 1
    q["radius"] = (7.0 * sf[k1["octave"][keep]]).astype(np.float32)
    q["min_level"], q["max_level"], q["blocks"], q["angle"] = k1["octave"][keep] - 1, k1["octave"][keep] + 1, 1, k1["angle"][keep]
    bq = np.zeros(len(q), orb_slam2_amd.BEST_QUERY_DTYPE)
    bq["x"], bq["y"], bq["radius"], bq["level"] = q["x"], q["y"], q["radius"], k1["octave"][keep]
    inv = (1.0 / (sf * sf)).astype(np.float32)
    m = orb_slam2_amd.ORBmatcher(0.9, True)
    out = {
        "features": [len(k1), len(k2)], "queries": len(q),
        "search_for_initialization_ms": timed(lambda: m.SearchForInitialization(k1, d1, k2, d2, W, H, windowSize=100)),
        "search_by_projection_last_frame_ms": timed(lambda: orb_slam2_amd.search_by_projection(k2, d2, W, H, q, d1[keep], 1, nnratio=0.9)),
        "search_by_projection_local_map_ms": timed(lambda: orb_slam2_amd.search_by_projection(k2, d2, W, H, q, d1[keep], 0, nnratio=0.8)),
        "search_best_in_window_ms": timed(lambda: orb_slam2_amd.search_best_in_window(k2, d2, W, H, inv, bq, d1[keep], True)),
        "extract_single_frame_ms": timed(lambda: ex(seq[0])),
    }
    # M4 / M5: the BoW-guided matchers on FeatureVectors of the small golden vocabulary (k = 6, L = 3: 216 words, levelsup 2 -> 6 nodes of
    # ~330 features each; the real ORBvoc at levelsup 4 gives ~100 nodes of ~20 features, i.e. far fewer pairs per node), Fuse's / SearchBySim3's window search
    voc = orb_slam2_amd.ORBVocabulary(os.path.join(ROOT, "tests", "golden", "voc_k6_L3_ref.txt"))
    fv1, fv2 = voc.transform(d1, 2)[2:], voc.transform(d2, 2)[2:]
    valid1 = (rng.random(len(k1)) < 0.8).astype(np.uint8); valid2 = (rng.random(len(k2)) < 0.8).astype(np.uint8)
    out["search_by_bow_keyframe_frame_ms"] = timed(lambda: orb_slam2_amd.search_by_bow(0, d1, k1["angle"], valid1, fv1, d2, k2["angle"], None, fv2, nnratio=0.7), reps=10)
    out["search_by_bow_keyframe_keyframe_ms"] = timed(lambda: orb_slam2_amd.search_by_bow(1, d1, k1["angle"], valid1, fv1, d2, k2["angle"], valid2, fv2, nnratio=0.75), reps=10)
    # ... and on a partition shaped like ORBvoc's at levelsup 4 (k = 10, L = 6: the 100 nodes of level 2, ~20 features each): features of both frames dealt to
    # 100 nodes by the same rule (their first descriptor byte mod 100), so that a node holds similar counts on both sides - the matcher's work per node
    def partition(desc):
        node = desc[:, 0].astype(np.int64) % 100
        order = np.argsort(node, kind="stable")
        ids, counts = np.unique(node, return_counts=True)
        return ids.astype(np.uint32), np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), order.astype(np.uint32)
    pv1, pv2 = partition(d1), partition(d2)
    out["search_by_bow_keyframe_frame_100_nodes_ms"] = timed(lambda: orb_slam2_amd.search_by_bow(0, d1, k1["angle"], valid1, pv1, d2, k2["angle"], None, pv2, nnratio=0.7), reps=10)
    out["search_by_bow_keyframe_keyframe_100_nodes_ms"] = timed(lambda: orb_slam2_amd.search_by_bow(1, d1, k1["angle"], valid1, pv1, d2, k2["angle"], valid2, pv2, nnratio=0.75), reps=10)
    F12 = np.array([[0, -1e-3, 1.0 / 300], [1e-3, 0, -3.0 / 300], [-1.0 / 300, 3.0 / 300, 0]], np.float32)
    no1 = (rng.random(len(k1)) < 0.3).astype(np.uint8); no2 = (rng.random(len(k2)) < 0.3).astype(np.uint8)
    z1, z2 = np.zeros(len(k1), np.uint8), np.zeros(len(k2), np.uint8)
    out["search_for_triangulation_ms"] = timed(lambda: orb_slam2_amd.search_for_triangulation(d1, k1, no1, z1, fv1, d2, k2, no2, z2, fv2, F12, 620.0, 190.0, sf, sf * sf), reps=10)
    out["search_for_triangulation_100_nodes_ms"] = timed(lambda: orb_slam2_amd.search_for_triangulation(d1, k1, no1, z1, pv1, d2, k2, no2, z2, pv2, F12, 620.0, 190.0, sf, sf * sf), reps=10)
    out["fuse_window_search_ms"] = timed(lambda: orb_slam2_amd.search_best_in_window(k2, d2, W, H, inv, bq, d1[keep], True))
    out["search_by_sim3_two_window_searches_ms"] = round(2 * timed(lambda: orb_slam2_amd.search_best_in_window(k2, d2, W, H, inv, bq, d1[keep], False)), 3)
    ex.extract_batch(seq)                                  # frame 1 (k2, d2) is on the device again: only queries and results travel
    out["search_by_projection_frame_on_device_ms"] = timed(lambda: ex.search_by_projection(1, len(k2), q, d1[keep], 0, nnratio=0.8))
    out["search_best_in_window_frame_on_device_ms"] = timed(lambda: ex.search_best_in_window(1, len(k2), bq, d1[keep], True))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
