#!/usr/bin/env python3
"""The single-image kernels under the profiler (tools/gpu_pmc.sh <tag> <passes> single): what ORB_SLAM2 drives - 30 single-image extraction calls, 30 stereo pairs
as one call each with a motion-model and a local-map search on the resident frame behind them (k_pyramid_cascade, k_fast_cells, k_blur_quadtree, k_describe,
k_stereo_*, k_match_grid, k_proj_candidates / k_proj_select), and a few stateless back-end calls (k_best_in_window, k_bow_match, k_bow_triangulate)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import orb_slam2_amd as A  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

W, H, N = 1241, 376, 2000
L, R, _, _ = synth.stereo_sequence(W, H, 4, 718.856, 386.1448, seed=5)
one = A.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=1, blur_round_mode=1)
for i in range(30):
    one(L[i % 4])
pair = A.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=2, blur_round_mode=1)
sf = pair.GetScaleFactors()
prev = None
for i in range(30):
    kl, dl, kr, dr, u, d = pair.extract_stereo(L[i % 4], R[i % 4], 386.1448, 386.1448 / 718.856)
    if prev is not None:
        pk, pd = prev
        q = np.zeros(len(pk), A.PROJ_QUERY_DTYPE)
        q["x"], q["y"] = pk["x"] - 2.0, pk["y"] - 1.0
        q["radius"] = (7.0 * sf[pk["octave"]]).astype(np.float32); q["ur"] = q["x"] - 8.0
        q["min_level"], q["max_level"], q["blocks"], q["angle"] = pk["octave"] - 1, pk["octave"] + 1, 1, pk["angle"]
        pair.search_by_projection(0, len(kl), q, pd, 1, nnratio=0.9, use_u_right=True)
        q["min_level"], q["max_level"] = pk["octave"] - 1, pk["octave"]
        pair.search_by_projection(0, len(kl), q, pd, 0, nnratio=0.8, use_u_right=True)
    prev = (kl, dl)
kl, dl = prev
bq = np.zeros(len(kl), A.BEST_QUERY_DTYPE)
bq["x"], bq["y"], bq["radius"], bq["level"] = kl["x"] - 2, kl["y"] - 1, (3.0 * sf[kl["octave"]]).astype(np.float32), kl["octave"]
inv = (1.0 / (sf * sf)).astype(np.float32)
voc = A.ORBVocabulary(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "voc_k6_L3_ref.txt"))
fv = voc.transform(dl, 2)[2:]
v = np.ones(len(kl), np.uint8)
for _ in range(10):
    A.search_best_in_window(kl, dl, W, H, inv, bq, dl, True)
    A.search_by_bow(1, dl, kl["angle"], v, fv, dl, kl["angle"], v, fv, nnratio=0.75)
    A.search_for_triangulation(dl, kl, 1 - v, 1 - v, fv, dl, kl, 1 - v, 1 - v, fv, np.eye(3, dtype=np.float32) * 1e-3, 620.0, 190.0, sf, sf * sf, check_ori=False)
print("pmc single workload done")
