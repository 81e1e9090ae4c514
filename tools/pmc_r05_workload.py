#!/usr/bin/env python3
"""Round 5's new kernels under the profiler (tools/gpu_pmc.sh <tag> <passes> r05): bag of words for one frame on an ORBvoc-shaped vocabulary (k_bow_descend,
k_bow_assemble: 1000 and 2000 features), SearchByBoW / SearchForTriangulation on its 100-node partition (k_bow_match, k_bow_triangulate), stereo pairs as one call
(k_stereo_rows on 1024 threads), and two 2000-query scans of a 2 M-row descriptor database on the FP4 matrix path (k_hamming_nn_fp4)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import orb_slam2_amd as A  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402
from secondary_units import write_voc  # noqa: E402

p = "/tmp/voc_k10_L6_pmc.txt"
if not os.path.exists(p):
    write_voc(p, 10, 6)
voc = A.ORBVocabulary(p)
rng = np.random.default_rng(1)
W, H, N = 1241, 376, 2000
L, R, _, _ = synth.stereo_sequence(W, H, 3, 718.856, 386.1448, seed=5)
pair = A.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=2, blur_round_mode=1)
frames = []
for i in range(12):
    kl, dl, kr, dr, u, d = pair.extract_stereo(L[i % 3], R[i % 3], 386.1448, 386.1448 / 718.856)
    frames.append((kl, dl))
for n in (1000, 2000):
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for _ in range(10):
        voc.transform(d, 4)
(k1, d1), (k2, d2) = frames[0], frames[1]
fv1, fv2 = voc.transform(d1, 4)[2:], voc.transform(d2, 4)[2:]
v1 = (rng.random(len(k1)) < 0.8).astype(np.uint8); v2 = (rng.random(len(k2)) < 0.8).astype(np.uint8)
for _ in range(10):
    A.search_by_bow(0, d1, k1["angle"], v1, fv1, d2, k2["angle"], None, fv2, nnratio=0.7, check_ori=True)
    A.search_by_bow(1, d1, k1["angle"], v1, fv1, d2, k2["angle"], v2, fv2, nnratio=0.75, check_ori=True)
sf = pair.GetScaleFactors()
F = np.array([[0, -1e-3, 1.0 / 300], [1e-3, 0, -3.0 / 300], [-1.0 / 300, 3.0 / 300, 0]], np.float32)
has1 = (rng.random(len(k1)) < 0.3).astype(np.uint8); has2 = (rng.random(len(k2)) < 0.3).astype(np.uint8)
st = np.zeros(len(k1), np.uint8); st2 = np.zeros(len(k2), np.uint8)
for _ in range(10):
    A.search_for_triangulation(d1, k1, has1, st, fv1, d2, k2, has2, st2, fv2, F, np.float32(600), np.float32(180), sf, (sf * sf).astype(np.float32), only_stereo=False, check_ori=True)
NDB, NQ = 2_000_000, 2000
db_h = rng.integers(0, 256, (NDB, 32), dtype=np.uint8)
q_h = db_h[rng.integers(0, NDB, NQ)].copy()
db = A.DeviceBuffer.from_array(db_h); q = A.DeviceBuffer.from_array(q_h)
bi = A.DeviceBuffer(NQ * 8); bd = A.DeviceBuffer(NQ * 4); sd = A.DeviceBuffer(NQ * 4)
for _ in range(3):
    A.hamming_nn_device(None, q.ptr, NQ, db.ptr, NDB, bi.ptr, bd.ptr, sd.ptr)
    A.device_synchronize()
print("pmc workload done")
