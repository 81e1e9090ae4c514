"""Rounds, rescans and us per round of k_proj_select (instrumented build: tools/trace_builds.py pjtrace; run with ORBHIP_LIBRARY=$PWD/ab/liborbhip_pjtrace.so).  Measurement aid."""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, orb_slam2_amd
from orb_slam2_amd import synth, orbhip
W, H, N = 1241, 376, 2000
seq = synth.sequence(W, H, 2, seed=5)
ex = orb_slam2_amd.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=2)
ks, ds = ex.extract_batch(seq); sf = ex.GetScaleFactors()
(k1, d1), (k2, d2) = (ks[0], ds[0]), (ks[1], ds[1])
rng = np.random.default_rng(1)
L = ex.L
for frac in (0.75, 0.4):
  keep = rng.random(len(k1)) < frac
  q = np.zeros(int(keep.sum()), orb_slam2_amd.PROJ_QUERY_DTYPE)
  q["x"], q["y"] = k1["x"][keep] - 3, k1["y"][keep] - 1
  q["radius"] = (7.0 * sf[k1["octave"][keep]]).astype(np.float32)
  q["min_level"], q["max_level"], q["blocks"], q["angle"] = k1["octave"][keep] - 1, k1["octave"][keep] + 1, 1, k1["angle"][keep]
  for mode, nn in ((1, 0.9), (0, 0.8)):
    t = np.zeros(8, np.uint64); L.orbhip_debug_pj(t.ctypes.data_as(C.c_void_p))
    for i in range(10): r = orb_slam2_amd.search_by_projection(k2, d2, W, H, q, d1[keep], mode, nnratio=nn)
    L.orbhip_debug_pj(t.ctypes.data_as(C.c_void_p)); t = t.astype(np.int64); c = t[0]
    print(f"mode {mode} nq {t[6]//c}: steps of 256 {t[1]/c:.0f} rounds {t[2]/c:.1f} rescans {t[3]/c:.1f} | prologue {t[4]/c/100:.1f} us, selection loop {t[5]/c/100:.1f} us ({t[5]/t[1]/100:.2f} us per step, {t[5]/t[2]/100:.2f} per round)")
