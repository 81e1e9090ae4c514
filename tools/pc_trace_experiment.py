"""Phase times inside k_pyramid_cascade.  Needs an instrumented build: `tools/trace_builds.py pctrace` (s_memrealtime stamps of eight workgroups) builds
ab/liborbhip_pctrace.so; run with ORBHIP_LIBRARY pointing at it.  Measurement aid, not part of the product."""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, orb_slam2_amd
from orb_slam2_amd import synth
W, H, N = 1241, 376, 2000
img = synth.frame(W, H, seed=3)
ex = orb_slam2_amd.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=1, blur_round_mode=1)
t = np.zeros(8 * 32, np.uint64)
for i in range(6): ex(img)
ex.L.orbhip_debug_pc_trace(t.ctypes.data_as(C.c_void_p))
t = t.reshape(8, 32).astype(np.int64)
live = [k for k in range(8) if t[k, 0] > 0]
t0 = min(t[k, 0] for k in live)
for k in live:
    r = t[k]
    print("tile %3d start +%.2f us | round 1 %.2f | round 2 loads %.2f | stores+barrier %.2f | levels" % (11 * k, (r[0] - t0) / 100, (r[1] - r[0]) / 100, (r[2] - r[1]) / 100, (r[3] - r[2]) / 100),
          [round(float(r[3 + l] - r[2 + l]) / 100, 2) for l in range(1, 8)], "| end +%.2f" % ((r[10] - t0) / 100))
