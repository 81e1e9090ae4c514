"""(instrumented build: tools/trace_builds.py; run with ORBHIP_LIBRARY=$PWD/ab/liborbhip_<name>.so)  Phase times inside k_fast_cells for a single frame (an instrumented build, ab/liborbhip_fctrace.so: s_memrealtime stamps of eight workgroups).  Measurement aid."""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, orb_slam2_amd
from orb_slam2_amd import synth
W, H, N = 1241, 376, 2000
img = synth.frame(W, H, seed=3)
ex = orb_slam2_amd.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=1, blur_round_mode=1)
t = np.zeros(8 * 16, np.uint64)
for i in range(6): ex(img)
ex.L.orbhip_debug_fc_trace(t.ctypes.data_as(C.c_void_p))
t = t.reshape(8, 16).astype(np.int64)
live = [k for k in range(8) if t[k, 0] > 0]
t0 = min(t[k, 0] for k in live)
names = ["cell desc", "patch DMA", "pretest (stage a)", "barrier", "scores (stage b)", "barrier", "NMS + emit (stage c)", "minTh call + count"]
for k in live:
    r = t[k]
    print("workgroup %3d start +%.2f us | " % (20 * k, (r[0] - t0) / 100) + " | ".join("%s %.2f" % (names[i], (r[i + 1] - r[i]) / 100) for i in range(8)) + " | end +%.2f" % ((r[8] - t0) / 100))
