"""(instrumented build: tools/trace_builds.py; run with ORBHIP_LIBRARY=$PWD/ab/liborbhip_<name>.so)  Start / end of every workgroup of the kernels of one single-image call on the GPU's own 100 MHz clock (instrumented build ab/liborbhip_span.so; one store per
workgroup, no atomics): how long the kernels' workgroups really run and what lies BETWEEN the kernels of the dependent chain.  Measurement aid."""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, orb_slam2_amd
from orb_slam2_amd import synth
W, H, N = 1241, 376, 2000
img = synth.frame(W, H, seed=3)
ex = orb_slam2_amd.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=1, blur_round_mode=1)
for rep in range(3):
    t = np.zeros(4 * 2 * 256, np.uint64)
    for i in range(4): ex(img)
    ex.L.orbhip_debug_wg(t.ctypes.data_as(C.c_void_p))
    t = t.reshape(4, 2, 256).astype(np.int64)
    def span(k):
        s, e = t[k, 0], t[k, 1]; m = s > 0
        return s[m].min(), e[m].max() if e[m].max() > 0 else 0, s[m], e[m]
    names = ["pyramid cascade", "FAST", "blur+quadtree", "describe"]
    prev_end = None; out = []
    for k in range(4):
        s0, e1, s, e = span(k)
        if prev_end: out.append("gap %.2f" % ((s0 - prev_end) / 100))
        if k < 3:
            d = (e - s) / 100.0
            out.append("%s: first start -> last end %.2f us (%d workgroups: run %.1f-%.1f us each, starts spread over %.2f us, ends %.2f..%.2f)" % (names[k], (e1 - s0) / 100, len(s), d.min(), d.max(), (s.max() - s0) / 100, (e.min() - s0) / 100, (e1 - s0) / 100))
            prev_end = e1
    print(" | ".join(out))
    if rep == 2:
        s0, e1, s, e = span(1); d = (e - s) / 100.0; late = np.argsort(e)[-6:]
        print("FAST: the six workgroups that end last:", [(int(i), round(float((s[i] - s0) / 100), 2), round(float(d[i]), 2)) for i in late], "(index, start, run us)")
        s0, e1, s, e = span(2); print("blur+quadtree: quadtree workgroups (level: run us)", [(l, round(float((e[l] - s[l]) / 100), 1)) for l in range(8)])
