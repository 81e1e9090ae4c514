# instrumented build: tools/trace_builds.py qttrace; run with ORBHIP_LIBRARY=$PWD/ab/liborbhip_qttrace.so
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, orb_slam2_amd
from orb_slam2_amd import synth
for (W, H, N) in ((1241, 376, 2000), (752, 480, 1200)):
    img = synth.frame(W, H, seed=3)
    ex = orb_slam2_amd.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=1, blur_round_mode=1)
    for i in range(5): ex(img)
    t = np.zeros(64 * 8, np.uint64)
    ex.L.orbhip_debug_qt_trace(t.ctypes.data_as(C.c_void_p))
    t = t.reshape(8, 64)
    for lvl in range(8):
        r = t[lvl].astype(np.int64); n = int(r[63]); npass = int(r[62]); K = int(r[61]); m = int(r[60])
        us = lambda a, b: (r[b] - r[a]) / 100.0
        st = [5 + k for k in range(min(npass, 30))] + [44]
        passes = [round(us(st[k], st[k + 1]), 1) for k in range(len(st) - 1)]
        print(f"{W}x{H} level {lvl}: n={n} m={m} total {us(0,45):.1f} us | cells+scan {us(0,1):.1f} | dense copy {us(1,2):.1f} | read back + path codes {us(2,3):.1f} | jump (K={K}) {us(3,4):.1f} | setup {us(4, 5) if npass else us(4,44):.1f} | {npass} passes {passes} | best+out {us(44,45):.1f}")
        if lvl in (0, 3):
            names = ["child histogram (16 LDS atomics / key) + barrier", "flags + scan", "compaction + rank", "barrier", "children + barrier", "scan", "break position + barrier", "marks, flags (3 barriers)", "scan", "new nodes", "barrier", "keys to their new nodes", "count, clear, barrier"]
            idx = [5, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32]
            jn = ["entry", "histogram sums", "wave sums + atomics", "barrier", "K", "flags + scan", "node sizes + barrier", "keys to their nodes", "barrier"]
            ji = [3, 33, 34, 35, 36, 37, 38, 39, 40, 4]
            print("      jump: " + " | ".join("%s %.1f" % (jn[i], us(ji[i], ji[i + 1])) for i in range(9)))
            print("      final-phase pass: " + " | ".join("%s %.1f" % (names[i], us(idx[i], idx[i + 1])) for i in range(13)))
    ex.close()
