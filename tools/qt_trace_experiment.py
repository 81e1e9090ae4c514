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
        r = t[lvl].astype(np.int64); n = int(r[63]); npass = int(r[62])
        us = lambda a, b: (r[b] - r[a]) / 100.0
        passes = [round(us(4 + k, 5 + k) if k else us(4, 5), 1) for k in range(min(npass, 40))]
        ms = [(int(r[46 + k]) >> 1, int(r[46 + k]) & 1) for k in range(min(npass, 16))]
        print(f"{W}x{H} level {lvl}: n={n} total {us(0,45):.1f} us | cells+scan {us(0,1):.1f} | dense copy {us(1,2):.1f} | read back + path codes {us(2,3):.1f} | roots {us(3,4):.1f} | {npass} passes {passes} (m, modeB) {ms} | best+out {us(44,45):.1f}")
    ex.close()
