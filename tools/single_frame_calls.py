#!/usr/bin/env python3
"""40 single-frame drop-in calls (ORBextractor.__call__ on one pageable 1241x376 frame) — the thing Frame::ExtractORB sees.  Run under
rocprofv3 --sys-trace to get the timeline of one call; prints the mean call latency itself."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import orb_slam2_amd
from orb_slam2_amd import synth
W, H = 1241, 376
img = [synth.frame(W, H, seed=s) for s in range(4)]
ex = orb_slam2_amd.ORBextractor(2000, 1.2, 8, 20, 7, W, H, max_batch=1, blur_round_mode=1)
for i in range(8):
    ex(img[i % 4])
t0 = time.perf_counter()
for i in range(40):
    ex(img[i % 4])
print("single_frame_call_ms", round((time.perf_counter() - t0) / 40 * 1e3, 4))
