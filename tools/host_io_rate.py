#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in boundary: host images in, host keypoints + descriptors out (orbhip_extract_batch),
pageable numpy buffers, synchronous.  Reported in DESIGN.md next to the HBM-resident bench value; never the bench `value`."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import orb_slam2_amd  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

W, H, B = 1241, 376, 64
frames = [synth.frame(W, H, seed=s % 8, t=s // 8) for s in range(B)]
ex = orb_slam2_amd.ORBextractor(2000, 1.2, 8, 20, 7, W, H, max_batch=B)
ex.extract_batch(frames)
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    ks, ds = ex.extract_batch(frames)
dt = time.perf_counter() - t0
one = orb_slam2_amd.ORBextractor(2000, 1.2, 8, 20, 7, W, H, max_batch=1)
one(frames[0])
t1 = time.perf_counter()
for i in range(50):
    one(frames[i % B])
lat = (time.perf_counter() - t1) / 50
print(json.dumps({"host_io_batch64_frames_per_s": round(B * reps / dt, 1), "single_frame_latency_ms": round(lat * 1e3, 3),
                  "note": "extract only (no match), pageable host buffers, includes H2D of images and D2H of keypoints+descriptors"}))
