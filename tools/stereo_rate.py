#!/usr/bin/env python3
"""Secondary unit of SURVEY.md §8d: stereo pair = 2 x extract + Frame::ComputeStereoMatches, frames resident in HBM."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import orb_slam2_amd  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

W, H, B = 1241, 376, 128
pitch = 1280
m = 64
left = np.zeros((B, H, pitch), np.uint8)
right = np.zeros((B, H, pitch), np.uint8)
for s in range(8):
    sc = synth.scene(W, H, seed=200 + s)
    L = synth.frame_from_scene(sc, W, H, t=0, seed=200 + s)
    rng = np.random.default_rng(300 + s)
    R = np.clip(np.rint(sc[m // 2:m // 2 + H, m // 2 + 14:m // 2 + 14 + W]) + rng.integers(-6, 7, size=(H, W)), 0, 255).astype(np.uint8)
    for b in range(s, B, 8):
        left[b, :, :W], right[b, :, :W] = L, R
dl, dr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
torch.cuda.synchronize()
xl = orb_slam2_amd.ORBextractor(2000, 1.2, 8, 20, 7, W, H, max_batch=B)
xr = orb_slam2_amd.ORBextractor(2000, 1.2, 8, 20, 7, W, H, max_batch=B)


def step():
    xl.extract_device(dl.data_ptr(), B, H * pitch, pitch)
    xr.extract_device(dr.data_ptr(), B, H * pitch, pitch)
    return xl.ComputeStereoMatches(xr, 386.1448, 386.1448 / 718.856, nimg=B)


u, d = step()
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    u, d = step()
dt = time.perf_counter() - t0
print(json.dumps({"stereo_pairs_per_s": round(B * reps / dt, 1), "matched_per_pair": int((u[0] >= 0).sum()),
                  "note": "2 x extract (2000 features) + ComputeStereoMatches per pair, inputs resident in HBM, mvuRight/mvDepth copied to the host"}))
