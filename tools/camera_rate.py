#!/usr/bin/env python3
"""Secondary measurement: the camera-geometry stages either side of the extractor on the GPU.

  tum    640x480 frames of a distorted monocular camera (TUM1.yaml, 1000 features): extract + match_prev with and without the
         camera attached -> what Frame::UndistortKeyPoints on the device costs per batch; per-kernel times from the library's events.
  euroc  752x480 raw stereo frames (EuRoC.yaml geometry, 1200 features) rectified on the device in front of the pyramid
         (orbhip_extract_device_rectify) against the same batch entering already rectified -> what cv::remap on the device costs.
Prints one JSON line.  (Parity is tests/test_parity_camera.py's job; this tool only measures.)"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TUM1 = (517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314)


_KEEP = []


def _upload(hip, host):
    import orb_slam2_amd
    b = orb_slam2_amd.DeviceBuffer.from_array(host)          # the library's own HIP runtime
    _KEEP.append(b)
    return b.ptr


def _ms(prof, name):
    p = prof.get(name)
    return round(p["total_ms"] / p["launches"], 4) if p and p["launches"] else None


def _timed(fn, sync, steps):
    fn(); fn(); sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import orb_slam2_amd
    from orb_slam2_amd import synth
    hip = None
    B = args.batch
    out = {"batch": B}

    # ---- distorted monocular camera
    W, H, N = 640, 480, 1000
    pitch = (W + 63) // 64 * 64
    steps = [np.zeros((B, H, pitch), np.uint8) for _ in range(2)]
    seqs = [synth.sequence(W, H, 2, seed=500 + s) for s in range(min(B, 8))]
    for t in range(2):
        for b in range(B):
            steps[t][b, :, :W] = seqs[b % len(seqs)][t]
    d = [_upload(hip, s) for s in steps]
    ex = orb_slam2_amd.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=B)
    state = {"t": 0}

    def step():
        ex.extract_device(d[state["t"] & 1], B, H * pitch, pitch, match_prev=True, window=100, nnratio=0.9, check_ori=True)
        state["t"] += 1
    dt_plain = _timed(step, ex.sync, args.steps)
    ex.set_camera(TUM1)
    ex.profile_enable(True); ex.profile_reset()
    dt_cam = _timed(step, ex.sync, args.steps)
    prof = ex.profile()
    ks, _ = ex.fetch(1)
    un = ex.fetch_undistorted(1, [len(ks[0])])[0]
    out["tum_640x480"] = {"features_per_frame": len(ks[0]), "frames_per_s_undistorted_camera": round(B / dt_plain, 1), "frames_per_s_distorted_camera": round(B / dt_cam, 1),
                          "k_undistort_keys_ms_per_launch": _ms(prof, "k_undistort_keys"),
                          "max_shift_px": round(float(max(np.abs(un["x"] - ks[0]["x"]).max(), np.abs(un["y"] - ks[0]["y"]).max())), 2),
                          "bounds": [round(float(v), 3) for v in ex.bounds()]}
    ex.close()

    # ---- raw stereo input rectified on the device
    W, H, N = 752, 480, 1200
    pitch = (W + 63) // 64 * 64
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    th = 0.012
    dx, dy = xx - 367.2, yy - 248.4
    r2 = (dx * dx + dy * dy) / (458.0 * 458.0)
    mx = (367.2 + (np.cos(th) * dx - np.sin(th) * dy) * (1 - 0.28 * r2 + 0.07 * r2 * r2)).astype(np.float32)          # EuRoC-like: barrel distortion + small rotation
    my = (248.4 + (np.sin(th) * dx + np.cos(th) * dy) * (1 - 0.28 * r2 + 0.07 * r2 * r2)).astype(np.float32)
    raw = np.zeros((B, H, pitch), np.uint8)
    scenes = [synth.frame(W, H, seed=700 + s) for s in range(min(B, 8))]
    for b in range(B):
        raw[b, :, :W] = scenes[b % len(scenes)]
    d_raw = _upload(hip, raw)
    ex = orb_slam2_amd.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=B)
    ex.set_rectification(mx, my, W, H)
    dt_plain = _timed(lambda: ex.extract_device(d_raw, B, H * pitch, pitch), ex.sync, args.steps)
    ex.profile_enable(True); ex.profile_reset()
    dt_rect = _timed(lambda: ex.extract_device_rectify(d_raw, B, H * pitch, pitch), ex.sync, args.steps)
    prof = ex.profile()
    ms = _ms(prof, "k_remap")
    out["euroc_752x480"] = {"frames_per_s_rectified_input": round(B / dt_plain, 1), "frames_per_s_raw_input_remap_on_device": round(B / dt_rect, 1),
                            "k_remap_ms_per_launch": ms,
                            "k_remap_GBps_algorithmic": None if not ms else round(B * W * H * (1 + 1) / (ms * 1e-3) / 1e9, 1),      # 1 B gathered + 1 B written per pixel; the maps (8 B/px) stay in L2 across the batch
                            "k_remap_GBps_with_maps": None if not ms else round((B * W * H * 2 + W * H * 8) / (ms * 1e-3) / 1e9, 1)}
    ex.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
