#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in boundary (orbhip_submit / orbhip_collect, chunked orbhip_extract_batch): host images in, host
key points + descriptors out.  Sweeps batch size, chunk size, pinned / pageable caller buffers and the synchronous entry; also the
single-frame latency.  Reported in DESIGN.md next to the HBM-resident bench value; never the bench `value`.
usage: host_io_rate.py [seconds per point]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import orb_slam2_amd  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

W, H = 1241, 376
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
base = np.stack([synth.frame(W, H, seed=s % 16, t=s // 16) for s in range(64)])


def rate(B, chunk, kind, mode):
    if chunk:
        os.environ["ORBHIP_HOST_CHUNK"] = str(chunk)
    else:
        os.environ.pop("ORBHIP_HOST_CHUNK", None)
    ex = orb_slam2_amd.ORBextractor(2000, 1.2, 8, 20, 7, W, H, max_batch=B)
    cap = ex.capacity
    if kind == "pinned":
        src = orb_slam2_amd.pinned_array((B, H, W), np.uint8)
        bufs = [(orb_slam2_amd.pinned_array((B, cap), orb_slam2_amd.KEYPOINT_DTYPE), orb_slam2_amd.pinned_array((B, cap, 32), np.uint8), np.zeros(B, np.int32)) for _ in range(2)]
    else:
        src = np.zeros((B, H, W), np.uint8)
        bufs = [(np.zeros((B, cap), orb_slam2_amd.KEYPOINT_DTYPE), np.zeros((B, cap, 32), np.uint8), np.zeros(B, np.int32)) for _ in range(2)]
    src[:] = np.resize(base, (B, H, W))
    imgs = [src[f] for f in range(B)]
    import ctypes as C
    ptrs = (C.c_void_p * B)(*[im.ctypes.data for im in imgs])
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    done = 0
    if mode == "sync":       # orbhip_extract_batch: chunks pipeline inside one synchronous call
        st = ex.L.orbhip_extract_batch(ex.h, B, ptrs, W, p(bufs[0][0]), p(bufs[0][1]), cap, p(bufs[0][2])); assert st == 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget:
            st = ex.L.orbhip_extract_batch(ex.h, B, ptrs, W, p(bufs[0][0]), p(bufs[0][1]), cap, p(bufs[0][2])); assert st == 0
            done += 1
    else:                    # two batches in flight
        ex.collect(ex.submit(imgs), out=bufs[0])
        t0 = time.perf_counter()
        pending = [ex.submit(imgs)]
        while time.perf_counter() - t0 < budget:
            pending.append(ex.submit(imgs))
            ex.collect(pending.pop(0), out=bufs[done % 2]); done += 1
        ex.collect(pending.pop(0), out=bufs[done % 2]); done += 1
    dt = time.perf_counter() - t0
    ex.close()
    return round(done * B / dt, 1)


rows = []
ONLY = os.environ.get("HOST_IO_ONLY")             # e.g. "256": just the configuration bench.py's host_io object quotes (batch 256, auto chunks, ring)
for B, chunk in ((64, 0), (64, 16), (128, 0), (128, 32), (256, 0), (256, 64)):
    if ONLY and (B != int(ONLY) or chunk):
        continue
    for kind in ("pinned", "pageable"):
        for mode in ("ring", "sync"):
            if ONLY and mode != "ring":
                continue
            rows.append({"batch": B, "chunk": chunk or "auto", "buffers": kind, "mode": mode, "frames_per_s": rate(B, chunk, kind, mode)})
            print(json.dumps(rows[-1]), flush=True)
os.environ.pop("ORBHIP_HOST_CHUNK", None)
one = orb_slam2_amd.ORBextractor(2000, 1.2, 8, 20, 7, W, H, max_batch=1)
one(base[0])
t1 = time.perf_counter()
for i in range(200):
    one(base[i % 64])
lat = (time.perf_counter() - t1) / 200
best = {k: max(r["frames_per_s"] for r in rows if r["buffers"] == k) for k in ("pinned", "pageable")}
print(json.dumps({"summary": True, "best_pinned_frames_per_s": best["pinned"], "best_pageable_frames_per_s": best["pageable"], "single_frame_latency_ms": round(lat * 1e3, 4),
                  "copy_threads": os.environ.get("ORBHIP_COPY_THREADS", "default"),
                  "note": "extract only (no match), 1241x376 / 2000 features, H2D of images and D2H of key points + descriptors included"}))
