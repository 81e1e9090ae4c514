#!/usr/bin/env python3
"""Secondary measurement: Frame::ComputeBoW throughput on the GPU.

A synthetic vocabulary of ORBvoc.txt's shape (k = 10, L = 6: 1 111 110 nodes, 35 MB of node descriptors; random, untrained —
the real ORBvoc.txt blob is not in the reference checkout) is written in the reference's text format and loaded through
orbhip_voc_load_text; B frames of 1241x376 are extracted (2000 features) and transformed in place (orbhip_compute_bow,
levelsup 4).  Prints one JSON line.  (Parity of the transform is tests/test_bow.py's job; this tool only measures.)"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _digits(v, width):
    """[n] non-negative ints -> [n, width] ASCII digits with leading zeros (both `istream >> int` and strtol read them as decimal)"""
    out = np.empty((len(v), width), np.uint8)
    v = v.astype(np.int64)
    for c in range(width - 1, -1, -1):
        out[:, c] = 48 + v % 10
        v = v // 10
    return out


def write_vocabulary(path, k, L, seed=1):
    """Random k-ary tree of depth L in TemplatedVocabulary::saveToTextFile's format, fixed-width columns (vectorised writer)."""
    rng = np.random.default_rng(seed)
    with open(path, "wb") as f:
        f.write(f"{k} {L}  0 0\n".encode())
        first, count, nid = 0, 1, 0
        for depth in range(1, L + 1):
            n = count * k
            leaf = int(depth == L)
            row = np.full((n, 9 + 1 + 1 + 1 + 32 * 4 + 8 + 1), 32, np.uint8)
            row[:, 0:9] = _digits(np.repeat(np.arange(first, first + count), k), 9)
            row[:, 10] = 48 + leaf
            desc = rng.integers(0, 256, (n, 32))
            for b in range(32):
                row[:, 12 + 4 * b:12 + 4 * b + 3] = _digits(desc[:, b], 3)
            w = rng.integers(500, 9000, n) if leaf else np.zeros(n, np.int64)         # weight = w / 1000, written as dddd.ddd
            row[:, 140:144] = _digits(w // 1000, 4); row[:, 144] = 46; row[:, 145:148] = _digits(w % 1000, 3)
            row[:, -1] = 10
            f.write(row.tobytes())
            first, count = nid + 1, n
            nid += n
    return nid + 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--levels", type=int, default=6)
    args = ap.parse_args()
    import ctypes as C
    import orb_slam2_amd
    from orb_slam2_amd import synth

    W, H, N = 1241, 376, 2000
    path = os.path.join(tempfile.gettempdir(), f"voc_k{args.k}_L{args.levels}.txt")
    t0 = time.perf_counter()
    nnodes = write_vocabulary(path, args.k, args.levels)
    t_write = time.perf_counter() - t0
    t0 = time.perf_counter()
    voc = orb_slam2_amd.ORBVocabulary(path)
    t_load = time.perf_counter() - t0
    B = args.batch
    scenes = [synth.frame(W, H, seed=900 + s) for s in range(min(B, 8))]
    pitch = (W + 63) // 64 * 64
    host = np.zeros((B, H, pitch), np.uint8)
    for b in range(B):
        host[b, :, :W] = scenes[b % len(scenes)]
    dbuf = orb_slam2_amd.DeviceBuffer.from_array(host)

    class _P:
        value = dbuf.ptr
    dptr = _P()
    ex = orb_slam2_amd.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=B)
    ex.extract_device(dptr.value, B, H * pitch, pitch)
    voc.compute_bow(ex, B, 4)
    ex.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        voc.compute_bow(ex, B, 4)
    ex.sync()
    dt_bow = (time.perf_counter() - t0) / args.steps
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ex.extract_device(dptr.value, B, H * pitch, pitch)
        voc.compute_bow(ex, B, 4)
    ex.sync()
    dt_both = (time.perf_counter() - t0) / args.steps
    got = voc.fetch_bow(ex, 0)
    ks, ds = ex.fetch(1)
    t0 = time.perf_counter()
    again = voc.transform(ds[0], 4)                                  # host-pointer entry on the same descriptors: one frame, synchronous
    t_host = time.perf_counter() - t0
    ok = all(g.tobytes() == w.tobytes() for g, w in zip(got, again))
    print(json.dumps({"vocabulary": {"k": args.k, "L": args.levels, "nodes": nnodes, "words": voc.size(), "text_write_s": round(t_write, 1), "load_s": round(t_load, 2)},
                      "batch": B, "features_per_frame": len(ds[0]), "bow_entries_frame0": len(got[0]),
                      "compute_bow_ms_per_batch": round(dt_bow * 1e3, 3), "bow_frames_per_s": round(B / dt_bow, 1),
                      "extract_plus_bow_frames_per_s": round(B / dt_both, 1),
                      "host_entry_transform_ms_per_frame": round(t_host * 1e3, 3), "device_and_host_entries_agree": ok}))


if __name__ == "__main__":
    main()
