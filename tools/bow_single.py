#!/usr/bin/env python3
"""One frame's bag of words, many times (for kernel traces): python tools/bow_single.py [features] [repeats]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import orb_slam2_amd
from secondary_units import write_voc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
p = "/tmp/voc_k10_L6_bow_single.txt"
if not os.path.exists(p):
    write_voc(p, 10, 6)
v = orb_slam2_amd.ORBVocabulary(p)
d = np.random.default_rng(1).integers(0, 256, (n, 32), dtype=np.uint8)
for _ in range(20):
    v.transform(d, 4)
t0 = time.perf_counter()
for _ in range(reps):
    v.transform(d, 4)
print({"features": n, "transform_ms": round((time.perf_counter() - t0) / reps * 1e3, 4)})
