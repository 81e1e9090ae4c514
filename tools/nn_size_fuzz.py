#!/usr/bin/env python3
"""Brute-force NN (BASELINE.json config 5) at random database sizes: the matrix-core scan splits the rows behind its head into chunks sized by the database
(orbhip_launch_hamming_nn: rows per workgroup = whole rounds of the chip's workgroup slots), so sizes around every boundary of that rule are compared with the CPU
scan (oracle bf_nn) - bit form and registered (expanded) form, tie-heavy and near-duplicate databases, best row / best distance / second-best distance of every query.
usage: nn_size_fuzz.py [ncases] [seed]      (needs a GPU; the oracle is the checker, as in tests/)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import orb_slam2_amd  # noqa: E402
from oracle import orb_oracle as O  # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t_start = time.time()
for case in range(ncases):
    kind = str(rng.choice(["random", "ties", "dups", "few_values"]))
    edge = [32768, 32768 + 2048, 65536, 32768 + 512 * 2048, 1 << 20]
    if rng.random() < 0.4:
        ndb = max(1, int(rng.choice(edge)) + int(rng.integers(-300, 300)))
    else:
        ndb = int(rng.integers(1, int(rng.choice([5000, 70000, 400000, 1500000]))))
    nq = int(rng.choice([1, 31, 64, 127, 512, 513, 2000, 2100]))
    if kind == "random":
        db = rng.integers(0, 256, (ndb, 32), dtype=np.uint8)
    elif kind == "ties":                                   # every row one of a handful of descriptors: the lowest row of the best one must win in every chunk
        base = rng.integers(0, 256, (int(rng.integers(1, 6)), 32), dtype=np.uint8)
        db = base[rng.integers(0, len(base), ndb)]
    elif kind == "dups":                                   # random rows with exact copies of some of them scattered behind
        db = rng.integers(0, 256, (ndb, 32), dtype=np.uint8)
        src = rng.integers(0, ndb, max(1, ndb // 50)); dst = rng.integers(0, ndb, len(src)); db[dst] = db[src]
    else:                                                  # bytes from two values: many equal distances
        db = rng.choice(np.array([0x0F, 0xF0], np.uint8), (ndb, 32))
    q = db[rng.integers(0, ndb, nq)].copy()
    flip = rng.random(nq) < 0.6
    q[flip, int(rng.integers(0, 32))] ^= np.uint8(1 << int(rng.integers(0, 8)))
    far = rng.random(nq) < 0.1
    q[far] = rng.integers(0, 256, (int(far.sum()), 32), dtype=np.uint8)
    want = O.bf_nn(q, db)
    got = orb_slam2_amd.hamming_nn(q, db)
    ok = all(np.array_equal(a, b) for a, b in zip(got, want))
    # the registered form of the same database
    d_db = orb_slam2_amd.DeviceBuffer.from_array(db); d_q = orb_slam2_amd.DeviceBuffer.from_array(q)
    dx = orb_slam2_amd.DeviceBuffer(orb_slam2_amd.nn_expanded_size(ndb))
    bi = orb_slam2_amd.DeviceBuffer(nq * 8); bd = orb_slam2_amd.DeviceBuffer(nq * 4); sd = orb_slam2_amd.DeviceBuffer(nq * 4)
    orb_slam2_amd.nn_expand_device(None, d_db.ptr, ndb, dx.ptr)
    orb_slam2_amd.hamming_nn_device_expanded(None, d_q.ptr, nq, d_db.ptr, dx.ptr, ndb, bi.ptr, bd.ptr, sd.ptr)
    orb_slam2_amd.device_synchronize()
    gx = (bi.download((nq,), np.int64), bd.download((nq,), np.int32), sd.download((nq,), np.int32))
    okx = all(np.array_equal(a, b) for a, b in zip(gx, want))
    del d_db, d_q, dx, bi, bd, sd
    bad += (not ok) + (not okx)
    print(f"case {case}: ndb={ndb} nq={nq} {kind}: bit form {'OK' if ok else 'MISMATCH'}, registered form {'OK' if okx else 'MISMATCH'}", flush=True)
print(f"nn size fuzz: {ncases} cases, {bad} mismatches, {time.time() - t_start:.0f} s")
sys.exit(1 if bad else 0)
