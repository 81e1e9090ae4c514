#!/usr/bin/env python3
"""BASELINE.json config 5 at its FULL size, parity included: a 10 000-key-frame descriptor DB (20 M rows x 32 B = 640 MB, seed 7), a
2000-descriptor query with planted near-duplicates; the GPU answer (single scan, and the pool's row shards merged on the host) against
the CPU oracle's brute force for ALL 2000 queries (4 x 10^10 Hamming distances, spread over the host cores).  One JSON line.
usage: db_full_parity.py [keyframes]"""
import json
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import orb_slam2_amd  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

NKF = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
PER, NQ = 2000, 2000
_db = _q = None


def _cpu_chunk(span):
    from oracle import orb_oracle as O          # the checker: never the thing measured
    lo, hi = span
    return O.bf_nn(_q[lo:hi], _db, fast=True)


def main():
    global _db, _q
    t0 = time.perf_counter()
    _db = synth.descriptor_db(NKF, PER, seed=7)
    _q = synth.descriptor_query(_db, NQ, seed=7)
    t_gen = time.perf_counter() - t0
    out = {"keyframes": NKF, "rows": len(_db), "db_MB": round(_db.nbytes / 1e6, 1), "queries": NQ, "generate_s": round(t_gen, 2)}
    # ---- GPU: one device, host buffers in (upload included), then resident shards through the pool
    t0 = time.perf_counter(); single = orb_slam2_amd.hamming_nn(_q, _db); out["single_call_incl_upload_s"] = round(time.perf_counter() - t0, 3)
    g = orb_slam2_amd.device_count()
    shards = max(g, 2)                           # a 1-GPU box still exercises the shard merge at full size (two shards on GPU 0)
    pool = orb_slam2_amd.MultiGpuExtractor([i % g for i in range(shards)], shards, 500, 1.2, 8, 20, 7, 320, 240)
    t0 = time.perf_counter(); pool.db_load(_db); out["pool_load_s"] = round(time.perf_counter() - t0, 3)
    pool.db_query(_q[:8])
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps):
        sharded = pool.db_query(_q)
    dt = (time.perf_counter() - t0) / reps
    out.update(devices=g, shards=shards, pool_query_ms=round(dt * 1e3, 2), pair_distances_per_s=round(len(_db) * NQ / dt / 1e9, 1) * 1e9)
    out["sharded_equals_single_scan"] = bool(all(np.array_equal(a, b) for a, b in zip(single, sharded)))
    pool.close()
    # ---- CPU oracle on every query, fork-shared DB
    ncores = os.cpu_count() or 1
    step = max(1, NQ // (ncores * 2))
    spans = [(lo, min(lo + step, NQ)) for lo in range(0, NQ, step)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(ncores) as p:
        parts = p.map(_cpu_chunk, spans)
    out["oracle_s"] = round(time.perf_counter() - t0, 2); out["oracle_processes"] = ncores
    want = [np.concatenate([pt[k] for pt in parts]) for k in range(3)]
    bad = [int((np.asarray(a, np.int64) != np.asarray(b, np.int64)).sum()) for a, b in zip(single, want)]
    out.update(queries_compared=NQ, mismatches_best_idx=bad[0], mismatches_best_dist=bad[1], mismatches_second_dist=bad[2],
               planted_found=int((single[1] <= 20).sum()), parity="bit-exact" if sum(bad) == 0 else "MISMATCH")
    print(json.dumps(out))
    sys.exit(0 if sum(bad) == 0 and out["sharded_equals_single_scan"] else 1)


if __name__ == "__main__":
    main()
