#!/usr/bin/env python3
"""BASELINE.json config 5: brute-force 256-bit Hamming of a 2000-descriptor query frame against a keyframe descriptor DB
resident in HBM (10 000 keyframes x 2000 descriptors = 640 MB).  Reports pair distances/s, the DB streaming rate, and the roofline of
whichever kernel ran: the matrix-core scan (default from 32 K rows on: 512 i8 operations per pair against the dense i8 MFMA peak, ~5 POP/s
spec, 4.4 measured - MI355X_MICROARCH.md) or, with ORBHIP_NN=valu, the popcount kernel against the integer-VALU issue peak
(8 v_xor_b32 @2 cycles + 8 v_bcnt_u32_b32 @4 cycles per 64 pairs and SIMD).  No framework: device memory through the library."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import orb_slam2_amd  # noqa: E402

NKF = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
NQ, PER = 2000, 2000
rng = np.random.default_rng(7)
db_h = rng.integers(0, 256, (NKF * PER, 32), dtype=np.uint8)
q_h = db_h[rng.integers(0, NKF * PER, NQ)].copy()
q_h[::2, 0] ^= 0x5A                                            # half the queries are near-duplicates, half exact
db = orb_slam2_amd.DeviceBuffer.from_array(db_h)
q = orb_slam2_amd.DeviceBuffer.from_array(q_h)
bi = orb_slam2_amd.DeviceBuffer(NQ * 8); bd = orb_slam2_amd.DeviceBuffer(NQ * 4); sd = orb_slam2_amd.DeviceBuffer(NQ * 4)
orb_slam2_amd.device_synchronize()


EXPANDED = os.environ.get("DB_EXPANDED", "0") == "1"           # the database expanded once (orbhip_nn_expand_device), tiles staged by LDS-DMA
dx = None
if EXPANDED:
    dx = orb_slam2_amd.DeviceBuffer(orb_slam2_amd.nn_expanded_size(NKF * PER))
    t_e = time.perf_counter()
    orb_slam2_amd.nn_expand_device(None, db.ptr, NKF * PER, dx.ptr)
    orb_slam2_amd.device_synchronize()
    expand_ms = (time.perf_counter() - t_e) * 1e3


def run():
    if EXPANDED:
        orb_slam2_amd.hamming_nn_device_expanded(None, q.ptr, NQ, db.ptr, dx.ptr, NKF * PER, bi.ptr, bd.ptr, sd.ptr)
    else:
        orb_slam2_amd.hamming_nn_device(None, q.ptr, NQ, db.ptr, NKF * PER, bi.ptr, bd.ptr, sd.ptr)
    orb_slam2_amd.device_synchronize()


run()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    run()
dt = (time.perf_counter() - t0) / reps
pairs = NQ * NKF * PER
peak_pairs = 1024 * 64 / (8 * 2 + 8 * 4) * 2.4e9
assert os.environ.get("ORBHIP_NN_ABLATE", "0") != "0" or os.environ.get("ORBHIP_NN_BLOCK_VAR", "0") != "0" or int((bd.download((NQ,), np.int32) <= 4).sum()) == NQ      # (an ablated scan - measurement only - answers wrongly)
valu = os.environ.get("ORBHIP_NN") == "valu"
out = {"kernel": "k_hamming_nn (popcount)" if valu else ("k_hamming_nn_mfma" if os.environ.get("ORBHIP_NN") == "i8" else "k_hamming_nn_fp4"), "db_keyframes": NKF, "db_bytes": NKF * PER * 32, "query_ms": round(dt * 1e3, 2),
       "pair_distances_per_s": float(f"{pairs / dt:.4g}"), "db_stream_GBps": round(NKF * PER * 32 / dt / 1e9, 1), "queries_per_s_vs_full_db": round(NQ / dt, 1),
       "library": os.environ.get("ORBHIP_LIBRARY", "in-tree")}
if valu:
    out["frac_of_int_valu_issue_peak"] = round(pairs / dt / peak_pairs, 3)
else:
    out["matrix_TOPs"] = round(pairs * 512 / dt / 1e12, 1)
    if out["kernel"] == "k_hamming_nn_fp4":
        out["frac_of_fp4_mfma_rate_measured_9100_TOPs"] = round(pairs * 512 / dt / 9.1e15, 3); out["frac_of_fp4_mfma_spec_10000_TOPs"] = round(pairs * 512 / dt / 10e15, 3)
    else:
        out["frac_of_i8_mfma_peak_measured_4400_TOPs"] = round(pairs * 512 / dt / 4.4e15, 3); out["frac_of_i8_mfma_spec_5000_TOPs"] = round(pairs * 512 / dt / 5e15, 3)
if EXPANDED:
    out["expanded_db_bytes"] = dx.nbytes; out["expand_once_ms"] = round(expand_ms, 2)
    # the same queries through the ordinary scan: the answers must be identical
    want = [b.download((NQ,), t) for b, t in ((bi, np.int64), (bd, np.int32), (sd, np.int32))]
    orb_slam2_amd.hamming_nn_device(None, q.ptr, NQ, db.ptr, NKF * PER, bi.ptr, bd.ptr, sd.ptr); orb_slam2_amd.device_synchronize()
    out["equals_unexpanded_scan"] = all(np.array_equal(w, b.download((NQ,), t)) for w, (b, t) in zip(want, ((bi, np.int64), (bd, np.int32), (sd, np.int32))))
print(json.dumps(out))
