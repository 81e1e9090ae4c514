#!/usr/bin/env python3
"""Where does the host-buffer path lose against the link?  B = 256, three batches in flight, every combination of pinned / pageable frames in
and (pinned named at submit = DMA'd directly | pinned filled by collect's copy | pageable) results out, a few chunk sizes.  One JSON row each."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import orb_slam2_amd
from orb_slam2_amd import synth
W, H, B = 1241, 376, 256
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 1.2
base = np.stack([synth.frame(W, H, seed=s % 16, t=s // 16) for s in range(64)])

def rate(inp, outp, chunk):
    if chunk: os.environ["ORBHIP_HOST_CHUNK"] = str(chunk)
    else: os.environ.pop("ORBHIP_HOST_CHUNK", None)
    ex = orb_slam2_amd.ORBextractor(2000, 1.2, 8, 20, 7, W, H, max_batch=B, blur_round_mode=1)
    cap = ex.capacity
    pin = lambda shape, dt: orb_slam2_amd.pinned_array(shape, dt)
    src = pin((B, H, W), np.uint8) if inp == "pinned" else np.zeros((B, H, W), np.uint8)
    src[:] = np.resize(base, (B, H, W))
    mk = pin if outp.startswith("pinned") else (lambda shape, dt: np.zeros(shape, dt))
    bufs = [(mk((B, cap), orb_slam2_amd.KEYPOINT_DTYPE), mk((B, cap, 32), np.uint8), np.zeros(B, np.int32)) for _ in range(3)]
    named = outp == "pinned_named"
    sub = lambda i: ex.submit(src, out=bufs[i % 3]) if named else ex.submit(src)
    col = lambda t, i: ex.collect(t) if named else ex.collect(t, out=bufs[i % 3])
    col(sub(0), 0)
    s = d = 0
    t0 = time.perf_counter(); pending = []
    for _ in range(2): pending.append(sub(s)); s += 1
    while time.perf_counter() - t0 < budget:
        pending.append(sub(s)); s += 1
        col(pending.pop(0), d); d += 1
    while pending: col(pending.pop(0), d); d += 1
    dt = time.perf_counter() - t0
    ex.close()
    return round(d * B / dt, 1)

for chunk in (0, 32, 128):
    for inp in ("pinned", "pageable"):
        for outp in ("pinned_named", "pinned_copied", "pageable"):
            print(json.dumps({"chunk": chunk or "auto(64)", "frames_in": inp, "results_out": outp, "frames_per_s": rate(inp, outp, chunk)}), flush=True)
