"""Freezes what the reference's own src/ORBextractor.cc computes when it is built with the reference's OWN compiler flags
(-O3 -march=native: gcc contracts the pattern rotation of computeOrbDescriptor into FMAs; `make -C oracle ref_native`) on a frame where
that differs from the canonical -ffp-contract=off build.  The fixture travels to the GPU box, where that build cannot
(-march=native).  Run from the repo root where /root/reference is mounted: python tests/golden/make_golden_native.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orbextractor_ref as R  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

assert R.build() and R.build_native(), "needs /root/reference"
W, H, N, SF, NL = 640, 480, 800, 1.5, 5
nat, can = R.RefExtractor(N, SF, NL, 20, 7, native=True), R.RefExtractor(N, SF, NL, 20, 7)
for seed in range(200, 204):
    img = synth.frame(W, H, seed=seed)
    kn, dn = nat.extract(img)
    kc, dc = can.extract(img)
    diff = int(np.unpackbits(dn ^ dc).sum())
    if diff:
        np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "extract_native_flags_640x480.npz"), seed=seed, config=np.array([W, H, N, NL], np.int32),
                            scale=np.float32(SF), keypoints=kn, descriptors=dn, canonical_descriptors=dc)
        print(f"golden: seed {seed}, {len(kn)} key points, {diff} descriptor bit(s) differ between the native-flags and the canonical build")
        break
else:
    raise SystemExit("no frame with a contraction-dependent bit among the seeds tried")
