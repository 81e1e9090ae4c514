"""Generates the committed golden fixtures from the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

The fixtures pin the oracle against drift and give the GPU tests vectors that need neither the oracle nor the reference.
Where /root/reference is mounted the script also runs the reference's OWN src/ORBextractor.cc, src/Frame.cc and
src/ORBmatcher.cc (oracle/_ref builds; OpenCV replaced by a type stand-in + the oracle's four image primitives) on the same
inputs and refuses to write fixtures that differ from them.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orb_oracle as O  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
seq = synth.sequence(320, 240, 2, seed=21)
assert np.array_equal(seq[0], synth.frame(320, 240, seed=21))
ex = O.OracleExtractor(300, 1.2, 8, 20, 7)
k1, d1 = ex.extract(seq[0])
k2, d2 = ex.extract(seq[1])
np.savez_compressed(os.path.join(here, "extract_320x240_n300_seed21.npz"), image=seq[0], keypoints=k1, descriptors=d1)
n, m12, prev = O.search_for_initialization(k1, d1, k2, d2, 320, 240, window=100, nnratio=0.9)
np.savez_compressed(os.path.join(here, "match_320x240_n300_seed21.npz"), image2=seq[1], k1=k1, d1=d1, k2=k2, d2=d2,
                    nmatches=n, matches12=m12, prev=prev)
from oracle import orbslam_ref as S  # noqa: E402
if S.build():
    F1, F2 = S.RefFrame(seq[0], nfeatures=300), S.RefFrame(seq[1], nfeatures=300)
    assert F1.keys.tobytes() == k1.tobytes() and np.array_equal(F1.desc, d1) and F2.keys.tobytes() == k2.tobytes() and np.array_equal(F2.desc, d2)
    n_r, m_r, p_r = S.search_for_initialization(F1, F2, window=100, nnratio=0.9)
    assert n_r == n and np.array_equal(m_r, m12) and p_r.tobytes() == prev.tobytes()
    print("golden: identical to the reference's own Frame / ORBextractor / ORBmatcher code")
print("golden:", len(k1), len(k2), "keypoints,", n, "matches")
