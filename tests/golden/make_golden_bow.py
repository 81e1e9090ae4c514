"""Generates the BoW golden fixtures with the REFERENCE's own DBoW2 code (oracle/_ref/libdbow2_ref.so = Thirdparty/DBoW2 of
/root/reference compiled by `make -C oracle ref`; run from the repo root where /root/reference is mounted:
python tests/golden/make_golden_bow.py).

  voc_k6_L3_ref.txt        a vocabulary trained by TemplatedVocabulary::create (k-means++, DUtils::Random seed 20260921) on ORB
                           descriptors of 24 seeded synthetic frames and written by TemplatedVocabulary::saveToTextFile —
                           byte for byte what the reference wrote (TF_IDF weighting, L1_NORM scoring, like ORBvoc.txt)
  bow_k6_L3_ref.npz        reference outputs for the descriptors of tests/golden/extract_320x240_n300_seed21.npz and
                           match_..._seed21.npz: per-feature (word, weight, node), BowVector, FeatureVector at levelsup = 4
                           and 1, and the L1 score between the two frames.
These pin oracle/bow_oracle.cpp and the HIP path on machines where the reference is not mounted (the GPU box).
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dbow2_ref as R, orb_oracle as O  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
assert R.build(), "oracle/_ref/libdbow2_ref.so needs /root/reference"
ex = O.OracleExtractor(300, 1.2, 8, 20, 7)
train = [ex.extract(synth.frame(320, 240, seed=500 + s))[1] for s in range(24)]
voc = R.RefVocabulary()
voc.create(train, 6, 3, weighting=0, scoring=0, seed=20260921)
path = os.path.join(here, "voc_k6_L3_ref.txt")
voc.save_text(path)

# the reference's loader must not see the trailing newline its own writer emits (uninitialised reads)
with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as t:
    t.write(open(path).read().rstrip("\n"))
ref = R.RefVocabulary()
assert ref.load_text(t.name)
os.unlink(t.name)
g1 = np.load(os.path.join(here, "extract_320x240_n300_seed21.npz"))["descriptors"]
g2 = np.load(os.path.join(here, "match_320x240_n300_seed21.npz"))["d2"]
out = {"nwords": ref.size()}
for name, d in (("a", g1), ("b", g2)):
    for lu in (4, 1):
        w, v, nd = ref.transform_features(d, lu)
        bid, bval, fnode, foff, ffeat = ref.transform(d, lu)
        out.update({f"{name}{lu}_word": w, f"{name}{lu}_weight": v, f"{name}{lu}_node": nd, f"{name}{lu}_bow_id": bid, f"{name}{lu}_bow_val": bval,
                    f"{name}{lu}_fv_node": fnode, f"{name}{lu}_fv_off": foff, f"{name}{lu}_fv_feat": ffeat})
out["score_ab"] = ref.score(out["a4_bow_id"], out["a4_bow_val"], out["b4_bow_id"], out["b4_bow_val"])
np.savez_compressed(os.path.join(here, "bow_k6_L3_ref.npz"), **out)
print("golden bow:", ref.size(), "words,", len(out["a4_bow_id"]), "/", len(out["b4_bow_id"]), "bow entries, score", out["score_ab"])
