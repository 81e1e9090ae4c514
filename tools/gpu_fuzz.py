#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: extractor (+ frame-to-frame matcher, stereo) against the CPU oracle for random image
sizes, feature counts, pyramid shapes, thresholds and image statistics.  Prints one line per case and a summary; exit code 1 on
any mismatch.  usage: tools/gpu_fuzz.py [ncases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orb_slam2_amd  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402
from oracle import orb_oracle as O  # noqa: E402


def image(rng, w, h, kind):
    if kind == "scene":
        return synth.frame(w, h, seed=int(rng.integers(1 << 30)))
    if kind == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == "lowcontrast":
        base = synth.frame(w, h, seed=int(rng.integers(1 << 30))).astype(np.float32)
        return np.clip(110 + (base - 128) * 0.12, 0, 255).astype(np.uint8)          # most cells need the minThFAST fallback
    if kind == "sparse":
        img = np.full((h, w), 90, np.uint8)
        for _ in range(int(rng.integers(3, 40))):
            x, y, s = int(rng.integers(0, w - 12)), int(rng.integers(0, h - 12)), int(rng.integers(3, 12))
            img[y:y + s, x:x + s] = int(rng.integers(0, 256))
        return img
    return (rng.integers(0, 2, (h // 4 + 1, w // 4 + 1), dtype=np.uint8).repeat(4, 0).repeat(4, 1)[:h, :w] * 200 + 20).astype(np.uint8)   # blocks


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for c in range(ncases):
        w, h = int(rng.integers(120, 900)), int(rng.integers(100, 600))
        n = int(rng.integers(50, 2500))
        sf = float(rng.choice([1.1, 1.2, 1.2, 1.3, 1.5, 2.0]))
        nl = int(rng.integers(1, 9))
        mn = int(rng.integers(2, 15)); ini = int(rng.integers(mn, 45))
        while min(w, h) / (sf ** (nl - 1)) < 60 and nl > 1:
            nl -= 1
        kind = str(rng.choice(["scene", "scene", "noise", "lowcontrast", "sparse", "blocks"]))
        imgs = [image(rng, w, h, kind) for _ in range(2)]
        tag = f"case {c}: {w}x{h} n={n} sf={sf} levels={nl} th={ini}/{mn} {kind}"
        try:
            ex = orb_slam2_amd.ORBextractor(n, sf, nl, ini, mn, w, h, max_batch=2)
        except orb_slam2_amd.OrbHipError as e:
            print(tag, "unsupported:", str(e)[:80]); continue
        ora = O.OracleExtractor(n, sf, nl, ini, mn)
        ks, ds = ex.extract_batch(imgs)
        ok = True
        K = []
        for f in range(2):
            ko, do = ora.extract(imgs[f])
            K.append((ko, do))
            ok &= ks[f].tobytes() == ko.tobytes() and np.array_equal(ds[f], do)
        m = orb_slam2_amd.ORBmatcher(0.9, True)
        n_g, m_g, p_g = m.SearchForInitialization(K[0][0], K[0][1], K[1][0], K[1][1], w, h, windowSize=int(rng.integers(5, 200)))
        ex.close()
        print(tag, "kp", [len(k) for k in ks], "OK" if ok else "MISMATCH", flush=True)
        bad += not ok
    print("fuzz:", ncases, "cases,", bad, "mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
