"""Randomised parity sweep on the GPU: extractor + frame-to-frame matcher against the CPU oracle for random image sizes, feature
counts, pyramid shapes, thresholds and image statistics (textured scenes, white noise, low contrast = minThFAST fallback in most
cells, sparse shapes, blocks).  `python tests/test_fuzz_gpu.py [ncases] [seed]` runs a longer sweep by hand
(profiles/r01_gpu_fuzz_40cases.txt is such a run)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import orb_slam2_amd  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402


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


def run_case(rng, O, library=None):
    """-> (description, ok or None if the configuration is outside the supported envelope)"""
    w, h = int(rng.integers(120, 900)), int(rng.integers(100, 600))
    n = int(rng.integers(50, 2500))
    sf = float(rng.choice([1.1, 1.2, 1.2, 1.3, 1.5, 2.0]))
    nl = int(rng.integers(1, 9))
    mn = int(rng.integers(2, 15)); ini = int(rng.integers(mn, 45))
    while min(w, h) / (sf ** (nl - 1)) < 60 and nl > 1:
        nl -= 1
    kind = str(rng.choice(["scene", "scene", "noise", "lowcontrast", "sparse", "blocks"]))
    imgs = [image(rng, w, h, kind) for _ in range(2)]
    tag = f"{w}x{h} n={n} sf={sf} levels={nl} th={ini}/{mn} {kind}"
    try:
        ex = orb_slam2_amd.ORBextractor(n, sf, nl, ini, mn, w, h, max_batch=2, library=library)
    except orb_slam2_amd.OrbHipError as e:
        return tag + " unsupported: " + str(e)[:60], None
    ora = O.OracleExtractor(n, sf, nl, ini, mn)
    ks, ds = ex.extract_batch(imgs)
    ok = True
    K = []
    for f in range(2):
        ko, do = ora.extract(imgs[f])
        K.append((ko, do))
        ok &= ks[f].tobytes() == ko.tobytes() and np.array_equal(ds[f], do)
    win = int(rng.integers(5, 200))
    m = orb_slam2_amd.ORBmatcher(0.9, True, library=library)
    n_g, m_g, p_g = m.SearchForInitialization(K[0][0], K[0][1], K[1][0], K[1][1], w, h, windowSize=win)
    n_o, m_o, p_o = O.search_for_initialization(K[0][0], K[0][1], K[1][0], K[1][1], w, h, window=win, nnratio=0.9)
    ok &= n_g == n_o and np.array_equal(m_g, m_o) and p_g.tobytes() == p_o.tobytes()
    ex.close()
    return tag + f" kp {[len(k) for k in ks]} matches {n_o}", ok


def test_random_configurations_emulation(emu_lib, oracle, monkeypatch):
    """The same sweep on the CPU emulation of the kernel sources, every workgroup starting on garbage LDS (HIPEMU_POISON_LDS): a read of
    LDS the kernel never wrote changes results here as it would on the device (tools/fuzz_emu.py runs hundreds of cases this way)."""
    monkeypatch.setenv("HIPEMU_POISON_LDS", "1")
    rng = np.random.default_rng(5)
    ran = 0
    for _ in range(6):
        tag, ok = run_case(rng, oracle, emu_lib)
        assert ok is not False, tag
        ran += ok is True
    assert ran >= 3


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [7, 11, 23])
def test_random_configurations_gpu(gpu_lib, oracle, seed):
    rng = np.random.default_rng(seed)
    ran = 0
    for _ in range(8):
        tag, ok = run_case(rng, oracle, gpu_lib)
        assert ok is not False, tag
        ran += ok is True
    assert ran >= 4


if __name__ == "__main__":
    from oracle import orb_oracle as O
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for c in range(ncases):
        tag, ok = run_case(rng, O)
        print(f"case {c}: {tag}", "OK" if ok else ("-" if ok is None else "MISMATCH"), flush=True)
        bad += ok is False
    print("fuzz:", ncases, "cases,", bad, "mismatches")
    sys.exit(1 if bad else 0)
