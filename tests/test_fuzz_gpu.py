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


def run_single_image_sequence(rng, O, library=None, nframes=3):
    """What ORB_SLAM2 does, at a random geometry: one stereo pair at a time on two max_batch = 1 contexts, each pair followed by a random subset of
    the follow-ups (ComputeStereoMatches, a motion-model search, a local-map search) on the frame still in HBM - the frame epilogues of
    orbhip_api.hip (row table / feature grid built behind the extraction once a context has seen the follow-up) at sizes nobody picked by hand."""
    w, h = int(rng.integers(200, 800)), int(rng.integers(160, 520))
    n = int(rng.integers(150, 1800))
    sf, nl = float(rng.choice([1.2, 1.2, 1.3])), int(rng.integers(4, 9))
    while min(w, h) / (sf ** (nl - 1)) < 60 and nl > 1:
        nl -= 1
    tag = f"sequence {w}x{h} n={n} sf={sf} levels={nl}"
    try:
        xl = orb_slam2_amd.ORBextractor(n, sf, nl, 20, 7, w, h, max_batch=1, library=library)
        xr = orb_slam2_amd.ORBextractor(n, sf, nl, 20, 7, w, h, max_batch=1, library=library)
    except orb_slam2_amd.OrbHipError as e:
        return tag + " unsupported: " + str(e)[:60], None
    eL, eR = O.OracleExtractor(n, sf, nl, 20, 7), O.OracleExtractor(n, sf, nl, 20, 7)
    scale = xl.GetScaleFactors()
    mbf, mb = float(np.float32(40.0 * w / 640)), float(np.float32(0.1))
    ok = True
    for t in range(nframes):
        sc = synth.scene(w, h, seed=int(rng.integers(1 << 30)))
        disp = int(rng.integers(2, 30))
        L = np.clip(np.rint(sc[32:32 + h, 32:32 + w]) + rng.integers(-6, 7, (h, w)), 0, 255).astype(np.uint8)
        R = np.clip(np.rint(sc[32:32 + h, 32 + disp:32 + disp + w]) + rng.integers(-6, 7, (h, w)), 0, 255).astype(np.uint8)
        kc, dc = xl(L)
        xr(R)
        ko, do = eL.extract(L); eR.extract(R)
        ok &= kc.tobytes() == ko.tobytes() and np.array_equal(dc, do)
        nk = len(kc)
        u = None
        if rng.random() < 0.7 and nk:
            ug, dg = xl.ComputeStereoMatches(xr, mbf, mb)
            uo, dpo = O.stereo_matches(eL, eR, mbf, mb)
            ok &= ug[0, :nk].tobytes() == uo.tobytes() and dg[0, :nk].tobytes() == dpo.tobytes()
            u = ug[0, :nk]
        for mode in [m for m in (1, 0) if rng.random() < 0.8 and nk]:
            q = np.zeros(nk, O.PROJ_QUERY_DTYPE)
            q["x"] = kc["x"] + rng.normal(0, 1.5, nk).astype(np.float32); q["y"] = kc["y"] + rng.normal(0, 1.5, nk).astype(np.float32)
            q["radius"] = (np.float32(rng.choice([3.0, 7.0, 15.0])) * scale[kc["octave"]]).astype(np.float32)
            q["ur"] = (q["x"] - 20).astype(np.float32) if u is None else np.where(u > 0, u + rng.normal(0, 2.0, nk), q["x"] - 20).astype(np.float32)
            q["min_level"], q["max_level"] = kc["octave"] - 1, kc["octave"] + (0 if mode == 0 else 1)
            q["blocks"] = rng.random(nk) < 0.8
            q["angle"] = kc["angle"]
            keep = rng.random(nk) < 0.8
            qd = dc.copy()
            flips = rng.integers(0, 256, (nk, 12))
            for i in np.nonzero(rng.random(nk) < 0.7)[0]:
                for b in flips[i, :int(rng.integers(1, 12))]:
                    qd[i, b >> 3] ^= 1 << (b & 7)
            blocked = (rng.random(nk) < 0.1).astype(np.uint8)
            n_o, f_o = O.search_by_projection(kc, dc, w, h, q[keep], qd[keep], mode, nnratio=0.9, th_high=100, check_ori=True, u_right=u, blocked=blocked)
            n_g, f_g = xl.search_by_projection(0, nk, q[keep], qd[keep], mode, nnratio=0.9, th_high=100, check_ori=True, use_u_right=u is not None, blocked=blocked)
            ok &= n_g == n_o and np.array_equal(f_g, f_o)
    xl.close(); xr.close()
    return tag, ok


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
    tag, ok = run_single_image_sequence(rng, oracle, emu_lib, nframes=3)
    assert ok is not False, tag


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [31, 32, 33, 34])
def test_single_image_sequences_gpu(gpu_lib, oracle, seed):
    rng = np.random.default_rng(seed)
    ran = 0
    for _ in range(3):
        tag, ok = run_single_image_sequence(rng, oracle, gpu_lib, nframes=4)
        assert ok is not False, tag
        ran += ok is True
    assert ran >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [7, 11, 23, 41, 43, 47, 53, 59])
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
