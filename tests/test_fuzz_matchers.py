"""Randomised conflict-heavy cases for the order-dependent matcher loops (the part k_match_select / k_proj_select resolve 64 queries
per step from precomputed records): many queries competing for the same features, equal descriptors (ties), stolen matches,
blocking and non-blocking map points, initially occupied features, every level-filter form, windows from a few pixels to the whole
image.  Results must equal the oracle's one-query-at-a-time restatement of ORBmatcher.cc:45-129, 405-520, 1328-1470.
A few seeds on the CPU emulation of the kernels, more on the GPU; `python tests/test_fuzz_matchers.py <library> [ncases]` by hand."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import orb_slam2_amd  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

KD = orb_slam2_amd.KEYPOINT_DTYPE


def _flip(rng, d, maxbits):
    for i in range(len(d)):
        for b in rng.integers(0, 256, int(rng.integers(0, maxbits + 1))):
            d[i, b >> 3] ^= 1 << (b & 7)


def initialization_case(rng, O, lib):
    """SearchForInitialization on synthetic key points: clustered targets, few distinct descriptors (ties, steals)"""
    w, h = int(rng.integers(100, 700)), int(rng.integers(100, 500))
    n2, n1 = int(rng.integers(1, 900)), int(rng.integers(1, 900))
    k2 = np.zeros(n2, KD)
    k2["x"], k2["y"] = rng.uniform(0, w, n2), rng.uniform(0, h, n2)
    k2["octave"], k2["angle"] = rng.choice([0, 0, 0, 1, 2], n2), rng.uniform(0, 360, n2)
    d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
    if rng.random() < 0.5:
        base = rng.integers(0, 256, (int(rng.integers(1, 8)), 32), dtype=np.uint8)
        d2 = base[rng.integers(0, len(base), n2)].copy()
        _flip(rng, d2, 3)
    src = rng.integers(0, n2, n1)
    k1 = np.zeros(n1, KD)
    k1["x"], k1["y"] = k2["x"][src] + rng.normal(0, 4, n1), k2["y"][src] + rng.normal(0, 4, n1)
    k1["octave"], k1["angle"] = rng.choice([0, 0, 0, 0, 1], n1), k2["angle"][src] + rng.choice([0, 0, 0, 90], n1)
    o = np.argsort(k1["octave"], kind="stable")
    k1, src = k1[o], src[o]
    d1 = d2[src].copy()
    _flip(rng, d1, int(rng.integers(0, 30)))
    win, ratio, ori = int(rng.choice([5, 20, 60, 200])), float(rng.choice([0.6, 0.9, 1.0])), bool(rng.integers(0, 2))
    m = orb_slam2_amd.ORBmatcher(ratio, ori, library=lib)
    n_g, m_g, p_g = m.SearchForInitialization(k1, d1, k2, d2, w, h, windowSize=win)
    n_o, m_o, p_o = O.search_for_initialization(k1, d1, k2, d2, w, h, window=win, nnratio=ratio, check_ori=ori)
    return n_g == n_o and np.array_equal(m_g, m_o) and p_g.tobytes() == p_o.tobytes()


def projection_case(rng, O, lib, frames):
    """SearchByProjection (both modes, with / without the orientation check): queries clustered on a few spots of a real frame"""
    w, h, sf, (kl, dl), (kc, dc) = frames
    nq = int(rng.integers(1, 400))
    idx = rng.integers(0, len(kl), nq)
    q = np.zeros(nq, O.PROJ_QUERY_DTYPE)
    centers = rng.integers(0, len(kc), max(1, int(rng.integers(1, 40))))
    c = centers[rng.integers(0, len(centers), nq)]
    q["x"], q["y"] = kc["x"][c] + rng.normal(0, 2, nq).astype(np.float32), kc["y"][c] + rng.normal(0, 2, nq).astype(np.float32)
    q["radius"] = rng.choice([3.0, 8.0, 20.0, 60.0], nq).astype(np.float32)
    q["ur"] = q["x"] - 10
    lv = int(rng.integers(0, 3))
    if lv == 0:
        q["min_level"], q["max_level"] = 0, -1
    elif lv == 1:
        q["min_level"], q["max_level"] = kl["octave"][idx] - 1, kl["octave"][idx] + 1
    else:
        q["min_level"], q["max_level"] = -1, -1
    q["blocks"] = rng.random(nq) < rng.choice([0.0, 0.5, 0.9, 1.0])
    q["angle"] = kl["angle"][idx]
    qd = dc[c].copy()
    _flip(rng, qd, int(rng.integers(0, 40)))
    blocked = (rng.random(len(kc)) < rng.choice([0.0, 0.2])).astype(np.uint8)
    ok = True
    for mode in (0, 1):
        for ori in (True, False):
            n_o, f_o = O.search_by_projection(kc, dc, w, h, q, qd, mode, nnratio=0.9, th_high=100, check_ori=ori, blocked=blocked)
            n_g, f_g = orb_slam2_amd.search_by_projection(kc, dc, w, h, q, qd, mode, nnratio=0.9, th_high=100, check_ori=ori, blocked=blocked, library=lib)
            ok &= n_g == n_o and np.array_equal(f_g, f_o)
    return ok


@pytest.fixture(scope="module")
def frames(oracle):
    w, h, n = 480, 360, 700
    seq = synth.sequence(w, h, 2, seed=41)
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    return w, h, ora.params()["scale_factors"], ora.extract(seq[0]), ora.extract(seq[1])


def _ncases(backend, few, many):
    return few if backend.endswith("_emu.so") else many


def test_initialization_conflicts(backend, oracle):
    for t in range(_ncases(backend, 8, 60)):
        assert initialization_case(np.random.default_rng(7000 + t), oracle, backend), f"case {t}"


def test_projection_conflicts(backend, oracle, frames):
    for t in range(_ncases(backend, 5, 40)):
        assert projection_case(np.random.default_rng(1000 + t), oracle, backend, frames), f"case {t}"


if __name__ == "__main__":
    from oracle import orb_oracle as O
    lib = sys.argv[1]
    ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    w, h, n = 480, 360, 700
    seq = synth.sequence(w, h, 2, seed=41)
    ora = O.OracleExtractor(n, 1.2, 8, 20, 7)
    fr = (w, h, ora.params()["scale_factors"], ora.extract(seq[0]), ora.extract(seq[1]))
    bad = 0
    for t in range(ncases):
        bad += not initialization_case(np.random.default_rng(7000 + t), O, lib)
        bad += not projection_case(np.random.default_rng(1000 + t), O, lib, fr)
    print(f"{2 * ncases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)
