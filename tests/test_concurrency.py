"""Three-thread re-entrancy of the binding, as ORB_SLAM2 uses it ("re-entrant from >= 3 threads").

In the reference, LocalMapping (LocalMapping.cc:215, 237-268 SearchForTriangulation per neighbour; :483-514 Fuse per target; ProcessNewKeyFrame's
KeyFrame::ComputeBoW) and LoopClosing (LoopClosing.cc:239-375 SearchByBoW(KF, KF) / SearchBySim3 / SearchByProjection(KF, Scw, ...); :589-599
Fuse(KF, Scw, ...)) call ORBmatcher on their own threads WHILE Tracking (Tracking.cc:867-928, 1143-1193) extracts and searches on the same device.
oracle/orbslam_ref_wrap.cpp::orbslam_ref_concurrency runs exactly that on three std::threads around the reference's own Frame.cc / ORBmatcher.cc:

    T  the stereo front-end loop (stereo constructor with its two extractor threads, ComputeStereoMatches, both SearchByProjection overloads),
       round after round until the other two are done;
    L  SearchForTriangulation + Fuse (stereo chi-square branch) + ComputeBoW on key-frame pairs;
    C  SearchByBoW(KF, KF) + SearchBySim3 + SearchByProjection(KF, Scw) + Fuse(KF, Scw) + ComputeBoW, the vocabulary shared with L;

with random start offsets and pauses.  Every call's result (return value + whole output array) is hashed, iteration by iteration, and must equal
  (a) the same calls made one after another on one thread by the same build, and
  (b) the all-reference build (liborbslam_ref.so: the reference's own extractor and search loops on the host).
Here: the drop-in build on the CPU emulation of the kernels (kernel launches take turns there, the host side around them — per-thread scratch, the
vocabulary's lock, the classes' state — does not).  `-m gpu`: liborbslam_dropin_full_gpu.so on the MI355X at KITTI's shape, >= 200 iterations per
thread, ORBHIP_POISON set so that a buffer one thread reads before it wrote it shows.  tools/sanitize_concurrency.sh runs the emulation case under
AddressSanitizer and ThreadSanitizer."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from orb_slam2_amd import synth  # noqa: E402

SMALL = dict(w=400, h=300, n=500, fx=231.5, fy=231.5, cx=200.0, cy=150.0, bf=25.5, th_depth=35.0)
KITTI = dict(w=1241, h=376, n=2000, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448, th_depth=35.0)       # Examples/Stereo/KITTI00-02.yaml
VOC = os.path.join(HERE, "golden", "voc_k6_L3_ref.txt")


def build_cases(S, lib, cfg, seq, voc_path, ncases=4, seed=0):
    """Key frames (stereo Frames of the sequence's frames 0 and 2, one private set per thread) and ncases parameterisations of every call."""
    from oracle import orb_oracle as O
    lefts, rights = seq[0], seq[1]
    cam = dict(nfeatures=cfg["n"], fx=cfg["fx"], fy=cfg["fy"], cx=cfg["cx"], cy=cfg["cy"], bf=cfg["bf"], th_depth=cfg["th_depth"])
    frames = {}
    for who in "LC":
        frames[who] = [S.RefFrame(lefts[k], rights[k], library=lib, **cam) for k in (0, 2)]
    ov = O.OracleVocabulary(VOC)
    w, h = cfg["w"], cfg["h"]
    calls = {"L": S.ConcCalls(), "C": S.ConcCalls()}
    keep = frames["L"] + frames["C"]
    for who in "LC":
        A, B = frames[who]
        ka, da, kb, db = A.keys_un, A.desc, B.keys_un, B.desc
        na, nb = len(ka), len(kb)
        rng = np.random.default_rng(1000 * seed + (1 if who == "L" else 2))
        fx, fy, cx, cy = (np.float32(cfg[k]) for k in ("fx", "fy", "cx", "cy"))

        def world(k, dx, dy):                       # points whose projection (identity pose) lands near key points k shifted by the far plane's image motion
            px = (k["x"] + dx + rng.normal(0, 1.2, len(k))).astype(np.float32); py = (k["y"] + dy + rng.normal(0, 1.2, len(k))).astype(np.float32)
            px[:4] = -2.0; px[4:7] = w + 1.0; py[7:9] = h
            return ((px - cx) / fx).astype(np.float32), ((py - cy) / fy).astype(np.float32), np.clip(k["octave"] + rng.integers(0, 2, len(k)), 0, 7).astype(np.int32)

        for c in range(ncases):
            levelsup = 1 + c % 3
            fva, fvb = ov.transform(da, levelsup)[2:], ov.transform(db, levelsup)[2:]
            X, Y, lev = world(ka, -4.0, -2.0)
            bad = (rng.random(na) < 0.05).astype(np.uint8)
            Z = np.ones(na, np.float32)
            if who == "L":
                Fm = np.array([[0, -1e-3, 1.0 / 300], [1e-3, 0, -3.0 / 300], [-1.0 / 300, 3.0 / 300, 0]], np.float32) + rng.normal(0, 1e-5, (3, 3)).astype(np.float32)
                calls[who].triangulation(A, rng.random(na) < 0.3, fva, B, rng.random(nb) < 0.3, fvb, Fm, np.array((0.3, 0.1, 1.0) if c % 2 else (2.0, 1.0, 4.0), np.float32),
                                         only_stereo=bool(c % 2), check_ori=c % 3 != 2)
                calls[who].fuse(B, rng.choice([0, 0, 1, 2], nb), X, Y, Z, lev, rng.integers(0, 4, na), bad, da, th=3.0 + 2.0 * (c % 2))
                calls[who].compute_bow(A if c % 2 else B, voc_path)
            else:
                calls[who].bow(1, A, rng.random(na) < 0.75, rng.random(na) < 0.07, fva, B, rng.random(nb) < 0.85, rng.random(nb) < 0.07, fvb, nnratio=0.75 + 0.05 * c, check_ori=c % 2 == 0)
                X2, Y2, lev2 = world(kb, 4.0, 2.0)
                calls[who].sim3(A, rng.random(na) < 0.8, X, Y, Z, lev, da, B, rng.random(nb) < 0.8, X2, Y2, np.ones(nb, np.float32), lev2, db, th=7.5)
                calls[who].projection_kf(B, rng.random(nb) < 0.2, X, Y, Z, lev, bad, da, th=10)
                calls[who].fuse_sim3(B, rng.choice([0, 0, 1], nb), X, Y, Z, lev, bad, da, th=4.0)
                calls[who].compute_bow(B if c % 2 else A, voc_path)
    return calls, keep


def run(S, lib, cfg, nframes, threaded, iters, t_rounds, seed, voc_path):
    seq = synth.stereo_sequence(cfg["w"], cfg["h"], nframes, cfg["fx"], cfg["bf"], seed=3)
    S.RefFrame._geometry = None
    S.RefFrame._geometry_other.clear()
    calls, keep = build_cases(S, lib, cfg, seq, voc_path)
    try:
        return S.concurrency(lib, threaded, iters, t_rounds, seed, seq[0], seq[1], seq[2], seq[3], cfg["n"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["bf"], cfg["th_depth"],
                             calls["L"], calls["C"], kf_every=3)
    finally:
        for f in keep:
            f.close()
        S.RefFrame._geometry = None


def voc_without_final_newline(tmp_path):
    """(the reference's loader must not see the file's final newline)"""
    p = tmp_path / "voc_no_final_newline.txt"
    p.write_text(open(VOC).read().rstrip("\n"))
    return str(p)


def check(S, D, cfg, nframes, iters, t_rounds, voc_path, seeds):
    hT0, hL0, hC0, _, _ = run(S, S.lib(), cfg, nframes, False, iters, 1, 0, voc_path)                      # (b) the all-reference build, one thread
    assert len(set(hL0.tolist())) >= min(iters, 8) and len(set(hC0.tolist())) >= min(iters, 12) and len(set(hT0.tolist())) == nframes      # the hashes tell the cases apart
    hT1, hL1, hC1, _, _ = run(S, D, cfg, nframes, False, iters, 1, 0, voc_path)                            # (a) the drop-in build, one thread
    assert np.array_equal(hT1, hT0) and np.array_equal(hL1, hL0) and np.array_equal(hC1, hC0), "the drop-in build differs from the reference before any thread is involved"
    total_rounds = 0
    for seed in seeds:
        hT, hL, hC, rounds, differing = run(S, D, cfg, nframes, True, iters, t_rounds, seed, voc_path)
        assert rounds >= t_rounds, f"a thread failed (rc {rounds}; see stderr)"
        bad = dict(T=int((hT != hT0).sum()), L=int((hL != hL0).sum()), C=int((hC != hC0).sum()), T_rounds=differing)
        assert not any(bad.values()), f"seed {seed}: results under three threads differ from the serial run: {bad}"
        total_rounds += rounds
    return total_rounds


def test_three_threads_on_the_emulation(emu_lib, tmp_path):
    from oracle import orbslam_ref as S
    if not (S.build() and S.build_dropin()):
        pytest.skip("reference sources not mounted")
    iters = int(os.environ.get("ORBHIP_CONCURRENCY_ITERS", "16"))          # tools/sanitize_concurrency.sh raises it
    check(S, S.dropin_full_lib(), SMALL, nframes=4, iters=iters, t_rounds=1, voc_path=voc_without_final_newline(tmp_path), seeds=(1,))


@pytest.mark.gpu
def test_three_threads_on_the_device(gpu_lib, tmp_path, monkeypatch):
    """KITTI's shape, 24-frame loop on thread T, 400 matcher calls on each of L and C, three runs with different start offsets; allocations poisoned."""
    from oracle import orbslam_ref as S
    monkeypatch.setenv("ORBHIP_POISON", "165")
    if not (S.build() and S.build_dropin_gpu()):
        pytest.fail("oracle/_ref/liborbslam_dropin_full_gpu.so did not travel with the repository (build it with `make -C oracle dropin_gpu` where /root/reference is mounted)")
    rounds = check(S, S.dropin_gpu_lib(full=True), KITTI, nframes=24, iters=400, t_rounds=2, voc_path=voc_without_final_newline(tmp_path), seeds=(1, 2, 3))
    print(f"\n[concurrency] KITTI shape: 3 runs x (T: {rounds} loop rounds of 24 stereo frames in total, L: 400 calls, C: 400 calls) on three threads, every result equal to the serial run and to the reference")
