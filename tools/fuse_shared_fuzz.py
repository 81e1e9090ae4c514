"""GPU fuzz of orbhip_project_best_in_window_shared / _held (FuseBatch's entries) against the single-slot entry orbhip_project_best_in_window_bounds, which the
parity tests pin to the oracle: random numbers of slots (1..64, slots without key points among them), key points, points (none among them), skip masks, poses,
level tables, both roundings of R*x+t; every slot's answers on the points it was not told to skip, -1 / 256 on the others; then random held re-checks of random
slots (several in a row: a held call does not end the holding), and the holding ended by an ordinary call.     usage: python tools/fuse_shared_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orb_slam2_amd  # noqa: E402
from orb_slam2_amd import orbhip as H  # noqa: E402
import test_parity_projection_algebra as T  # noqa: E402

LIB = os.environ.get("ORBHIP_LIBRARY") or os.path.join(ROOT, "orb_slam2_amd", "liborbhip.so")


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    T._case.library = LIB
    rng = np.random.default_rng(seed)
    bad = 0; slots_total = held_total = 0
    for case in range(ncases):
        nlevels, scale = ((8, 1.2), (5, 1.5), (12, 1.1))[case % 3]
        gemm = int(rng.integers(0, 2))
        nsl = int(rng.choice([1, 2, 3, 5, 10, 33, 64])) if case % 7 else int(rng.integers(1, 65))
        npts = int(rng.choice([0, 1, 3, 64, 200, 777, 1200])) if case % 5 == 0 else int(rng.integers(1, 900))
        P0, pts, log_sf, bounds = T._case(rng, "fuse", gemm, nlevels, scale, npts=max(npts, 1))
        pts = pts[:npts]
        inv = (1.0 / (np.asarray(P0.scale_factors[:P0.nlevels], np.float32) ** 2)).astype(np.float32)
        frames, Ps = [], []
        for s in range(nsl):
            n = 0 if rng.random() < 0.08 else int(rng.integers(1, 1500))
            kps = np.zeros(n, orb_slam2_amd.KEYPOINT_DTYPE)
            kps["x"] = rng.uniform(0, T.W, n).astype(np.float32); kps["y"] = rng.uniform(0, T.HT, n).astype(np.float32)
            kps["octave"] = rng.integers(0, nlevels, n); kps["angle"] = rng.uniform(0, 360, n).astype(np.float32); kps["size"] = 31.0
            desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            ur = np.where(rng.random(n) < 0.6, kps["x"] - rng.uniform(1, 60, n), -1.0).astype(np.float32)
            frames.append((kps, desc, ur))
            Ps.append(P0 if s == 0 else T._case(np.random.default_rng(int(rng.integers(1 << 30))), "fuse", gemm, nlevels, scale, npts=1)[0])
        src = frames[int(rng.integers(0, nsl))][1]
        pdesc = (src[rng.integers(0, len(src), npts)] if len(src) else rng.integers(0, 256, (npts, 32), dtype=np.uint8)) ^ (rng.random((npts, 32)) < 0.04).astype(np.uint8)
        pdesc = np.ascontiguousarray(pdesc, np.uint8).reshape(npts, 32)
        skip = None if rng.random() < 0.15 else (rng.integers(0, 1 << 62, npts, dtype=np.uint64) & rng.integers(0, 1 << 62, npts, dtype=np.uint64)) | (rng.integers(0, 2, npts).astype(np.uint64) << np.uint64(63))
        sl = [dict(kps=f[0], desc=f[1], u_right=f[2], bounds=bounds, inv_level_sigma2=inv, proj=Ps[s]) for s, f in enumerate(frames)]
        outs = H.project_best_in_window_shared(sl, pts, pdesc, skip, True, library=LIB)
        held_checks = [(int(rng.integers(0, nsl)), rng.random(npts) < rng.uniform(0.05, 0.6)) for _ in range(int(rng.integers(0, 4)))] if npts else []
        held_out = []
        for s, m in held_checks:
            pd2 = pdesc[m] ^ (rng.random((int(m.sum()), 32)) < 0.03).astype(np.uint8)
            held_out.append((s, m, pd2) + tuple(H.project_best_in_window_held(s, Ps[s], pts[m], pd2, True, library=LIB)))
        ok = True; ordinary = 0
        for s in range(nsl):
            keep = np.ones(npts, bool) if skip is None else ((skip >> np.uint64(s)) & np.uint64(1)) == 0
            kps, desc, ur = frames[s]
            if len(kps) and keep.any():
                bi, bd, _ = H.project_best_in_window(kps, desc, bounds, inv, Ps[s], pts[keep], pdesc[keep], True, u_right=ur, library=LIB); ordinary += 1
            else:
                bi, bd = np.full(int(keep.sum()), -1, np.int32), np.full(int(keep.sum()), 256, np.int32)
            ok = ok and np.array_equal(outs[s][0][keep], bi) and np.array_equal(outs[s][1][keep], bd) and (outs[s][0][~keep] == -1).all() and (outs[s][1][~keep] == 256).all()
        slots_total += nsl
        for s, m, pd2, hb, hd in held_out:
            kps, desc, ur = frames[s]
            if len(kps) and m.any():
                bi, bd, _ = H.project_best_in_window(kps, desc, bounds, inv, Ps[s], pts[m], pd2, True, u_right=ur, library=LIB); ordinary += 1
            else:
                bi, bd = np.full(int(m.sum()), -1, np.int32), np.full(int(m.sum()), 256, np.int32)
            ok = ok and np.array_equal(hb, bi) and np.array_equal(hd, bd)
            held_total += 1
        if npts and ordinary:                                               # the ordinary calls above ended the holding
            st, _, _ = H.project_best_in_window_held(0, Ps[0], pts[:1], pdesc[:1], True, library=LIB, check=False)
            ok = ok and st == H.ERR_INVALID
        if not ok:
            bad += 1
        print(f"case {case}: slots {nsl} points {npts} gemm {gemm} levels {nlevels} skip {'none' if skip is None else 'masks'} held {len(held_out)} {'OK' if ok else 'MISMATCH'}")
    print(f"fuse shared / held fuzz: {ncases} cases, {slots_total} slots, {held_total} held re-checks, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
