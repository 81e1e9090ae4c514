#!/usr/bin/env python3
"""Every secondary unit of SURVEY.md §8 with the reference beside it (bench.py's `matcher_calls`, `config5`, `config4` objects).

matcher_calls  one call of every ORBmatcher member / Frame member the drop-in forwards, at KITTI's shape (1241x376, 2000 features), through the
               reference's OWN Frame.cc / ORBmatcher.cc callers: `ref_ms` = the all-reference build on one host thread (oracle/_ref/liborbslam_ref_fast.so,
               -O3), `gpu_ms` = the same member in the build integration/apply_dropin.py emits, on the MI355X (liborbslam_dropin_full_gpu.so).  The time is
               the member's alone (a timer inside the wrapper, orbslam_ref_last_call_ms), medians; `parity` = every output of the two builds equal.
               BoW-guided members run on FeatureVectors of a vocabulary of ORBvoc.txt's shape (k = 10, L = 6, random: the real file is not in the checkout).
config5        BASELINE.json configs[4]: 2000 query descriptors against a 10 000-key-frame (20 M row, 640 MB) descriptor DB resident in HBM, best and
               second best per query; the CPU beside it = the oracle's scan (the matcher's idiom, ORBmatcher.cc:102-114 / :1647-1663) over the SAME full DB
               on every host thread at once; parity = all 2000 answers equal.
config4        BASELINE.json configs[3]: an 8-camera rig at 1920x1080 / 4000 features through the one-process pool (orbhip_pool_*; the cameras share the
               one GPU of this box), host buffers in and out; the CPU beside it = the reference's ORBextractor on the same frames, one process per thread.

Prints one JSON object.  `--small`: tiny shapes on the CPU emulation of the kernels (the tool's own test, tests/test_secondary_units.py)."""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

KITTI = dict(w=1241, h=376, n=2000, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448, th_depth=35.0)
SMALL = dict(w=400, h=300, n=500, fx=231.5, fy=231.5, cx=200.0, cy=150.0, bf=25.5, th_depth=35.0)


def host_cpu():
    """physical cores / hardware threads / model of this box (north_star: "core count stated")"""
    threads = os.cpu_count() or 1
    cores, model, seen = None, None, set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        cores = len(seen) or None
    except OSError:
        pass
    return {"physical_cores": cores, "hardware_threads": threads, "model": model}


def _median_call(S, lib, fn, reps):
    out, ms, lms = None, [], []
    for _ in range(reps):
        out = fn()
        ms.append(S.last_call_ms(lib)); lms.append(S.last_call_lib_ms(lib))
    return out, float(np.median(ms)), float(np.median(lms))


def _same(a, b):
    if isinstance(a, (tuple, list)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return a.shape == b.shape and a.tobytes() == b.tobytes()
    return a == b


def matcher_calls(cfg, ref_lib, gpu_lib, voc_path, reps=7, ref_reps=3):
    from orb_slam2_amd import synth
    from oracle import orbslam_ref as S
    w, h = cfg["w"], cfg["h"]
    L, R, _, _ = synth.stereo_sequence(w, h, 3, cfg["fx"], cfg["bf"], seed=5)
    cam = dict(nfeatures=cfg["n"], fx=cfg["fx"], fy=cfg["fy"], cx=cfg["cx"], cy=cfg["cy"], bf=cfg["bf"], th_depth=cfg["th_depth"])
    out = {}
    sides = {}
    for name, lib in (("ref", ref_lib), ("gpu", gpu_lib)):
        S.RefFrame._geometry = None
        S.RefFrame._geometry_other.clear()
        m0, m1 = S.RefFrame(L[0], library=lib, **cam), S.RefFrame(L[2], library=lib, **cam)
        s0 = S.RefFrame(L[0], R[0], library=lib, **cam)
        s1 = S.RefFrame(L[2], R[2], library=lib, **cam)                    # the last frame the rig made: resident on the device in the drop-in build
        sides[name] = (lib, m0, m1, s0, s1)
    fx, fy, cx, cy = (np.float32(cfg[k]) for k in ("fx", "fy", "cx", "cy"))
    _, m0, m1, s0, s1 = sides["ref"]
    rng = np.random.default_rng(11)
    ka, da, kb = s0.keys_un, s0.desc, s1.keys_un
    na, nb = len(ka), len(kb)
    dx, dy = -4.0, -2.0                                                     # the far plane's image motion over two frames
    px = (ka["x"] + dx + rng.normal(0, 1.2, na)).astype(np.float32); py = (ka["y"] + dy + rng.normal(0, 1.2, na)).astype(np.float32)
    X, Y, Z = ((px - cx) / fx).astype(np.float32), ((py - cy) / fy).astype(np.float32), np.ones(na, np.float32)
    lev = np.clip(ka["octave"] + rng.integers(0, 2, na), 0, 7).astype(np.int32)
    bad = (rng.random(na) < 0.05).astype(np.uint8)
    has = (rng.random(na) < 0.85).astype(np.uint8)
    inview = (rng.random(na) < 0.85).astype(np.uint8); nobs = (rng.random(na) < 0.9).astype(np.int32)
    vc = np.where(rng.random(na) < 0.5, 0.9995, 0.9).astype(np.float32)
    state = rng.choice([0, 0, 0, 1, 2], nb).astype(np.uint8)
    kfstate = rng.choice([0, 0, 1, 2], nb).astype(np.uint8)
    nobs4 = rng.integers(0, 4, na).astype(np.int32)
    px2 = (kb["x"] - dx + rng.normal(0, 1.2, nb)).astype(np.float32); py2 = (kb["y"] - dy + rng.normal(0, 1.2, nb)).astype(np.float32)
    X2, Y2 = ((px2 - cx) / fx).astype(np.float32), ((py2 - cy) / fy).astype(np.float32)
    lev2 = np.clip(kb["octave"] + rng.integers(0, 2, nb), 0, 7).astype(np.int32)
    hs1 = (rng.random(na) < 0.8).astype(np.uint8); hs2 = (rng.random(nb) < 0.8).astype(np.uint8)
    matched = (rng.random(nb) < 0.2).astype(np.uint8)
    found = (rng.random(na) < 0.1).astype(np.uint8)
    Fm = np.array([[0, -1e-3, 1.0 / 300], [1e-3, 0, -3.0 / 300], [-1.0 / 300, 3.0 / 300, 0]], np.float32)
    t2w = np.array((0.3, 0.1, 1.0), np.float32)
    hmp1 = (rng.random(na) < 0.3).astype(np.uint8); hmp2 = (rng.random(nb) < 0.3).astype(np.uint8)
    v1 = (rng.random(na) < 0.75).astype(np.uint8); b1 = (rng.random(na) < 0.07).astype(np.uint8)
    v2 = (rng.random(nb) < 0.85).astype(np.uint8); b2 = (rng.random(nb) < 0.07).astype(np.uint8)
    res = {"ref": {}, "gpu": {}}
    for name in ("ref", "gpu"):
        lib, m0, m1, s0, s1 = sides[name]
        r = reps if name == "gpu" else ref_reps
        call = lambda fn: _median_call(S, lib, fn, r)
        o = res[name]
        def stereo_ctor():
            f = S.RefFrame(L[2], R[2], library=lib, **cam)
            r = (f.keys.copy(), f.desc.copy(), f.u_right.copy(), f.depth.copy())
            f.close()
            return r
        o["Frame::Frame(imLeft, imRight, ...) incl. ComputeStereoMatches (S1)"] = call(stereo_ctor)
        s1b = S.RefFrame(L[2], R[2], library=lib, **cam); s1.close(); s1 = s1b; sides[name] = (lib, m0, m1, s0, s1)      # the last frame the rig made again
        o["SearchByProjection(Current, Last) (M2)"] = call(lambda: S.search_by_projection_last(s1, s0, has, X, Y, Z, da, cur_state=state, th=7.0, mono=False, nnratio=0.9, check_ori=True))
        o["SearchByProjection(Frame, MapPoints) (M3)"] = call(lambda: S.search_by_projection_points(s1, px, py, (px - 9.0).astype(np.float32), lev, vc, inview, bad, nobs, da, state, th=3.0, nnratio=0.8))
        o["SearchForInitialization (M1)"] = call(lambda: S.search_for_initialization(m0, m1, window=100, nnratio=0.9, check_ori=True))
        bow0 = call(lambda: s0.compute_bow(voc_path))
        bow1 = call(lambda: s1.compute_bow(voc_path))
        o["Frame::ComputeBoW"] = bow1
        fv0, fv1 = bow0[0][2:], bow1[0][2:]
        o["SearchByBoW(KF, Frame) (M4)"] = call(lambda: S.search_by_bow(0, s0, v1, b1, fv0, s1, None, None, fv1, nnratio=0.7, check_ori=True))
        o["SearchByBoW(KF, KF) (M4)"] = call(lambda: S.search_by_bow(1, s0, v1, b1, fv0, s1, v2, b2, fv1, nnratio=0.75, check_ori=True))
        o["SearchForTriangulation (M4)"] = call(lambda: S.search_for_triangulation(s0, hmp1, fv0, s1, hmp2, fv1, Fm, t2w, only_stereo=False, check_ori=True))
        o["Fuse(KF, MapPoints) (M5)"] = call(lambda: S.fuse(s1, kfstate, X, Y, Z, lev, nobs4, bad, da, th=3.0))
        o["Fuse(KF, Scw, MapPoints) (M5)"] = call(lambda: S.fuse_sim3(s1, kfstate, X, Y, Z, lev, bad, da, th=4.0))
        o["SearchBySim3 (M5)"] = call(lambda: S.search_by_sim3(s0, hs1, X, Y, Z, lev, da, s1, hs2, X2, Y2, np.ones(nb, np.float32), lev2, s1.desc, th=7.5))
        o["SearchByProjection(KF, Scw, MapPoints) (M5)"] = call(lambda: S.search_by_projection_kf(s1, matched, X, Y, Z, lev, bad, da, th=10))
        o["SearchByProjection(Frame, KF) (M5, relocalisation)"] = call(lambda: S.search_by_projection_reloc(s1, s0, has, X, Y, Z, lev, bad, found, da, state, th=10.0, orb_dist=100, nnratio=0.9, check_ori=True))
    shape = f"{w}x{h} stereo, {na} / {nb} features, ~{int(has.sum())} map points per call"
    for k in res["ref"]:
        (ro, rms, _), (go, gms, glib) = res["ref"][k], res["gpu"][k]
        out[k] = {"gpu_ms": round(gms, 4), "gpu_ms_inside_the_library": round(glib, 4), "ref_ms": round(rms, 4), "ref_over_gpu": round(rms / gms, 2) if gms > 0 else None, "parity": bool(_same(ro, go))}
    nodes = len(res["ref"]["Frame::ComputeBoW"][0][2])
    # ---- the C ABI alone: ten per-call entries vs one batch entry on the same arrays (no reference code around them)
    abi = None
    try:
        import orb_slam2_amd as A
        _, _, _, g0, g1 = sides["gpu"]
        k0, d0, k1, d1 = g0.keys_un, g0.desc, g1.keys_un, g1.desc
        fva, fvb = res["gpu"]["Frame::ComputeBoW"][0][2:], res["gpu"]["Frame::ComputeBoW"][0][2:]
        fva = g0.compute_bow(voc_path)[2:]; fvb = g1.compute_bow(voc_path)[2:]
        sf = (1.2 ** np.arange(8)).astype(np.float32)
        def t(fn, reps=7):
            fn(); ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            return float(np.median(ts)) * 1e3
        NB = 10
        z0, z1 = np.zeros(len(k0), np.uint8), np.zeros(len(k1), np.uint8)
        kf1 = dict(desc=d0, kps=k0, has_mp=hmp1, stereo=z0, fv=fva, scale_factors=sf, level_sigma2=sf * sf)
        nbs = [dict(kf=dict(desc=d1, kps=k1, has_mp=hmp2, stereo=z1, fv=fvb, scale_factors=sf, level_sigma2=sf * sf), F12=Fm, ex=620.0, ey=190.0) for _ in range(NB)]
        tri_call = t(lambda: [A.search_for_triangulation(d0, k0, hmp1, z0, fva, d1, k1, hmp2, z1, fvb, Fm, 620.0, 190.0, sf, sf * sf, check_ori=False) for _ in range(NB)])
        tri_batch = t(lambda: A.search_for_triangulation_batch(kf1, nbs, check_ori=False))
        frame_side = dict(desc=d1, angle=k1["angle"], valid=None, fv=fvb)
        cands = [dict(desc=d0, angle=k0["angle"], valid=v1, fv=fva) for _ in range(5)]
        bow_call = t(lambda: [A.search_by_bow(0, d0, k0["angle"], v1, fva, d1, k1["angle"], None, fvb, nnratio=0.75) for _ in range(5)])
        bow_batch = t(lambda: A.search_by_bow_batch(0, [(c, frame_side) for c in cands], nnratio=0.75))
        q = np.zeros(na, A.BEST_QUERY_DTYPE); q["x"], q["y"], q["radius"], q["ur"], q["level"] = px, py, (3.0 * sf[lev]).astype(np.float32), px - 9.0, lev
        inv = (1.0 / (sf * sf)).astype(np.float32)
        slot = dict(kps=k1, desc=d1, u_right=g1.u_right, bounds=(0.0, 0.0, float(w), float(h)), inv_level_sigma2=inv, queries=q, qdesc=d0)
        win_call = t(lambda: [A.search_best_in_window(k1, d1, w, h, inv, q, d0, True, u_right=g1.u_right) for _ in range(NB)])
        win_batch = t(lambda: A.search_best_in_window_batch([slot] * NB, True))
        abi = {"orbhip_search_for_triangulation x 10 vs _batch(10)": {"per_call_ms": round(tri_call, 4), "batch_ms": round(tri_batch, 4), "ratio": round(tri_call / tri_batch, 2)},
               "orbhip_search_by_bow x 5 vs _batch(5)": {"per_call_ms": round(bow_call, 4), "batch_ms": round(bow_batch, 4), "ratio": round(bow_call / bow_batch, 2)},
               "orbhip_search_best_in_window x 10 vs _batch(10)": {"per_call_ms": round(win_call, 4), "batch_ms": round(win_batch, 4), "ratio": round(win_call / win_batch, 2)},
               "note": "host arrays in, host arrays out, through the ctypes mirror (its per-call marshalling included on both sides)"}
    except Exception as e:                                            # noqa: BLE001
        abi = {"error": str(e)[:300]}
    for name in sides:
        for f in sides[name][1:]:
            f.close()
    # ---- LocalMapping's two loops over ten neighbours: the reference's per-call loops vs the single device passes of include/ORBmatcherBatch.h
    loops = None
    try:
        nn = 10
        Ls, Rs, _, _ = synth.stereo_sequence(w, h, nn + 1, cfg["fx"], cfg["bf"], seed=9)
        rngl = np.random.default_rng(2)
        F12 = np.stack([Fm + rngl.normal(0, 1e-5, (3, 3)).astype(np.float32) for _ in range(nn)])
        b = cfg["bf"] / cfg["fx"]
        t2 = np.stack([np.array([-0.25 * b * (i + 1), -0.125 * b * (i + 1), 0.0], np.float32) for i in range(nn)])
        lres = {}
        for name in ("ref", "gpu"):
            lib = sides[name][0]
            S.RefFrame._geometry = None; S.RefFrame._geometry_other.clear()
            fr = [S.RefFrame(Ls[i], Rs[i], library=lib, **cam) for i in range(nn + 1)]
            runs = [S.local_mapping_loops(fr, F12, t2, voc_path, point_depth=cfg["bf"] / 8.0) for _ in range(3 if name == "gpu" else 1)]
            lres[name] = (runs[-1], float(np.median([r[3][0] for r in runs])), float(np.median([r[3][1] for r in runs])))
            for f in fr:
                f.close()
        (rr, rt, rf), (gr, gt, gf) = lres["ref"], lres["gpu"]
        same = all(np.array_equal(a, b2) for a, b2 in zip(rr[0], gr[0])) and np.array_equal(rr[1], gr[1]) and rr[2] == gr[2]
        loops = {"neighbours": nn, "CreateNewMapPoints: SearchForTriangulation over all neighbours": {"gpu_batch_ms": round(gt, 4), "ref_loop_ms": round(rt, 4), "pairs": int(sum(len(p) for p in rr[0]))},
                 "SearchInNeighbors: Fuse over all targets": {"gpu_batch_ms": round(gf, 4), "ref_loop_ms": round(rf, 4), "fused": int(rr[2])}, "parity": bool(same),
                 "note": "the reference's loops (map points created / replaced between key frames) vs SearchForTriangulationBatch + TriangulationPairs and FuseBatch (include/ORBmatcherBatch.h), whole loop incl. the reference's host code"}
    except Exception as e:                                            # noqa: BLE001
        loops = {"error": str(e)[:300]}
    S.RefFrame._geometry = None
    return {"shape": shape, "feature_vector_nodes": nodes, "timer": "the member call alone (inside the wrapper), median of %d (gpu) / %d (ref) calls; ref = one host thread; gpu_ms_inside_the_library = the part of gpu_ms spent in liborbhip's entry points (orbhip_thread_api_ms), the rest is the reference's own host code of the member" % (reps, ref_reps),
            "members": out, "all_parity": all(v["parity"] for v in out.values()), "back_end_loops": loops, "batch_entries": abi}


# ---- config 5
_DB = None


def _bf_worker(args):
    lo, hi, q = args
    from oracle import orb_oracle as O
    bi, bd, sd = O.bf_nn(q, _DB[lo:hi], fast=True)
    return lo, bi, bd, sd


def config5(nkf, ncores, per=2000, nq=2000):
    global _DB
    import orb_slam2_amd
    from oracle import orb_oracle as O
    O.build()
    rng = np.random.default_rng(7)
    ndb = nkf * per
    _DB = rng.integers(0, 256, (ndb, 32), dtype=np.uint8)
    q_h = _DB[rng.integers(0, ndb, nq)].copy()
    q_h[::2, 0] ^= 0x5A                                           # half the queries are near-duplicates of a row, half exact copies
    # the CPU beside it first (fork before this process touches the device): every host thread scans its own row range of the SAME DB
    nw = min(ncores, max(ndb // 1024, 1))
    edges = np.linspace(0, ndb, nw + 1).astype(np.int64)
    ctx = mp.get_context("fork")
    with ctx.Pool(nw) as pool:
        pool.map(_bf_worker, [(0, 64, q_h[:4])] * nw)            # workers up, library loaded
        t0 = time.perf_counter()
        parts = pool.map(_bf_worker, [(int(edges[i]), int(edges[i + 1]), q_h) for i in range(nw)])
        ref_s = time.perf_counter() - t0
    bi = np.full(nq, -1, np.int64); bd = np.full(nq, 257, np.int64); sd = np.full(nq, 257, np.int64)
    for lo, pbi, pbd, psd in sorted(parts, key=lambda p: p[0]):   # the matcher's rule: strict '<', the lowest row wins ties; second = second smallest of the union
        pbd = pbd.astype(np.int64); psd = psd.astype(np.int64)
        better = pbd < bd
        sd = np.where(better, np.minimum(bd, psd), np.minimum(sd, pbd))
        bi = np.where(better, pbi.astype(np.int64) + lo, bi); bd = np.where(better, pbd, bd)
    db = orb_slam2_amd.DeviceBuffer.from_array(_DB)
    q = orb_slam2_amd.DeviceBuffer.from_array(q_h)
    gbi = orb_slam2_amd.DeviceBuffer(nq * 8); gbd = orb_slam2_amd.DeviceBuffer(nq * 4); gsd = orb_slam2_amd.DeviceBuffer(nq * 4)
    orb_slam2_amd.device_synchronize()

    # The database of a map is registered once and asked at every relocalisation / loop query: it is kept EXPANDED beside its bit form (orbhip_nn_expand_device,
    # what orbhip_pool_db_load does for its shards) and the query multiplies that form directly.  The bit form (a caller without the 4 x memory) is timed beside it.
    dx = orb_slam2_amd.DeviceBuffer(orb_slam2_amd.nn_expanded_size(ndb))
    t0 = time.perf_counter()
    orb_slam2_amd.nn_expand_device(None, db.ptr, ndb, dx.ptr); orb_slam2_amd.device_synchronize()
    expand_ms = (time.perf_counter() - t0) * 1e3

    def run(expanded):
        if expanded:
            orb_slam2_amd.hamming_nn_device_expanded(None, q.ptr, nq, db.ptr, dx.ptr, ndb, gbi.ptr, gbd.ptr, gsd.ptr)
        else:
            orb_slam2_amd.hamming_nn_device(None, q.ptr, nq, db.ptr, ndb, gbi.ptr, gbd.ptr, gsd.ptr)
        orb_slam2_amd.device_synchronize()
    times, parity = {}, True
    for expanded in (False, True):
        run(expanded)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); run(expanded); ts.append(time.perf_counter() - t0)
        times[expanded] = float(np.median(ts))
        g_bi, g_bd, g_sd = gbi.download((nq,), np.int64), gbd.download((nq,), np.int32), gsd.download((nq,), np.int32)
        parity = parity and bool(np.array_equal(g_bi, bi) and np.array_equal(g_bd, bd) and np.array_equal(g_sd, sd))
    dt = times[True]
    pairs = nq * ndb
    nn_form = os.environ.get("ORBHIP_NN", "")
    xbytes = dx.nbytes
    for b in (db, dx, q, gbi, gbd, gsd):
        b.free()
    _DB = None
    return {"workload": f"{nq} query descriptors x {nkf} key frames x {per} descriptors ({ndb * 32 / 1e6:.0f} MB DB resident in HBM), best + second best per query",
            "kernel": {"valu": "k_hamming_nn (popcount)", "i8": "k_hamming_nn_mfma (i8 matrix cores)"}.get(nn_form, "k_hamming_nn_fp4b (FP4 matrix cores, %s)" % (nn_form or "hand-ordered superstep of eight tiles, seeded two-pass scan, bounds shared between workgroups")),
            "database_form": "expanded once at registration (orbhip_nn_expand_device: 128 B per row beside the 32), tiles staged by 16-byte LDS-DMA",
            "query_ms": round(dt * 1e3, 3), "query_ms_bit_form": round(times[False] * 1e3, 3), "expand_once_ms": round(expand_ms, 2), "expanded_db_bytes": xbytes, "pair_distances_per_s": float(f"{pairs / dt:.4g}"), "db_stream_GBps": round(ndb * 32 / dt / 1e9, 1),
            "matrix_TOPs": round(pairs * 512 / dt / 1e12, 1), "matrix_peak": "FP4 (v_mfma_scale_f32_32x32x64_f8f6f4): 10 PF spec, 9.1 PF measured; i8: 5 PF spec, 4.4 measured (MI355X_MICROARCH.md)",
            "frac_of_fp4_mfma_peak": round(pairs * 512 / dt / 10e15, 3), "frac_of_fp4_mfma_rate_measured_9100_TOPs": round(pairs * 512 / dt / 9.1e15, 3),
            "frac_of_i8_mfma_peak": round(pairs * 512 / dt / 5e15, 3),
            "ref_ms_all_threads": round(ref_s * 1e3, 1), "ref_threads": nw, "ref_kind": "port: oracle bf_nn (-O3 -march=x86-64-v3), the whole DB split by rows over all host threads, partial answers merged with the matcher's tie rule",
            "ref_over_gpu": round(ref_s / dt, 1), "parity_sample": {"queries_compared": nq, "rows": ndb, "equal": parity, "compared": "best row, best distance, second-best distance of every query, GPU (expanded form and bit form) vs the CPU scan of the whole DB"}}


# ---- config 4
def _rig_cpu_worker(args):
    """the reference's own Frame constructor (ORBextractor::operator() + UndistortKeyPoints + the feature grid) on one camera's frames, like bench.py's cpu_baseline"""
    frames, budget_s, n, fast = args
    from oracle import orbslam_ref as S
    S.use_fast_build(fast)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        S.RefFrame(frames[done % len(frames)], nfeatures=n).close()
        done += 1
    return done, time.perf_counter() - t0


def config4(w, h, n, ncam, ncores, rounds=12, cpu_budget_s=5.0, fast=True):
    import orb_slam2_amd
    from orb_slam2_amd import synth
    from oracle import orb_oracle as O
    frames = [synth.frame(w, h, seed=700 + c) for c in range(ncam)]
    nw = ncores
    ctx = mp.get_context("fork")
    with ctx.Pool(nw) as pool:                                   # before this process creates its pool of device contexts
        res = pool.map(_rig_cpu_worker, [([frames[i % ncam]], cpu_budget_s, n, fast) for i in range(nw)])
    cpu_rate = sum(r[0] / r[1] for r in res)
    one = res[0][0] / res[0][1]
    g = max(orb_slam2_amd.device_count(), 1)
    devices = [c % g for c in range(min(ncam, 8))]
    pool = orb_slam2_amd.MultiGpuExtractor(devices, ncam, n, 1.2, 8, 20, 7, w, h)
    src = orb_slam2_amd.pinned_array((ncam, h, w), np.uint8)
    for c in range(ncam):
        src[c] = frames[c]
    cap = pool.capacity
    bufs = [(orb_slam2_amd.pinned_array((ncam, cap), orb_slam2_amd.KEYPOINT_DTYPE), orb_slam2_amd.pinned_array((ncam, cap, 32), np.uint8), np.zeros(ncam, np.int32)) for _ in range(2)]
    imgs = [src[c] for c in range(ncam)]
    for i in range(2):
        pool.collect(pool.submit(imgs), out=bufs[i & 1])
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        pending = [pool.submit(imgs)]
        for i in range(1, rounds):
            pending.append(pool.submit(imgs))
            pool.collect(pending.pop(0), out=bufs[(i - 1) & 1])
        nout = pool.collect(pending.pop(0), out=bufs[(rounds - 1) & 1])
        ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    last = (rounds - 1) & 1
    ora = O.OracleExtractor(n, 1.2, 8, 20, 7)
    mism = 0
    for c in (0, ncam - 1):
        ko, do = ora.extract(frames[c]); k = int(nout[c])
        mism += 0 if (k == len(ko) and bufs[last][0][c, :k].tobytes() == ko.tobytes() and np.array_equal(bufs[last][1][c, :k], do)) else 1
    pool.close()
    return {"workload": f"{ncam}-camera rig, {w}x{h}, {n} features per camera, host buffers in / out, through the one-process pool ({len(devices)} worker contexts on {g} GPU(s) of this box)",
            "rig_frames_per_s": round(rounds / dt, 1), "camera_frames_per_s": round(rounds * ncam / dt, 1), "ms_per_rig_frame": round(dt / rounds * 1e3, 3),
            "keypoints_per_camera": int(np.mean(nout)),
            "ref_camera_frames_per_s_all_threads": round(cpu_rate, 1), "ref_camera_frames_per_s_one_thread": round(one, 2), "ref_threads": nw,
            "ref_kind": "reference: its own Frame constructor (ORBextractor::operator()) from oracle/_ref/liborbslam_ref_fast.so on the same frames, one process per hardware thread",
            "gpu_over_ref_all_threads": round(rounds * ncam / dt / cpu_rate, 1) if cpu_rate > 0 else None,
            "parity_sample": {"cameras_compared": 2, "mismatches": mism, "compared": "key points + descriptors of the last round vs oracle/"}}


def write_voc(path, k, L):
    from bow_rate import write_vocabulary
    n = write_vocabulary(path, k, L)
    with open(path, "rb+") as f:                                  # the reference's loader must not see the file's final newline
        f.seek(-1, os.SEEK_END)
        if f.read(1) == b"\n":
            f.seek(-1, os.SEEK_END); f.truncate()
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small", action="store_true", help="tiny shapes on the CPU emulation of the kernels (the tool's own test)")
    ap.add_argument("--only", default="", help="comma-separated subset of matcher_calls,config5,config4")
    ap.add_argument("--blur-round-mode", type=int, default=1)
    args = ap.parse_args()
    os.environ["ORB_REF_BLUR_ROUND_MODE"] = str(args.blur_round_mode)
    want = set(filter(None, args.only.split(","))) or {"matcher_calls", "config5", "config4"}
    from oracle import orbslam_ref as S
    cpu = host_cpu()
    out = {"host_cpu": cpu}
    threads = cpu["hardware_threads"]
    if args.small:
        import orb_slam2_amd
        emu = os.path.join(ROOT, "tests", "emu", "liborbhip_emu.so")
        os.environ["ORBHIP_LIBRARY"] = emu
    # the forked CPU legs first: a process that has touched the device must not fork workers
    if "config5" in want:
        out["config5"] = config5(20 if args.small else 10000, min(threads, 8) if args.small else threads, per=200 if args.small else 2000, nq=64 if args.small else 2000)
    if "config4" in want:
        out["config4"] = config4(*((480, 360, 600, 3, 2, 3, 1.0, False) if args.small else (1920, 1080, 4000, 8, threads)))
    if "matcher_calls" in want:
        voc = os.path.join(tempfile.gettempdir(), "orbhip_voc_k4_L3.txt" if args.small else "orbhip_voc_k10_L6_nonl.txt")
        write_voc(voc, 4 if args.small else 10, 3 if args.small else 6)
        ref_lib = S._bind(C.CDLL(S.PATH if args.small else S.FAST_PATH))
        gpu_lib = S.dropin_full_lib() if args.small else S.dropin_gpu_lib(full=True)
        out["matcher_calls"] = matcher_calls(SMALL if args.small else KITTI, ref_lib, gpu_lib, voc, reps=2 if args.small else 7, ref_reps=1 if args.small else 3)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
