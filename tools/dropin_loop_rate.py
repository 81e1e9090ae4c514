#!/usr/bin/env python3
"""The front-end loop (tests/test_dropin_loop.py) as a measurement: ms per stereo frame of Tracking's per-frame sequence through the
reference's own Frame.cc / ORBmatcher.cc, once with the reference's CPU extractor / stereo matcher / projection matchers
(oracle/_ref/liborbslam_ref_fast.so, the -O3 build) and once with this repository's on the MI355X (liborbslam_dropin_full_gpu.so =
integration/apply_dropin.py applied to the same sources, linked to liborbhip.so).  Prints one JSON object per shape; bench.py's
`dropin_loop` calls measure() for the KITTI shape.  The first pass of each build captures every frame and compares them bit for bit
(parity); the timed passes run without the capture copies."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

SHAPES = {
    "euroc": dict(w=752, h=480, n=1200, fx=435.2047, fy=435.2047, cx=367.4517, cy=252.2008, bf=47.9064, th_depth=35.0),       # Examples/Stereo/EuRoC.yaml
    "kitti": dict(w=1241, h=376, n=2000, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448, th_depth=35.0),       # Examples/Stereo/KITTI00-02.yaml
}


def measure(shape="kitti", nframes=24, passes=5, ref_passes=1, seed=5, kf_every=5):
    from orb_slam2_amd import synth
    from oracle import orbslam_ref as S
    cfg = SHAPES[shape]
    if not (os.path.exists(S.FAST_PATH) and os.path.exists(S.DROPIN_FULL_GPU_PATH)):
        return {"shape": shape, "skipped": "oracle/_ref/liborbslam_ref_fast.so / liborbslam_dropin_full_gpu.so did not travel with the repository"}
    ref_lib = S._bind(C.CDLL(S.FAST_PATH))
    gpu_lib = S.dropin_gpu_lib(full=True)
    L, R, T, P = synth.stereo_sequence(cfg["w"], cfg["h"], nframes, cfg["fx"], cfg["bf"], seed=seed)
    args = (L, R, T, P, cfg["n"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["bf"], cfg["th_depth"])
    ref = S.tracking_loop(*args, kf_every=kf_every, library=ref_lib)
    got = S.tracking_loop(*args, kf_every=kf_every, library=gpu_lib)
    mismatched = [k for k, (a, b) in enumerate(zip(ref, got)) if not a.same(b)]
    keys = ("ms", "ms_ctor", "ms_motion", "ms_local", "ms_frustum", "ms_copy")

    def timed(lib, n):
        runs = [S.tracking_loop(*args, kf_every=kf_every, capture=False, library=lib) for _ in range(n)]
        per_frame = {k: [np.median([getattr(r[f], k) for r in runs]) for f in range(1, nframes)] for k in keys}       # frame 0 is the initialisation
        return {k: round(float(np.median(v)), 4) for k, v in per_frame.items()}, round(float(np.max(per_frame["ms"])), 4)
    g, gmax = timed(gpu_lib, passes)
    r, rmax = timed(ref_lib, ref_passes)
    return {
        "shape": f"{cfg['w']}x{cfg['h']} stereo, {cfg['n']} features, 8 levels, {nframes} frames, new points every {kf_every} frames",
        "what": "per stereo pair: reference Frame constructor (two extractor threads, ComputeStereoMatches, UndistortKeyPoints, grid) + SearchByProjection(Current, Last, 7) "
                "+ isInFrustum over the local map + SearchByProjection(Frame, MapPoints, 1) + Frame copy; pose from the sequence instead of the optimiser",
        "ms_per_frame_gpu": g["ms"], "ms_per_frame_ref": r["ms"], "speedup": round(r["ms"] / g["ms"], 1),
        "gpu_parts_ms": {"frame_constructor": g["ms_ctor"], "motion_model_search": g["ms_motion"], "local_map_search": g["ms_local"], "of_which_isInFrustum_loop": g["ms_frustum"], "frame_copy": g["ms_copy"]},
        "ref_parts_ms": {"frame_constructor": r["ms_ctor"], "motion_model_search": r["ms_motion"], "local_map_search": r["ms_local"], "of_which_isInFrustum_loop": r["ms_frustum"], "frame_copy": r["ms_copy"]},
        "worst_frame_ms_gpu": gmax, "worst_frame_ms_ref": rmax,
        "ref_build": "oracle/_ref/liborbslam_ref_fast.so (-O3 -march=x86-64-v3, the reference's two extractor threads)", "passes_gpu": passes, "passes_ref": ref_passes,
        "parity": {"frames_compared": nframes, "frames_mismatched": len(mismatched), "compared": "keys, mvKeysUn, descriptors, mvuRight, mvDepth, map point per feature after each matcher, counters"},
        "median_features": int(np.median([f.N for f in ref])), "median_motion_matches": int(np.median([f.n_motion for f in ref[1:]])),
        "median_local_points_in_frustum": int(np.median([f.n_to_match for f in ref[1:]])), "local_map_points_at_end": ref[-1].n_local_points,
    }


TUM1 = dict(w=640, h=480, n=1000, fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, bf=40.0, th_depth=40.0)       # Examples/Monocular/TUM1.yaml (BASELINE.json configs[0])
TUM1_DIST = (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)


def measure_sequence(sensor, nframes=24, passes=5, ref_passes=1, seed=3, kf_every=4, lost_every=5):
    """tests/test_sequences.py as a measurement: the monocular / RGB-D matcher sequences with relocalisation at configs[0]'s shape (640x480, 1000 features)."""
    import tempfile
    from orb_slam2_amd import synth
    from oracle import orbslam_ref as S
    cfg = TUM1
    if not (os.path.exists(S.FAST_PATH) and os.path.exists(S.DROPIN_FULL_GPU_PATH)):
        return {"shape": sensor, "skipped": "oracle/_ref/liborbslam_ref_fast.so / liborbslam_dropin_full_gpu.so did not travel with the repository"}
    ref_lib = S._bind(C.CDLL(S.FAST_PATH))
    gpu_lib = S.dropin_gpu_lib(full=True)
    # a vocabulary of ORBvoc.txt's shape (k = 10, L = 6, random; the real file is not in the checkout): FeatureVectors at levelsup 4 then have ~100 nodes, as in
    # ORB_SLAM2 - the 216-word golden vocabulary of the tests puts a whole frame into ONE node, which makes SearchByBoW a 1000 x 1000 scan on one wavefront
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from secondary_units import write_voc
    voc = os.path.join(tempfile.gettempdir(), "orbhip_voc_k10_L6_nonl.txt")
    if not os.path.exists(voc):
        write_voc(voc, 10, 6)
    L, R, T, P, depth = synth.stereo_sequence(cfg["w"], cfg["h"], nframes, cfg["fx"], cfg["bf"], seed=seed, return_depth=True)
    args = (sensor, L, depth, T, P, cfg["n"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["bf"], cfg["th_depth"], voc)
    kw = dict(dist=TUM1_DIST if sensor == "rgbd" else None, kf_every=kf_every, lost_every=lost_every)
    ref = S.sequence_loop(*args, library=ref_lib, **kw)
    got = S.sequence_loop(*args, library=gpu_lib, **kw)
    mismatched = [k for k, (a, b) in enumerate(zip(ref, got)) if not a.same(b)]
    names = {0: "TrackWithMotionModel", 1: "TrackWithMotionModel (2*th retry)", 2: "TrackReferenceKeyFrame", 4: "Relocalization"}

    def timed(lib, n):
        runs = [S.sequence_loop(*args, capture=False, library=lib, **kw) for _ in range(n)]
        per = {}
        for f in range(2, nframes):
            mode = ref[f].used_wide
            if mode in names:
                per.setdefault(mode, []).append(float(np.median([r[f].ms for r in runs])))
        allms = [float(np.median([r[f].ms for r in runs])) for f in range(2, nframes)]
        ctor = [float(np.median([r[f].ms_ctor for r in runs])) for f in range(2, nframes)]
        return round(float(np.median(allms)), 4), round(float(np.median(ctor)), 4), {names[m]: round(float(np.median(v)), 4) for m, v in per.items()}
    g, gc, gparts = timed(gpu_lib, passes)
    r, rc, rparts = timed(ref_lib, ref_passes)
    return {"shape": f"{cfg['w']}x{cfg['h']} {sensor}, {cfg['n']} features, 8 levels, {nframes} frames, a key frame every {kf_every}, lost every {lost_every}" + (", TUM1 distortion" if sensor == "rgbd" else ", 2 x nFeatures until initialised"),
            "ms_per_frame_gpu": g, "ms_per_frame_ref": r, "speedup": round(r / g, 1), "frame_constructor_ms": {"gpu": gc, "ref": rc},
            "ms_per_frame_by_mode_gpu": gparts, "ms_per_frame_by_mode_ref": rparts,
            "vocabulary": "k = 10, L = 6 (ORBvoc.txt's shape), random", "parity": {"frames_compared": nframes, "frames_mismatched": len(mismatched), "compared": "keys, mvKeysUn, descriptors, depth columns, map point per feature after each matcher, counters, bag-of-words hashes"},
            "modes": [int(f.used_wide) for f in ref]}


if __name__ == "__main__":
    for name in (sys.argv[1:] or list(SHAPES)):
        print(json.dumps(measure_sequence(name) if name in ("mono", "rgbd") else measure(name)), flush=True)
