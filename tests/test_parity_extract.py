"""Parity of the HIP extractor (through the C ABI) with the CPU oracle: bit-exact keypoints, angles, descriptors and
every intermediate stage, on seeded synthetic frames and the degenerate inputs of SURVEY.md §8d.

backend = "emu" (kernel sources under the test-only fiber emulation, CPU) or "gpu" (real liborbhip.so, marked gpu).
"""
import numpy as np
import pytest

import orb_slam2_amd
from orb_slam2_amd import synth


def _same(kg, dg, ko, do):
    assert len(kg) == len(ko), (len(kg), len(ko))
    for f in ko.dtype.names:           # raw IEEE-754 bits of pt.x pt.y size angle response, ints octave class_id
        assert np.array_equal(kg[f].view(np.int32), ko[f].view(np.int32)), f
    assert np.array_equal(dg, do)


@pytest.mark.parametrize("w,h,n,seed", [(320, 240, 500, 1), (400, 250, 300, 2), (640, 480, 1000, 3)])
def test_extract_bit_exact_with_stages(backend, oracle, w, h, n, seed):
    img = synth.frame(w, h, seed=seed)
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    ko, do = ora.extract(img)
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, library=backend)
    kg, dg = ex(img)
    p = ora.params()
    assert np.array_equal(ex.GetScaleFactors(), p["scale_factors"]) and np.array_equal(ex.GetInverseScaleFactors(), p["inv_scale_factors"])
    assert np.array_equal(ex.GetScaleSigmaSquares(), p["sigma2"]) and np.array_equal(ex.GetInverseScaleSigmaSquares(), p["inv_sigma2"])
    assert np.array_equal(ex.features_per_level(), p["features_per_level"]) and ex.GetLevels() == 8
    for l in range(8):
        assert ex.level_size(l) == ora.level_size(l)
        assert np.array_equal(ex.mvImagePyramid(l), ora.level(l)), f"pyramid level {l}"
        assert np.array_equal(ex.candidates(l), ora.candidates(l)), f"FAST candidates level {l}"
        b = ora.blurred(l)
        if b is not None:
            assert np.array_equal(ex.blurred_level(l), b), f"blurred level {l}"
    _same(kg, dg, ko, do)
    ex.close()


@pytest.mark.parametrize("ini,mn", [(0, 0), (1, 0), (255, 200), (255, 255), (300, -5), (-3, -9), (254, 0)])
def test_fast_threshold_extremes(backend, oracle, ini, mn):
    """iniThFAST / minThFAST at and beyond the ends of [0, 255]: cv::FAST clamps its threshold into that range (OpenCV 3.2 fast.cpp, FAST_t), so
    the reference accepts any integer from the settings file.  Threshold 0 is where the one-polarity exact score needs its premise (a margin that
    is not positive is never a corner) and the low-contrast frame sends most cells through the minThFAST call."""
    w, h, n = 320, 240, 500
    img = synth.frame(w, h, seed=3)
    low = np.clip(110 + (img.astype(np.float32) - 128) * 0.1, 0, 255).astype(np.uint8)
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 4, ini, mn, w, h, library=backend)
    ora = oracle.OracleExtractor(n, 1.2, 4, ini, mn)
    for im in (img, low):
        ko, do = ora.extract(im)
        kg, dg = ex(im)
        for l in range(4):
            assert np.array_equal(ex.candidates(l), ora.candidates(l)), f"FAST candidates level {l}"
        _same(kg, dg, ko, do)
    ex.close()


def test_dense_candidates_overflow_lds_keys(backend, oracle):
    """White noise: > 4096 FAST candidates on level 0, so the quadtree keeps its per-candidate keys in the HBM workspace
    instead of LDS, and the quadtree is ~40x over-subscribed."""
    w, h, n = 352, 288, 600
    img = np.random.default_rng(17).integers(0, 256, (h, w), dtype=np.uint8)
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    ko, do = ora.extract(img)
    assert len(ora.candidates(0)) > 4096
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, library=backend)
    kg, dg = ex(img)
    for l in range(8):
        assert np.array_equal(ex.candidates(l), ora.candidates(l))
    _same(kg, dg, ko, do)
    ex.close()


def _clusters(w, h, boxes, seed):
    """flat background with boxes of dense texture (a scene at quarter scale plus noise): every FAST candidate of the frame lies inside them"""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 110, np.uint8)
    tex = synth.frame(4 * w, 4 * h, seed=seed)[::4, ::4]
    for (x, y, bw, bh) in boxes:
        img[y:y + bh, x:x + bw] = np.clip(tex[y:y + bh, x:x + bw].astype(np.int32) + rng.integers(-25, 26, (bh, bw)), 0, 255).astype(np.uint8)
    return img


@pytest.mark.parametrize("case", ["one_cluster", "two_roots", "corner_cluster", "tiny_N", "N_between_passes", "wide_many_roots", "dense_big_N", "deep_cluster", "two_deep_clusters"])
def test_quadtree_regular_pass_jump_and_its_exits(backend, oracle, case):
    """k_quadtree resolves the passes in which every node divides in one step (per-cell key counts -> list positions) and replays the rest.
    Candidate sets that leave that regime at every possible point: all keys in one small region (a pass after which the list did not grow:
    the loop ends right behind the jump), two regions in different roots with empty roots between them, a cluster in the image corner (single
    child chains), so few features that the list reaches N after the first pass, an N that falls between two regular passes (the final phase
    starts behind the jump), a wide image with many roots, white noise with a large N (keys beyond the register budget), and many features wanted
    from one or two regions (a four-pass jump, then passes deeper than the path digits computed up front: those are recomputed on demand)."""
    w, h, n, nl = 480, 320, 400, 4
    if case == "one_cluster":
        img = _clusters(w, h, [(200, 130, 44, 40)], 1)
    elif case == "two_roots":
        img = _clusters(w, h, [(30, 40, 60, 50), (400, 230, 50, 60)], 2)
    elif case == "corner_cluster":
        img = _clusters(w, h, [(20, 20, 36, 36)], 3)
    elif case == "tiny_N":
        img, n = synth.frame(w, h, seed=4), 8
    elif case == "N_between_passes":
        img, n = synth.frame(w, h, seed=5), 70
    elif case == "wide_many_roots":
        w, h, n = 900, 130, 600
        img = synth.frame(w, h, seed=6)
    elif case == "deep_cluster":          # many features wanted from one region: the tree goes past the path digits computed up front (and jumps four passes)
        img, n, nl = _clusters(w, h, [(150, 100, 110, 90)], 9), 1500, 2
    elif case == "two_deep_clusters":
        img, n, nl = _clusters(w, h, [(40, 60, 90, 70), (300, 200, 80, 70)], 9), 2000, 3
    else:
        w, h, n, nl = 640, 480, 5000, 3
        img = np.random.default_rng(7).integers(0, 256, (h, w), dtype=np.uint8)
    ora = oracle.OracleExtractor(n, 1.2, nl, 20, 7)
    ko, do = ora.extract(img)
    ex = orb_slam2_amd.ORBextractor(n, 1.2, nl, 20, 7, w, h, library=backend)
    kg, dg = ex(img)
    assert kg.tobytes() == ko.tobytes() and np.array_equal(dg, do), case
    assert len(ko) >= 4, (case, len(ko))          # (clustered candidates end the replay early: the reference returns a handful of key points per level)
    ex.close()


def test_quadtree_final_phase_ranking_both_forms(emu_lib, oracle, monkeypatch):
    """The final phase ranks the expandable nodes by (size, list position): as one packed number per node where sizes fit 18 bits, by three compares
    for levels with 2^18 candidates or more.  ORBHIP_TEST_QT_UNPACKED=1 (emulation build only) sends ordinary frames through the second form."""
    monkeypatch.setenv("ORBHIP_TEST_QT_UNPACKED", "1")
    for (w, h, n, seed) in ((640, 480, 1000, 3), (480, 320, 70, 5), (752, 480, 1200, 8)):
        img = synth.frame(w, h, seed=seed)
        ko, do = oracle.OracleExtractor(n, 1.2, 8, 20, 7).extract(img)
        ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, library=emu_lib)
        kg, dg = ex(img)
        assert kg.tobytes() == ko.tobytes() and np.array_equal(dg, do), (w, h, n)
        ex.close()


@pytest.mark.parametrize("name", ["zeros", "checkerboard", "ramp", "low_texture", "saturated"])
def test_degenerate_inputs(backend, oracle, name):
    w, h, n = 320, 240, 400
    img = {"zeros": synth.zeros(w, h), "checkerboard": synth.checkerboard(w, h, cell=8), "ramp": synth.ramp(w, h),
           "low_texture": synth.low_texture(w, h), "saturated": np.full((h, w), 255, np.uint8)}[name]
    ko, do = oracle.OracleExtractor(n, 1.2, 8, 20, 7).extract(img)
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, library=backend)
    kg, dg = ex(img)
    _same(kg, dg, ko, do)
    if name in ("zeros", "saturated"):
        assert len(kg) == 0 and dg.shape == (0, 32)          # descriptors released, keypoints cleared
    k0, d0 = ex(None)                                         # empty image: silent return (ORBextractor.cc:1046-1047)
    assert len(k0) == 0
    ex.close()


def test_batch_strides_thresholds_and_round_mode(backend, oracle):
    w, h, n = 352, 288, 350
    seq = synth.sequence(w, h, 3, seed=7)
    # non-default thresholds / levels / scale factor, SSE2 rounding mode of the blur
    ora = oracle.OracleExtractor(n, 1.25, 6, 12, 5, blur_round_mode=1)
    ex = orb_slam2_amd.ORBextractor(n, 1.25, 6, 12, 5, w, h, max_batch=3, blur_round_mode=1, library=backend)
    ks, ds = ex.extract_batch(seq)
    for f in range(3):
        ko, do = ora.extract(seq[f])
        _same(ks[f], ds[f], ko, do)
    # a frame alone == the same frame inside a batch (slots are independent)
    k1, d1 = ex(seq[1])
    _same(k1, d1, ks[1], ds[1])
    # padded row stride through the raw C ABI
    import ctypes as C
    padded = np.zeros((h, w + 37), np.uint8)
    padded[:, :w] = seq[2]
    cap = ex.capacity
    kps = np.zeros(cap, orb_slam2_amd.KEYPOINT_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    nout = C.c_int()
    st = ex.L.orbhip_extract(ex.h, padded.ctypes.data_as(C.c_void_p), w + 37, kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap, C.byref(nout))
    assert st == 0
    _same(kps[:nout.value], desc[:nout.value], ks[2], ds[2])
    # too-small caller buffer: ORBHIP_ERR_CAPACITY, n_out still reports the true count
    st = ex.L.orbhip_extract(ex.h, padded.ctypes.data_as(C.c_void_p), w + 37, kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), 10, C.byref(nout))
    assert st == 3 and nout.value == len(ks[2])
    ex.close()


def test_large_scale_factor_uses_direct_pyramid(backend, oracle):
    """scaleFactor 1.6: the 256-px output tile's source footprint exceeds the LDS stage -> direct-from-global pyramid kernel."""
    w, h, n = 640, 480, 300
    img = synth.frame(w, h, seed=5)
    ora = oracle.OracleExtractor(n, 1.6, 4, 20, 7)
    ko, do = ora.extract(img)
    ex = orb_slam2_amd.ORBextractor(n, 1.6, 4, 20, 7, w, h, library=backend)
    kg, dg = ex(img)
    for l in range(4):
        assert np.array_equal(ex.mvImagePyramid(l), ora.level(l))
    _same(kg, dg, ko, do)
    ex.close()


@pytest.mark.parametrize("w,h,levels", [(91, 75, 2), (124, 62, 1), (150, 97, 3)])
def test_tiny_images_single_cell_levels(backend, oracle, w, h, levels):
    """Smallest supported shapes: one FAST cell per level up to 59 px wide (65-px sub-image, 17 dwords per patch row),
    levels narrower than one blur / pyramid tile, more quadtree roots than features."""
    rng = np.random.default_rng(w * 1000 + h)
    img = np.clip(synth.frame(max(w, 320), max(h, 240), seed=13)[:h, :w].astype(int) + rng.integers(-20, 21, (h, w)), 0, 255).astype(np.uint8)
    for n in (5, 60):
        ora = oracle.OracleExtractor(n, 1.2, levels, 20, 7)
        ko, do = ora.extract(img)
        ex = orb_slam2_amd.ORBextractor(n, 1.2, levels, 20, 7, w, h, library=backend)
        kg, dg = ex(img)
        for l in range(levels):
            assert np.array_equal(ex.mvImagePyramid(l), ora.level(l)) and np.array_equal(ex.candidates(l), ora.candidates(l))
            b = ora.blurred(l)
            if b is not None:
                assert np.array_equal(ex.blurred_level(l), b)
        _same(kg, dg, ko, do)
        ex.close()


@pytest.mark.parametrize("channels,rgb,w,h", [(3, True, 321, 243), (3, False, 320, 240), (4, True, 402, 250), (4, False, 319, 241)])
def test_colour_input_converted_on_device(backend, oracle, channels, rgb, w, h):
    """Tracking::GrabImage* (Tracking.cc:172-198) converts colour frames with cvtColor before the extractor; the colour entry
    points do that on the device.  Widths not divisible by 4 exercise the ragged row end, 3-channel odd widths the unaligned rows."""
    n = 400
    planes = [synth.frame(w, h, seed=40 + c) for c in range(channels)]
    col = np.stack(planes, axis=-1)
    gray = oracle.cvt_gray(col, rgb=rgb)
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, max_batch=2, library=backend)
    ks, ds = ex.extract_batch_color([col, col[::-1].copy()], rgb=rgb)
    assert np.array_equal(ex.mvImagePyramid(0, frame=0), gray)
    assert np.array_equal(ex.mvImagePyramid(0, frame=1), gray[::-1])
    ko, do = ora.extract(gray)
    _same(ks[0], ds[0], ko, do)
    ko, do = ora.extract(np.ascontiguousarray(gray[::-1]))
    _same(ks[1], ds[1], ko, do)
    with pytest.raises(orb_slam2_amd.OrbHipError):
        ex.extract_batch_color([col[..., :2].copy()], rgb=rgb)
    ex.close()


def test_unsupported_and_invalid_configs(backend):
    with pytest.raises(orb_slam2_amd.OrbHipError):
        orb_slam2_amd.ORBextractor(500, 1.2, 8, 20, 7, 200, 120, library=backend)     # top level < 62 px: the reference itself divides by zero
    with pytest.raises(orb_slam2_amd.OrbHipError):
        orb_slam2_amd.ORBextractor(500, 1.2, 4, 20, 7, 240, 640, library=backend)     # portrait: zero quadtree roots (ORBextractor.cc:543)
    with pytest.raises(orb_slam2_amd.OrbHipError):
        orb_slam2_amd.ORBextractor(500, 1.0, 8, 20, 7, 640, 480, library=backend)     # scale factor must exceed 1


def test_golden_fixture(backend):
    """Committed golden vectors (tests/golden/): the HIP path reproduces them without the oracle library present."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "extract_320x240_n300_seed21.npz"))
    ex = orb_slam2_amd.ORBextractor(300, 1.2, 8, 20, 7, 320, 240, library=backend)
    k, d = ex(g["image"])
    assert k.tobytes() == g["keypoints"].tobytes() and np.array_equal(d, g["descriptors"])
    ex.close()
