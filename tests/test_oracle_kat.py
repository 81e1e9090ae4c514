"""Known-answer tests that pin the CPU oracle (the reference ships no tests or golden vectors: SURVEY.md §4).

Hand-derivable anchors: umax table, features-per-level split, scale tables, level sizes for the four
BASELINE configs, the fixed-point Gaussian kernel, Hamming extremes, plus per-primitive checks of the restated OpenCV
functions and a bit-for-bit sweep of the sincosf restatement against this box's libm.
"""
import math
import os

import numpy as np
import pytest

from orb_slam2_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_constructor_tables(oracle):
    ex = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    p = ex.params()
    assert p["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]      # ORBextractor.cc:452-469
    assert int(2 * p["umax"][1:].sum() * 2 + 4 * 0 + (2 * 15 + 1) + 2 * len(p["umax"][1:])) == 749   # patch pixels
    assert p["features_per_level"].tolist() == [434, 362, 302, 251, 209, 175, 145, 122]             # :435-446
    want = [1, 1.2000000477, 1.4400000572, 1.7280001640, 2.0736002922, 2.4883203506, 2.9859845638, 3.5831816196]
    assert np.allclose(p["scale_factors"], want, rtol=0, atol=1e-7)
    assert np.array_equal(p["sigma2"], p["scale_factors"] * p["scale_factors"])
    assert np.array_equal(p["inv_scale_factors"], np.float32(1.0) / p["scale_factors"])
    for n, want in ((1000, [217, 181, 151, 126, 105, 87, 73, 60]), (1200, [261, 217, 181, 151, 126, 105, 87, 72]),
                    (4000, [869, 724, 603, 503, 419, 349, 291, 242])):
        assert oracle.OracleExtractor(n, 1.2, 8, 20, 7).params()["features_per_level"].tolist() == want


@pytest.mark.parametrize("w,h,sizes", [
    (1241, 376, [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]),
    (640, 480, [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]),
    (752, 480, [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]),
])
def test_level_sizes(oracle, w, h, sizes):
    ex = oracle.OracleExtractor(500, 1.2, 8, 20, 7)
    ex.extract(np.zeros((h, w), np.uint8))
    assert [ex.level_size(l) for l in range(8)] == sizes                                          # ORBextractor.cc:1111-1112


def test_gaussian_kernel_and_blur(oracle):
    assert oracle.gauss_kernel().tolist() == [18, 34, 49, 55, 49, 34, 18]                         # cvRound(getGaussianKernel(7,2)*256)
    for c in (0, 1, 77, 200, 254, 255):                                                           # constant image: (c*257*257 + 2^15) >> 16, saturated
        img = np.full((40, 52), c, np.uint8)
        want = min((c * 257 * 257 + 32768) >> 16, 255)
        assert np.all(oracle.blur(img) == want)
    # single impulse away from the border: separable kernel outer product
    img = np.zeros((31, 33), np.uint8)
    img[15, 16] = 255
    k = np.array([18, 34, 49, 55, 49, 34, 18])
    want = np.zeros_like(img)
    want[12:19, 13:20] = np.minimum((np.outer(k, k) * 255 + 32768) >> 16, 255)
    assert np.array_equal(oracle.blur(img), want)
    # BORDER_REFLECT_101: a column ramp mirrored about the first/last pixel (not duplicated)
    img = np.tile(np.arange(40, dtype=np.uint8) * 5, (20, 1))
    b = oracle.blur(img)
    row = img[0].astype(np.int64)
    ext = np.concatenate([row[3:0:-1], row, row[-2:-5:-1]])
    h = np.array([(k * ext[i:i + 7]).sum() for i in range(40)])
    assert np.array_equal(b[10], np.minimum((h * 257 + 32768) >> 16, 255))
    # rounding modes only differ where the 16-bit remainder is exactly one half
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (64, 67), dtype=np.uint8)
    d = oracle.blur(img, 0).astype(int) - oracle.blur(img, 1).astype(int)
    assert set(np.unique(d)) <= {0, 1}


def test_resize_fixed_point(oracle):
    for c in (0, 3, 128, 255):                                                                     # coefficient pairs sum to 2048: constants stay constant
        assert np.all(oracle.resize(np.full((50, 60), c, np.uint8), 50, 42) == c)
    # 1-D hand computation of the OpenCV fixed-point recipe on a horizontal ramp
    src = np.tile((np.arange(60) * 4).astype(np.uint8), (12, 1))
    dw = 50
    out = oracle.resize(src, dw, 10)
    scale = 1.0 / (dw / 60.0)
    for dx in range(dw):
        fx = np.float32((dx + 0.5) * scale - 0.5)
        sx = int(math.floor(fx))
        fx = np.float32(fx - sx)
        a1 = int(np.rint(np.float32(fx * np.float32(2048))))
        a0 = int(np.rint(np.float32((np.float32(1) - fx) * np.float32(2048))))
        hsum = int(src[0, sx]) * a0 + int(src[0, min(sx + 1, 59)]) * a1
        # rows are identical, so the vertical pass blends two equal row sums with b0 + b1 = 2048
        vals = {((b0 * (hsum >> 4)) >> 16) + (((2048 - b0) * (hsum >> 4)) >> 16) + 2 >> 2 for b0 in range(0, 2049)}
        assert int(out[5, dx]) in vals
    assert out.shape == (10, 50)
    # exactly half size in both directions: OpenCV takes its INTER_AREA fast path, the 2x2 box average — identical to this path's result
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    box = ((a[0::2, 0::2].astype(int) + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert np.array_equal(oracle.resize(a, 32, 24), box)


def test_cvt_gray_fixed_point(oracle):
    """cvtColor(*2GRAY) of Tracking.cc:172-198: 14-bit fixed point, coefficients 4899/9617/1868, alpha skipped."""
    rng = np.random.default_rng(5)
    for ch in (3, 4):
        src = rng.integers(0, 256, (37, 53, ch), dtype=np.uint8)
        s = src.astype(np.int64)
        for rgb in (True, False):
            r, b = (s[..., 0], s[..., 2]) if rgb else (s[..., 2], s[..., 0])
            want = ((r * 4899 + s[..., 1] * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)
            assert np.array_equal(oracle.cvt_gray(src, rgb=rgb), want)
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [128, 128, 128]]], np.uint8)
    assert oracle.cvt_gray(px, rgb=True).tolist() == [[76, 150, 29, 255, 0, 128]]
    assert oracle.cvt_gray(px, rgb=False).tolist() == [[29, 150, 76, 255, 0, 128]]


def test_undistort_points_known_answers(oracle):
    """cv::undistortPoints(.., K, D, Mat(), K): hand-derivable cases of the 5-iteration inverse of the Brown model."""
    cam = (500.0, 480.0, 320.0, 240.0, -0.2, 0.05, 0.001, -0.0005, 0.01)
    # the principal point is a fixed point; without distortion the map is the identity for points whose normalisation is exact
    assert oracle.undistort_points(cam, [[320.0, 240.0]]).tolist() == [[320.0, 240.0]]
    flat = cam[:4] + (0.0, 0.0, 0.0, 0.0, 0.0)
    pts = np.array([[320.0 + 500.0 * 0.25, 240.0 - 480.0 * 0.5], [70.0, 0.0], [0.0, 0.0]], np.float32)
    assert oracle.undistort_points(flat, pts).tolist() == pts.tolist()
    # one radial coefficient, point on the x axis: x' solves x = x'(1 + k1 x'^2); 5 iterations of x <- x0 / (1 + k1 x^2) in double
    k1, x0 = 0.1, 0.5
    x = x0
    for _ in range(5):
        x = x0 * (1 / (1 + k1 * (x * x)))
    got = oracle.undistort_points((400.0, 400.0, 100.0, 100.0, k1, 0.0, 0.0, 0.0), [[100.0 + 400.0 * x0, 100.0]])
    assert got[0, 0] == np.float32(400.0 * x + 100.0) and got[0, 1] == np.float32(100.0)
    # re-distorting the result gives the input back (the iteration has converged well inside the image)
    rng = np.random.default_rng(0)
    p = np.stack([rng.uniform(170, 470, 200), rng.uniform(120, 360, 200)], axis=1).astype(np.float32)
    u = oracle.undistort_points(cam, p).astype(np.float64)
    xn, yn = (u[:, 0] - 320) / 500, (u[:, 1] - 240) / 480
    r2 = xn * xn + yn * yn
    rad = 1 + (-0.2) * r2 + 0.05 * r2 ** 2 + 0.01 * r2 ** 3
    xd = xn * rad + 2 * 0.001 * xn * yn + (-0.0005) * (r2 + 2 * xn * xn)
    yd = yn * rad + 0.001 * (r2 + 2 * yn * yn) + 2 * (-0.0005) * xn * yn
    assert np.abs(xd * 500 + 320 - p[:, 0]).max() < 2e-3 and np.abs(yd * 480 + 240 - p[:, 1]).max() < 2e-3
    # Frame::ComputeImageBounds: k1 == 0 -> the image rectangle
    assert oracle.image_bounds(flat, 640, 480).tolist() == [0.0, 0.0, 640.0, 480.0]
    b = oracle.image_bounds(cam, 640, 480)
    c = oracle.undistort_points(cam, [[0, 0], [640, 0], [0, 480], [640, 480]])
    assert b.tolist() == [min(c[0, 0], c[2, 0]), min(c[0, 1], c[1, 1]), max(c[1, 0], c[3, 0]), max(c[2, 1], c[3, 1])]


def test_remap_fixed_point(oracle):
    """cv::remap INTER_LINEAR 8U with CV_32FC1 maps: 5-bit fractions, (sum w p + 2^14) >> 15, constant 0 border."""
    rng = np.random.default_rng(4)
    src = rng.integers(0, 256, (9, 12), dtype=np.uint8)
    yy, xx = np.mgrid[0:9, 0:12].astype(np.float32)
    assert np.array_equal(oracle.remap(src, xx, yy), src)                                         # identity map
    half = oracle.remap(src, xx + 0.5, yy)                                                        # a = 16: (p0 + p1 + 1) >> 1, last column reads the border
    want = (src.astype(np.int32) + np.concatenate([src[:, 1:], np.zeros((9, 1), np.uint8)], axis=1) + 1) >> 1
    assert np.array_equal(half, want)
    # map quantisation: cvRound(32 m) is round-half-even -> m = 3 + 1/64 stays on pixel 3, m = 3 + 3/64 uses a = 2
    one = lambda mx, my: int(oracle.remap(src, np.array([[mx]], np.float32), np.array([[my]], np.float32))[0, 0])
    assert one(3 + 1 / 64, 2.0) == int(src[2, 3])
    assert one(3 + 3 / 64, 2.0) == (30 * int(src[2, 3]) + 2 * int(src[2, 4]) + 16) >> 5
    a, b = 7, 21                                                                                  # general bilinear case
    p = src[4:6, 5:7].astype(np.int64)
    acc = 32 * ((32 - b) * ((32 - a) * p[0, 0] + a * p[0, 1]) + b * ((32 - a) * p[1, 0] + a * p[1, 1]))
    assert one(5 + a / 32, 4 + b / 32) == (acc + (1 << 14)) >> 15
    # border: windows touching the image read 0 outside, windows entirely outside give 0
    assert one(-0.5, 0.0) == (int(src[0, 0]) + 1) >> 1 and one(-1.0, 0.0) == 0 and one(-1.5, 3.0) == 0
    assert one(11.0, 8.0) == int(src[8, 11]) and one(11.5, 8.5) == (int(src[8, 11]) * 256 + 512) >> 10 and one(12.0, 3.0) == 0


def test_fast_known_corner(oracle):
    # isolated bright pixel: all 16 ring pixels are darker by exactly 100 -> corner with score 100 - 1, no other corner
    img = np.full((21, 23), 100, np.uint8)
    img[9, 12] = 200
    for nms, sc in ((False, 0), (True, 99)):                                       # without NMS cv::FAST leaves response = 0
        assert oracle.fast(img, 20, nms=nms).tolist() == [[12, 9, sc]]
        assert oracle.fast(img, 99, nms=nms).tolist() == [[12, 9, sc]]            # corner at t  <=>  score >= t
        assert len(oracle.fast(img, 100, nms=nms)) == 0
    # L-shaped step: the inner-corner pixel sees an 11-pixel dark arc, its diagonal neighbour a 9-pixel one, both with
    # margin 80 -> equal scores 79 annihilate each other under the strict '>' NMS
    img = np.full((21, 21), 100, np.uint8)
    img[10:, 10:] = 180
    raw = oracle.fast(img, 20, nms=False)[:, :2].tolist()
    assert [10, 10] in raw and [11, 11] in raw
    assert oracle.fast(img, 79, nms=False)[:, :2].tolist() == raw                  # every arc pixel differs by exactly 80 -> score 79
    assert len(oracle.fast(img, 80, nms=False)) == 0
    assert [10, 10] not in oracle.fast(img, 20, nms=True)[:, :2].tolist()
    # flat image and 6x6 image: nothing
    assert len(oracle.fast(np.full((30, 30), 9, np.uint8), 7)) == 0
    assert len(oracle.fast(np.zeros((6, 6), np.uint8), 7)) == 0
    # row-major output order and the 3-px examined border
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (40, 47), dtype=np.uint8)
    k = oracle.fast(img, 10, nms=True)
    order = k[:, 1] * 1000 + k[:, 0]
    assert np.all(np.diff(order) > 0)
    assert k[:, 0].min() >= 3 and k[:, 0].max() <= 47 - 4 and k[:, 1].min() >= 3 and k[:, 1].max() <= 40 - 4
    # an X-junction of a checkerboard is not a FAST-9 corner (no 9-pixel arc of one sign)
    assert len(oracle.fast(synth.checkerboard(64, 64, cell=8), 20, nms=False)) == 0


def test_fast_atan2(oracle):
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = (float(v) for v in rng.integers(-200000, 200000, 2))
        a = oracle.fastatan2(y, x)
        ref = math.degrees(math.atan2(y, x)) % 360.0
        assert 0.0 <= a <= 360.0
        assert min(abs(a - ref), 360 - abs(a - ref)) < 0.3                                        # documented accuracy of cv::fastAtan2
    assert oracle.fastatan2(0.0, 0.0) == 0.0
    assert oracle.fastatan2(0.0, 5.0) == 0.0 and oracle.fastatan2(5.0, 0.0) == 90.0
    assert oracle.fastatan2(0.0, -5.0) == 180.0 and oracle.fastatan2(-5.0, 0.0) == 270.0


def test_sincosf_matches_this_box_libm(oracle):
    """The glibc-2.35 sincosf restatement is bit-identical to the real libm of this machine over [0, 2*pi]
    (exhaustive sweep, a few seconds with the -O3 build)."""
    two_pi_bits = 0x40C90FDB
    assert oracle.lib(fast=True).orb_oracle_sincosf_vs_libm(0, two_pi_bits, 1) == 0     # exhaustive: all 1 086 918 620 floats in [0, 2*pi]
    assert oracle.lib().orb_oracle_sincosf_vs_libm(0x3F000000, 0x3F800000, 1) == 0                # dense around pi/4 (branch switch)
    s, c = oracle.sincosf(0.0)
    assert (s, c) == (0.0, 1.0)


def test_hamming(oracle):
    z, o = np.zeros(32, np.uint8), np.full(32, 255, np.uint8)
    assert oracle.hamming(z, z) == 0 and oracle.hamming(z, o) == 256 and oracle.hamming(o, o) == 0
    rng = np.random.default_rng(1)
    for _ in range(200):
        a, b = rng.integers(0, 256, 32, dtype=np.uint8), rng.integers(0, 256, 32, dtype=np.uint8)
        assert oracle.hamming(a, b) == int(np.unpackbits(a ^ b).sum())


def test_quadtree_small_cases(oracle):
    # two far-apart points, N large: both survive, list order = reverse creation order
    out = oracle.distribute([[10, 10, 50], [300, 200, 60]], 16, 16 + 400, 16, 16 + 300, 10)
    assert sorted(map(tuple, out)) == [(10, 10, 50), (300, 200, 60)]
    # one cluster in one leaf: the best response wins, the FIRST of equal responses wins
    pts = [[100, 100, 30], [101, 100, 90], [100, 101, 90], [101, 101, 10]]
    out = oracle.distribute(pts, 16, 16 + 400, 16, 16 + 300, 1)
    assert len(out) >= 1
    big = oracle.distribute(pts, 16, 16 + 400, 16, 16 + 300, 0)
    assert all(tuple(p) in set(map(tuple, pts)) for p in big)
    # no candidates -> no keypoints
    assert len(oracle.distribute(np.zeros((0, 3), np.int32), 16, 416, 16, 316, 50)) == 0
    # the result can exceed N by up to 3 (ORBextractor.cc:730-731 breaks after the split that reaches N)
    rng = np.random.default_rng(9)
    xy = np.unique(rng.integers(3, 290, (4000, 2)), axis=0)
    pts = np.column_stack([xy, rng.integers(7, 255, len(xy))])
    for N in (1, 17, 100, 434):
        out = oracle.distribute(pts, 16, 16 + 1200, 16, 16 + 300, N)
        assert N <= len(out) <= N + 3
        assert len(set(map(tuple, out[:, :2]))) == len(out)


def test_timing_build_equals_parity_build(oracle):
    img = synth.frame(640, 480, seed=4)
    a = oracle.OracleExtractor(1000, 1.2, 8, 20, 7).extract(img)
    b = oracle.OracleExtractor(1000, 1.2, 8, 20, 7, fast=True).extract(img)
    assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1])


def test_degenerate_images(oracle):
    ex = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
    k, d = ex.extract(synth.zeros(640, 480))
    assert len(k) == 0 and d.shape == (0, 32)                                                       # descriptors released (ORBextractor.cc:1064-1065)
    k, d = ex.extract(synth.low_texture(640, 480))
    assert 0 < len(k) < 1000                                                                        # fewer than nfeatures on low texture
    assert (k["response"] < 20).any()                                                               # some cells fell back to minThFAST
    k, d = ex.extract(synth.frame(640, 480, seed=1))
    assert 1000 <= len(k) <= 1000 + 3 * 8
    assert np.all(k["class_id"] == -1) and np.all(k["size"] == np.floor(31 * ex.params()["scale_factors"][k["octave"]]))
    assert np.all(np.diff(k["octave"]) >= 0)                                                        # level-major output


def test_golden_fixtures(oracle):
    """Committed golden vectors (tests/golden/make_golden.py, generated by this oracle): guards the oracle against drift."""
    g = np.load(os.path.join(GOLDEN, "extract_320x240_n300_seed21.npz"))
    img = synth.frame(320, 240, seed=21)
    assert np.array_equal(img, g["image"]), "synthetic generator drifted"
    k, d = oracle.OracleExtractor(300, 1.2, 8, 20, 7).extract(img)
    assert k.tobytes() == g["keypoints"].tobytes() and np.array_equal(d, g["descriptors"])
    g = np.load(os.path.join(GOLDEN, "match_320x240_n300_seed21.npz"))
    n, m12, prev = oracle.search_for_initialization(g["k1"], g["d1"], g["k2"], g["d2"], 320, 240, window=100, nnratio=0.9)
    assert n == int(g["nmatches"]) and np.array_equal(m12, g["matches12"]) and prev.tobytes() == g["prev"].tobytes()


def test_fast_score_needs_one_polarity_only():
    """The identity k_fast_cells' exact-score stage rests on (orbhip_kernels_extract.hip, fast_score_pair): with ring values R_0..R_15, centre C and
    m = max_k min(R_k, R_k+8), the cornerScore  S = max(C - min_arcs max R, max_arcs min R - C) - 1  equals the one-sided score of the pixel
    (complemented when m <= C) whenever S >= 0, and the one-sided score is negative whenever S is - so "score >= threshold" and the score itself agree
    for every threshold >= 0 (cv::FAST clamps its threshold to [0, 255]).  Checked on random, near-flat, arc-shaped, bright-arc-on-dark and two-valued rings."""
    rng = np.random.default_rng(20260922)
    idx = (np.arange(16)[:, None] + np.arange(9)[None, :]) % 16
    n = 200_000

    def both(R, C):
        A = R[:, idx]
        two_sided = np.maximum(C - A.max(axis=2).min(axis=1), A.min(axis=2).max(axis=1) - C) - 1
        dark = np.minimum(R[:, :8], R[:, 8:]).max(axis=1) <= C
        Q = np.where(dark[:, None], 255 - R, R)
        one_sided = Q[:, idx].min(axis=2).max(axis=1) - np.where(dark, 255 - C, C) - 1
        return two_sided, one_sided

    cases = []
    cases.append((rng.integers(0, 256, (n, 16)), rng.integers(0, 256, n)))
    cases.append((rng.integers(100, 140, (n, 16)), rng.integers(100, 140, n)))
    C = rng.integers(0, 256, n)
    R = np.clip(C[:, None] + rng.integers(-40, 41, (n, 16)), 0, 255)
    k = (np.arange(16)[None, :] - rng.integers(0, 16, n)[:, None]) % 16
    arc = k < rng.integers(7, 12, n)[:, None]
    cases.append((np.where(arc, np.clip(C + rng.choice([-1, 1], n) * rng.integers(1, 80, n), 0, 255)[:, None], R), C))
    C = rng.integers(20, 236, n)
    cases.append((np.where(arc, np.clip(C[:, None] + rng.integers(1, 60, (n, 16)), 0, 255), np.clip(C[:, None] - rng.integers(1, 60, (n, 16)), 0, 255)), C))
    cases.append((rng.choice([0, 255], (n, 16)), rng.choice([0, 1, 127, 128, 254, 255], n)))
    for R, C in cases:
        s2, s1 = both(R.astype(np.int32), C.astype(np.int32))
        assert np.array_equal(s1[s2 >= 0], s2[s2 >= 0])
        assert (s1[s2 < 0] < 0).all()
