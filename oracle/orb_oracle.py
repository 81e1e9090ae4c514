"""ctypes binding of the CPU oracle (oracle/orb_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never from the
orb_slam2_amd package (the product path fails loudly without its HIP library instead of falling back).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# == cv::KeyPoint memory layout (28 B), see include/orbhip.h orbhip_keypoint
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def build(force=False):
    """Compile the oracle with its Makefile (g++ only; no reference sources, no OpenCV)."""
    so = os.path.join(_HERE, "liborb_oracle.so")
    if force or not os.path.exists(so) or not os.path.exists(os.path.join(_HERE, "liborb_oracle_fast.so")) \
            or os.path.getmtime(so) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("orb_oracle.cpp", "bow_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


_libs = {}


def lib(fast=False):
    name = "liborb_oracle_fast.so" if fast else "liborb_oracle.so"
    if name in _libs:
        return _libs[name]
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    u8p, i32p, f32p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_void_p
    L.orb_oracle_create.restype = vp
    L.orb_oracle_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
    L.orb_oracle_destroy.argtypes = [vp]
    L.orb_oracle_set_blur_round_mode.argtypes = [vp, C.c_int]
    L.orb_oracle_set_fp_contract.argtypes = [vp, C.c_int]
    L.orb_oracle_extract.restype = C.c_int
    L.orb_oracle_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int]
    L.orb_oracle_get_params.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.orb_oracle_level_size.argtypes = [vp, C.c_int, i32p, i32p]
    L.orb_oracle_get_level.argtypes = [vp, C.c_int, vp]
    L.orb_oracle_get_blurred.argtypes = [vp, C.c_int, vp]
    L.orb_oracle_get_candidates.argtypes = [vp, C.c_int, vp, C.c_int]
    L.orb_oracle_get_level_keypoints.argtypes = [vp, C.c_int, vp, C.c_int]
    L.orb_oracle_resize.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int]
    L.orb_oracle_blur.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.orb_oracle_voc_load.argtypes = [C.c_char_p]; L.orb_oracle_voc_load.restype = vp
    L.orb_oracle_voc_free.argtypes = [vp]
    L.orb_oracle_voc_info.argtypes = [vp, vp]
    L.orb_oracle_voc_transform_features.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.orb_oracle_voc_transform.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    L.orb_oracle_voc_transform.restype = C.c_int
    L.orb_oracle_voc_score.argtypes = [C.c_int, vp, vp, C.c_int, vp, vp, C.c_int]
    L.orb_oracle_voc_score.restype = C.c_double
    L.orb_oracle_forb_distance.argtypes = [vp, vp]
    L.orb_oracle_search_for_triangulation.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, C.c_float, C.c_float, vp, vp, C.c_int, C.c_int, vp]
    L.orb_oracle_search_for_triangulation.restype = C.c_int
    L.orb_oracle_search_best_in_window.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp]
    L.orb_oracle_search_by_bow.argtypes = [C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_float, C.c_int, vp]
    L.orb_oracle_search_by_bow.restype = C.c_int
    L.orb_oracle_cvt_gray.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.orb_oracle_gauss_kernel.argtypes = [vp]
    L.orb_oracle_set_image_bounds.argtypes = [vp]
    L.orb_oracle_undistort_points.argtypes = [vp, vp, vp, C.c_int, vp]
    L.orb_oracle_image_bounds.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.orb_oracle_stereo_from_rgbd.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp, vp]
    L.orb_oracle_remap.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]
    L.orb_oracle_fast.restype = C.c_int
    L.orb_oracle_fast.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.orb_oracle_fastatan2.restype = C.c_float
    L.orb_oracle_fastatan2.argtypes = [C.c_float, C.c_float]
    L.orb_oracle_sincosf.argtypes = [C.c_float, f32p, f32p]
    L.orb_oracle_sincosf_vs_libm.restype = C.c_long
    L.orb_oracle_sincosf_vs_libm.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.orb_oracle_distribute.restype = C.c_int
    L.orb_oracle_distribute.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.orb_oracle_hamming.restype = C.c_int
    L.orb_oracle_hamming.argtypes = [vp, vp]
    L.orb_oracle_bf_nn.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp]
    L.orb_oracle_search_by_projection.restype = C.c_int
    L.orb_oracle_project_points.argtypes = [vp, vp, C.c_int, C.c_float, vp]
    L.orb_oracle_predict_scale_of_ratio.argtypes = [C.c_float, C.c_float, C.c_int]
    L.orb_oracle_search_by_projection.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, vp]
    L.orb_oracle_stereo_matches.restype = C.c_int
    L.orb_oracle_stereo_matches.argtypes = [vp, vp, C.c_float, C.c_float, vp, vp, C.c_int]
    L.orb_oracle_search_for_initialization.restype = C.c_int
    L.orb_oracle_search_for_initialization.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp,
                                                       C.c_int, C.c_float, C.c_int]
    L.orb_oracle_features_in_area.restype = C.c_int
    L.orb_oracle_features_in_area.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                              C.c_int, C.c_int, vp, C.c_int]
    _libs[name] = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleExtractor:
    """CPU restatement of ORB_SLAM2::ORBextractor (ORBextractor.h:45-111)."""

    def __init__(self, nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, fast=False,
                 blur_round_mode=0, fp_contract=0):
        self.L = lib(fast)
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = self.L.orb_oracle_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
        self.L.orb_oracle_set_blur_round_mode(self.h, blur_round_mode)
        self.L.orb_oracle_set_fp_contract(self.h, fp_contract)      # 1 = the FMA forms gcc emits for the reference's own flags (H3)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orb_oracle_destroy(self.h)
            self.h = None

    def extract(self, img):
        """img: HxW uint8 (any row stride).  Returns (keypoints[KEYPOINT_DTYPE], descriptors[N,32] uint8)."""
        assert img.dtype == np.uint8 and img.ndim == 2 and img.strides[1] == 1
        cap = self.nfeatures + 3 * self.nlevels + 64
        while True:
            kps = np.zeros(cap, KEYPOINT_DTYPE)
            desc = np.zeros((cap, 32), np.uint8)
            n = self.L.orb_oracle_extract(self.h, _p(img), img.shape[1], img.shape[0], img.strides[0], _p(kps), _p(desc), cap)
            if n <= cap:
                return kps[:n].copy(), desc[:n].copy()
            cap = n

    def params(self):
        n = self.nlevels
        fpl = np.zeros(n, np.int32)
        sf, isf, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        umax = np.zeros(16, np.int32)
        self.L.orb_oracle_get_params(self.h, _p(fpl), _p(sf), _p(isf), _p(s2), _p(is2), _p(umax))
        return dict(features_per_level=fpl, scale_factors=sf, inv_scale_factors=isf, sigma2=s2, inv_sigma2=is2, umax=umax)

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        assert self.L.orb_oracle_level_size(self.h, level, C.byref(w), C.byref(h)) == 0
        return w.value, h.value

    def level(self, level):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        self.L.orb_oracle_get_level(self.h, level, _p(out))
        return out

    def blurred(self, level):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        ok = self.L.orb_oracle_get_blurred(self.h, level, _p(out))
        return out if ok else None

    def candidates(self, level):
        cap = 1 << 16
        while True:
            out = np.zeros((cap, 3), np.int32)
            n = self.L.orb_oracle_get_candidates(self.h, level, _p(out), cap)
            if n <= cap:
                return out[:n].copy()
            cap = n

    def level_keypoints(self, level):
        cap = self.nfeatures + 64
        out = np.zeros(cap, KEYPOINT_DTYPE)
        n = self.L.orb_oracle_get_level_keypoints(self.h, level, _p(out), cap)
        assert n <= cap
        return out[:n].copy()


PROJ_QUERY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("radius", "<f4"), ("ur", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"),
                             ("blocks", "<i4"), ("angle", "<f4")])


def search_by_projection(kps, desc, imw, imh, queries, qdesc, mode, nnratio=0.8, th_high=100, check_ori=True, u_right=None, blocked=None):
    """Search core of SearchByProjection(Frame, MapPoints) (mode 0) / SearchByProjection(Current, Last) (mode 1) on flat data.
    Returns (nmatches, feature_query int32[n])."""
    kps = np.ascontiguousarray(kps)
    desc = np.ascontiguousarray(desc, np.uint8)
    queries = np.ascontiguousarray(queries, PROJ_QUERY_DTYPE)
    qdesc = np.ascontiguousarray(qdesc, np.uint8)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    bl = None if blocked is None else np.ascontiguousarray(blocked, np.uint8)
    fq = np.full(max(len(kps), 1), -1, np.int32)
    n = lib().orb_oracle_search_by_projection(_p(kps), _p(desc), None if ur is None else _p(ur), None if bl is None else _p(bl), len(kps), imw, imh,
                                              _p(queries), _p(qdesc), len(queries), mode, nnratio, th_high, int(check_ori), _p(fq))
    return n, fq[:len(kps)]


def project_points(projection_bytes, points, log_scale_factor):
    """The per-point projection of the pose-guided ORBmatcher members (orb_oracle.cpp: ProjectPoint) for one call: `projection_bytes` = the bytes of an
    orbhip_projection, `points` = orbhip_map_point records -> float32 [n, 8]: live, u, v, radius, ur, level, min_level, max_level."""
    pts = np.ascontiguousarray(points)
    assert pts.dtype.itemsize == 60
    buf = np.frombuffer(bytes(projection_bytes), np.uint8).copy()
    out = np.zeros((len(pts), 8), np.float32)
    lib().orb_oracle_project_points(_p(buf), _p(pts), len(pts), float(log_scale_factor), _p(out))
    return out


def predict_scale_of_ratio(ratio, log_scale_factor, nlevels):
    """MapPoint::PredictScale (MapPoint.cc:385-421) for a distance ratio, with this machine's logf"""
    return lib().orb_oracle_predict_scale_of_ratio(float(ratio), float(log_scale_factor), int(nlevels))


def stereo_matches(left, right, mbf, mb):
    """Frame::ComputeStereoMatches on the last extract() of two OracleExtractor instances: (mvuRight, mvDepth) float32[N]."""
    cap = left.nfeatures + 3 * left.nlevels + 64
    u = np.zeros(cap, np.float32)
    d = np.zeros(cap, np.float32)
    n = left.L.orb_oracle_stereo_matches(left.h, right.h, mbf, mb, _p(u), _p(d), cap)
    assert n <= cap
    return u[:n].copy(), d[:n].copy()


def resize(src, dw, dh):
    out = np.zeros((dh, dw), np.uint8)
    lib().orb_oracle_resize(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(out), dw, dh)
    return out


def blur(src, round_mode=0):
    out = np.zeros(src.shape, np.uint8)
    lib().orb_oracle_blur(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(out), round_mode)
    return out


def cvt_gray(src, rgb=True):
    """cv::cvtColor(src, RGB2GRAY/BGR2GRAY/RGBA2GRAY/BGRA2GRAY) for [H,W,3|4] uint8 (Tracking.cc:172-198)."""
    src = np.ascontiguousarray(src, np.uint8)
    h, w, ch = src.shape
    out = np.zeros((h, w), np.uint8)
    lib().orb_oracle_cvt_gray(_p(src), w, h, src.strides[0], ch, int(rgb), _p(out), w)
    return out


def _camera(camera):
    """(fx, fy, cx, cy, k1, k2, p1, p2[, k3]) -> K4, D5 float arrays (mK / mDistCoef, Tracking.cc:60-82)"""
    c = [float(v) for v in camera]
    assert len(c) in (8, 9)
    return np.array(c[:4], np.float32), np.array((c[4:] + [0.0])[:5], np.float32)


def undistort_points(camera, xy):
    """cv::undistortPoints(xy, xy, K, D, Mat(), K) for [n,2] float points (Frame.cc:421)"""
    K4, D5 = _camera(camera)
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.zeros_like(xy)
    lib().orb_oracle_undistort_points(_p(K4), _p(D5), _p(xy), len(xy), _p(out))
    return out


def image_bounds(camera, w, h):
    """Frame::ComputeImageBounds (Frame.cc:436-464): (mnMinX, mnMinY, mnMaxX, mnMaxY)"""
    K4, D5 = _camera(camera)
    out = np.zeros(4, np.float32)
    lib().orb_oracle_image_bounds(_p(K4), _p(D5), w, h, _p(out))
    return out


def undistort_keypoints(camera, kps):
    """Frame::UndistortKeyPoints (Frame.cc:404-434): mvKeysUn from mvKeys"""
    K4, D5 = _camera(camera)
    un = np.array(kps, copy=True)
    if D5[0] != 0.0 and len(un):
        xy = undistort_points(camera, np.stack([kps["x"], kps["y"]], axis=1))
        un["x"], un["y"] = xy[:, 0], xy[:, 1]
    return un


class image_bounds_set:
    """with image_bounds_set(b): the frames the matcher entry points build inside use the bounds b of a distorted camera"""

    def __init__(self, bounds):
        self.b = None if bounds is None else np.ascontiguousarray(bounds, np.float32)

    def __enter__(self):
        lib().orb_oracle_set_image_bounds(_p(self.b))

    def __exit__(self, *a):
        lib().orb_oracle_set_image_bounds(None)


def remap(src, map_x, map_y):
    """cv::remap(src, dst, map_x, map_y, INTER_LINEAR) with CV_32FC1 maps, constant 0 border (stereo_euroc.cc:136-137)"""
    src = np.ascontiguousarray(src, np.uint8)
    mx = np.ascontiguousarray(map_x, np.float32); my = np.ascontiguousarray(map_y, np.float32)
    assert mx.shape == my.shape
    out = np.zeros(mx.shape, np.uint8)
    lib().orb_oracle_remap(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(mx), _p(my), mx.shape[1], _p(out), mx.shape[1], mx.shape[0], mx.shape[1])
    return out


def stereo_from_rgbd(keys, keys_un, depth_map, depth_factor, mbf):
    """Frame::ComputeStereoFromRGBD (Frame.cc:643-665) + the convertTo of Tracking::GrabImageRGBD; depth_map float32 or uint16 [H,W]"""
    keys = np.ascontiguousarray(keys); keys_un = np.ascontiguousarray(keys_un)
    dm = np.ascontiguousarray(depth_map)
    assert dm.dtype in (np.float32, np.uint16)
    u = np.zeros(len(keys), np.float32); z = np.zeros(len(keys), np.float32)
    lib().orb_oracle_stereo_from_rgbd(_p(keys), _p(keys_un), len(keys), _p(dm), dm.shape[1], dm.shape[0], dm.strides[0], int(dm.dtype == np.uint16),
                                      float(depth_factor), float(mbf), _p(u), _p(z))
    return u, z


def gauss_kernel():
    k = np.zeros(7, np.int32)
    lib().orb_oracle_gauss_kernel(_p(k))
    return k


def fast(img, threshold, nms=True):
    cap = img.size
    out = np.zeros((cap, 3), np.int32)
    n = lib().orb_oracle_fast(_p(img), img.shape[1], img.shape[0], img.strides[0], threshold, int(nms), _p(out), cap)
    return out[:n].copy()


def fastatan2(y, x):
    return lib().orb_oracle_fastatan2(y, x)


def sincosf(a):
    s, c = C.c_float(), C.c_float()
    lib().orb_oracle_sincosf(a, C.byref(s), C.byref(c))
    return s.value, c.value


def distribute(xys, minX, maxX, minY, maxY, N):
    xys = np.ascontiguousarray(xys, np.int32)
    cap = max(N + 64, 4 * 64)
    out = np.zeros((cap, 3), np.int32)
    n = lib().orb_oracle_distribute(_p(xys), len(xys), minX, maxX, minY, maxY, N, _p(out), cap)
    assert n <= cap
    return out[:n].copy()


def hamming(a, b):
    return lib().orb_oracle_hamming(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)))


def bf_nn(q, db, fast=False):
    q = np.ascontiguousarray(q, np.uint8)
    db = np.ascontiguousarray(db, np.uint8)
    bi = np.zeros(len(q), np.int32)
    bd = np.zeros(len(q), np.int32)
    sd = np.zeros(len(q), np.int32)
    lib(fast).orb_oracle_bf_nn(_p(q), len(q), _p(db), len(db), _p(bi), _p(bd), _p(sd))
    return bi, bd, sd


def reloc_candidates(best_idx, best_dist, second_dist, row_keyframe, nkf, th_dist=50, ratio=0.75, top_k=10):
    """Definition (this repository's) of the relocalisation candidate list built from the brute-force descriptor DB
    (BASELINE.json config 5) where Tracking::Relocalization (Tracking.cc:1344-1348) asks KeyFrameDatabase::DetectRelocalizationCandidates
    (KeyFrameDatabase.cc:199-309) for key frames: a query descriptor votes for the key frame owning its nearest row iff it passes the
    matcher's own acceptance idiom (best <= th and best < ratio * second in float, ORBmatcher.cc:102-114); key frames ranked by votes,
    ties by ascending id, key frames without votes not listed.  Plain Python loop: the checker, not a product path."""
    votes = [0] * max(nkf, 1)
    for i in range(len(best_idx)):
        r = int(best_idx[i])
        if r < 0 or r >= len(row_keyframe):
            continue
        if int(best_dist[i]) > th_dist:
            continue
        if not (np.float32(best_dist[i]) < np.float32(ratio) * np.float32(second_dist[i])):
            continue
        votes[int(row_keyframe[r])] += 1
    order = sorted([k for k in range(nkf) if votes[k] > 0], key=lambda k: (-votes[k], k))[:top_k]
    return np.array(order, np.int32), np.array([votes[k] for k in order], np.int32)


def search_for_initialization(kps1, desc1, kps2, desc2, imw, imh, prev=None, window=100, nnratio=0.9, check_ori=True,
                              fast=False):
    """ORBmatcher(nnratio, check_ori).SearchForInitialization(F1, F2, prev, matches12, window).
    Returns (nmatches, matches12[int32 N1], prev_updated[N1,2] float32)."""
    kps1 = np.ascontiguousarray(kps1)
    kps2 = np.ascontiguousarray(kps2)
    desc1 = np.ascontiguousarray(desc1)
    desc2 = np.ascontiguousarray(desc2)
    if prev is None:
        prev = np.stack([kps1["x"], kps1["y"]], axis=1)
    prev = np.ascontiguousarray(prev, np.float32).copy()
    m12 = np.full(len(kps1), -1, np.int32)
    n = lib(fast).orb_oracle_search_for_initialization(_p(kps1), _p(desc1), len(kps1), _p(kps2), _p(desc2), len(kps2),
                                                       imw, imh, _p(prev), _p(m12), window, nnratio, int(check_ori))
    return n, m12, prev


def features_in_area(kps, imw, imh, x, y, r, min_level, max_level):
    kps = np.ascontiguousarray(kps)
    out = np.zeros(len(kps) + 1, np.int32)
    n = lib().orb_oracle_features_in_area(_p(kps), len(kps), imw, imh, x, y, r, min_level, max_level, _p(out), len(out))
    return out[:n].copy()


class OracleVocabulary:
    """DBoW2 vocabulary (TemplatedVocabulary<FORB>) restated: text loader, transform, scoring (oracle/bow_oracle.cpp)."""

    def __init__(self, path):
        self.h = lib().orb_oracle_voc_load(str(path).encode())
        if not self.h:
            raise ValueError(f"not a vocabulary text file: {path}")
        info = np.zeros(6, np.int32)
        lib().orb_oracle_voc_info(self.h, _p(info))
        self.k, self.L, self.scoring, self.weighting, self.nnodes, self.nwords = [int(v) for v in info]

    def close(self):
        if self.h:
            lib().orb_oracle_voc_free(self.h)
            self.h = None

    def transform_features(self, desc, levelsup):
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        w = np.zeros(n, np.uint32); v = np.zeros(n, np.float64); nd = np.zeros(n, np.uint32)
        lib().orb_oracle_voc_transform_features(self.h, _p(desc), n, levelsup, _p(w), _p(v), _p(nd))
        return w, v, nd

    def transform(self, desc, levelsup):
        """-> (bow ids, bow values, featvec node ids, featvec offsets [len+1], featvec feature indices), all in std::map order"""
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        bid = np.zeros(max(n, 1), np.uint32); bval = np.zeros(max(n, 1), np.float64)
        fnode = np.zeros(max(n, 1), np.uint32); foff = np.zeros(n + 2, np.int32); ffeat = np.zeros(max(n, 1), np.uint32)
        nfv = C.c_int(0)
        m = lib().orb_oracle_voc_transform(self.h, _p(desc), n, levelsup, _p(bid), _p(bval), C.byref(nfv), _p(fnode), _p(foff), _p(ffeat))
        q = nfv.value
        return bid[:m].copy(), bval[:m].copy(), fnode[:q].copy(), foff[:q + 1].copy(), ffeat[:foff[q]].copy()

    def score(self, id1, val1, id2, val2, scoring=None):
        return voc_score(self.scoring if scoring is None else scoring, id1, val1, id2, val2)


def voc_score(scoring, id1, val1, id2, val2):
    id1 = np.ascontiguousarray(id1, np.uint32); id2 = np.ascontiguousarray(id2, np.uint32)
    val1 = np.ascontiguousarray(val1, np.float64); val2 = np.ascontiguousarray(val2, np.float64)
    return lib().orb_oracle_voc_score(int(scoring), _p(id1), _p(val1), len(id1), _p(id2), _p(val2), len(id2))


def forb_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().orb_oracle_forb_distance(_p(a), _p(b))


def search_by_bow(mode, desc1, angle1, valid1, fv1, desc2, angle2, valid2, fv2, nnratio=0.7, check_ori=True):
    """ORBmatcher::SearchByBoW on flat data (mode 0: KeyFrame vs Frame, ORBmatcher.cc:159-288; mode 1: KeyFrame vs KeyFrame,
    :522-655).  fv = (node ids, offsets, feature indices) of a FeatureVector.  -> (nmatches, match12[n1])"""
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    angle1 = np.ascontiguousarray(angle1, np.float32); angle2 = np.ascontiguousarray(angle2, np.float32)
    valid1 = np.ascontiguousarray(valid1, np.uint8)
    valid2 = np.ones(len(desc2), np.uint8) if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
    f1 = [np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32)]
    f2 = [np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32)]
    m12 = np.full(len(desc1), -1, np.int32)
    n = lib().orb_oracle_search_by_bow(mode, _p(desc1), _p(angle1), _p(valid1), len(desc1), _p(f1[0]), _p(f1[1]), _p(f1[2]), len(f1[0]),
                                       _p(desc2), _p(angle2), _p(valid2), len(desc2), _p(f2[0]), _p(f2[1]), _p(f2[2]), len(f2[0]),
                                       nnratio, int(check_ori), _p(m12))
    return n, m12


def _kp4(k):
    return np.ascontiguousarray(np.stack([k["x"], k["y"], k["angle"], k["octave"].astype(np.float32)], axis=1), np.float32)


def search_for_triangulation(desc1, kps1, has_mp1, stereo1, fv1, desc2, kps2, has_mp2, stereo2, fv2, F12, ex, ey, scale_factors2, level_sigma2_2,
                             only_stereo=False, check_ori=True):
    """ORBmatcher::SearchForTriangulation (ORBmatcher.cc:657-823) on flat data.  kps = KEYPOINT arrays (mvKeysUn); -> (nmatches, match12)"""
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    k1, k2 = _kp4(kps1), _kp4(kps2)
    a = [np.ascontiguousarray(v, np.uint8) for v in (has_mp1, stereo1, has_mp2, stereo2)]
    f1 = [np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32)]
    f2 = [np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32)]
    F = np.ascontiguousarray(F12, np.float32).reshape(9)
    sc = np.ascontiguousarray(scale_factors2, np.float32); sg = np.ascontiguousarray(level_sigma2_2, np.float32)
    m12 = np.full(len(desc1), -1, np.int32)
    n = lib().orb_oracle_search_for_triangulation(_p(desc1), _p(k1), _p(a[0]), _p(a[1]), len(desc1), _p(f1[0]), _p(f1[1]), _p(f1[2]), len(f1[0]),
                                                  _p(desc2), _p(k2), _p(a[2]), _p(a[3]), len(desc2), _p(f2[0]), _p(f2[1]), _p(f2[2]), len(f2[0]),
                                                  _p(F), float(ex), float(ey), _p(sc), _p(sg), int(only_stereo), int(check_ori), _p(m12))
    return n, m12


BEST_QUERY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("radius", "<f4"), ("ur", "<f4"), ("level", "<i4")])


def search_best_in_window(kps, desc, imw, imh, inv_level_sigma2, queries, qdesc, chi2_gate, u_right=None):
    """Candidate loop of ORBmatcher::Fuse / SearchBySim3 on flat data -> (best_idx[nq], best_dist[nq])"""
    kps = np.ascontiguousarray(kps); desc = np.ascontiguousarray(desc, np.uint8)
    queries = np.ascontiguousarray(queries, BEST_QUERY_DTYPE); qdesc = np.ascontiguousarray(qdesc, np.uint8)
    inv = np.ascontiguousarray(inv_level_sigma2, np.float32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    bi = np.full(len(queries), -1, np.int32); bd = np.full(len(queries), 256, np.int32)
    lib().orb_oracle_search_best_in_window(_p(kps), _p(desc), None if ur is None else _p(ur), len(kps), imw, imh, _p(inv), _p(queries), _p(qdesc), len(queries),
                                           int(chi2_gate), _p(bi), _p(bd))
    return bi, bd
