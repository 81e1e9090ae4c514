"""ctypes binding of oracle/_ref/liborbslam_ref.so: the REFERENCE's own src/Frame.cc + src/ORBmatcher.cc (+ ORBextractor.cc),
compiled from /root/reference by `make -C oracle ref` (OpenCV image primitives = the oracle's restatements, MapPoint / KeyFrame
accessors = getters in the wrapper).  Pins the matcher / stereo / feature-grid restatements of orb_oracle.cpp against the
reference's real code.  Test infrastructure."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "_ref", "liborbslam_ref.so")
FAST_PATH = os.path.join(HERE, "_ref", "liborbslam_ref_fast.so")      # -O3 timing build of the same sources (bench.py cpu_baseline)
_use_fast = False


def use_fast_build(on=True):
    """Select the timing build; call before the first frame is made in this process."""
    global _use_fast, _lib
    assert _lib is None or _use_fast == on, "library already loaded"
    _use_fast = on

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
_lib = None


def available():
    return os.path.exists(PATH)

def _locked_make(args):
    """`make` under an exclusive file lock: pytest-xdist workers (and parallel tools) reach these builds at the same moment."""
    import fcntl
    import subprocess
    import tempfile
    if os.environ.get("ORBHIP_NO_MAKE"):      # sanitizer runs (tools/sanitize_concurrency.sh builds everything first): fork() under a preloaded ThreadSanitizer runtime can deadlock
        return
    with open(os.path.join(tempfile.gettempdir(), "orbhip_test_make.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            subprocess.check_call(["make"] + list(args))
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)



def build():
    import subprocess
    if os.path.isdir("/root/reference/src"):
        _locked_make(["-C", HERE, "-s", "ref"])
    return available()


NATIVE_PATH = os.path.join(HERE, "_ref", "liborbslam_ref_native.so")   # the same sources with the reference's own flags (-O3 -march=native: FMA contraction); CPU only
_native = None


def build_native():
    if os.path.isdir("/root/reference/src"):
        _locked_make(["-C", HERE, "-s", "ref_native_slam"])
    return os.path.exists(NATIVE_PATH)


DROPIN_NATIVE_PATH = os.path.join(HERE, "_ref", "liborbslam_dropin_full_native.so")   # the all-steps drop-in build with the same flags (CPU emulation of the kernels)
_dropin_native = None


def build_dropin_native():
    if os.path.isdir("/root/reference/src"):
        _locked_make(["-C", HERE, "-s", "dropin_native"])
    return os.path.exists(DROPIN_NATIVE_PATH)


def dropin_native_lib():
    global _dropin_native
    if _dropin_native is None:
        _dropin_native = _bind(C.CDLL(DROPIN_NATIVE_PATH))
    return _dropin_native


def native_lib():
    global _native
    if _native is None:
        _native = _bind(C.CDLL(NATIVE_PATH))
    return _native


DROPIN_PATH = os.path.join(HERE, "_ref", "liborbslam_dropin.so")      # reference Frame.cc / ORBmatcher.cc calling THIS repo's drop-in ORBextractor class
_dropin = None


def build_dropin():
    """reference callers + drop-in extractor class on the CPU emulation of the kernels (needs /root/reference and tests/emu/liborbhip_emu.so)"""
    import subprocess
    if os.path.isdir("/root/reference/src"):
        _locked_make(["-C", HERE, "-s", "dropin"])
    return os.path.exists(DROPIN_PATH) and os.path.exists(DROPIN_FULL_PATH)


def dropin_lib():
    global _dropin
    if _dropin is None:
        _dropin = _bind(C.CDLL(DROPIN_PATH))
    return _dropin


DROPIN_FULL_PATH = os.path.join(HERE, "_ref", "liborbslam_dropin_full.so")   # + Frame's stereo / undistortion / RGB-D members forwarded to the extractor
_dropin_full = None


def dropin_full_lib():
    """ORBSLAM_DROPIN_FULL_LIB: another build of the same library (the sanitizer builds of tools/sanitize_concurrency.sh)"""
    global _dropin_full
    if _dropin_full is None:
        _dropin_full = _bind(C.CDLL(os.environ.get("ORBSLAM_DROPIN_FULL_LIB") or DROPIN_FULL_PATH))
    return _dropin_full


# the same two builds linked to the real liborbhip.so (make dropin_gpu): built where /root/reference is mounted, run by the -m gpu tests
DROPIN_GPU_PATH = os.path.join(HERE, "_ref", "liborbslam_dropin_gpu.so")
DROPIN_FULL_GPU_PATH = os.path.join(HERE, "_ref", "liborbslam_dropin_full_gpu.so")
_dropin_gpu = {}


def build_dropin_gpu():
    import subprocess
    if os.path.isdir("/root/reference/src"):
        _locked_make(["-C", HERE, "-s", "dropin_gpu"])
    return os.path.exists(DROPIN_GPU_PATH) and os.path.exists(DROPIN_FULL_GPU_PATH)


def dropin_gpu_lib(full=False):
    path = DROPIN_FULL_GPU_PATH if full else DROPIN_GPU_PATH
    if path not in _dropin_gpu:
        _dropin_gpu[path] = _bind(C.CDLL(path))
    return _dropin_gpu[path]


def lib():
    global _lib
    if _lib is None:
        _lib = _bind(C.CDLL(FAST_PATH if _use_fast else PATH))
    return _lib


def _bind(L):
    if True:
        vp, i, f = C.c_void_p, C.c_int, C.c_float
        L.orbslam_ref_frame_mono.restype = vp
        L.orbslam_ref_frame_mono.argtypes = [vp, i, i, i, i, f, i, i, i, f, f, f, f, f, f, i]
        L.orbslam_ref_frame_mono_dist.restype = vp
        L.orbslam_ref_frame_mono_dist.argtypes = [vp, i, i, i, i, f, i, i, i, f, f, f, f, vp, i, f, f, i]
        L.orbslam_ref_frame_rgbd.restype = vp
        L.orbslam_ref_frame_rgbd.argtypes = [vp, vp, i, i, i, i, f, i, i, i, f, f, f, f, vp, i, f, f, i]
        L.orbslam_ref_frame_compute_bow.argtypes = [vp, C.c_char_p, vp, vp, vp, vp, vp, vp, vp]
        L.orbslam_ref_frame_bounds.argtypes = [vp]
        L.orbslam_ref_frame_stereo.restype = vp
        L.orbslam_ref_frame_stereo.argtypes = [vp, vp, i, i, i, i, f, i, i, i, f, f, f, f, f, f, i]
        L.orbslam_ref_frame_delete.argtypes = [vp]
        L.orbslam_ref_last_call_ms.restype = C.c_double
        L.orbslam_ref_last_call_lib_ms.restype = C.c_double
        L.orbslam_ref_frame_stereo_matches_again.argtypes = [vp, vp, vp]
        L.orbslam_ref_frame_n.argtypes = [vp]
        L.orbslam_ref_frame_get.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orbslam_ref_features_in_area.argtypes = [vp, f, f, f, i, i, vp, i]
        L.orbslam_ref_search_for_initialization.argtypes = [vp, vp, vp, vp, i, f, i]
        L.orbslam_ref_descriptor_distance.argtypes = [vp, vp]
        L.orbslam_ref_search_by_projection_points.argtypes = [vp, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f, f, vp]
        L.orbslam_ref_search_by_projection_last.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f, i, f, i, vp]
        L.orbslam_ref_search_by_bow.argtypes = [i, vp, vp, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, vp, i, f, i, vp]
        L.orbslam_ref_search_for_triangulation.argtypes = [vp, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, i, vp, vp, i, i, vp]
        L.orbslam_ref_fuse.argtypes = [vp, vp, i, vp, vp, vp, vp, vp, vp, vp, f, vp]
        L.orbslam_ref_fuse_sim3.argtypes = [vp, vp, i, vp, vp, vp, vp, vp, vp, f, vp]
        L.orbslam_ref_search_by_projection_kf.argtypes = [vp, vp, i, vp, vp, vp, vp, vp, vp, i, vp]
        L.orbslam_ref_search_by_projection_reloc.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f, i, f, i, vp]
        L.orbslam_ref_search_by_sim3.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f, vp]
        L.orbslam_ref_tracking_loop.argtypes = [i, vp, vp, i, i, i, i, f, i, i, i, f, f, f, f, f, f, vp, vp, i, i]
        L.orbslam_ref_loop_get.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.orbslam_ref_local_mapping_loops.argtypes = [i, vp, vp, vp, C.c_char_p, f, f, i, vp, vp, vp, vp, vp]
        L.orbslam_ref_local_mapping_ms.argtypes = [vp, vp]
        L.orbslam_ref_sequence_loop.argtypes = [i, i, vp, vp, i, i, i, i, f, i, i, i, f, f, f, f, vp, i, f, f, vp, vp, i, i, C.c_char_p, i]
        L.orbslam_ref_loop_bow_hash.argtypes = [i]
        L.orbslam_ref_loop_bow_hash.restype = C.c_uint64
        L.orbslam_ref_concurrency.argtypes = [i, i, i, C.c_uint, i, vp, vp, i, i, i, i, f, i, i, i, f, f, f, f, f, f, vp, vp, i, vp, i, vp, i, vp, vp, vp, vp]
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class RefFrame:
    """ORB_SLAM2::Frame built by the reference's own constructor."""
    _geometry = None                 # of the default library (tests reset it to force ComputeImageBounds)
    _geometry_other = {}             # of other builds of the same sources (the drop-in build), by handle

    def __init__(self, img, right=None, nfeatures=1000, scale=1.2, nlevels=8, ini_th=20, min_th=7, fx=1.0, fy=1.0, cx=0.0, cy=0.0, bf=40.0, th_depth=35.0,
                 dist=None, depth=None, library=None):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        geom = (w, h, fx, fy, cx, cy, None if dist is None else tuple(float(v) for v in dist))
        self.L = L = lib() if library is None else library
        if library is None:
            new = RefFrame._geometry != geom
            RefFrame._geometry = geom
        else:
            new = RefFrame._geometry_other.get(id(L)) != geom
            RefFrame._geometry_other[id(L)] = geom
        if depth is not None:                 # RGB-D sensor: CV_32F depth map (already converted, Tracking.cc:226-227)
            assert right is None
            d = np.ascontiguousarray(np.zeros(4) if dist is None else dist, np.float32)
            dm = np.ascontiguousarray(depth, np.float32)
            assert dm.shape == (h, w)
            self.h = L.orbslam_ref_frame_rgbd(_p(img), _p(dm), w, h, img.strides[0], nfeatures, scale, nlevels, ini_th, min_th, fx, fy, cx, cy, _p(d), len(d), bf, th_depth, int(new))
        elif dist is not None:                # distorted mono / RGB-D camera (mDistCoef, Tracking.cc:70-82)
            assert right is None
            d = np.ascontiguousarray(dist, np.float32)
            self.h = L.orbslam_ref_frame_mono_dist(_p(img), w, h, img.strides[0], nfeatures, scale, nlevels, ini_th, min_th, fx, fy, cx, cy, _p(d), len(d), bf, th_depth, int(new))
        elif right is None:
            self.h = L.orbslam_ref_frame_mono(_p(img), w, h, img.strides[0], nfeatures, scale, nlevels, ini_th, min_th, fx, fy, cx, cy, bf, th_depth, int(new))
        else:
            right = np.ascontiguousarray(right, np.uint8)
            self.h = L.orbslam_ref_frame_stereo(_p(img), _p(right), w, h, img.strides[0], nfeatures, scale, nlevels, ini_th, min_th, fx, fy, cx, cy, bf, th_depth, int(new))
        self.N = L.orbslam_ref_frame_n(self.h)
        self.keys = np.zeros(self.N, KEYPOINT_DTYPE); self.keys_un = np.zeros(self.N, KEYPOINT_DTYPE)
        self.desc = np.zeros((self.N, 32), np.uint8); self.u_right = np.zeros(self.N, np.float32); self.depth = np.zeros(self.N, np.float32)
        L.orbslam_ref_frame_get(self.h, _p(self.keys), _p(self.keys_un), _p(self.desc), _p(self.u_right), _p(self.depth))

    @staticmethod
    def bounds(library=None):
        """(mnMinX, mnMinY, mnMaxX, mnMaxY): the static image bounds the first Frame of the current geometry computed"""
        out = np.zeros(4, np.float32)
        (lib() if library is None else library).orbslam_ref_frame_bounds(_p(out))
        return out

    def compute_bow(self, voc_path):
        """Frame::ComputeBoW with the vocabulary file -> (bow ids, bow values, fv nodes, fv offsets, fv features), map order"""
        n = max(self.N, 1)
        bid = np.zeros(n, np.uint32); bval = np.zeros(n, np.float64); nb = C.c_int()
        fnode = np.zeros(n, np.uint32); foff = np.zeros(n + 1, np.int32); ffeat = np.zeros(n, np.uint32); nf = C.c_int()
        rc = self.L.orbslam_ref_frame_compute_bow(self.h, str(voc_path).encode(), _p(bid), _p(bval), C.byref(nb), _p(fnode), _p(foff), _p(ffeat), C.byref(nf))
        assert rc == 0, "vocabulary not loaded"
        return bid[:nb.value].copy(), bval[:nb.value].copy(), fnode[:nf.value].copy(), foff[:nf.value + 1].copy(), ffeat[:foff[nf.value]].copy()

    def stereo_matches_again(self):
        """Frame::ComputeStereoMatches once more (this frame must be the last one its rig made) -> (mvuRight, mvDepth); timed (last_call_ms)"""
        u = np.zeros(self.N, np.float32); d = np.zeros(self.N, np.float32)
        self.L.orbslam_ref_frame_stereo_matches_again(self.h, _p(u), _p(d))
        return u, d

    def close(self):
        if self.h:
            self.L.orbslam_ref_frame_delete(self.h)
            self.h = None

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        out = np.zeros(max(self.N, 1), np.int32)
        n = self.L.orbslam_ref_features_in_area(self.h, x, y, r, min_level, max_level, _p(out), len(out))
        return out[:n].copy()


def last_call_ms(library=None):
    """wall time of the ORBmatcher / Frame member the last wrapper call on this thread made (the member alone)"""
    return float((lib() if library is None else library).orbslam_ref_last_call_ms())


def last_call_lib_ms(library=None):
    """... and the part of it spent inside liborbhip's entry points (0 in the all-reference build)"""
    return float((lib() if library is None else library).orbslam_ref_last_call_lib_ms())


def search_for_initialization(f1, f2, prev=None, window=100, nnratio=0.9, check_ori=True):
    prev = np.ascontiguousarray(np.stack([f1.keys_un["x"], f1.keys_un["y"]], axis=1) if prev is None else prev, np.float32).copy()
    m12 = np.full(f1.N, -1, np.int32)
    n = f1.L.orbslam_ref_search_for_initialization(f1.h, f2.h, _p(prev), _p(m12), window, nnratio, int(check_ori))
    return n, m12, prev


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().orbslam_ref_descriptor_distance(_p(a), _p(b))


def search_by_projection_points(frame, px, py, pxr, level, viewcos, inview, bad, nobs, desc, feature_state=None, th=1.0, nnratio=0.8):
    a = [np.ascontiguousarray(px, np.float32), np.ascontiguousarray(py, np.float32), np.ascontiguousarray(pxr, np.float32), np.ascontiguousarray(level, np.int32),
         np.ascontiguousarray(viewcos, np.float32), np.ascontiguousarray(inview, np.uint8), np.ascontiguousarray(bad, np.uint8), np.ascontiguousarray(nobs, np.int32),
         np.ascontiguousarray(desc, np.uint8)]
    st = None if feature_state is None else np.ascontiguousarray(feature_state, np.uint8)
    fq = np.full(frame.N, -1, np.int32)
    n = frame.L.orbslam_ref_search_by_projection_points(frame.h, len(a[0]), *[_p(v) for v in a], _p(st), th, nnratio, _p(fq))
    return n, fq


def search_by_projection_last(cur, last, has_point, X, Y, Z, desc, outlier=None, bad=None, cur_state=None, th=7.0, mono=True, nnratio=0.9, check_ori=True):
    hp = np.ascontiguousarray(has_point, np.uint8)
    X, Y, Z = [np.ascontiguousarray(v, np.float32) for v in (X, Y, Z)]
    desc = np.ascontiguousarray(desc, np.uint8)
    out = None if outlier is None else np.ascontiguousarray(outlier, np.uint8)
    bd = None if bad is None else np.ascontiguousarray(bad, np.uint8)
    st = None if cur_state is None else np.ascontiguousarray(cur_state, np.uint8)
    fq = np.full(cur.N, -1, np.int32)
    n = cur.L.orbslam_ref_search_by_projection_last(cur.h, last.h, _p(hp), _p(X), _p(Y), _p(Z), _p(desc), _p(out), _p(bd), _p(st), th, int(mono), nnratio, int(check_ori), _p(fq))
    return n, fq


def search_by_bow(mode, f1, has1, bad1, fv1, f2, has2, bad2, fv2, nnratio=0.7, check_ori=True):
    """ORBmatcher::SearchByBoW through real KeyFrame / Frame objects; fv = (node ids, offsets, feature indices)"""
    u8 = lambda a, n: np.zeros(n, np.uint8) if a is None else np.ascontiguousarray(a, np.uint8)
    has1, bad1, has2, bad2 = u8(has1, f1.N), u8(bad1, f1.N), u8(has2, f2.N), u8(bad2, f2.N)
    a1 = [np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32)]
    a2 = [np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32)]
    m12 = np.full(f1.N, -1, np.int32)
    n = f1.L.orbslam_ref_search_by_bow(mode, f1.h, _p(has1), _p(bad1), _p(a1[0]), _p(a1[1]), _p(a1[2]), len(a1[0]),
                                        f2.h, _p(has2), _p(bad2), _p(a2[0]), _p(a2[1]), _p(a2[2]), len(a2[0]), nnratio, int(check_ori), _p(m12))
    return n, m12


def search_for_triangulation(f1, has1, fv1, f2, has2, fv2, F12, t2w, only_stereo=False, check_ori=True):
    """ORBmatcher::SearchForTriangulation through real KeyFrame objects (KF1 at the origin, KF2 = [I | t2w])"""
    has1 = np.ascontiguousarray(has1, np.uint8); has2 = np.ascontiguousarray(has2, np.uint8)
    a1 = [np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32)]
    a2 = [np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32)]
    F = np.ascontiguousarray(F12, np.float32).reshape(9); t = np.ascontiguousarray(t2w, np.float32)
    m12 = np.full(f1.N, -1, np.int32)
    n = f1.L.orbslam_ref_search_for_triangulation(f1.h, _p(has1), _p(a1[0]), _p(a1[1]), _p(a1[2]), len(a1[0]), f2.h, _p(has2), _p(a2[0]), _p(a2[1]), _p(a2[2]), len(a2[0]),
                                                   _p(F), _p(t), int(only_stereo), int(check_ori), _p(m12))
    return n, m12


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def set_test_poses(A=None, B=None, s12=1.0, R12=None, t12=None, library=None):
    """Poses of the member-level calls below (orbslam_ref_set_test_poses): A = 4x4 pose of the frame / key frame searched, B = of the other one, the similarity handed
    to SearchBySim3.  None restores identity.  Per library (the all-reference build and a drop-in build are two shared objects): call it on both."""
    L = library or lib()
    L.orbslam_ref_set_test_poses.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    if A is None:
        L.orbslam_ref_set_test_poses(None, None, 1.0, None, None)
        return
    a = _f32(A).reshape(16); b = a if B is None else _f32(B).reshape(16)
    r = None if R12 is None else _f32(R12).reshape(9); t = None if t12 is None else _f32(t12).reshape(3)
    L.orbslam_ref_set_test_poses(_p(a), _p(b), float(s12), None if r is None else _p(r), None if t is None else _p(t))


def is_in_frustum(frame, X, Y, Z, level, viewing_cos_limit=0.5):
    """Frame::isInFrustum for every point under the frame's test pose A -> float32 [nq, 6]: in view, mTrackProjX, mTrackProjY, mTrackProjXR, level, mTrackViewCos"""
    nq = len(X)
    out = np.zeros((nq, 6), np.float32)
    frame.L.orbslam_ref_is_in_frustum.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    frame.L.orbslam_ref_is_in_frustum(frame.h, nq, _p(_f32(X)), _p(_f32(Y)), _p(_f32(Z)), _p(np.ascontiguousarray(level, np.int32)), viewing_cos_limit, _p(out))
    return out


def gemm_mode(library=None):
    """how the drop-in ORBmatcher.cc of `library` found its cv::Mat algebra to round R*x+t (-1: the build has no drop-in matcher)"""
    L = library or lib()
    L.orbslam_ref_gemm_mode.restype = C.c_int
    return L.orbslam_ref_gemm_mode()


def _u8(a, n):
    return np.zeros(n, np.uint8) if a is None else np.ascontiguousarray(a, np.uint8)


def fuse(frame, kf_state, X, Y, Z, level, nobs, bad, desc, th=3.0):
    """ORBmatcher::Fuse(pKF, vpMapPoints, th) with the key frame (made of `frame`) at the origin -> (nFused, best_idx[nq])"""
    nq = len(X)
    best = np.full(nq, -1, np.int32)
    a = [_f32(X), _f32(Y), _f32(Z), np.ascontiguousarray(level, np.int32), np.ascontiguousarray(nobs, np.int32), _u8(bad, nq), np.ascontiguousarray(desc, np.uint8)]
    st = _u8(kf_state, frame.N)
    n = frame.L.orbslam_ref_fuse(frame.h, _p(st), nq, *[_p(v) for v in a], th, _p(best))
    return n, best


def fuse_sim3(frame, kf_state, X, Y, Z, level, bad, desc, th=4.0):
    """ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) with Scw = identity -> (nFused, key point per candidate point)"""
    nq = len(X)
    best = np.full(nq, -1, np.int32)
    a = [_f32(X), _f32(Y), _f32(Z), np.ascontiguousarray(level, np.int32), _u8(bad, nq), np.ascontiguousarray(desc, np.uint8)]
    n = frame.L.orbslam_ref_fuse_sim3(frame.h, _p(_u8(kf_state, frame.N)), nq, *[_p(v) for v in a], th, _p(best))
    return n, best


def search_by_projection_kf(frame, matched_state, X, Y, Z, level, bad, desc, th=10):
    nq = len(X)
    fq = np.full(frame.N, -1, np.int32)
    a = [_f32(X), _f32(Y), _f32(Z), np.ascontiguousarray(level, np.int32), _u8(bad, nq), np.ascontiguousarray(desc, np.uint8)]
    n = frame.L.orbslam_ref_search_by_projection_kf(frame.h, _p(_u8(matched_state, frame.N)), nq, *[_p(v) for v in a], int(th), _p(fq))
    return n, fq


def search_by_projection_reloc(cur, kf_frame, has_point, X, Y, Z, level, bad, found, desc, cur_state, th=10.0, orb_dist=100, nnratio=0.9, check_ori=True):
    n1 = kf_frame.N
    fq = np.full(cur.N, -1, np.int32)
    a = [_u8(has_point, n1), _f32(X), _f32(Y), _f32(Z), np.ascontiguousarray(level, np.int32), _u8(bad, n1), _u8(found, n1), np.ascontiguousarray(desc, np.uint8), _u8(cur_state, cur.N)]
    n = cur.L.orbslam_ref_search_by_projection_reloc(cur.h, kf_frame.h, *[_p(v) for v in a], th, int(orb_dist), nnratio, int(check_ori), _p(fq))
    return n, fq


def search_by_sim3(f1, has1, X1, Y1, Z1, level1, desc1, f2, has2, X2, Y2, Z2, level2, desc2, already12=None, th=7.5):
    m12 = np.full(f1.N, -1, np.int32)
    al = np.full(f1.N, -1, np.int32) if already12 is None else np.ascontiguousarray(already12, np.int32)
    a1 = [_u8(has1, f1.N), _f32(X1), _f32(Y1), _f32(Z1), np.ascontiguousarray(level1, np.int32), np.ascontiguousarray(desc1, np.uint8)]
    a2 = [_u8(has2, f2.N), _f32(X2), _f32(Y2), _f32(Z2), np.ascontiguousarray(level2, np.int32), np.ascontiguousarray(desc2, np.uint8)]
    n = f1.L.orbslam_ref_search_by_sim3(f1.h, *[_p(v) for v in a1], f2.h, *[_p(v) for v in a2], _p(al), th, _p(m12))
    return n, m12


def _collect_loop(L, n, capture):
    out = []
    for k in range(n):
        cnt = np.zeros(8, np.int32); ms = np.zeros(6, np.float64)
        L.orbslam_ref_loop_get(k, _p(cnt), _p(ms), None, None, None, None, None, None, None)
        fr = LoopFrame()
        for name, v in zip(LoopFrame.FIELDS + ("n_extra",), cnt):
            setattr(fr, name, int(v))
        fr.bow_hash = int(L.orbslam_ref_loop_bow_hash(k))
        fr.ms, fr.ms_ctor, fr.ms_motion, fr.ms_local = (float(v) for v in ms[:4])
        N = fr.N if capture else 0
        fr.keys = np.zeros(N, KEYPOINT_DTYPE); fr.keys_un = np.zeros(N, KEYPOINT_DTYPE); fr.desc = np.zeros((N, 32), np.uint8)
        fr.u_right = np.zeros(N, np.float32); fr.depth = np.zeros(N, np.float32); fr.mp_motion = np.full(N, -1, np.int32); fr.mp_final = np.full(N, -1, np.int32)
        if N:
            L.orbslam_ref_loop_get(k, None, None, _p(fr.keys), _p(fr.keys_un), _p(fr.desc), _p(fr.u_right), _p(fr.depth), _p(fr.mp_motion), _p(fr.mp_final))
        out.append(fr)
    return out


def sequence_loop(sensor, images, gt_depth, Tcw, Tpred, nfeatures, fx, fy, cx, cy, bf, th_depth, voc_path, dist=None, scale=1.2, nlevels=8, ini_th=20, min_th=7, kf_every=5, lost_every=0,
                  capture=True, library=None):
    """Tracking's monocular (sensor "mono") / RGB-D ("rgbd") matcher sequences with relocalisation through the reference's own Frame.cc / ORBmatcher.cc
    (orbslam_ref_sequence_loop in orbslam_ref_wrap.cpp).  -> list of LoopFrame; `used_wide` carries the frame's mode (10 / 11 / 12: initialisation without
    / with the initial frame / map created; 2 TrackReferenceKeyFrame; 0 / 1 TrackWithMotionModel (1: the 2*th retry); 4 Relocalization), `n_motion` the
    first matcher's return value, `n_extra` Relocalization's two projection searches, `bow_hash` the frame's bag of words where one was computed."""
    L = lib() if library is None else library
    n = len(images)
    images = [np.ascontiguousarray(a, np.uint8) for a in images]; gt = [np.ascontiguousarray(a, np.float32) for a in gt_depth]
    h, w = images[0].shape
    ip = (C.c_void_p * n)(*[a.ctypes.data for a in images]); dp = (C.c_void_p * n)(*[a.ctypes.data for a in gt])
    tc = np.ascontiguousarray(np.stack(Tcw), np.float32); tp = np.ascontiguousarray(np.stack(Tpred), np.float32)
    d = np.zeros(0, np.float32) if dist is None else np.ascontiguousarray(dist, np.float32)
    RefFrame._geometry = None
    RefFrame._geometry_other.clear()
    got = L.orbslam_ref_sequence_loop(dict(mono=0, rgbd=1)[sensor], n, ip, dp, w, h, w, nfeatures, scale, nlevels, ini_th, min_th, fx, fy, cx, cy, _p(d) if len(d) else None, len(d), bf, th_depth,
                                      _p(tp), _p(tc), kf_every, lost_every, None if voc_path is None else str(voc_path).encode(), int(capture))
    assert got == n, f"orbslam_ref_sequence_loop returned {got}"
    return _collect_loop(L, n, capture)


def local_mapping_loops(frames, F12, t2w, voc_path, fuse_th=3.0, point_depth=1.0):
    """LocalMapping's CreateNewMapPoints + SearchInNeighbors matcher loops on frames[0] (the current key frame) and its neighbours frames[1:] through the
    reference's own ORBmatcher.cc (orbslam_ref_local_mapping_loops): per-call loops in the all-reference / steps 1-3 builds, the single device passes of
    include/ORBmatcherBatch.h in the all-steps build.  -> (pairs per neighbour [(idx1, idx2) array], map point id per feature of every key frame, nFused, (tri_ms, fuse_ms))"""
    L = frames[0].L
    nn = len(frames) - 1
    cap = max(f.N for f in frames)
    hs = (C.c_void_p * (nn + 1))(*[f.h for f in frames])
    Fm = np.ascontiguousarray(F12, np.float32).reshape(nn, 9); tw = np.ascontiguousarray(t2w, np.float32).reshape(nn, 3)
    p1 = np.full((nn, cap), -1, np.int32); p2 = np.full((nn, cap), -1, np.int32); npairs = np.zeros(nn, np.int32); pts = np.full((nn + 1, cap), -1, np.int32); nf = C.c_int(0)
    rc = L.orbslam_ref_local_mapping_loops(nn, hs, _p(Fm), _p(tw), str(voc_path).encode(), fuse_th, point_depth, cap, _p(p1), _p(p2), _p(npairs), _p(pts), C.byref(nf))
    assert rc == 0, "orbslam_ref_local_mapping_loops failed (vocabulary?)"
    a, b = C.c_double(0), C.c_double(0)
    L.orbslam_ref_local_mapping_ms(C.byref(a), C.byref(b))
    return [np.stack([p1[i, :npairs[i]], p2[i, :npairs[i]]], axis=1) for i in range(nn)], pts, nf.value, (a.value, b.value)


class LoopFrame:
    """what one frame of tracking_loop left behind: features, stereo columns, MapPoint::mnId per feature after TrackWithMotionModel
    (mp_motion) and after SearchLocalPoints (mp_final), the counters and the wall time of the frame"""
    FIELDS = ("N", "n_motion", "used_wide", "n_to_match", "n_local", "n_new_points", "n_local_points")

    def same(self, o):
        return (all(getattr(self, k) == getattr(o, k) for k in self.FIELDS) and getattr(self, "n_extra", 0) == getattr(o, "n_extra", 0) and getattr(self, "bow_hash", 0) == getattr(o, "bow_hash", 0)
                and self.keys.tobytes() == o.keys.tobytes() and self.keys_un.tobytes() == o.keys_un.tobytes()
                and np.array_equal(self.desc, o.desc) and self.u_right.tobytes() == o.u_right.tobytes() and self.depth.tobytes() == o.depth.tobytes()
                and np.array_equal(self.mp_motion, o.mp_motion) and np.array_equal(self.mp_final, o.mp_final))


def tracking_loop(lefts, rights, Tcw, Tpred, nfeatures, fx, fy, cx, cy, bf, th_depth, scale=1.2, nlevels=8, ini_th=20, min_th=7, kf_every=5, capture=True, library=None):
    """Tracking's per-frame sequence on a stereo stream through the reference's own Frame.cc / ORBmatcher.cc (orbslam_ref_tracking_loop in
    orbslam_ref_wrap.cpp); `library` = another build of the same sources (the drop-in build).  -> list of LoopFrame"""
    L = lib() if library is None else library
    n = len(lefts)
    lefts = [np.ascontiguousarray(a, np.uint8) for a in lefts]; rights = [np.ascontiguousarray(a, np.uint8) for a in rights]
    h, w = lefts[0].shape
    assert all(a.shape == (h, w) for a in lefts + rights)
    lp = (C.c_void_p * n)(*[a.ctypes.data for a in lefts]); rp = (C.c_void_p * n)(*[a.ctypes.data for a in rights])
    tc = np.ascontiguousarray(np.stack(Tcw), np.float32); tp = np.ascontiguousarray(np.stack(Tpred), np.float32)
    assert tc.shape == (n, 4, 4) and tp.shape == (n, 4, 4)
    RefFrame._geometry = None
    RefFrame._geometry_other.clear()
    got = L.orbslam_ref_tracking_loop(n, lp, rp, w, h, w, nfeatures, scale, nlevels, ini_th, min_th, fx, fy, cx, cy, bf, th_depth, _p(tp), _p(tc), kf_every, int(capture))
    assert got == n
    out = []
    for k in range(n):
        cnt = np.zeros(8, np.int32); ms = np.zeros(6, np.float64)
        L.orbslam_ref_loop_get(k, _p(cnt), _p(ms), None, None, None, None, None, None, None)
        fr = LoopFrame()
        for name, v in zip(LoopFrame.FIELDS, cnt):
            setattr(fr, name, int(v))
        fr.ms, fr.ms_ctor, fr.ms_motion, fr.ms_local, fr.ms_frustum, fr.ms_copy = (float(v) for v in ms)
        N = fr.N if capture else 0
        fr.keys = np.zeros(N, KEYPOINT_DTYPE); fr.keys_un = np.zeros(N, KEYPOINT_DTYPE); fr.desc = np.zeros((N, 32), np.uint8)
        fr.u_right = np.zeros(N, np.float32); fr.depth = np.zeros(N, np.float32); fr.mp_motion = np.full(N, -1, np.int32); fr.mp_final = np.full(N, -1, np.int32)
        if N:
            L.orbslam_ref_loop_get(k, None, None, _p(fr.keys), _p(fr.keys_un), _p(fr.desc), _p(fr.u_right), _p(fr.depth), _p(fr.mp_motion), _p(fr.mp_final))
        out.append(fr)
    return out


class ConcCall(C.Structure):
    """one matcher call of orbslam_ref_concurrency (struct ConcCall in orbslam_ref_wrap.cpp)"""
    _fields_ = [("fn", C.c_int32), ("i", C.c_int32 * 6), ("f", C.c_float * 2), ("p", C.c_void_p * 16)]


class ConcCalls:
    """A list of ConcCall that keeps every array it points to alive."""
    FN = dict(triangulation=0, fuse=1, bow=2, sim3=3, projection_kf=4, compute_bow=5, fuse_sim3=6)

    def __init__(self):
        self.calls, self.keep = [], []

    def _add(self, fn, ptrs, ints=(), floats=()):
        c = ConcCall()
        c.fn = self.FN[fn]
        for k, v in enumerate(ints):
            c.i[k] = int(v)
        for k, v in enumerate(floats):
            c.f[k] = float(v)
        for k, v in enumerate(ptrs):
            if isinstance(v, RefFrame):
                c.p[k] = v.h
            elif isinstance(v, bytes):
                self.keep.append(v); c.p[k] = C.cast(C.c_char_p(v), C.c_void_p)
            elif v is None:
                c.p[k] = None
            else:
                self.keep.append(v); c.p[k] = v.ctypes.data
        self.calls.append(c)

    @staticmethod
    def _fv(fv):
        return [np.ascontiguousarray(fv[0], np.uint32), np.ascontiguousarray(fv[1], np.int32), np.ascontiguousarray(fv[2], np.uint32)]

    def triangulation(self, f1, has1, fv1, f2, has2, fv2, F12, t2w, only_stereo=False, check_ori=True):
        a1, a2 = self._fv(fv1), self._fv(fv2)
        self._add("triangulation", [f1, _u8(has1, f1.N), *a1, f2, _u8(has2, f2.N), *a2, np.ascontiguousarray(F12, np.float32).reshape(9), np.ascontiguousarray(t2w, np.float32)],
                  ints=(len(a1[0]), len(a2[0]), int(only_stereo), int(check_ori)))

    def fuse(self, frame, kf_state, X, Y, Z, level, nobs, bad, desc, th=3.0):
        nq = len(X)
        self._add("fuse", [frame, _u8(kf_state, frame.N), _f32(X), _f32(Y), _f32(Z), np.ascontiguousarray(level, np.int32), np.ascontiguousarray(nobs, np.int32), _u8(bad, nq),
                           np.ascontiguousarray(desc, np.uint8)], ints=(nq,), floats=(th,))

    def bow(self, mode, f1, has1, bad1, fv1, f2, has2, bad2, fv2, nnratio=0.7, check_ori=True):
        a1, a2 = self._fv(fv1), self._fv(fv2)
        self._add("bow", [f1, _u8(has1, f1.N), _u8(bad1, f1.N), *a1, f2, _u8(has2, f2.N), _u8(bad2, f2.N), *a2], ints=(mode, len(a1[0]), len(a2[0]), int(check_ori)), floats=(nnratio,))

    def sim3(self, f1, has1, X1, Y1, Z1, level1, desc1, f2, has2, X2, Y2, Z2, level2, desc2, already12=None, th=7.5):
        al = np.full(f1.N, -1, np.int32) if already12 is None else np.ascontiguousarray(already12, np.int32)
        self._add("sim3", [f1, _u8(has1, f1.N), _f32(X1), _f32(Y1), _f32(Z1), np.ascontiguousarray(level1, np.int32), np.ascontiguousarray(desc1, np.uint8),
                           f2, _u8(has2, f2.N), _f32(X2), _f32(Y2), _f32(Z2), np.ascontiguousarray(level2, np.int32), np.ascontiguousarray(desc2, np.uint8), al], floats=(th,))

    def projection_kf(self, frame, matched_state, X, Y, Z, level, bad, desc, th=10):
        nq = len(X)
        self._add("projection_kf", [frame, _u8(matched_state, frame.N), _f32(X), _f32(Y), _f32(Z), np.ascontiguousarray(level, np.int32), _u8(bad, nq), np.ascontiguousarray(desc, np.uint8)],
                  ints=(nq, int(th)))

    def compute_bow(self, frame, voc_path):
        self._add("compute_bow", [frame, str(voc_path).encode() + b"\0"])

    def fuse_sim3(self, frame, kf_state, X, Y, Z, level, bad, desc, th=4.0):
        nq = len(X)
        self._add("fuse_sim3", [frame, _u8(kf_state, frame.N), _f32(X), _f32(Y), _f32(Z), np.ascontiguousarray(level, np.int32), _u8(bad, nq), np.ascontiguousarray(desc, np.uint8)],
                  ints=(nq,), floats=(th,))

    def array(self):
        return (ConcCall * max(len(self.calls), 1))(*self.calls)


def concurrency(library, threaded, iters, t_rounds, seed, lefts, rights, Tcw, Tpred, nfeatures, fx, fy, cx, cy, bf, th_depth, lcalls, ccalls,
                scale=1.2, nlevels=8, ini_th=20, min_th=7, kf_every=5):
    """orbslam_ref_concurrency: Tracking's stereo loop, LocalMapping's and LoopClosing's matcher calls - one after another on this thread (threaded = False) or
    on three threads at once.  -> (hashes of T's frames, of L's calls, of C's calls, T rounds run, T rounds that differed from the first)"""
    n = len(lefts)
    lefts = [np.ascontiguousarray(a, np.uint8) for a in lefts]; rights = [np.ascontiguousarray(a, np.uint8) for a in rights]
    h, w = lefts[0].shape
    lp = (C.c_void_p * n)(*[a.ctypes.data for a in lefts]); rp = (C.c_void_p * n)(*[a.ctypes.data for a in rights])
    tc = np.ascontiguousarray(np.stack(Tcw), np.float32); tp = np.ascontiguousarray(np.stack(Tpred), np.float32)
    hT = np.zeros(n, np.uint64); hL = np.zeros(iters, np.uint64); hC = np.zeros(iters, np.uint64); differing = C.c_int(0)
    la, ca = lcalls.array(), ccalls.array()
    rounds = library.orbslam_ref_concurrency(int(threaded), iters, t_rounds, seed, n, lp, rp, w, h, w, nfeatures, scale, nlevels, ini_th, min_th, fx, fy, cx, cy, bf, th_depth,
                                             _p(tp), _p(tc), kf_every, la, len(lcalls.calls), ca, len(ccalls.calls), _p(hT), _p(hL), _p(hC), C.byref(differing))
    return hT, hL, hC, rounds, differing.value
