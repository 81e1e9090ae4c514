"""ctypes binding of oracle/_ref/liborbextractor_ref.so: the REFERENCE's own src/ORBextractor.cc, compiled from /root/reference
by `make -C oracle ref` with the OpenCV image primitives replaced by the oracle's restatements (oracle/ref_shim/cv_image_shim.h).
Pins the oracle's (and thereby the HIP path's) extractor logic against the reference's real code.  Test infrastructure."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "_ref", "liborbextractor_ref.so")
NATIVE_PATH = os.path.join(HERE, "_ref", "liborbextractor_ref_native.so")     # the reference's own flags: -O3 -march=native, contraction on (make ref_native)
STOCK_PATH = os.path.join(HERE, "_ref", "liborbextractor_ref_stock.so")       # glibc's allocator instead of the bump arena (make ref_stock): H1 as a maintainer's binary has it
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
_lib = None


def available():
    return os.path.exists(PATH)


def build():
    import subprocess
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", HERE, "-s", "ref"])
    return available()


_native = None


def build_native():
    """-march=native: only meaningful (and only attempted) on the machine that runs it"""
    import subprocess
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", HERE, "-s", "ref_native"])
        return True
    return False


_stock = None


def build_stock():
    import subprocess
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", HERE, "-s", "ref_stock"])
    return os.path.exists(STOCK_PATH)


def lib(native=False, stock=False):
    global _lib, _native, _stock
    if stock:
        if _stock is None:
            _stock = _bind(C.CDLL(STOCK_PATH))
        return _stock
    if native:
        if _native is None:
            _native = _bind(C.CDLL(NATIVE_PATH))
        return _native
    if _lib is None:
        _lib = _bind(C.CDLL(PATH))
    return _lib


def _bind(L):
    if True:
        vp = C.c_void_p
        L.orbextractor_ref_new.restype = vp
        L.orbextractor_ref_new.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orbextractor_ref_delete.argtypes = [vp]
        L.orbextractor_ref_params.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.orbextractor_ref_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int]
        L.orbextractor_ref_level.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class RefExtractor:
    def __init__(self, nfeatures, scale, nlevels, ini_th, min_th, native=False, stock=False):
        self.nlevels = nlevels
        self.L = lib(native, stock)
        self.h = self.L.orbextractor_ref_new(nfeatures, scale, nlevels, ini_th, min_th)
        self.cap = nfeatures * 2 + 64

    def close(self):
        if self.h:
            self.L.orbextractor_ref_delete(self.h)
            self.h = None

    def params(self):
        n = self.nlevels
        f = np.zeros(n, np.int32); a, b, c, d = [np.zeros(n, np.float32) for _ in range(4)]; u = np.zeros(16, np.int32)
        self.L.orbextractor_ref_params(self.h, _p(f), _p(a), _p(b), _p(c), _p(d), _p(u))
        return {"features_per_level": f, "scale_factors": a, "inv_scale_factors": b, "sigma2": c, "inv_sigma2": d, "umax": u}

    def extract(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        k = np.zeros(self.cap, KEYPOINT_DTYPE); d = np.zeros((self.cap, 32), np.uint8)
        n = self.L.orbextractor_ref_extract(self.h, _p(img), img.shape[1], img.shape[0], img.strides[0], _p(k), _p(d), self.cap)
        assert n <= self.cap
        return k[:n].copy(), d[:n].copy()

    def level(self, l):
        w, h = C.c_int(0), C.c_int(0)
        if not self.L.orbextractor_ref_level(self.h, l, None, C.byref(w), C.byref(h)):
            return None
        out = np.zeros((h.value, w.value), np.uint8)
        self.L.orbextractor_ref_level(self.h, l, _p(out), C.byref(w), C.byref(h))
        return out
