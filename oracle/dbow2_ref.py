"""ctypes binding of oracle/_ref/libdbow2_ref.so: the REFERENCE's own DBoW2 (Thirdparty/DBoW2 compiled from
/root/reference by `make -C oracle ref`).  Test infrastructure: pins the BoW restatement in orb_oracle.cpp and the HIP path."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "_ref", "libdbow2_ref.so")
_lib = None


def available():
    return os.path.exists(PATH)


def build():
    """Only possible where the reference sources are mounted."""
    import subprocess
    if os.path.isdir("/root/reference/Thirdparty/DBoW2"):
        subprocess.check_call(["make", "-C", HERE, "-s", "ref"])
    return available()


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(PATH)
        vp = C.c_void_p
        L.dbow2_ref_new.restype = vp
        L.dbow2_ref_delete.argtypes = [vp]
        L.dbow2_ref_load_text.argtypes = [vp, C.c_char_p]
        L.dbow2_ref_save_text.argtypes = [vp, C.c_char_p]
        L.dbow2_ref_create.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.dbow2_ref_size.argtypes = [vp]
        L.dbow2_ref_transform_features.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp]
        L.dbow2_ref_transform.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
        L.dbow2_ref_transform.restype = C.c_int
        L.dbow2_ref_score.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int]
        L.dbow2_ref_score.restype = C.c_double
        L.dbow2_ref_distance.argtypes = [vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class RefVocabulary:
    def __init__(self):
        self.h = lib().dbow2_ref_new()

    def close(self):
        if self.h:
            lib().dbow2_ref_delete(self.h)
            self.h = None

    def load_text(self, path):
        return bool(lib().dbow2_ref_load_text(self.h, path.encode()))

    def save_text(self, path):
        lib().dbow2_ref_save_text(self.h, path.encode())

    def create(self, desc_per_image, k, L, weighting=0, scoring=0, seed=1):
        counts = np.array([len(d) for d in desc_per_image], np.int32)
        allv = np.ascontiguousarray(np.concatenate(desc_per_image), np.uint8)
        lib().dbow2_ref_create(self.h, _p(allv), _p(counts), len(counts), k, L, weighting, scoring, seed)

    def size(self):
        return lib().dbow2_ref_size(self.h)

    def transform_features(self, desc, levelsup):
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        w = np.zeros(n, np.uint32); v = np.zeros(n, np.float64); nd = np.zeros(n, np.uint32)
        lib().dbow2_ref_transform_features(self.h, _p(desc), n, levelsup, _p(w), _p(v), _p(nd))
        return w, v, nd

    def transform(self, desc, levelsup):
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        bid = np.zeros(max(n, 1), np.uint32); bval = np.zeros(max(n, 1), np.float64)
        fnode = np.zeros(max(n, 1), np.uint32); foff = np.zeros(n + 2, np.int32); ffeat = np.zeros(max(n, 1), np.uint32)
        nfv = C.c_int(0)
        m = lib().dbow2_ref_transform(self.h, _p(desc), n, levelsup, _p(bid), _p(bval), C.byref(nfv), _p(fnode), _p(foff), _p(ffeat))
        q = nfv.value
        return bid[:m].copy(), bval[:m].copy(), fnode[:q].copy(), foff[:q + 1].copy(), ffeat[:foff[q]].copy()

    def score(self, id1, val1, id2, val2):
        id1 = np.ascontiguousarray(id1, np.uint32); id2 = np.ascontiguousarray(id2, np.uint32)
        val1 = np.ascontiguousarray(val1, np.float64); val2 = np.ascontiguousarray(val2, np.float64)
        return lib().dbow2_ref_score(self.h, _p(id1), _p(val1), len(id1), _p(id2), _p(val2), len(id2))


def distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().dbow2_ref_distance(_p(a), _p(b))
