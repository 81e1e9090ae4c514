"""The product against frames FROZEN FROM THE REFERENCE's own constructors (tests/golden/frames_ref.npz, written by tests/golden/
make_golden_frames.py with oracle/_ref/liborbslam_ref.so = src/Frame.cc + src/ORBextractor.cc + src/ORBmatcher.cc of the reference).
Needs neither the oracle nor the reference at run time: on the GPU box this compares the HIP path with the reference's outputs directly —
monocular frame of a distorted camera (mvKeys, mvKeysUn, mDescriptors, image bounds), RGB-D frame (mvuRight, mvDepth), stereo frame
(key points, descriptors, mvuRight, mvDepth from two extractors + ComputeStereoMatches)."""
import os
import sys

import numpy as np

import orb_slam2_amd
from orb_slam2_amd import synth

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_parity_stereo import stereo_pair  # noqa: E402

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames_ref.npz"))


def _inputs():
    w, h = 384, 288
    img = synth.frame(w, h, seed=77)
    rng = np.random.default_rng(77)
    depth = (np.float32(1.0) + np.float32(2.0) * rng.random((h, w)).astype(np.float32)).astype(np.float32)
    depth[rng.random((h, w)) < 0.2] = 0.0
    L, R = stereo_pair(400, 300, 8, 7)
    sums = [int(img.astype(np.uint64).sum()), int(np.float64(depth.astype(np.float64).sum()) * 1000), int(L.astype(np.uint64).sum()), int(R.astype(np.uint64).sum())]
    assert sums == G["input_checksums"].tolist(), "the seeded input generators drifted: regenerate tests/golden/frames_ref.npz"
    return img, depth, L, R


def test_frames_equal_the_reference_constructors(backend):
    img, depth, L, R = _inputs()
    cam = tuple(float(v) for v in G["camera"])
    ex = orb_slam2_amd.ORBextractor(400, 1.2, 8, 20, 7, 384, 288, library=backend)
    ex.set_camera(cam)
    ks, ds = ex.extract_batch([img])
    assert ks[0].tobytes() == G["mono_keys"].tobytes() and np.array_equal(ds[0], G["mono_desc"])
    assert ex.fetch_undistorted(1, [len(ks[0])])[0].tobytes() == G["mono_keys_un"].tobytes()
    assert ex.bounds().tobytes() == G["mono_bounds"].tobytes()
    u, z = ex.ComputeStereoFromRGBD([depth], 1.0, float(G["rgbd_mbf"]))
    nk = len(ks[0])
    assert u[0, :nk].tobytes() == G["rgbd_u_right"].tobytes() and z[0, :nk].tobytes() == G["rgbd_depth"].tobytes()
    ex.close()
    xl = orb_slam2_amd.ORBextractor(500, 1.2, 8, 20, 7, 400, 300, library=backend)
    xr = orb_slam2_amd.ORBextractor(500, 1.2, 8, 20, 7, 400, 300, library=backend)
    kl, dl = xl.extract_batch([L])
    xr.extract_batch([R])
    assert kl[0].tobytes() == G["stereo_keys"].tobytes() and np.array_equal(dl[0], G["stereo_desc"])
    bf, fx = np.float32(G["stereo_bf"]), np.float32(G["stereo_fx"])
    u, z = xl.ComputeStereoMatches(xr, float(bf), float(bf / fx), nimg=1)
    ns = len(kl[0])
    assert u[0, :ns].tobytes() == G["stereo_u_right"].tobytes() and z[0, :ns].tobytes() == G["stereo_depth"].tobytes()
    xl.close(); xr.close()


def test_native_flags_build_fixture(backend):
    """tests/golden/extract_native_flags_640x480.npz: a frame on which the reference's ORBextractor.cc built with its own flags (FMA
    contraction) and the canonical two-rounding build disagree in a descriptor bit.  fp_contract = 1 reproduces the former, 0 the latter."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "extract_native_flags_640x480.npz"))
    w, h, n, nl = [int(v) for v in g["config"]]
    img = synth.frame(w, h, seed=int(g["seed"]))
    ex = orb_slam2_amd.ORBextractor(n, float(g["scale"]), nl, 20, 7, w, h, library=backend)
    assert not np.array_equal(g["descriptors"], g["canonical_descriptors"])
    for mode, want in ((1, g["descriptors"]), (0, g["canonical_descriptors"])):
        ex.SetFpContract(mode)
        k, d = ex(img)
        assert k.tobytes() == g["keypoints"].tobytes() and np.array_equal(d, want), f"fp_contract {mode}"
    ex.close()
