"""Generates tests/golden/frames_ref.npz with the REFERENCE's own Frame constructors (oracle/_ref/liborbslam_ref.so = src/Frame.cc,
src/ORBextractor.cc, src/ORBmatcher.cc of /root/reference compiled by `make -C oracle ref`; run from the repo root where /root/reference
is mounted: python tests/golden/make_golden_frames.py).

Three frames as the reference builds them, frozen for machines without the reference (the GPU box):
  mono   Frame(imGray, ...) of a TUM1-distorted camera (scaled to 384x288): mvKeys, mvKeysUn, mDescriptors, the image bounds
  rgbd   Frame(imGray, imDepth, ...) of the same camera and image with a seeded CV_32F depth map: mvuRight, mvDepth
  stereo Frame(imLeft, imRight, ...) of a KITTI-like rectified pair (400x300, uniform 7 px disparity): mvKeys, mDescriptors, mvuRight, mvDepth
The image primitives inside that build are the oracle's restatements (OpenCV is not vendored); everything else is the reference's code."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import orbslam_ref as S  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402
from test_parity_stereo import stereo_pair  # noqa: E402

assert S.build(), "oracle/_ref/liborbslam_ref.so needs /root/reference"
here = os.path.dirname(os.path.abspath(__file__))
w, h, n = 384, 288, 400
camera = np.array([517.306408 * w / 640, 516.469215 * h / 480, 318.643040 * w / 640, 255.313989 * h / 480,
                   0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)
cam = [float(v) for v in camera]
img = synth.frame(w, h, seed=77)
rng = np.random.default_rng(77)
depth = (np.float32(1.0) + np.float32(2.0) * rng.random((h, w)).astype(np.float32)).astype(np.float32)
depth[rng.random((h, w)) < 0.2] = 0.0
mbf = np.float32(40.0)
S.RefFrame._geometry = None
m = S.RefFrame(img, nfeatures=n, fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3], dist=cam[4:])
bounds = S.RefFrame.bounds()
r = S.RefFrame(img, nfeatures=n, fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3], bf=float(mbf), dist=cam[4:], depth=depth)
assert r.keys.tobytes() == m.keys.tobytes() and r.keys_un.tobytes() == m.keys_un.tobytes()
sw, sh, sn = 400, 300, 500
L, R = stereo_pair(sw, sh, 8, 7)
fx, bf = np.float32(718.856), np.float32(386.1448)
S.RefFrame._geometry = None
s = S.RefFrame(L, R, nfeatures=sn, fx=float(fx), fy=float(fx), cx=607.1928, cy=185.2157, bf=float(bf))
# inputs are regenerated from their seeds by the test (synth.frame(384, 288, seed=77), default_rng(77) depth map, stereo_pair(400, 300, 8, 7));
# checksums guard against a generator that drifts
np.savez_compressed(os.path.join(here, "frames_ref.npz"), camera=camera, mono_keys=m.keys, mono_keys_un=m.keys_un, mono_desc=m.desc,
                    mono_bounds=bounds, rgbd_mbf=mbf, rgbd_u_right=r.u_right, rgbd_depth=r.depth,
                    stereo_fx=fx, stereo_bf=bf, stereo_keys=s.keys, stereo_desc=s.desc, stereo_u_right=s.u_right, stereo_depth=s.depth,
                    input_checksums=np.array([int(img.astype(np.uint64).sum()), int(np.float64(depth.astype(np.float64).sum()) * 1000), int(L.astype(np.uint64).sum()),
                                              int(R.astype(np.uint64).sum())], np.int64))
print("golden frames:", m.N, "mono key points (max shift %.1f px)," % float(np.abs(m.keys_un["x"] - m.keys["x"]).max()), int((r.depth > 0).sum()),
      "RGB-D depths,", s.N, "stereo key points,", int((s.depth > 0).sum()), "stereo depths")
S.RefFrame._geometry = None
