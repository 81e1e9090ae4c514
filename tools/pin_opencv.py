#!/usr/bin/env python3
"""pin_opencv.py — closes the one hole the parity chain declares: the OpenCV boundary.

The reference calls seven OpenCV primitives on this path (unpinned dependency, CMakeLists.txt:31-37):
    cv::resize INTER_LINEAR        ORBextractor.cc:1120          cv::cvtColor *2GRAY            Tracking.cc:172-198
    cv::FAST (+ NMS)               ORBextractor.cc:809-815       cv::remap INTER_LINEAR         Examples/Stereo/stereo_euroc.cc:136-137
    cv::GaussianBlur 7x7, sigma 2  ORBextractor.cc:1086          cv::undistortPoints            Frame.cc:421, 450
    cv::fastAtan2                  ORBextractor.cc:103
oracle/orb_oracle.cpp restates each from OpenCV 3.2's published algorithm; the container this repository is built in has no OpenCV, so no byte
of a real one was ever compared.  Run this where `import cv2` works:

    python tools/pin_opencv.py                 # prints the OpenCV version and a per-primitive verdict, writes tests/golden/opencv_<version>.npz

It runs every primitive of the installed OpenCV on the synthetic set AND on the two natural photographs (tests/golden/natural_images.npz),
diffs against oracle/ primitive by primitive (bit for bit; for GaussianBlur it reports which of the two rounding modes of DESIGN.md H2 this build
is), and stores OpenCV's outputs.  tests/test_opencv_pin.py then replays every such file it finds against the oracle on any machine — a
one-command check that turns "parity unpinned at the OpenCV boundary" into a pinned fact for that OpenCV version.  Exit code 1 if a primitive differs.
OpenCV >= 3.4 replaced GaussianBlur's 8-bit path by a bit-exact fixed-point one and 4.x reworked resize's SIMD paths: a mismatch against such a
version is information about that version (ORB_SLAM2 pins none), not necessarily a defect of the restatement of 3.2."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def inputs():
    """name -> gray image; the same set tests/test_opencv_pin.py rebuilds (seeds only, plus the frozen photographs)"""
    from orb_slam2_amd import synth
    from oracle import orb_oracle as O
    z = np.load(os.path.join(ROOT, "tests", "golden", "natural_images.npz"))
    imgs = {"synth_320x240_s1": synth.frame(320, 240, seed=1), "synth_401x307_s2": synth.frame(401, 307, seed=2), "checker_320x240": synth.checkerboard(320, 240),
            "low_texture_320x240": synth.low_texture(320, 240)}
    rgb = {"china": np.ascontiguousarray(z["china_rgb"]), "flower": np.ascontiguousarray(z["flower_rgb"])}
    for k, v in rgb.items():
        imgs[k] = O.cvt_gray(v, True)
    return imgs, rgb


def pyramid_sizes(w, h, nlevels=4, sf=1.2):
    out, s = [], 1.0
    for _ in range(1, nlevels):
        s = np.float32(s * np.float32(sf))
        inv = np.float32(1.0) / s
        out.append((int(round(float(np.float32(w) * inv))), int(round(float(np.float32(h) * inv)))))       # cvRound((float)cols*scale), ORBextractor.cc:1112
    return out


def remap_maps(w, h):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    mx = (xx + np.float32(3.25) * np.sin(yy / np.float32(37.0))).astype(np.float32)
    my = (yy + np.float32(2.5) * np.cos(xx / np.float32(53.0)) - np.float32(1.125)).astype(np.float32)
    return mx, my


CAMERA = (517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314)        # Examples/Monocular/TUM1.yaml


def run_oracle(imgs, rgb):
    """every primitive of the oracle on the input set -> {key: array}; keys are what run_opencv writes too"""
    from oracle import orb_oracle as O
    out = {}
    for name, im in imgs.items():
        h, w = im.shape
        src = im
        for l, (dw, dh) in enumerate(pyramid_sizes(w, h), 1):
            src = O.resize(src, dw, dh)
            out[f"resize/{name}/L{l}"] = src
        for mode in (0, 1):
            out[f"blur{mode}/{name}"] = O.blur(im, mode)
        for th in (20, 7):
            f = O.fast(im, th, True)
            out[f"fast{th}/{name}"] = f[np.lexsort((f[:, 0], f[:, 1]))].astype(np.int32)
        mx, my = remap_maps(w, h)
        out[f"remap/{name}"] = O.remap(im, mx, my)
    for name, c in rgb.items():
        out[f"gray_rgb/{name}"] = O.cvt_gray(c, True)
        out[f"gray_bgr/{name}"] = O.cvt_gray(c, False)
        out[f"gray_rgba/{name}"] = O.cvt_gray(np.concatenate([c, np.full(c.shape[:2] + (1,), 255, np.uint8)], axis=2), True)
    ys, xs = np.meshgrid(np.linspace(-300, 300, 41, dtype=np.float32), np.linspace(-300, 300, 41, dtype=np.float32))
    out["fastatan2"] = np.array([O.fastatan2(float(y), float(x)) for y, x in zip(ys.ravel(), xs.ravel())], np.float32)
    rng = np.random.default_rng(4)
    pts = np.stack([rng.uniform(0, 640, 500), rng.uniform(0, 480, 500)], axis=1).astype(np.float32)
    out["undistort"] = O.undistort_points(CAMERA, pts)
    return out


def run_opencv(imgs, rgb):
    import cv2
    out = {}
    for name, im in imgs.items():
        h, w = im.shape
        src = im
        for l, (dw, dh) in enumerate(pyramid_sizes(w, h), 1):
            src = cv2.resize(src, (dw, dh), 0, 0, cv2.INTER_LINEAR)
            out[f"resize/{name}/L{l}"] = src
        out[f"blur/{name}"] = cv2.GaussianBlur(im, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
        for th in (20, 7):
            det = cv2.FastFeatureDetector_create(threshold=th, nonmaxSuppression=True)
            kp = det.detect(im, None)
            f = np.array([[int(k.pt[0]), int(k.pt[1]), int(k.response)] for k in kp], np.int32).reshape(-1, 3)
            out[f"fast{th}/{name}"] = f[np.lexsort((f[:, 0], f[:, 1]))]
        mx, my = remap_maps(w, h)
        out[f"remap/{name}"] = cv2.remap(im, mx, my, cv2.INTER_LINEAR)
    for name, c in rgb.items():
        out[f"gray_rgb/{name}"] = cv2.cvtColor(c, cv2.COLOR_RGB2GRAY)
        out[f"gray_bgr/{name}"] = cv2.cvtColor(c, cv2.COLOR_BGR2GRAY)
        out[f"gray_rgba/{name}"] = cv2.cvtColor(np.concatenate([c, np.full(c.shape[:2] + (1,), 255, np.uint8)], axis=2), cv2.COLOR_RGBA2GRAY)
    ys, xs = np.meshgrid(np.linspace(-300, 300, 41, dtype=np.float32), np.linspace(-300, 300, 41, dtype=np.float32))
    out["fastatan2"] = np.array([cv2.fastAtan2(float(y), float(x)) for y, x in zip(ys.ravel(), xs.ravel())], np.float32)
    rng = np.random.default_rng(4)
    pts = np.stack([rng.uniform(0, 640, 500), rng.uniform(0, 480, 500)], axis=1).astype(np.float32)
    K = np.array([[CAMERA[0], 0, CAMERA[2]], [0, CAMERA[1], CAMERA[3]], [0, 0, 1]], np.float32)
    D = np.array(CAMERA[4:], np.float32).reshape(-1, 1)
    out["undistort"] = cv2.undistortPoints(pts.reshape(-1, 1, 2), K, D, None, K).reshape(-1, 2).astype(np.float32)
    return out


def compare(cv, ora):
    """-> (verdicts {primitive: 'match' | 'differs: ...'}, blur_mode or None)"""
    verdict, blur_mode = {}, None
    groups = {}
    for key in cv:
        groups.setdefault(key.split("/")[0], []).append(key)
    for prim, keys in sorted(groups.items()):
        if prim == "blur":
            ok = {m: all(np.array_equal(cv[k], ora[k.replace("blur/", f"blur{m}/")]) for k in keys) for m in (0, 1)}
            blur_mode = 0 if ok[0] else 1 if ok[1] else None
            if blur_mode is None:
                worst = max(int(np.abs(cv[k].astype(int) - ora[k.replace("blur/", "blur0/")].astype(int)).max()) for k in keys)
                verdict[prim] = f"differs from BOTH rounding modes (max |diff| vs mode 0: {worst})"
            else:
                verdict[prim] = f"match (rounding mode {blur_mode}: {'generic C++ path' if blur_mode == 0 else 'SSE2 column filter'})"
            continue
        bad = []
        for k in keys:
            a, b = cv[k], ora[k]
            same = a.shape == b.shape and (a.tobytes() == b.tobytes() if a.dtype.kind == "f" else np.array_equal(a, b))
            if not same:
                bad.append(k + (f" (max |diff| {np.abs(a.astype(np.float64) - b.astype(np.float64)).max():g}, {int((a != b).sum())} of {a.size} entries)" if a.shape == b.shape else f" (shape {a.shape} vs {b.shape})"))
        verdict[prim] = "match" if not bad else "differs: " + "; ".join(bad[:4]) + (f" ... +{len(bad) - 4} more" if len(bad) > 4 else "")
    return verdict, blur_mode


def main():
    try:
        import cv2
    except ImportError:
        raise SystemExit("pin_opencv.py needs OpenCV's Python module (import cv2); this machine has none — run it where OpenCV is installed")
    from oracle import orb_oracle as O
    O.build()
    imgs, rgb = inputs()
    cv, ora = run_opencv(imgs, rgb), run_oracle(imgs, rgb)
    verdict, blur_mode = compare(cv, ora)
    print(f"OpenCV {cv2.__version__}  ({cv2.getBuildInformation().split('CPU/HW features')[1].split(chr(10))[1].strip() if 'CPU/HW features' in cv2.getBuildInformation() else ''})")
    for prim, v in verdict.items():
        print(f"  {prim:12s} {v}")
    path = os.path.join(ROOT, "tests", "golden", f"opencv_{cv2.__version__}.npz")
    np.savez_compressed(path, __version__=np.array(cv2.__version__), __blur_mode__=np.array(-1 if blur_mode is None else blur_mode),
                        __verdict__=np.array([f"{k}={v}" for k, v in verdict.items()]), **{k.replace("/", "|"): v for k, v in cv.items()})
    print("wrote", path, "- commit it: tests/test_opencv_pin.py replays it against the oracle on every machine")
    sys.exit(0 if all(v.startswith("match") for v in verdict.values()) else 1)


if __name__ == "__main__":
    main()
