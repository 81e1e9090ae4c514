"""Replays every tests/golden/opencv_<version>.npz (written by tools/pin_opencv.py on a machine that has OpenCV) against the oracle's
restatements of the seven OpenCV primitives on this path.  A primitive the file records as matching that OpenCV version must still match
bit for bit; one it records as differing is reported, not asserted (it documents that OpenCV version).  Without such a file the OpenCV boundary
stays "parity unpinned"  and the replay test is skipped; the self-check below runs regardless."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
FILES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "opencv_*.npz")))


def test_pin_tool_self_check(oracle):
    """the tool's own plumbing, without OpenCV: the oracle's outputs compared with themselves through compare() (blur as rounding mode 0 and 1),
    one altered pixel must be reported"""
    import pin_opencv as P
    imgs, rgb = P.inputs()
    small = {k: imgs[k] for k in ("synth_320x240_s1", "china")}
    ora = P.run_oracle(small, {"china": rgb["china"]})
    for mode in (0, 1):
        cv = {(k.replace(f"blur{mode}/", "blur/")): v for k, v in ora.items() if not k.startswith(f"blur{1 - mode}/")}
        verdict, bm = P.compare(cv, ora)
        assert bm == mode and all(v.startswith("match") for v in verdict.values()), verdict
    cv["resize/china/L2"] = cv["resize/china/L2"].copy(); cv["resize/china/L2"][5, 7] ^= 1
    verdict, _ = P.compare(cv, ora)
    assert verdict["resize"].startswith("differs: resize/china/L2") and verdict["remap"] == "match"
    assert len(ora["fast20/china"]) > 500 and len(ora["fast7/china"]) > len(ora["fast20/china"])


@pytest.mark.skipif(not FILES, reason="no tests/golden/opencv_<version>.npz: run tools/pin_opencv.py where OpenCV is installed (the OpenCV boundary is unpinned until then)")
@pytest.mark.parametrize("path", FILES or [None])
def test_oracle_matches_recorded_opencv(oracle, path):
    import pin_opencv as P
    z = np.load(path)
    cv = {k.replace("|", "/"): z[k] for k in z.files if not k.startswith("__")}
    recorded = dict(v.split("=", 1) for v in z["__verdict__"].tolist())
    imgs, rgb = P.inputs()
    ora = P.run_oracle(imgs, rgb)
    verdict, blur_mode = P.compare(cv, ora)
    for prim, was in recorded.items():
        if was.startswith("match"):
            assert verdict[prim].startswith("match"), f"OpenCV {z['__version__']}: {prim} matched when the file was written, now {verdict[prim]}"
        else:
            print(f"OpenCV {z['__version__']}: {prim} recorded as '{was}' (now: {verdict[prim]})")
    assert int(z["__blur_mode__"]) == (-1 if blur_mode is None else blur_mode)
