"""The oracle's extractor against the REFERENCE's own src/ORBextractor.cc.

oracle/_ref/liborbextractor_ref.so is the reference file compiled where it lies under /root/reference (`make -C oracle ref`);
the four OpenCV image primitives it calls (resize, GaussianBlur, FAST, fastAtan2) forward to the oracle's restatements, every
other line — constructor tables, pyramid sequencing, the per-cell FAST loop with its threshold fallback, DistributeOctTree /
DivideNode on a real std::list, IC_Angle, the steered BRIEF with the file's own pattern table and cvRound, the output assembly —
is the reference's code.  std::list nodes come from a bump arena, under which the reference's (size, pointer) sort is the
canonical tie-break H1.  Skipped where /root/reference is not mounted and no prebuilt library travelled (the GPU box has it)."""
import numpy as np
import pytest

import orb_slam2_amd
from orb_slam2_amd import synth


@pytest.fixture(scope="module")
def ref():
    from oracle import orbextractor_ref as R
    if not R.build():
        pytest.skip("reference sources not mounted (oracle/_ref/liborbextractor_ref.so absent)")
    return R


CONFIGS = [(320, 240, 500, 1, 1.2, 8, 20, 7), (400, 250, 300, 2, 1.2, 8, 20, 7), (640, 480, 1000, 3, 1.2, 8, 20, 7),
           (1241, 376, 2000, 4, 1.2, 8, 20, 7), (333, 250, 300, 19, 1.2, 6, 20, 7), (752, 480, 1200, 5, 1.2, 8, 20, 7),
           (640, 480, 800, 6, 1.5, 5, 30, 10), (352, 288, 600, 17, 1.2, 8, 12, 4)]


@pytest.mark.parametrize("w,h,n,seed,sf,nl,ini,mn", CONFIGS)
def test_oracle_equals_reference_extractor(ref, oracle, w, h, n, seed, sf, nl, ini, mn):
    img = synth.frame(w, h, seed=seed)
    r, o = ref.RefExtractor(n, sf, nl, ini, mn), oracle.OracleExtractor(n, sf, nl, ini, mn)
    pr, po = r.params(), o.params()
    for key in pr:
        assert np.array_equal(pr[key], po[key]), key                      # ctor tables (ORBextractor.cc:410-470)
    kr, dr = r.extract(img)
    ko, do = o.extract(img)
    assert len(kr) > n // 2
    assert kr.tobytes() == ko.tobytes()                                     # raw bits of pt, size, angle, response, octave, class_id
    assert np.array_equal(dr, do)
    for l in range(nl):
        assert np.array_equal(r.level(l), o.level(l))                       # mvImagePyramid
    r.close()


@pytest.mark.parametrize("name", ["zeros", "checkerboard", "ramp", "low_texture", "noise"])
def test_oracle_equals_reference_extractor_degenerate(ref, oracle, name):
    w, h, n = 320, 240, 400
    img = np.random.default_rng(17).integers(0, 256, (h, w), dtype=np.uint8) if name == "noise" else getattr(synth, name)(w, h)
    r, o = ref.RefExtractor(n, 1.2, 8, 20, 7), oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    kr, dr = r.extract(img)
    ko, do = o.extract(img)
    assert kr.tobytes() == ko.tobytes() and dr.tobytes() == do.tobytes()
    r.close()


@pytest.mark.parametrize("boxes,n", [([(200, 130, 44, 40)], 400), ([(30, 40, 60, 50), (400, 230, 50, 60)], 400), ([(20, 20, 36, 36)], 400), ([(100, 100, 200, 120)], 70)])
def test_oracle_equals_reference_extractor_on_clustered_candidates(ref, oracle, boxes, n):
    """candidate sets that leave the quadtree's "every node divides" regime early (tests/test_parity_extract.py runs the kernels on the same images):
    the reference's own DistributeOctTree on its std::list is the authority for what the oracle - and through it the kernel's jump - must produce"""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_parity_extract import _clusters
    img = _clusters(480, 320, boxes, len(boxes) + n)
    r, o = ref.RefExtractor(n, 1.2, 4, 20, 7), oracle.OracleExtractor(n, 1.2, 4, 20, 7)
    kr, dr = r.extract(img)
    ko, do = o.extract(img)
    assert len(kr) >= 4 and kr.tobytes() == ko.tobytes() and np.array_equal(dr, do)
    r.close()


def test_product_equals_reference_extractor(ref, emu_lib):
    """The HIP kernel sources (CPU emulation build) against the reference's code directly, two batched frames."""
    w, h, n = 400, 300, 500
    imgs = [synth.frame(w, h, seed=s) for s in (31, 32)]
    r = ref.RefExtractor(n, 1.2, 8, 20, 7)
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, max_batch=2, library=emu_lib)
    ks, ds = ex.extract_batch(imgs)
    for f in range(2):
        kr, dr = r.extract(imgs[f])
        assert ks[f].tobytes() == kr.tobytes() and np.array_equal(ds[f], dr)
    ex.close(); r.close()


@pytest.mark.gpu
def test_gpu_equals_reference_extractor(ref, gpu_lib):
    """KITTI-shaped frame on the real GPU library against the reference's code (the .so built from the reference travels)."""
    w, h, n = 1241, 376, 2000
    img = synth.frame(w, h, seed=4)
    r = ref.RefExtractor(n, 1.2, 8, 20, 7)
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, library=gpu_lib)
    kg, dg = ex(img)
    kr, dr = r.extract(img)
    assert kg.tobytes() == kr.tobytes() and np.array_equal(dg, dr)
    ex.close(); r.close()
