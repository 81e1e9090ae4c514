"""The extractor on natural photographs (tests/golden/natural_images.npz: scikit-learn's china.jpg and flower.jpg, frozen as raw arrays by
tests/golden/make_natural_images.py) — every other image under tests/ is synthetic.  Real camera content: lens blur, JPEG block edges,
foliage, sky gradients (large regions where only the minThFAST fallback finds corners), saturated highlights.

  * the oracle against the REFERENCE's own src/ORBextractor.cc on both photographs, two configurations each, gray from either channel order
  * the HIP kernels (CPU emulation here, the MI355X under -m gpu) against the oracle: key points, descriptors, every pyramid plane,
    colour frames converted on the device, and a frame-to-frame match between the photograph and a shifted crop of it
The OpenCV primitives under both sides are still the oracle's restatements (tools/pin_opencv.py compares those with a real OpenCV where one
is installed)."""
import os

import numpy as np
import pytest

import orb_slam2_amd

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def photos(oracle):
    z = np.load(os.path.join(HERE, "golden", "natural_images.npz"))
    out = {}
    for name in ("china", "flower"):
        rgb = np.ascontiguousarray(z[name + "_rgb"])
        out[name] = (rgb, oracle.cvt_gray(rgb, True))
    return out


@pytest.mark.parametrize("name,n,sf,nl", [("china", 1000, 1.2, 8), ("flower", 1000, 1.2, 8), ("china", 2000, 1.2, 8), ("flower", 500, 1.3, 6)])
def test_oracle_equals_reference_on_photographs(oracle, photos, name, n, sf, nl):
    from oracle import orbextractor_ref as R
    if not R.build():
        pytest.skip("reference sources not mounted (oracle/_ref/liborbextractor_ref.so absent)")
    gray = photos[name][1]
    r, o = R.RefExtractor(n, sf, nl, 20, 7), oracle.OracleExtractor(n, sf, nl, 20, 7)
    kr, dr = r.extract(gray)
    ko, do = o.extract(gray)
    assert len(kr) > n // 2 and kr.tobytes() == ko.tobytes() and np.array_equal(dr, do)
    for l in range(nl):
        assert np.array_equal(r.level(l), o.level(l))
    assert len(set(ko["octave"])) == nl                                    # every level contributes on real content
    r.close()


@pytest.mark.parametrize("name,n", [("china", 1000), ("flower", 1500)])
def test_kernels_equal_oracle_on_photographs(backend, oracle, photos, name, n):
    rgb, gray = photos[name]
    h, w = gray.shape
    shifted = np.ascontiguousarray(gray[3:, 5:])                           # the same scene 5 px / 3 px further: a second "frame"
    shifted = np.pad(shifted, ((0, 3), (0, 5)), mode="edge")
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, max_batch=2, library=backend)
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    kg, dg = ex.extract_batch([gray, shifted])
    for f, img in enumerate((gray, shifted)):
        ko, do = ora.extract(img)
        assert kg[f].tobytes() == ko.tobytes() and np.array_equal(dg[f], do), f"frame {f}"
        for l in range(8):
            assert np.array_equal(ex.mvImagePyramid(l, frame=f), ora.level(l)), (f, l)
    k0, d0 = ora.extract(gray); k1, d1 = ora.extract(shifted)
    m = orb_slam2_amd.ORBmatcher(0.9, True, library=backend)
    n_g, m_g, p_g = m.SearchForInitialization(kg[0], dg[0], kg[1], dg[1], w, h, windowSize=30)
    n_o, m_o, p_o = oracle.search_for_initialization(k0, d0, k1, d1, w, h, window=30, nnratio=0.9)
    assert n_g == n_o and np.array_equal(m_g, m_o) and p_g.tobytes() == p_o.tobytes() and n_o > 100
    # colour frames: cvtColor on the device, both channel orders, then the same extraction
    for rgb_order in (True, False):
        src = rgb if rgb_order else np.ascontiguousarray(rgb[:, :, ::-1])
        kc, dc = ex.extract_batch_color([src], rgb=rgb_order)
        assert kc[0].tobytes() == k0.tobytes() and np.array_equal(dc[0], d0)
    ex.close()
