"""k_pyramid_cascade: every pyramid level of a single-image call in ONE launch (orbhip_kernels_extract.hip; a workgroup owns a tile of the last
level and computes the rectangle of every level that tile descends from, in LDS).  Level planes and the extraction must be what the seven
launches give and what the oracle gives (ORBextractor.cc:1107-1132: each level is cv::resize of the level before), for every tile shape, for
widths of every residue mod 4 (the last 4-pixel group is partial), for two to eight levels, and for batches of up to eight frames."""
import numpy as np
import pytest

import orb_slam2_amd
from orb_slam2_amd import synth


def _planes(ex, nl):
    return [ex.mvImagePyramid(l).copy() for l in range(nl)]


@pytest.mark.parametrize("tile", ["32x8", "16x4", "64x16", "8x1", "128x32", "0"])
@pytest.mark.parametrize("w,h,nl", [(640, 480, 8), (321, 243, 6), (322, 241, 5), (323, 250, 4), (193, 244, 2)])
def test_cascade_planes_and_extraction(backend, oracle, monkeypatch, tile, w, h, nl):
    monkeypatch.setenv("ORBHIP_PC_TILE", tile)          # read when the context is created; "0" = the seven launches
    img = synth.frame(w, h, seed=w * 7 + nl)
    ora = oracle.OracleExtractor(400, 1.2, nl, 20, 7)
    ko, do = ora.extract(img)
    ex = orb_slam2_amd.ORBextractor(400, 1.2, nl, 20, 7, w, h, library=backend)
    kg, dg = ex(img)
    if tile == "0":
        assert ex.pyramid_cascade_tiles() == 0
    elif tile in ("32x8", "16x4", "8x1"):              # (64x16 and 128x32 hold more table entries per workgroup than the kernel stages: those contexts take the level kernels)
        assert ex.pyramid_cascade_tiles() > 0, "the cascade's tables failed their self-check"
    for l in range(nl):
        assert np.array_equal(ex.mvImagePyramid(l), ora.level(l)), f"pyramid level {l}"
    assert kg.tobytes() == ko.tobytes() and np.array_equal(dg, do)
    ex.close()


def test_cascade_small_batches(backend, oracle, monkeypatch):
    """up to eight frames take the cascade too (one grid over tiles x frames)"""
    w, h, nl = 400, 300, 7
    imgs = [synth.frame(w, h, seed=40 + i) for i in range(5)]
    ex = orb_slam2_amd.ORBextractor(500, 1.2, nl, 20, 7, w, h, max_batch=5, library=backend)
    ks, ds = ex.extract_batch(imgs)
    ora = oracle.OracleExtractor(500, 1.2, nl, 20, 7)
    for f, im in enumerate(imgs):
        ko, do = ora.extract(im)
        assert ks[f].tobytes() == ko.tobytes() and np.array_equal(ds[f], do), f"frame {f}"
        for l in range(nl):
            assert np.array_equal(ex.mvImagePyramid(l, frame=f), ora.level(l)), f"frame {f} level {l}"
    ex.close()


def test_cascade_is_skipped_where_it_does_not_apply(backend, oracle):
    """scale factor 1.7: a 4-pixel group's taps span more than 8 source bytes - the level kernels run; results as ever"""
    w, h = 480, 360
    img = synth.frame(w, h, seed=5)
    ex = orb_slam2_amd.ORBextractor(300, 1.7, 3, 20, 7, w, h, library=backend)
    kg, dg = ex(img)
    assert ex.pyramid_cascade_tiles() == 0
    ko, do = oracle.OracleExtractor(300, 1.7, 3, 20, 7).extract(img)
    assert kg.tobytes() == ko.tobytes() and np.array_equal(dg, do)
    ex.close()


@pytest.mark.parametrize("w,h,n", [(1241, 376, 2000), (752, 480, 1200), (640, 480, 1000), (1920, 1080, 4000)])
def test_cascade_serves_the_baseline_shapes(emu_lib, w, h, n):
    """BASELINE.json's four image shapes at scale 1.2 / 8 levels: the tables pass their self-check and the launch fits its LDS"""
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, library=emu_lib)
    assert ex.pyramid_cascade_tiles() > 0
    ex.close()
