"""Both blur kernels produce the oracle's bytes: the matrix-core one (k_blur_mfma, default) and the all-VALU one (ORBHIP_BLUR=valu, the
library's fallback when the taps do not fit the i8 form) in both GaussianBlur rounding modes, on widths that exercise every border case of
the 224-column tiles (w % 4 = 0..3, a tile that ends at the border, one that is a single block wide); the serial schedule (ORBHIP_SERIAL=1, the
profiling aid) beside the default one; key point slots whose count is not a multiple of the four slots a describing wavefront takes.

backend = "emu" (kernel sources under the test-only fiber emulation, CPU) or "gpu" (real liborbhip.so, marked gpu).
"""
import os

import numpy as np
import pytest

import orb_slam2_amd
from orb_slam2_amd import synth


def _same(kg, dg, ko, do):
    assert len(kg) == len(ko), (len(kg), len(ko))
    for f in ko.dtype.names:
        assert np.array_equal(kg[f].view(np.int32), ko[f].view(np.int32)), f
    assert np.array_equal(dg, do)


@pytest.fixture
def env(monkeypatch):
    def set_(**kw):
        for k, v in kw.items():
            monkeypatch.setenv(k, str(v))
    return set_


@pytest.mark.parametrize("blur", ["mfma", "valu"])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("w,h", [(224 + 62, 230), (333, 250), (450, 224), (227, 231), (1241, 376)])
def test_blur_kernels_and_rounding_modes(backend, oracle, env, blur, mode, w, h):
    if (w, h) == (1241, 376) and (backend.endswith("emu.so") and (blur, mode) != ("mfma", 1)):
        pytest.skip("full-size frame once on the emulation")
    env(ORBHIP_BLUR=blur)
    n = 600
    img = synth.frame(w, h, seed=w + h + mode)
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7, blur_round_mode=mode)
    ko, do = ora.extract(img)
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, library=backend, blur_round_mode=mode)
    kg, dg = ex(img)
    for l in range(8):
        b = ora.blurred(l)
        if b is not None:
            assert np.array_equal(ex.blurred_level(l), b), f"blurred level {l} ({blur}, mode {mode})"
    _same(kg, dg, ko, do)
    ex.close()


@pytest.mark.parametrize("serial", [0, 1])
def test_second_stream_and_serial_schedule(backend, oracle, env, serial):
    """a batch of more than eight frames puts the blur on the context's second stream beside the quadtree; ORBHIP_SERIAL=1 keeps every kernel on one"""
    env(ORBHIP_SERIAL=serial)
    w, h, n, B = 320, 240, 300, 11
    imgs = np.stack([synth.frame(w, h, seed=40 + s) for s in range(B)])
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, max_batch=B, library=backend)
    kps, descs = ex.extract_batch(imgs)
    for i in (0, 7, 8, 10):
        ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
        ko, do = ora.extract(imgs[i])
        _same(kps[i], descs[i], ko, do)
    ex.close()


@pytest.mark.parametrize("n", [37, 101, 250])
def test_slot_counts_not_multiples_of_four(backend, oracle, n):
    """k_describe takes four consecutive key point slots per wavefront: levels whose capacity / fill leaves ragged groups."""
    w, h = 320, 240
    img = synth.frame(w, h, seed=n)
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    ko, do = ora.extract(img)
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, library=backend)
    kg, dg = ex(img)
    _same(kg, dg, ko, do)
    ex.close()


@pytest.mark.parametrize("w,h", [(1230, 260), (617, 300)])      # level 1 = 1025 and 514 columns
def test_pyramid_tiles_that_end_in_the_last_columns(backend, oracle, w, h):
    """Level widths one or two columns past a multiple of the 256-column tile: the last tile stages a footprint that starts within three
    bytes of the source row's end (the LDS-DMA lanes are clamped into the row, the ragged dword is patched byte by byte)."""
    n = 400
    img = synth.frame(w, h, seed=w)
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    ko, do = ora.extract(img)
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, library=backend)
    kg, dg = ex(img)
    for l in range(1, 8):
        assert np.array_equal(ex.mvImagePyramid(l), ora.level(l)), f"pyramid level {l} of {ora.level_size(l)}"
    _same(kg, dg, ko, do)
    ex.close()
