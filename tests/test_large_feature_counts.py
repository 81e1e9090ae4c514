"""nFeatures beyond what the matchers' LDS tables hold (the reference takes any nFeatures, Tracking.cc:113-125): 12000 features on a 1920x1080 frame -
extraction, SearchForInitialization (context path and host-array path) and SearchByProjection against the oracle, bit for bit.  The order-dependent
select kernels then keep their per-feature tables in device memory (k_match_select_big / k_proj_select_big); the same form at small sizes is covered by the
`select_tables` parameter of tests/test_parity_match.py and tests/test_parity_projection.py."""
import numpy as np
import pytest

import orb_slam2_amd
from orb_slam2_amd import synth


def _case(backend, oracle, w, h, n):
    seq = synth.sequence(w, h, 2, seed=21)
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    K = [ora.extract(im) for im in seq]
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, max_batch=1, library=backend)
    for im, (ko, do) in zip(seq, K):
        kg, dg = ex(im)
        assert kg.tobytes() == ko.tobytes() and np.array_equal(dg, do)
    ex.close()
    # SearchForInitialization on host arrays (ORBmatcher.cc:405-520)
    m = orb_slam2_amd.ORBmatcher(0.9, True, library=backend)
    n_o, m_o, p_o = oracle.search_for_initialization(K[0][0], K[0][1], K[1][0], K[1][1], w, h, window=100, nnratio=0.9, check_ori=True)
    n_g, m_g, p_g = m.SearchForInitialization(K[0][0], K[0][1], K[1][0], K[1][1], w, h, windowSize=100)
    assert n_g == n_o and np.array_equal(m_g, m_o) and p_g.tobytes() == p_o.tobytes() and n_o > 100
    # ... and behind the extraction of a resident pipeline (the context's own tables)
    exd = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, max_batch=1, library=backend)
    dbuf = orb_slam2_amd.DeviceBuffer(h * w, library=backend)            # device memory through the library's own runtime: plumbing, not product
    for t, im in enumerate(seq):
        exd.sync()
        dbuf.upload(np.ascontiguousarray(im))
        exd.extract_device(dbuf.ptr, 1, h * w, w, match_prev=(t > 0), window=100, nnratio=0.9, check_ori=True)
        ks, ds = exd.fetch(1)
        assert ks[0].tobytes() == K[t][0].tobytes()
    m12, nm = exd.fetch_matches(1)
    assert nm[0] == n_o and np.array_equal(m12[0], m_o)
    exd.close()
    # SearchByProjection (ORBmatcher.cc:1328-1470): every key point of frame 0 projected near itself into frame 1
    (kl, dl), (kc, dc) = K
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    rng = np.random.default_rng(4)
    q = np.zeros(len(kl), oracle.PROJ_QUERY_DTYPE)
    q["x"] = kl["x"] + rng.normal(0, 2.0, len(kl)).astype(np.float32); q["y"] = kl["y"] + rng.normal(0, 2.0, len(kl)).astype(np.float32)
    q["radius"] = (np.float32(7.0) * sf[kl["octave"]]).astype(np.float32)
    q["min_level"], q["max_level"] = kl["octave"] - 1, kl["octave"] + 1
    q["blocks"] = rng.random(len(kl)) < 0.9
    q["angle"] = kl["angle"]
    for mode in (0, 1):
        n_o, f_o = oracle.search_by_projection(kc, dc, w, h, q, dl, mode, nnratio=0.9, th_high=100, check_ori=True)
        n_g, f_g = orb_slam2_amd.search_by_projection(kc, dc, w, h, q, dl, mode, nnratio=0.9, th_high=100, check_ori=True, library=backend)
        assert n_g == n_o and np.array_equal(f_g, f_o) and n_o > 500, mode
    return len(kc)


@pytest.mark.gpu
def test_nfeatures_12000_at_1080p(oracle):
    from conftest import GPU_LIB
    n = _case(GPU_LIB, oracle, 1920, 1080, 12000)
    assert n > 9000


def test_contexts_for_large_feature_counts_are_created(emu_lib):
    """What used to be ORBHIP_ERR_UNSUPPORTED ("nfeatures too large for the LDS matcher"): the context exists; the quadtree's own limit is further out."""
    for n in (9000, 12000):
        ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, 1920, 1080, library=emu_lib)
        assert ex.capacity >= n
        ex.close()
    with pytest.raises(orb_slam2_amd.OrbHipError):
        orb_slam2_amd.ORBextractor(16000, 1.2, 8, 20, 7, 1920, 1080, library=emu_lib)
