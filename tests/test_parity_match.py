"""Parity of the HIP matcher side (Hamming NN, SearchForInitialization, device pipeline) with the CPU oracle: match
indices identical, vbPrevMatched bit-identical."""
import os

import numpy as np
import pytest

import orb_slam2_amd
from orb_slam2_amd import synth, sharding

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def frames(oracle):
    w, h, n = 400, 300, 500
    seq = synth.sequence(w, h, 3, seed=12)
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    return w, h, n, seq, [ora.extract(im) for im in seq]


@pytest.mark.parametrize("window,nnratio,ori", [(100, 0.9, True), (30, 0.6, False), (10, 0.75, True)])
def test_search_for_initialization(backend, oracle, frames, select_tables, window, nnratio, ori):
    w, h, n, seq, K = frames
    m = orb_slam2_amd.ORBmatcher(nnratio, ori, library=backend)
    for a, b in ((0, 1), (1, 2), (2, 0)):
        n_o, m_o, p_o = oracle.search_for_initialization(K[a][0], K[a][1], K[b][0], K[b][1], w, h, window=window, nnratio=nnratio, check_ori=ori)
        n_g, m_g, p_g = m.SearchForInitialization(K[a][0], K[a][1], K[b][0], K[b][1], w, h, windowSize=window)
        assert n_g == n_o and np.array_equal(m_g, m_o) and p_g.tobytes() == p_o.tobytes()
    assert n_o > 0 or window == 10


def test_search_for_initialization_prev_matched_carries_over(backend, oracle, frames):
    """vbPrevMatched is an in/out argument: feeding the updated positions into the next call (Tracking.cc:590-600)."""
    w, h, n, seq, K = frames
    m = orb_slam2_amd.ORBmatcher(0.9, True, library=backend)
    n_o, m_o, p_o = oracle.search_for_initialization(K[0][0], K[0][1], K[1][0], K[1][1], w, h, window=50)
    n_g, m_g, p_g = m.SearchForInitialization(K[0][0], K[0][1], K[1][0], K[1][1], w, h, windowSize=50)
    assert p_g.tobytes() == p_o.tobytes()
    n_o2, m_o2, p_o2 = oracle.search_for_initialization(K[0][0], K[0][1], K[2][0], K[2][1], w, h, prev=p_o, window=50)
    n_g2, m_g2, p_g2 = m.SearchForInitialization(K[0][0], K[0][1], K[2][0], K[2][1], w, h, vbPrevMatched=p_g, windowSize=50)
    assert n_g2 == n_o2 and np.array_equal(m_g2, m_o2) and p_g2.tobytes() == p_o2.tobytes()


def test_search_for_initialization_edge_cases(backend, oracle, frames):
    w, h, n, seq, K = frames
    m = orb_slam2_amd.ORBmatcher(0.9, True, library=backend)
    empty_k, empty_d = np.zeros(0, orb_slam2_amd.KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8)
    n_g, m_g, _ = m.SearchForInitialization(K[0][0], K[0][1], empty_k, empty_d, w, h, windowSize=100)      # nothing to match against
    assert n_g == 0 and np.all(m_g == -1)
    n_g, m_g, _ = m.SearchForInitialization(empty_k, empty_d, K[0][0], K[0][1], w, h, windowSize=100)
    assert n_g == 0 and len(m_g) == 0
    # identical frames: every level-0 keypoint matches itself at distance 0 unless the ratio test fails
    n_o, m_o, p_o = oracle.search_for_initialization(K[0][0], K[0][1], K[0][0], K[0][1], w, h, window=100)
    n_g, m_g, p_g = m.SearchForInitialization(K[0][0], K[0][1], K[0][0], K[0][1], w, h, windowSize=100)
    assert n_g == n_o and np.array_equal(m_g, m_o) and n_g > 0
    # keypoints NOT level-major (shuffled): the level-0 filter and the candidate order follow indices, not levels
    rng = np.random.default_rng(4)
    p1, p2 = rng.permutation(len(K[1][0])), rng.permutation(len(K[2][0]))
    a = (K[1][0][p1], K[1][1][p1])
    b = (K[2][0][p2], K[2][1][p2])
    n_o, m_o, p_o = oracle.search_for_initialization(a[0], a[1], b[0], b[1], w, h, window=100)
    n_g, m_g, p_g = m.SearchForInitialization(a[0], a[1], b[0], b[1], w, h, windowSize=100)
    assert n_g == n_o and np.array_equal(m_g, m_o) and p_g.tobytes() == p_o.tobytes()
    # collisions: many F1 keypoints compete for one F2 descriptor (steal-back bookkeeping ORBmatcher.cc:463-471)
    k1 = K[0][0][K[0][0]["octave"] == 0][:40].copy()
    d1 = np.tile(K[0][1][0], (len(k1), 1))
    for i in range(len(k1)):
        d1[i, i % 32] ^= np.uint8(1 << (i % 7))            # distances 0/1 to the same target
    k2 = k1.copy()
    d2 = np.tile(K[0][1][0], (len(k1), 1))
    d2[1:] = K[0][1][1:len(k1)]
    k1["x"], k1["y"] = k2["x"][0], k2["y"][0]
    n_o, m_o, p_o = oracle.search_for_initialization(k1, d1, k2, d2, w, h, window=100)
    n_g, m_g, p_g = m.SearchForInitialization(k1, d1, k2, d2, w, h, windowSize=100)
    assert n_g == n_o and np.array_equal(m_g, m_o) and p_g.tobytes() == p_o.tobytes()


def test_search_for_initialization_long_candidate_lists(backend, oracle, select_tables):
    """Window larger than the image: every level-0 keypoint of F2 is a candidate of every level-0 keypoint of F1, more
    candidate records than the matcher stages in LDS (the lists are then read from HBM)."""
    w, h, n = 640, 480, 1000
    seq = synth.sequence(w, h, 2, seed=14)
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    (k1, d1), (k2, d2) = ora.extract(seq[0]), ora.extract(seq[1])
    assert int((k1["octave"] == 0).sum()) * int((k2["octave"] == 0).sum()) > 20480
    m = orb_slam2_amd.ORBmatcher(0.9, True, library=backend)
    n_o, m_o, p_o = oracle.search_for_initialization(k1, d1, k2, d2, w, h, window=700, nnratio=0.9, check_ori=True)
    n_g, m_g, p_g = m.SearchForInitialization(k1, d1, k2, d2, w, h, windowSize=700)
    assert n_g == n_o and np.array_equal(m_g, m_o) and p_g.tobytes() == p_o.tobytes()


def test_brute_force_nn(backend, oracle):
    db = synth.descriptor_db(3, 600, seed=7)
    q = synth.descriptor_query(db, 257, seed=7)
    for got, want in zip(orb_slam2_amd.hamming_nn(q, db, library=backend), oracle.bf_nn(q, db)):
        assert np.array_equal(got, want)
    # ties: duplicated DB rows -> the lowest index wins and the second-best equals the best (matcher idiom :447-456)
    db2 = np.concatenate([db[:50], db[:50], db[:50]])
    bi, bd, sd = orb_slam2_amd.hamming_nn(db[:50], db2, library=backend)
    assert np.array_equal(bi, np.arange(50)) and np.all(bd == 0) and np.all(sd == 0)
    # extremes and tiny shapes
    z, o = np.zeros((1, 32), np.uint8), np.full((1, 32), 255, np.uint8)
    bi, bd, sd = orb_slam2_amd.hamming_nn(z, o, library=backend)
    assert bi[0] == 0 and bd[0] == 256 and sd[0] == 2 ** 31 - 1        # one DB row: no second best
    bi, bd, sd = orb_slam2_amd.hamming_nn(z, np.zeros((0, 32), np.uint8), library=backend)
    assert bi[0] == -1 and bd[0] == 2 ** 31 - 1
    # more than one DB chunk + index base + shard merge == single scan
    db = synth.descriptor_db(9, 1900, seed=8)
    q = synth.descriptor_query(db, 40, seed=3)
    want = oracle.bf_nn(q, db)
    got = orb_slam2_amd.hamming_nn(q, db, library=backend)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    parts = []
    for r in range(3):
        lo, hi = sharding.db_shard(len(db), r, 3)
        parts.append(orb_slam2_amd.hamming_nn(q, db[lo:hi], index_base=lo, library=backend))
    merged = sharding.merge_nn(parts)
    assert all(np.array_equal(a, b) for a, b in zip(merged, want))


@pytest.mark.parametrize("form", ["default", "i8", "fp4:2:2:13:1", "fp4:3:2:13:1", "fp4:4:2:15:4", "fp4:4:2:15:6:unseeded", "fp4:4:2:15:3", "fp4:2:3:15:4", "fp4:4:2:16:4", "fp4:4:2:15:8", "fp4:9:9:9:9", "valu"])
def test_brute_force_nn_matrix_core_scan(backend, oracle, monkeypatch, form):
    """Databases from four chunks (32 K rows) on are scanned on the matrix cores - k_hamming_nn_mfma (<+-1, +-1> = 256 - 2 Hamming as i8 products) or
    k_hamming_nn_fp4 (the same as FP4 products on v_mfma_scale_f32_32x32x64_f8f6f4; query tiles per wave : workgroups per CU : log2 rows per workgroup); ORBHIP_NN picks the form:
    ragged last tile and chunk, a query count that fills neither a tile nor a workgroup, planted exact matches, duplicated rows
    (lowest index wins, second = best), an index base; every form against the oracle (the popcount kernel included)."""
    # the default shape (fp4:4:2:15:6: four query tiles per wave, two workgroups per CU, 2^15 rows per workgroup, six tiles per barrier = the pipelined loop on
    # three accumulator pairs) scans in two passes: the head's second-best distance per query seeds the skip threshold of every later chunk
    # (ORBHIP_NN_SEED=0: one pass).  Both here: the duplicates of rows 100..104 in the last tile tie with the head's rows and must lose to them.
    # fp4:9:9:9:9 is not a built shape: the default one scans (and says so on stderr) instead of a failed call
    if form.endswith(":unseeded"):
        form = form[:-len(":unseeded")]
        monkeypatch.setenv("ORBHIP_NN_SEED", "0")
    if form != "default":
        monkeypatch.setenv("ORBHIP_NN", form)
    if backend.endswith("_emu.so") and form.startswith("fp4") and (form not in ("fp4:3:2:13:1",) or "ORBHIP_NN_SEED" in os.environ):
        pytest.skip("the emulation's FP4 matrix product is slow: one FP4 form is enough here, all run on the GPU")
    rng = np.random.default_rng(5)
    n = 4 * 8192 + 1000 + 13 if backend.endswith("_emu.so") else 3 * 65536 + 8192 + 1000 + 13      # (several workgroups of the largest chunk on the GPU)
    db = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (70 if backend.endswith("_emu.so") else 530, 32), dtype=np.uint8)      # (the GPU run fills more than one workgroup of every form)
    q[:20] = db[rng.integers(0, n, 20)]                       # exact matches somewhere in the database
    q[20:30] ^= 1                                               # and near ones
    db[n - 5:] = db[100:105]; q[30:35] = db[100:105]            # duplicates in the last, ragged tile: index 100..104 must win, second == 0
    want = oracle.bf_nn(q, db, fast=True)
    got = orb_slam2_amd.hamming_nn(q, db, library=backend)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert np.array_equal(got[0][30:35], np.arange(100, 105)) and np.all(got[1][30:35] == 0) and np.all(got[2][30:35] == 0)
    gb = orb_slam2_amd.hamming_nn(q[:3], db, index_base=10 ** 10, library=backend)
    assert np.array_equal(gb[0], want[0][:3].astype(np.int64) + 10 ** 10) and np.array_equal(gb[1], want[1][:3]) and np.array_equal(gb[2], want[2][:3])


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("expanded", [False, True])
def test_brute_force_nn_ties_across_chunks(backend, oracle, expanded, waves, monkeypatch):
    """The hand-ordered FP4 scan (k_hamming_nn_fp4b) skips a tile when none of its distances beats a bound - the head's second best (strict: the head's rows have
    the lowest indices) or the second best ANY workgroup has found so far + 1 (shared through device memory).  A database made of a few prototypes with a few
    flipped bits is nothing but ties: hundreds of rows at the best and at the second-best distance of every query, in every chunk, the lowest index among them
    often NOT in the head.  Best row (lowest index), best and second-best distance against the oracle, on the bit form and on the expanded database."""
    emu = backend.endswith("_emu.so")
    if waves == 8:
        # ORBHIP_NN_WAVES=8 (measurement form: eight wavefronts per workgroup on the same staged tiles) must answer like the default
        if emu:
            pytest.skip("the eight-wave form runs on the GPU only (the emulation's FP4 product is slow; checked by hand: profiles/r06_exp_config5_eight_waves.txt)")
        monkeypatch.setenv("ORBHIP_NN_WAVES", "8")
    rng = np.random.default_rng(21)
    n = 32768 + 8192 + 500 + 7 if emu else 5 * 32768 + 8192 + 500 + 7
    proto = rng.integers(0, 256, (6, 32), dtype=np.uint8)
    flips = np.zeros((16, 32), np.uint8)
    for f in range(1, 16):
        for b in rng.integers(0, 256, int(rng.integers(1, 5))):
            flips[f, b >> 3] ^= np.uint8(1 << (b & 7))
    db = proto[rng.integers(0, 6, n)] ^ flips[rng.integers(0, 16, n)]
    db[:32768] = rng.integers(0, 256, (32768, 32), dtype=np.uint8)          # the head knows nothing of the prototypes: every good row lies behind it ...
    db[5000] = proto[0]; db[5001] = proto[0] ^ flips[3]                        # ... except these two
    nq = 70 if emu else 530
    q = proto[rng.integers(0, 6, nq)] ^ flips[rng.integers(0, 16, nq)]
    q[::7] = rng.integers(0, 256, (len(q[::7]), 32), dtype=np.uint8)           # and some queries near nothing
    want = oracle.bf_nn(q, db, fast=True)
    D = orb_slam2_amd.DeviceBuffer
    ddb, dq = D.from_array(db, library=backend), D.from_array(q, library=backend)
    bi, bd, sd = D(nq * 8, library=backend), D(nq * 4, library=backend), D(nq * 4, library=backend)
    if expanded:
        dx = D(orb_slam2_amd.nn_expanded_size(n, library=backend), library=backend)
        orb_slam2_amd.nn_expand_device(None, ddb.ptr, n, dx.ptr, library=backend)
        orb_slam2_amd.hamming_nn_device_expanded(None, dq.ptr, nq, ddb.ptr, dx.ptr, n, bi.ptr, bd.ptr, sd.ptr, library=backend)
    else:
        orb_slam2_amd.hamming_nn_device(None, dq.ptr, nq, ddb.ptr, n, bi.ptr, bd.ptr, sd.ptr, library=backend)
    orb_slam2_amd.device_synchronize(library=backend)
    got = bi.download((nq,), np.int64), bd.download((nq,), np.int32), sd.download((nq,), np.int32)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert (got[1] == got[2]).sum() > nq // 3                                  # (the case is what it claims: best == second best for many queries)


def test_brute_force_nn_expanded_database(backend, oracle):
    """orbhip_nn_expand_device + orbhip_hamming_nn_device_expanded: the database expanded once into the FP4 scan's own tile layout (128 B per row), tiles staged by
    16-byte LDS-DMA - the same answers as the scan that expands every row per query group, and as the oracle: ragged last tile and chunk, planted exact matches,
    duplicates in the last tile (lowest index wins), an index base; the expansion itself byte for byte (bit k of a row -> nibble k: set 0x2, clear 0xA; rows past
    the end zero)."""
    emu = backend.endswith("_emu.so")
    rng = np.random.default_rng(9)
    n = 4 * 8192 + 1000 + 13 if emu else 3 * 65536 + 8192 + 1000 + 13
    db = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (70 if emu else 530, 32), dtype=np.uint8)
    q[:20] = db[rng.integers(0, n, 20)]; q[20:30] ^= 1
    db[n - 5:] = db[100:105]; q[30:35] = db[100:105]
    want = oracle.bf_nn(q, db, fast=True)
    D = orb_slam2_amd.DeviceBuffer
    ddb, dq = D.from_array(db, library=backend), D.from_array(q, library=backend)
    nx = orb_slam2_amd.nn_expanded_size(n, library=backend)
    assert nx == (n + 31) // 32 * 4096
    dx = D(nx, library=backend)
    orb_slam2_amd.nn_expand_device(None, ddb.ptr, n, dx.ptr, library=backend)
    orb_slam2_amd.device_synchronize(library=backend)
    x = dx.download((nx,), np.uint8)
    # tile T: [dword d of the row][row i] x 16 bytes; byte b of the dword -> 4 bytes = its 8 bits as nibbles, low nibble first
    for row in (0, 31, 32, 12345, n - 1):
        T, i = divmod(row, 32)
        for d in (0, 3, 7):
            got = x[T * 4096 + (d * 32 + i) * 16:T * 4096 + (d * 32 + i) * 16 + 16]
            bits = np.unpackbits(db[row, 4 * d:4 * d + 4], bitorder="little")
            nib = np.where(bits == 1, 0x2, 0xA).astype(np.uint8)
            assert np.array_equal(got, nib[0::2] | (nib[1::2] << 4)), (row, d)
    if n % 32:
        assert not x[(n // 32) * 4096:].reshape(8, 32, 16)[:, n % 32:, :].any()          # rows past the end: zero
    bi, bd, sd = D(len(q) * 8, library=backend), D(len(q) * 4, library=backend), D(len(q) * 4, library=backend)
    for base in (0, 10 ** 10):
        orb_slam2_amd.hamming_nn_device_expanded(None, dq.ptr, len(q), ddb.ptr, dx.ptr, n, bi.ptr, bd.ptr, sd.ptr, index_base=base, library=backend)
        orb_slam2_amd.device_synchronize(library=backend)
        got = bi.download((len(q),), np.int64), bd.download((len(q),), np.int32), sd.download((len(q),), np.int32)
        assert np.array_equal(got[0], want[0].astype(np.int64) + base) and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    assert np.array_equal(got[0][30:35] - 10 ** 10, np.arange(100, 105))
    for b in (ddb, dq, dx, bi, bd, sd):
        b.free()


@pytest.mark.parametrize("num_streams", [1, 2])
def test_device_pipeline_extract_and_match(backend, oracle, select_tables, num_streams):
    """orbhip_extract_device on two camera slots over four time steps, matched against each slot's previous frame
    (num_streams = 2: the two slots run concurrently on two HIP streams)."""
    w, h, n = 400, 300, 500
    seqs = [synth.sequence(w, h, 4, seed=s) for s in (12, 15)]
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    K = [[ora.extract(im) for im in s] for s in seqs]
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, max_batch=2, library=backend, num_streams=num_streams)
    pitch = 448
    dbuf = orb_slam2_amd.DeviceBuffer(2 * h * pitch, library=backend)     # device memory through the library's own runtime: plumbing, not product
    for t in range(4):
        host = np.zeros((2, h, pitch), np.uint8)
        host[0, :, :w], host[1, :, :w] = seqs[0][t], seqs[1][t]
        ex.sync()
        dbuf.upload(host)
        ptr = dbuf.ptr
        ex.extract_device(ptr, 2, h * pitch, pitch, match_prev=(t > 0), window=100, nnratio=0.9, check_ori=True)
        ks, ds = ex.fetch(2)
        for s in range(2):
            assert ks[s].tobytes() == K[s][t][0].tobytes() and np.array_equal(ds[s], K[s][t][1])
        if t > 0:
            m12, nm = ex.fetch_matches(2)
            for s in range(2):
                n_o, m_o, _ = oracle.search_for_initialization(K[s][t - 1][0], K[s][t - 1][1], K[s][t][0], K[s][t][1], w, h, window=100, nnratio=0.9)
                assert nm[s] == n_o and np.array_equal(m12[s], m_o)
    prof = ex.profile()
    assert set(prof) >= {"k_pyramid_level", "k_fast_cells", "k_blur", "k_quadtree", "k_describe"}
    assert ex.algorithmic_bytes_per_frame() > 0
    ex.close()


def test_device_pipeline_unaligned_input(backend, oracle):
    """Caller buffer with an odd base address and an odd row pitch: every kernel that prefers aligned 32-bit loads on level 0
    (FAST patch staging, blur, pyramid level 1) must take its byte path and still be bit-exact."""
    w, h, n = 333, 250, 300
    img = synth.frame(w, h, seed=19)
    ko, do = oracle.OracleExtractor(n, 1.2, 6, 20, 7).extract(img)
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 6, 20, 7, w, h, max_batch=1, library=backend)
    pitch = w + 6                                            # 339: not a multiple of 4
    host = np.zeros(1 + h * pitch + 64, np.uint8)
    host[1:1 + h * pitch].reshape(h, pitch)[:, :w] = img   # frame starts at byte offset 1
    dbuf = orb_slam2_amd.DeviceBuffer.from_array(host, library=backend)
    ptr = dbuf.ptr + 1
    ex.extract_device(ptr, 1, h * pitch, pitch)
    ks, ds = ex.fetch(1)
    assert ks[0].tobytes() == ko.tobytes() and np.array_equal(ds[0], do)
    ex.close()


def test_device_pipeline_colour_unaligned_with_matching(backend, oracle):
    """orbhip_extract_device_color on BGR frames resident in device memory at an odd base address and odd row pitch (byte path of
    the conversion), two time steps with frame-to-frame matching: equals cvtColor + extract + SearchForInitialization."""
    w, h, n = 322, 246, 300
    seq = [np.stack([synth.sequence(w, h, 2, seed=31 + c)[t] for c in range(3)], axis=-1) for t in range(2)]
    ora = oracle.OracleExtractor(n, 1.2, 8, 20, 7)
    K = [ora.extract(oracle.cvt_gray(col, rgb=False)) for col in seq]
    ex = orb_slam2_amd.ORBextractor(n, 1.2, 8, 20, 7, w, h, max_batch=1, library=backend)
    pitch = 3 * w + 5
    dbuf = orb_slam2_amd.DeviceBuffer(1 + h * pitch + 64, library=backend)
    for t in range(2):
        host = np.zeros(1 + h * pitch + 64, np.uint8)
        host[1:1 + h * pitch].reshape(h, pitch)[:, :3 * w] = seq[t].reshape(h, 3 * w)
        ex.sync()
        dbuf.upload(host)
        ptr = dbuf.ptr + 1
        ex.extract_device_color(ptr, 1, h * pitch, pitch, 3, rgb=False, match_prev=(t > 0), window=100, nnratio=0.9, check_ori=True)
        ks, ds = ex.fetch(1)
        assert ks[0].tobytes() == K[t][0].tobytes() and np.array_equal(ds[0], K[t][1])
    m12, nm = ex.fetch_matches(1)
    n_o, m_o, _ = oracle.search_for_initialization(K[0][0], K[0][1], K[1][0], K[1][1], w, h, window=100, nnratio=0.9)
    assert nm[0] == n_o and np.array_equal(m12[0], m_o)
    ex.close()


def test_golden_match_fixture(backend):
    g = np.load(os.path.join(GOLDEN, "match_320x240_n300_seed21.npz"))
    m = orb_slam2_amd.ORBmatcher(0.9, True, library=backend)
    n, m12, prev = m.SearchForInitialization(g["k1"], g["d1"], g["k2"], g["d2"], 320, 240, windowSize=100)
    assert n == int(g["nmatches"]) and np.array_equal(m12, g["matches12"]) and prev.tobytes() == g["prev"].tobytes()
