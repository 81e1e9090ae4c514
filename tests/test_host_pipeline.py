"""The pipelined host-buffer path and the node-level pool through the C ABI (orbhip_submit / orbhip_collect, chunked
orbhip_extract_batch, orbhip_pool_*, orbhip_reloc_candidates): what every real caller of Frame::ExtractORB (Frame.cc:247-253) and
the stereo constructor's two extractor threads (Frame.cc:78-81) sit on.  Results must be bit-identical to the oracle whatever the
chunking, the ring position, the pinned / pageable source or the device a camera lands on."""
import os

import numpy as np
import pytest

import orb_slam2_amd
from orb_slam2_amd import synth

W, H, N = 320, 240, 300


@pytest.fixture(scope="module")
def frames():
    return [synth.frame(W, H, seed=60 + s % 5, t=s // 5) for s in range(12)]


@pytest.fixture(scope="module")
def want(oracle, frames):
    ora = oracle.OracleExtractor(N, 1.2, 8, 20, 7)
    return [ora.extract(im) for im in frames]


def _same(got_k, got_d, want, idx):
    for k, d, i in zip(got_k, got_d, idx):
        assert k.tobytes() == want[i][0].tobytes(), f"keypoints of frame {i}"
        assert np.array_equal(d, want[i][1]), f"descriptors of frame {i}"


@pytest.mark.parametrize("chunk", [0, 1, 5])
def test_chunked_extract_batch(backend, frames, want, chunk, monkeypatch):
    if chunk:
        monkeypatch.setenv("ORBHIP_HOST_CHUNK", str(chunk))          # frames per chunk: upload k+1 | kernels k | download k-1
    ex = orb_slam2_amd.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=12, library=backend)
    for n in (12, 1, 7):
        k, d = ex.extract_batch(frames[:n])
        _same(k, d, want, range(n))
    # "the last call" consumers see the whole batch whatever the chunking
    k, d = ex.fetch(7)
    _same(k, d, want, range(7))
    assert np.array_equal(ex.mvImagePyramid(0, frame=0), frames[0])


def test_submit_collect_ring(backend, frames, want, monkeypatch):
    monkeypatch.setenv("ORBHIP_HOST_CHUNK", "2")
    ex = orb_slam2_amd.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=4, library=backend)
    depth = ex.L.orbhip_ring_depth()
    assert depth == 3
    batches = [list(range(0, 4)), list(range(4, 8)), list(range(8, 11)), [11, 0, 1], [2]]
    tickets = []
    done = 0
    for b in batches:
        if len(tickets) - done == depth:                          # ring full: the oldest ticket must be collected first
            with pytest.raises(orb_slam2_amd.OrbHipError, match="ring full"):
                ex.submit([frames[i] for i in b])
            k, d = ex.collect(tickets[done])
            _same(k, d, want, batches[done])
            done += 1
        tickets.append(ex.submit([frames[i] for i in b]))
    with pytest.raises(orb_slam2_amd.OrbHipError, match="not the oldest"):
        ex.L.orbhip_collect.restype                                  # noqa: B018 (keeps the linter quiet)
        orb_slam2_amd.orbhip._check(ex.L.orbhip_collect(ex.h, tickets[-1], None, None, 0, orb_slam2_amd.orbhip._p(np.zeros(4, np.int32))), "orbhip_collect", ex.L)
    with pytest.raises(orb_slam2_amd.OrbHipError, match="in flight"):
        ex.extract_batch(frames[:1])                              # the synchronous entry refuses to jump the queue
    with pytest.raises(orb_slam2_amd.OrbHipError, match="in flight"):
        ex.fetch(1)                                               # ... and so does everything else that downloads into the context's own mirrors (= staging set 0)
    while done < len(tickets):
        k, d = ex.collect(tickets[done])
        _same(k, d, want, batches[done])
        done += 1
    k, d = ex.extract_batch(frames[:2])
    _same(k, d, want, range(2))


def test_pinned_sources_and_destinations(backend, frames, want, monkeypatch):
    """Pinned caller buffers are moved by DMA directly (no staging copy).  On the emulation HIPEMU_ALL_PINNED turns that branch on."""
    if backend.endswith("_emu.so"):
        monkeypatch.setenv("HIPEMU_ALL_PINNED", "1")
    monkeypatch.setenv("ORBHIP_HOST_CHUNK", "3")
    ex = orb_slam2_amd.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=8, library=backend)
    src = orb_slam2_amd.pinned_array((8, H, W), np.uint8, library=backend)
    src[:] = np.stack(frames[:8])
    cap = ex.capacity
    kps = orb_slam2_amd.pinned_array((8, cap), orb_slam2_amd.KEYPOINT_DTYPE, library=backend)
    desc = orb_slam2_amd.pinned_array((8, cap, 32), np.uint8, library=backend)
    nout = np.zeros(8, np.int32)
    import ctypes as C
    ptrs = (C.c_void_p * 8)(*[src[f].ctypes.data for f in range(8)])
    st = ex.L.orbhip_extract_batch(ex.h, 8, ptrs, W, kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap, nout.ctypes.data_as(C.c_void_p))
    assert st == 0, ex.L.orbhip_last_error()
    _same([kps[f, :nout[f]] for f in range(8)], [desc[f, :nout[f]] for f in range(8)], want, range(8))
    # a smaller caller capacity: the strided download path, and the capacity error of the reference-sized call
    small = int(nout.max()) + 2
    kps2 = orb_slam2_amd.pinned_array((8, small), orb_slam2_amd.KEYPOINT_DTYPE, library=backend)
    desc2 = orb_slam2_amd.pinned_array((8, small, 32), np.uint8, library=backend)
    st = ex.L.orbhip_extract_batch(ex.h, 8, ptrs, W, kps2.ctypes.data_as(C.c_void_p), desc2.ctypes.data_as(C.c_void_p), small, nout.ctypes.data_as(C.c_void_p))
    assert st == 0, ex.L.orbhip_last_error()
    _same([kps2[f, :nout[f]] for f in range(8)], [desc2[f, :nout[f]] for f in range(8)], want, range(8))
    t = ex.submit([src[f] for f in range(4)])
    k, d = ex.collect(t)
    _same(k, d, want, range(4))


def _pool_devices(backend, monkeypatch, n):
    if backend.endswith("_emu.so"):
        monkeypatch.setenv("HIPEMU_DEVICE_COUNT", str(n))         # n pretend devices on the emulation
        return list(range(n))
    g = orb_slam2_amd.device_count(backend)                      # not torch: its bundled HIP runtime must not enter this test process
    return [i % g for i in range(n)]                              # a 1-GPU box runs both contexts on GPU 0; an 8-GPU node gets 0, 1


def test_pool_cameras_round_robin(backend, frames, want, monkeypatch):
    devices = _pool_devices(backend, monkeypatch, 2)
    pool = orb_slam2_amd.MultiGpuExtractor(devices, 5, N, 1.2, 8, 20, 7, W, H, library=backend)
    assert [pool.device_of(c) for c in range(5)] == [devices[c % 2] for c in range(5)] and pool.device_of(5) == -1
    k, d = pool.extract(frames[:5])
    _same(k, d, want, range(5))
    # a camera without a frame this round; rounds in flight; tickets in order
    t0 = pool.submit([frames[5], None, frames[6], frames[7], None])
    t1 = pool.submit(frames[7:12])
    k, d = pool.collect(t0)
    assert len(k[1]) == 0 and len(k[4]) == 0
    _same([k[0], k[2], k[3]], [d[0], d[2], d[3]], want, [5, 6, 7])
    k, d = pool.collect(t1)
    _same(k, d, want, range(7, 12))
    pool.close()


def test_pool_survives_partial_submit_failure(backend, frames, want, monkeypatch):
    """One device refuses a round (injected): the round keeps its ticket (the other device's part must be collected in order), collect
    delivers the healthy cameras and reports the refusal, and the pool keeps working — a context ticket nobody collects would wedge its
    ring for good.  A round NO device takes fails at submit, without a ticket."""
    if not backend.endswith("_emu.so"):
        pytest.skip("the fault-injection hook is compiled into the CPU emulation build only (-DORBHIP_TEST_HOOKS), not into liborbhip.so")
    devices = _pool_devices(backend, monkeypatch, 2)
    pool = orb_slam2_amd.MultiGpuExtractor(devices, 4, N, 1.2, 8, 20, 7, W, H, library=backend)
    t0 = pool.submit(frames[:4])
    monkeypatch.setenv("ORBHIP_TEST_FAIL_SUBMIT_WORKER", "1")
    t1 = pool.submit(frames[4:8])                                 # worker 1 (cameras 1, 3) refuses; worker 0 took cameras 0, 2
    with pytest.raises(orb_slam2_amd.OrbHipError, match="injected submit failure"):
        pool.submit([None, frames[5], None, frames[7]])           # only worker 1 has work and it refuses: no ticket
    monkeypatch.delenv("ORBHIP_TEST_FAIL_SUBMIT_WORKER")
    t2 = pool.submit(frames[8:12])
    k, d = pool.collect(t0)
    _same(k, d, want, range(4))
    with pytest.raises(orb_slam2_amd.OrbHipError, match="injected submit failure"):
        pool.collect(t1)
    k, d = pool.collect(t2)
    _same(k, d, want, range(8, 12))
    for _ in range(4):                                            # more rounds than the ring is deep: nothing is left in flight
        k, d = pool.extract(frames[4:8])
        _same(k, d, want, range(4, 8))
    pool.close()


def test_pool_round_refused_by_every_device_leaves_nothing_behind(backend, frames, want, monkeypatch):
    """A round EVERY device refuses (a row stride below the width: ERR_INVALID on all workers) fails at submit without a ticket - and the next,
    healthy round must not inherit the second device's refusal (only the first failing worker's status used to be cleared)."""
    import ctypes as C
    devices = _pool_devices(backend, monkeypatch, 2)
    pool = orb_slam2_amd.MultiGpuExtractor(devices, 4, N, 1.2, 8, 20, 7, W, H, library=backend)
    imgs, ptrs = pool._ptrs(frames[:4])
    t = C.c_int(-1)
    for _ in range(2):
        assert pool.L.orbhip_pool_submit(pool.h, ptrs, W - 1, C.byref(t)) == orb_slam2_amd.orbhip.ERR_INVALID and t.value == -1
        assert b"stride" in pool.L.orbhip_last_error()
    for lo in (0, 4):
        t0 = pool.submit(frames[lo:lo + 4])
        k, d = pool.collect(t0)                                   # no "orbhip_pool_submit of this round (device ...)" error from the refused rounds
        _same(k, d, want, range(lo, lo + 4))
    node, bound = pool.numa_node(0)
    assert node >= -1 and (bound is False or node >= 0) and pool.numa_node(7) == (-1, False)
    pool.close()


def test_pool_db_shards_and_reloc_candidates(backend, oracle, monkeypatch):
    devices = _pool_devices(backend, monkeypatch, 3)
    pool = orb_slam2_amd.MultiGpuExtractor(devices, 3, N, 1.2, 8, 20, 7, W, H, library=backend)
    nkf, per = 40, 250
    db = synth.descriptor_db(nkf, per, seed=11)
    q = synth.descriptor_query(db, 300, seed=11)
    pool.db_load(db)
    spans = [pool.db_shard(r) for r in range(3)]
    assert spans[0][0] == 0 and spans[-1][1] == len(db) and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    got = pool.db_query(q)
    wantnn = oracle.bf_nn(q, db)
    assert all(np.array_equal(a, b) for a, b in zip(got, wantnn))            # sharded answer == one left-to-right scan
    rk = (np.arange(len(db)) // per).astype(np.int32)
    for th, ratio, k in ((50, 0.75, 10), (100, 0.9, 3), (20, 0.6, 100)):
        g = orb_slam2_amd.reloc_candidates(*got, rk, nkf, th, ratio, k, library=backend)
        w = oracle.reloc_candidates(*wantnn, rk, nkf, th, ratio, k)
        assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1])
    # tiny DB: fewer rows than devices (empty shards)
    pool.db_load(db[:2])
    got = pool.db_query(q[:5])
    assert all(np.array_equal(a, b) for a, b in zip(got, oracle.bf_nn(q[:5], db[:2])))
    pool.close()


def test_pool_rejects_missing_devices(backend):
    with pytest.raises(orb_slam2_amd.OrbHipError, match="out of range"):
        orb_slam2_amd.MultiGpuExtractor([0, 97], 2, N, 1.2, 8, 20, 7, W, H, library=backend)


@pytest.mark.parametrize("pinned", [False, True])
def test_submit_to_named_result_buffers(backend, frames, want, monkeypatch, pinned):
    """orbhip_submit_to: result buffers named at submit (pinned: filled by DMA, collect copies nothing), frames handed over as ONE array
    (arithmetic pointer table), three batches in flight, and a collect with other buffers than the submit named is refused without losing
    the ticket."""
    if backend.endswith("_emu.so") and pinned:
        monkeypatch.setenv("HIPEMU_ALL_PINNED", "1")
    monkeypatch.setenv("ORBHIP_HOST_CHUNK", "2")
    ex = orb_slam2_amd.ORBextractor(N, 1.2, 8, 20, 7, W, H, max_batch=4, library=backend)
    cap = ex.capacity
    mk = (lambda shape, dt: orb_slam2_amd.pinned_array(shape, dt, library=backend)) if pinned else (lambda shape, dt: np.zeros(shape, dt))
    src = mk((12, H, W), np.uint8)
    src[:] = np.stack(frames[:12])
    bufs = [(mk((4, cap), orb_slam2_amd.KEYPOINT_DTYPE), mk((4, cap, 32), np.uint8), np.zeros(4, np.int32)) for _ in range(3)]
    tickets = [ex.submit(src[4 * b:4 * b + 4], out=bufs[b]) for b in range(3)]
    other = (mk((4, cap), orb_slam2_amd.KEYPOINT_DTYPE), mk((4, cap, 32), np.uint8), np.zeros(4, np.int32))
    if pinned:
        with pytest.raises(orb_slam2_amd.OrbHipError, match="same buffers"):
            orb_slam2_amd.orbhip._check(ex.L.orbhip_collect(ex.h, tickets[0], orb_slam2_amd.orbhip._p(other[0]), orb_slam2_amd.orbhip._p(other[1]), cap,
                                                            orb_slam2_amd.orbhip._p(other[2])), "orbhip_collect", ex.L)
    for b in range(3):
        nout = ex.collect(tickets[b])
        k, d, _ = bufs[b]
        _same([k[f, :nout[f]] for f in range(4)], [d[f, :nout[f]] for f in range(4)], want, range(4 * b, 4 * b + 4))
    k, d = ex.extract_batch(frames[:2])
    _same(k, d, want, range(2))


def test_cpulist_parser_is_reentrant(emu_lib):
    """The NUMA binding of the pool workers and of the copy helpers parses /sys/devices/system/node/node<N>/cpulist from several threads at the same
    moment (orbhip_pool_create posts the workers in parallel; each CopyPool spawns its helpers at once): the parser must keep no shared state
    (it used strtok).  Eight threads parse different lists 2000 times each; every result must be that thread's own list."""
    import ctypes as C
    import threading
    L = C.CDLL(emu_lib)
    L.orbhip_test_parse_cpulist.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
    cases = [("0-63,128-191\n", list(range(0, 64)) + list(range(128, 192))), ("64-127,192-255\n", list(range(64, 128)) + list(range(192, 256))),
             ("3\n", [3]), ("0,2,4,6-9", [0, 2, 4, 6, 7, 8, 9]), ("", []), ("10-12,12-14\n", [10, 11, 12, 13, 14]), ("1-1,5", [1, 5]), ("200-203\n", [200, 201, 202, 203])]
    bad = []

    def work(text, want):
        out = (C.c_int * 1024)()
        for _ in range(2000):
            n = L.orbhip_test_parse_cpulist(text.encode(), out, 1024)
            if n != len(want) or list(out[:n]) != want:
                bad.append((text, n))
                return

    ts = [threading.Thread(target=work, args=c) for c in cases]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad, bad
