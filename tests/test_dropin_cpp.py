"""The drop-in C++ classes (include/ORBextractor.h, include/ORBmatcher.h + orb_slam2_amd/cpp/*.cc) used exactly like the
reference's callers use them — Tracking constructs the extractor, Frame::Frame calls ExtractORB, Tracking calls
SearchForInitialization — compiled against the in-repo `cv` type shim (OpenCV is absent here) and checked against the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

import orb_slam2_amd
from orb_slam2_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _run(exe, tmp_path, oracle, threads=False, blur=None):
    """blur: ORBHIP_BLUR_ROUNDING for the classes (None = their own default: the x86 SSE2 rounding on this host)"""
    w, h, n = 400, 300, 500
    seq = synth.sequence(w, h, 2, seed=31)
    for k in range(2):
        seq[k].tofile(str(tmp_path / f"in{k}.raw"))
    out = tmp_path / "out.bin"
    voc_path = os.path.join(ROOT, "tests", "golden", "voc_k6_L3_ref.txt")
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)                      # a rectification-like warp, partly leaving the raw image
    th = 0.015
    mx = (w / 2 + (np.cos(th) * (xx - w / 2) - np.sin(th) * (yy - h / 2)) * 1.03 + 2.25).astype(np.float32)
    my = (h / 2 + (np.sin(th) * (xx - w / 2) + np.cos(th) * (yy - h / 2)) * 1.03 - 1.4).astype(np.float32)
    np.concatenate([mx.ravel(), my.ravel()]).tofile(str(tmp_path / "maps.bin"))
    subprocess.check_call([exe, str(w), str(h), str(n), str(tmp_path / "in0.raw"), str(tmp_path / "in1.raw"), str(out), voc_path, ("threads" if threads is True else f"threads{threads}") if threads else "-",
                           str(tmp_path / "maps.bin")], env=dict(os.environ, **({} if blur is None else {"ORBHIP_BLUR_ROUNDING": str(blur)})))
    blur = 1 if blur is None else blur                     # this container and the GPU box are x86-64
    mk = lambda: oracle.OracleExtractor(n, 1.2, 8, 20, 7, blur_round_mode=blur)
    buf = out.read_bytes()
    off = 0
    frames = []
    for _ in range(2):
        (nk,) = struct.unpack_from("<i", buf, off); off += 4
        k = np.frombuffer(buf, orb_slam2_amd.KEYPOINT_DTYPE, nk, off); off += 28 * nk
        d = np.frombuffer(buf, np.uint8, 32 * nk, off).reshape(nk, 32); off += 32 * nk
        frames.append((k, d))
    (nm,) = struct.unpack_from("<i", buf, off); off += 4
    n1 = len(frames[0][0])
    m12 = np.frombuffer(buf, np.int32, n1, off); off += 4 * n1
    prev = np.frombuffer(buf, np.float32, 2 * n1, off).reshape(n1, 2); off += 8 * n1
    (lv,) = struct.unpack_from("<i", buf, off); off += 4
    pyr = []
    for _ in range(lv):
        pw, ph = struct.unpack_from("<ii", buf, off); off += 8
        pyr.append(np.frombuffer(buf, np.uint8, pw * ph, off).reshape(ph, pw)); off += pw * ph
    (dd,) = struct.unpack_from("<i", buf, off); off += 4
    (ns,) = struct.unpack_from("<i", buf, off); off += 4
    u_right = np.frombuffer(buf, np.float32, ns, off); off += 4 * ns
    depth = np.frombuffer(buf, np.float32, ns, off); off += 4 * ns
    (nc,) = struct.unpack_from("<i", buf, off); off += 4
    kc = np.frombuffer(buf, orb_slam2_amd.KEYPOINT_DTYPE, nc, off); off += 28 * nc
    dc = np.frombuffer(buf, np.uint8, 32 * nc, off).reshape(nc, 32); off += 32 * nc
    gray0 = np.frombuffer(buf, np.uint8, w * h, off).reshape(h, w)
    off += w * h
    bows = []
    for _ in range(3):
        (nb,) = struct.unpack_from("<i", buf, off); off += 4
        rec = np.frombuffer(buf, np.dtype([("id", "<u4"), ("val", "<f8")]), nb, off); off += 12 * nb
        (nf,) = struct.unpack_from("<i", buf, off); off += 4
        fnode, foff, ffeat = [], [0], []
        for _ in range(nf):
            nid, cnt = struct.unpack_from("<Ii", buf, off); off += 8
            ffeat.extend(np.frombuffer(buf, np.uint32, cnt, off).tolist()); off += 4 * cnt
            fnode.append(nid); foff.append(len(ffeat))
        bows.append((rec["id"].copy(), rec["val"].copy(), np.array(fnode, np.uint32), np.array(foff, np.int32), np.array(ffeat, np.uint32)))
    (score12,) = struct.unpack_from("<d", buf, off); off += 8
    (nwords,) = struct.unpack_from("<I", buf, off); off += 4
    if threads:                                           # rounds of concurrent left / right extraction reproduced the sequential results
        assert struct.unpack_from("<i", buf, off)[0] == 1
        off += 4
    nd1, nd2 = struct.unpack_from("<ii", buf, off); off += 8
    un1 = np.frombuffer(buf, orb_slam2_amd.KEYPOINT_DTYPE, nd1, off); off += 28 * nd1
    un2 = np.frombuffer(buf, orb_slam2_amd.KEYPOINT_DTYPE, nd2, off); off += 28 * nd2
    bnd = np.frombuffer(buf, np.float32, 4, off); off += 16
    (nm_d,) = struct.unpack_from("<i", buf, off); off += 4
    m12_d = np.frombuffer(buf, np.int32, nd1, off); off += 4 * nd1
    u_rgbd = np.frombuffer(buf, np.float32, nd2, off); off += 4 * nd2
    z_rgbd = np.frombuffer(buf, np.float32, nd2, off); off += 4 * nd2
    (nr,) = struct.unpack_from("<i", buf, off); off += 4
    kr = np.frombuffer(buf, orb_slam2_amd.KEYPOINT_DTYPE, nr, off); off += 28 * nr
    dr = np.frombuffer(buf, np.uint8, 32 * nr, off).reshape(nr, 32); off += 32 * nr
    rect0 = np.frombuffer(buf, np.uint8, w * h, off).reshape(h, w); off += w * h
    batch_flags = struct.unpack_from("<5i", buf, off); off += 20
    assert batch_flags == (1, 1, 1, 1, 1), f"Submit/Collect (same results, refuse context growth in flight, refuse undistort, refuse pyramid, valid again): {batch_flags}"
    (size_same,) = struct.unpack_from("<i", buf, off); off += 4
    (ncrop,) = struct.unpack_from("<i", buf, off); off += 4
    kcrop = np.frombuffer(buf, orb_slam2_amd.KEYPOINT_DTYPE, ncrop, off); off += 28 * ncrop
    dcrop = np.frombuffer(buf, np.uint8, 32 * ncrop, off).reshape(ncrop, 32); off += 32 * ncrop
    size_ms = struct.unpack_from("<6d", buf, off); off += 48
    assert size_same == 1, "one extractor object alternating between two image sizes: results or follow-ups changed between rounds"
    print(f"\n[image-size freedom] full / crop / full / crop / full / crop calls on one extractor: " + " / ".join(f"{v:.2f}" for v in size_ms) +
          " ms (the first call of each size creates its device context; later size changes switch between kept contexts)")
    assert off == len(buf)

    ora = mk()
    kcrop_o, dcrop_o = mk().extract(np.ascontiguousarray(seq[0][:h - 24, :w - 32]))
    assert kcrop.tobytes() == kcrop_o.tobytes() and np.array_equal(dcrop, dcrop_o)
    ref = [ora.extract(im) for im in seq]
    for f in range(2):
        assert frames[f][0].tobytes() == ref[f][0].tobytes() and np.array_equal(frames[f][1], ref[f][1])
    for l in range(8):                                    # mvImagePyramid of the LAST call (second frame)
        assert np.array_equal(pyr[l], ora.level(l))
    n_o, m_o, p_o = oracle.search_for_initialization(ref[0][0], ref[0][1], ref[1][0], ref[1][1], w, h, window=100, nnratio=0.9)
    assert nm == n_o and np.array_equal(m12, m_o) and prev.tobytes() == p_o.tobytes()
    assert dd == oracle.hamming(ref[0][1][0], ref[1][1][0])
    eL, eR = mk(), mk()
    eL.extract(seq[0])
    eR.extract(seq[1])
    uo, do = oracle.stereo_matches(eL, eR, np.float32(386.1448), np.float32(386.1448) / np.float32(718.856))
    assert ns == len(uo) and u_right.tobytes() == uo.tobytes() and depth.tobytes() == do.tobytes()
    gray = oracle.cvt_gray(np.stack([seq[0], seq[1], seq[0]], axis=-1), rgb=False)
    kco, dco = mk().extract(gray)
    assert np.array_equal(gray0, gray) and kc.tobytes() == kco.tobytes() and np.array_equal(dc, dco)
    # distorted camera: Frame(im, extractor, K, distCoef) -> mvKeysUn, image bounds, SearchForInitialization over them
    cam = tuple(np.float32(v) for v in (np.float32(517.306408) * w / 640, np.float32(516.469215) * h / 480, np.float32(318.643040) * w / 640,
                                        np.float32(255.313989) * h / 480, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314))
    U = [oracle.undistort_keypoints(cam, ref[f][0]) for f in range(2)]
    assert un1.tobytes() == U[0].tobytes() and un2.tobytes() == U[1].tobytes()
    assert bnd.tobytes() == oracle.image_bounds(cam, w, h).tobytes()
    with oracle.image_bounds_set(bnd):
        n_d, m_d, _ = oracle.search_for_initialization(U[0], ref[0][1], U[1], ref[1][1], w, h, window=100, nnratio=0.9)
    assert nm_d == n_d and np.array_equal(m12_d, m_d) and n_d > 40
    yy, xx = np.mgrid[0:h, 0:w]
    dm = np.where((xx + yy) % 7 == 0, 0, (xx * 7 + yy * 13) % 9000 + 2000).astype(np.uint16)
    u_o, z_o = oracle.stereo_from_rgbd(ref[1][0], U[1], dm, np.float32(1.0) / np.float32(5000.0), 40.0)
    assert u_rgbd.tobytes() == u_o.tobytes() and z_rgbd.tobytes() == z_o.tobytes() and (z_o > 0).sum() > 200
    # raw input rectified on the device
    rect = oracle.remap(seq[0], mx, my)
    kro, dro = mk().extract(rect)
    assert np.array_equal(rect0, rect) and kr.tobytes() == kro.tobytes() and np.array_equal(dr, dro)
    ov = oracle.OracleVocabulary(voc_path)
    want = [ov.transform(ref[0][1], 4), ov.transform(ref[1][1], 4), ov.transform(ref[1][1], 4)]
    for got, exp in zip(bows, want):
        assert all(g.tobytes() == e.astype(g.dtype).tobytes() for g, e in zip(got, exp))
    assert score12 == ov.score(want[0][0], want[0][1], want[1][0], want[1][1]) and nwords == ov.nwords


def _build(target):
    from oracle.orbslam_ref import _locked_make
    _locked_make(["-C", CPP, "-s", target])
    return os.path.join(CPP, target)


def test_dropin_classes_emulation(tmp_path, oracle, emu_lib):
    _run(_build("test_dropin_emu"), tmp_path, oracle, threads=2)         # two rounds: the emulation serialises kernel launches of the two host threads


@pytest.mark.gpu
@pytest.mark.parametrize("blur", [None, 0])
def test_dropin_classes_gpu(tmp_path, oracle, gpu_lib, blur):
    exe = _build("test_dropin_gpu")               # make: a no-op when the binary built by __graft_entry__.build() is current
    _run(exe, tmp_path, oracle, threads=True, blur=blur)
