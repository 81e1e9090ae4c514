"""The five projection-guided ORBmatcher members under GENERAL poses - rotations, translations, a similarity scale - through the reference's own callers:
the all-reference build (the reference's ORBmatcher.cc, its cv::Mat statements evaluated per point on the host) against the all-steps drop-in build
(orb_slam2_amd/cpp/ORBmatcher.cc: the same statements as flat float code ON THE DEVICE, orbhip_project_search_* / orbhip_project_best_in_window_*).

The member-level tests of tests/test_reference_dropin.py keep every camera at the origin, where `Rcw*p3Dw+tcw` is exact whatever the rounding of a matrix
product; here the products round, and the depth / bounds / distance / viewing-angle gates and MapPoint::PredictScale see distances that are not the test's
inputs.  Every call's whole output (the map point of every feature / the feature of every map point, and the return value) must be identical.
Also checked: which `R*x+t` rounding the drop-in's probe found in the cv::Mat it is linked with (include/cvlite: the generic kernel, mode 0), and - reported,
not asserted - how often forcing the OTHER rounding changes a result (the sensitivity of this test to DESIGN.md H11).
CPU: kernels under the emulation; -m gpu: the same builds linked to liborbhip.so."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from conftest import gpu_session  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

W, H, N = 480, 360, 700
CAM = dict(fx=300.0, fy=300.0, cx=240.0, cy=180.0)


@pytest.fixture(scope="module", params=["all-steps", pytest.param("all-steps-gpu", marks=pytest.mark.gpu)])
def builds(request):
    from oracle import orbslam_ref as S
    if request.param.endswith("-gpu"):
        if not (S.build() and S.build_dropin_gpu()):
            pytest.fail("oracle/_ref/liborbslam_dropin_full_gpu.so did not travel with the repository")
        return S, S.dropin_gpu_lib(full=True)
    if gpu_session(request.config):
        pytest.skip("a -m gpu session maps liborbhip.so only")
    if not (S.build() and S.build_dropin()):
        pytest.skip("reference sources not mounted")
    return S, S.dropin_full_lib()


def _rot(rng, max_deg):
    a = np.deg2rad(rng.uniform(-max_deg, max_deg, 3))
    cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _pose(rng, max_deg=8.0, max_t=0.3):
    T = np.eye(4); T[:3, :3] = _rot(rng, max_deg); T[:3, 3] = rng.uniform(-max_t, max_t, 3)
    return T


def _world(rng, keys, T, dx=-3.0, dy=-1.0, sigma=1.2, scale=1.0):
    """world points that the camera [R | t / scale] sees near the given key points (a few outside the image, a few behind the camera)"""
    n = len(keys)
    u = keys["x"] + dx + rng.normal(0, sigma, n); v = keys["y"] + dy + rng.normal(0, sigma, n)
    u[:4] = -3.0; u[4:7] = W + 2.0; v[7:9] = H + 1.0
    z = rng.uniform(3.0, 9.0, n); z[9:12] *= -1.0
    cam = np.stack([(u - CAM["cx"]) / CAM["fx"] * z, (v - CAM["cy"]) / CAM["fy"] * z, z], 1)
    R, t = T[:3, :3], T[:3, 3] / scale
    w = (R.T @ (cam - t).T).T
    return w[:, 0].astype(np.float32), w[:, 1].astype(np.float32), w[:, 2].astype(np.float32)


def _run(S, D, strict=True, base=None):
    """every member under three random pose sets -> (matches compared per member, calls whose output differs per member); strict: assert equality.
    base: the library the reference side runs on (None: the canonical all-reference build)"""
    seq = synth.sequence(W, H, 2, seed=41)
    S.RefFrame._geometry = None
    S.RefFrame._geometry_other.clear()
    checked, differs = {}, {}
    try:
        for stereo in (False, True):
            kw = dict(nfeatures=N, bf=40.0, **CAM)
            mk = lambda im, lib: S.RefFrame(im, np.roll(im, -9, axis=1) if stereo else None, library=lib, **kw)
            R, F = [mk(im, base) for im in seq], [mk(im, D) for im in seq]
            kl, dl, kc = R[0].keys_un, R[0].desc, R[1].keys_un
            nq = len(kl)
            for seed in range(3):
                rng = np.random.default_rng(100 * seed + stereo)
                A, B = _pose(rng), _pose(rng)
                s12 = float(rng.uniform(0.9, 1.15))
                # the similarity between the two key frames: camera 1 = s12 * R12 * camera 2 + t12, consistent with A and B up to the scale
                T12 = A @ np.linalg.inv(B)
                R12, t12 = T12[:3, :3], T12[:3, 3]
                for L in (base, D):
                    S.set_test_poses(A, B, s12, R12, t12, library=L)
                level = np.clip(kl["octave"] + rng.integers(0, 2, nq), 0, 7).astype(np.int32)
                bad = (rng.random(nq) < 0.05).astype(np.uint8)

                def both(fn, *a, **k):
                    (n_r, o_r), (n_f, o_f) = fn(False, *a, **k), fn(True, *a, **k)
                    same = n_r == n_f and np.array_equal(o_r, o_f)
                    assert same or not strict, fn.__name__
                    differs[fn.__name__] = differs.get(fn.__name__, 0) + (0 if same else int((np.asarray(o_r) != np.asarray(o_f)).sum()) or 1)
                    return n_r

                # Frame::isInFrustum: the reference's member against ORBmatcher::IsInFrustum (--flat-frustum), every field of every point bit for bit
                if not stereo or seed == 0:
                    Xf, Yf, Zf = _world(rng, kl, A, sigma=30.0)
                    f_r, f_f = S.is_in_frustum(R[1], Xf, Yf, Zf, level), S.is_in_frustum(F[1], Xf, Yf, Zf, level)
                    assert f_r.tobytes() == f_f.tobytes() or not strict, "isInFrustum"
                    differs["frustum"] = differs.get("frustum", 0) + int((f_r != f_f).any(1).sum())
                    checked["frustum"] = checked.get("frustum", 0) + int(f_r[:, 0].sum())
                # TrackWithMotionModel: last frame's points under pose A (the last frame's own pose B decides forward / backward)
                X, Y, Z = _world(rng, kl, A)
                has = (rng.random(nq) < 0.85).astype(np.uint8); outl = (rng.random(nq) < 0.1).astype(np.uint8)
                state = rng.choice([0, 0, 0, 1, 2], len(kc)).astype(np.uint8)
                for th, ori in ((15.0, True), (7.0, False)):
                    def last(dropin):
                        fr = F if dropin else R
                        return S.search_by_projection_last(fr[1], fr[0], has, X, Y, Z, dl, outlier=outl, cur_state=state, th=th, mono=not stereo, nnratio=0.9, check_ori=ori)
                    checked["last"] = checked.get("last", 0) + both(last)
                if stereo:
                    continue                                    # the other members do not read mvuRight except Fuse, which gets its own stereo key frame below
                # Relocalization: a key frame's points under pose A
                found = (rng.random(nq) < 0.1).astype(np.uint8)
                for th, od, ori in ((10.0, 100, True), (3.0, 64, False)):
                    def reloc(dropin):
                        fr = F if dropin else R
                        return S.search_by_projection_reloc(fr[1], fr[0], has, X, Y, Z, level, bad, found, dl, state, th=th, orb_dist=od, nnratio=0.9, check_ori=ori)
                    checked["reloc"] = checked.get("reloc", 0) + both(reloc)
                # loop closing: Scw = [s12 * R_A | t_A], i.e. the camera [R_A | t_A / s12]
                Xs, Ys, Zs = _world(rng, kl, A, scale=s12)
                ms = (rng.random(len(kc)) < 0.2).astype(np.uint8)

                def kf(dropin):
                    return S.search_by_projection_kf((F if dropin else R)[1], ms, Xs, Ys, Zs, level, bad, dl, th=10)
                checked["kf_sim3"] = checked.get("kf_sim3", 0) + both(kf)
                st = rng.choice([0, 0, 1], len(kc)).astype(np.uint8)

                def fsim3(dropin):
                    return S.fuse_sim3((F if dropin else R)[1], st, Xs, Ys, Zs, level, bad, dl, th=4.0)
                checked["fuse_sim3"] = checked.get("fuse_sim3", 0) + both(fsim3)
                # Fuse into the key frame at pose A
                nobs = rng.integers(0, 4, nq).astype(np.int32)
                for th in (3.0, 7.0):
                    def fuse(dropin):
                        return S.fuse((F if dropin else R)[1], st, X, Y, Z, level, nobs, bad, dl, th=th)
                    checked["fuse"] = checked.get("fuse", 0) + both(fuse)
                # SearchBySim3: key frame 1 at A holds the last frame's features, key frame 2 at B the current frame's; each side's points are where the OTHER camera sees them
                M1 = np.eye(4); M1[:3, :3] = R12.T / s12; M1[:3, 3] = -(R12.T / s12) @ t12                      # camera 2 from camera 1
                X1, Y1, Z1 = _world(rng, kl, M1 @ A)
                M2 = np.eye(4); M2[:3, :3] = s12 * R12; M2[:3, 3] = t12                                          # camera 1 from camera 2
                X2, Y2, Z2 = _world(rng, kc, M2 @ B, dx=3.0, dy=1.0)
                # (the products above carry the scale inside the rotation block: _world's inverse needs the true inverse)
                def inv_world(keys, M, dx, dy):
                    n = len(keys)
                    u = keys["x"] + dx + rng.normal(0, 1.2, n); v = keys["y"] + dy + rng.normal(0, 1.2, n); z = rng.uniform(3.0, 9.0, n)
                    cam = np.stack([(u - CAM["cx"]) / CAM["fx"] * z, (v - CAM["cy"]) / CAM["fy"] * z, z, np.ones(n)], 1)
                    w = (np.linalg.inv(M) @ cam.T).T
                    return w[:, 0].astype(np.float32), w[:, 1].astype(np.float32), w[:, 2].astype(np.float32)
                X1, Y1, Z1 = inv_world(kl, M1 @ A, -3.0, -1.0)
                X2, Y2, Z2 = inv_world(kc, M2 @ B, 3.0, 1.0)
                lev2 = np.clip(kc["octave"] + rng.integers(0, 2, len(kc)), 0, 7).astype(np.int32)
                hs1 = (rng.random(nq) < 0.8).astype(np.uint8); hs2 = (rng.random(len(kc)) < 0.8).astype(np.uint8)

                def sim3(dropin):
                    fr = F if dropin else R
                    return S.search_by_sim3(fr[0], hs1, X1, Y1, Z1, level, dl, fr[1], hs2, X2, Y2, Z2, lev2, fr[1].desc, th=7.5)
                checked["sim3"] = checked.get("sim3", 0) + both(sim3)
            for f in R + F:
                f.close()
            S.RefFrame._geometry = None
            S.RefFrame._geometry_other.clear()
        # stereo key frame for Fuse's chi-square gate with the right coordinate (ur = u - bf*invz)
        kw = dict(nfeatures=N, bf=40.0, **CAM)
        ks_r = S.RefFrame(seq[1], np.roll(seq[1], -9, axis=1), library=base, **kw); ks_f = S.RefFrame(seq[1], np.roll(seq[1], -9, axis=1), library=D, **kw)
        kl = S.RefFrame(seq[0], library=base, **kw)
        rng = np.random.default_rng(9)
        A = _pose(rng)
        for L in (base, D):
            S.set_test_poses(A, A, 1.0, None, None, library=L)
        X, Y, Z = _world(rng, kl.keys_un, A)
        nq = len(X)
        a = (rng.choice([0, 0, 1], ks_r.N).astype(np.uint8), X, Y, Z, np.clip(kl.keys_un["octave"], 0, 7).astype(np.int32), rng.integers(0, 4, nq).astype(np.int32), np.zeros(nq, np.uint8), kl.desc)
        n_r, b_r = S.fuse(ks_r, *a, th=3.0)
        n_f, b_f = S.fuse(ks_f, *a, th=3.0)
        assert (n_r == n_f and np.array_equal(b_r, b_f) or not strict) and int((ks_r.u_right >= 0).sum()) > 100
        differs["fuse_stereo"] = int((b_r != b_f).sum())
        checked["fuse_stereo"] = n_r
        ks_r.close(); ks_f.close(); kl.close()
    finally:
        for L in (base, D):
            S.set_test_poses(None, library=L)
        S.RefFrame._geometry = None
        S.RefFrame._geometry_other.clear()
    return checked, differs


def test_members_under_general_poses(builds, capsys):
    S, D = builds
    assert S.gemm_mode(D) == 0, "include/cvlite multiplies with cv::gemm's generic kernel: the drop-in's probe must have found that"
    checked, _ = _run(S, D)
    with capsys.disabled():
        print("\nmatches compared under general poses (all identical): " + ", ".join(f"{k} {v}" for k, v in sorted(checked.items())))
    # the test is not vacuous: every member found matches through the rotated cameras
    assert checked["last"] > 800 and checked["reloc"] > 300 and checked["kf_sim3"] > 150 and checked["fuse"] > 100 and checked["fuse_sim3"] > 100 and checked["sim3"] > 100 and checked["fuse_stereo"] > 20 and checked["frustum"] > 1000, checked


def test_report_sensitivity_to_the_other_gemm_rounding(builds, capsys):
    """H11 measured: the same calls with the drop-in FORCED to the rounding its probe did not find (ORBHIP_GEMM_MODE=1: OpenCV's small-matrix path, against
    a cv::Mat that multiplies with the generic kernel).  A differing output is a window membership, a gate or a level that moved with the last bit of
    u / v / dist.  Reported (python -m pytest -s), never asserted: the number says how much a wrong guess about the linked OpenCV would cost, and that the
    test above can see it at all."""
    import json
    import subprocess
    S, D = builds
    gpu = "gpu" in builds[1]._name
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from oracle import orbslam_ref as S\nimport test_projection_poses as T\n"
            "D = S.dropin_gpu_lib(full=True) if %r else S.dropin_full_lib()\n"
            "c, d = T._run(S, D, strict=False)\nprint(json.dumps({'mode': S.gemm_mode(D), 'checked': c, 'differs': d}))\n") % (ROOT, os.path.join(ROOT, "tests"), gpu)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, ORBHIP_GEMM_MODE="1"))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stderr[-2000:]
    out = json.loads(lines[-1])
    assert out["mode"] == 1
    with capsys.disabled():
        print("\nforced to the other R*x+t rounding (ORBHIP_GEMM_MODE=1 against a generic-kernel cv::Mat): entries of the members' outputs that differ from the "
              "reference: " + ", ".join(f"{k} {v}" for k, v in sorted(out["differs"].items())) + "  (of " + ", ".join(f"{k} {v}" for k, v in sorted(out["checked"].items())) + " matches)")


def test_host_transform_fallback_is_exact(builds):
    """gemm_mode 2 - what the drop-in does when its probe recognises NEITHER rounding in the linked cv::Mat: `Rcw*p3Dw+tcw` stays the member's own cv::Mat
    expression on the host, the device starts from the camera-frame point.  Forced here (ORBHIP_GEMM_MODE=2); every output must equal the reference's."""
    import json
    import subprocess
    gpu = "gpu" in builds[1]._name
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from oracle import orbslam_ref as S\nimport test_projection_poses as T\n"
            "D = S.dropin_gpu_lib(full=True) if %r else S.dropin_full_lib()\n"
            "c, d = T._run(S, D, strict=True)\nprint(json.dumps({'mode': S.gemm_mode(D), 'checked': c, 'differs': d}))\n") % (ROOT, os.path.join(ROOT, "tests"), gpu)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, ORBHIP_GEMM_MODE="2"))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.stdout + r.stderr)[-2000:]
    out = json.loads(lines[-1])
    assert out["mode"] == 2 and not any(out["differs"].values()) and out["checked"]["last"] > 800


def test_gemm_probe_tells_the_roundings_apart():
    """include/orbhip_gemm_probe.h against three cv::Mat behaviours (tests/cpp/test_gemm_probe.cc): the generic kernel -> 0, OpenCV's small-matrix path with the
    fused addition -> 1, an algebra nobody restates -> 2 (the drop-in then keeps the transform on the host); ORBHIP_GEMM_MODE overrides.  Host code only."""
    import subprocess
    cpp = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-C", cpp, "-s", "gemm_probe"])
    for name, want in (("generic", 0), ("small", 1), ("odd", 2)):
        out = subprocess.run([os.path.join(cpp, "test_gemm_probe_" + name)], capture_output=True, text=True, check=True, env={k: v for k, v in os.environ.items() if k != "ORBHIP_GEMM_MODE"}).stdout
        assert out.strip() == f"gemm mode {want}", (name, out)
    out = subprocess.run([os.path.join(cpp, "test_gemm_probe_generic")], capture_output=True, text=True, check=True, env=dict(os.environ, ORBHIP_GEMM_MODE="1")).stdout
    assert out.strip() == "gemm mode 1"
