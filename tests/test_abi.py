"""The C-ABI shared library: loads, exports every symbol include/orbhip.h declares, fails loudly without a GPU (no compute here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "orbhip.h")
LIB = os.path.join(ROOT, "orb_slam2_amd", "liborbhip.so")


def _declared():
    txt = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(orbhip_[a-z0-9_]+)\s*\(", txt)))


def test_library_is_built_for_gfx950_and_exports_every_declared_symbol():
    assert os.path.exists(LIB), "run __graft_entry__.build() first"
    L = ctypes.CDLL(LIB)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/orbhip.h but not exported by liborbhip.so"
    from orb_slam2_amd import orbhip
    assert sorted(orbhip.SYMBOLS) == names, "orbhip.py binding list out of sync with the header"
    # the code object inside is gfx950 (no other offload target is ever built)
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as td:                      # llvm-objdump --offloading drops the extracted code object next to its input
        tmp = shutil.copy(LIB, td)
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", tmp], capture_output=True, text=True, cwd=td).stdout
    if out.strip():
        assert "gfx950" in out


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product path must fail loudly instead of computing on the CPU."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is visible here")
    import orb_slam2_amd
    with pytest.raises(orb_slam2_amd.OrbHipError) as e:
        orb_slam2_amd.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, library=LIB)
    assert "no HIP device" in str(e.value) or "failed" in str(e.value)
    import numpy as np
    q = np.zeros((4, 32), np.uint8)
    with pytest.raises(orb_slam2_amd.OrbHipError):
        orb_slam2_amd.hamming_nn(q, q, library=LIB)


def test_product_package_never_touches_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "orb_slam2_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "orb_oracle" not in txt and "liborb_oracle" not in txt, f"{f} references the oracle"


def test_descriptor_distance_host_helper():
    import numpy as np
    from orb_slam2_amd import ORBmatcher
    z, o = np.zeros(32, np.uint8), np.full(32, 255, np.uint8)
    assert ORBmatcher.DescriptorDistance(z, o, library=LIB) == 256 and ORBmatcher.DescriptorDistance(o, o, library=LIB) == 0
    rng = np.random.default_rng(2)
    a, b = rng.integers(0, 256, 32, dtype=np.uint8), rng.integers(0, 256, 32, dtype=np.uint8)
    assert ORBmatcher.DescriptorDistance(a, b, library=LIB) == int(np.unpackbits(a ^ b).sum())
    assert (ORBmatcher.TH_LOW, ORBmatcher.TH_HIGH, ORBmatcher.HISTO_LENGTH) == (50, 100, 30)     # ORBmatcher.cc:37-39
