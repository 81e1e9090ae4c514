"""Runs first (file name): is the BOX usable, and which HIP runtime is the product on?

Round 2's acceptance run on the driver's box died at its first device touch (an abort inside the vocabulary upload — a function that
launches no kernel — and a GPU memory fault one second into smoke()), while the same tree was green on every other lease.  These tests
make such a run attribute itself:

  test_box_probe_product_free       a 40-line HIP program with no product code (tests/cpp/hip_touch.hip: hipMalloc, pageable and pinned
                                    copies, one trivial kernel) in its own process.  Red here = the box or its runtime, not liborbhip.so.
  test_first_touch_through_library  hipSetDevice + hipMalloc + hipMemcpy both ways through the C ABI, no kernel of the product.
  test_single_hip_runtime           exactly one libamdhip64 is mapped, and it is the one the library's RUNPATH names.
  test_smoke_on_poisoned_memory     smoke() with every device allocation pre-filled with 0xFF / 0x7F bytes: a kernel that reads memory it
                                    never wrote depends on what a previous tenant of the GPU left there — it must not.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import orb_slam2_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "cpp", "hip_touch")

pytestmark = pytest.mark.gpu


def test_box_probe_product_free():
    assert os.path.exists(PROBE), f"{PROBE} missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
    r = subprocess.run([PROBE], capture_output=True, text=True, timeout=300)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0 and "hip_touch ok" in r.stdout, (
        "THE BOX IS BROKEN, NOT THE PRODUCT: a product-free HIP program (hipMalloc + hipMemcpy + one trivial kernel on the system runtime) "
        f"fails on this machine, exit code {r.returncode}:\n{tail}")


def test_first_touch_through_library(gpu_lib):
    info = orb_slam2_amd.runtime_info(gpu_lib)
    assert "gfx950" in info, info
    host = np.arange(1 << 16, dtype=np.uint32)
    buf = orb_slam2_amd.DeviceBuffer.from_array(host, library=gpu_lib)
    back = buf.download(host.shape, host.dtype)
    orb_slam2_amd.device_synchronize(0, gpu_lib)
    buf.free()
    assert np.array_equal(back, host)


def test_single_hip_runtime(gpu_lib):
    orb_slam2_amd.runtime_info(gpu_lib)                          # makes sure the runtime is loaded and initialised
    mapped = orb_slam2_amd.mapped_hip_runtimes()
    assert len(mapped) == 1, f"more than one HIP runtime in this process: {mapped}"
    assert mapped[0].startswith("/opt/rocm"), f"the product is not on the system runtime its RUNPATH names: {mapped}"
    assert "torch" not in sys.modules, "the GPU test process must stay framework-free (a bundled HIP runtime would shadow the system one)"


@pytest.mark.parametrize("poison", [255, 127])
def test_smoke_on_poisoned_memory(poison):
    env = dict(os.environ, ORBHIP_POISON=str(poison))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as e; e.smoke()"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "smoke ok" in r.stdout, f"smoke() on memory pre-filled with byte {poison} failed:\n{(r.stdout + r.stderr)[-2000:]}"
