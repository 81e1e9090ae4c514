"""Test configuration.

Markers:   gpu  — needs a real MI355X (run by the driver with `-m gpu`; everything else runs on CPU).
Backends:  the parity tests are written once and parametrised over
           "emu" — the product kernel sources compiled against the TEST-ONLY fiber emulation of HIP
                   (tests/emu/, CPU; checks kernel logic where no GPU exists; never shipped / benchmarked), and
           "gpu" — the real liborbhip.so (hipcc, gfx950) through the C ABI; marked gpu.
The oracle (oracle/) is only ever the checker.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMU_LIB = os.path.join(ROOT, "tests", "emu", "liborbhip_emu.so")
GPU_LIB = os.path.join(ROOT, "orb_slam2_amd", "liborbhip.so")
CSRC = os.path.join(ROOT, "orb_slam2_amd", "csrc")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (liborbhip.so built for gfx950)")


def gpu_session(config):
    """True when this run selects the gpu-marked tests (`-m gpu`): fixtures with both emulation and GPU parameters then skip their emulation
    parameters BEFORE loading anything, so that no emulation library is mapped into a process whose record says which native code ran."""
    expr = (config.getoption("-m") or "").strip()
    return "gpu" in expr and "not gpu" not in expr


def pytest_report_header(config):
    """GPU runs name their box in the log's first lines: the product-free probe's verdict and the runtime the product is on (a red run on
    a machine whose GPU faults for every process then says so itself)."""
    if not gpu_session(config):
        return None
    lines = []
    probe = os.path.join(ROOT, "tests", "cpp", "hip_touch")
    try:
        r = subprocess.run([probe], capture_output=True, text=True, timeout=300)
        ok = r.returncode == 0 and "hip_touch ok" in r.stdout
        facts = [l for l in r.stdout.splitlines() if l.startswith("devices ") or "gfx" in l]
        lines.append(f"box probe (product-free HIP program): {'ok' if ok else 'FAILED rc=%d -- THE BOX, NOT THE PRODUCT' % r.returncode}; " + "; ".join(facts))
        if not ok:
            lines.append("box probe tail: " + (r.stdout + r.stderr)[-600:].replace("\n", " | "))
    except Exception as e:                                             # noqa: BLE001 - the header must never stop the run
        lines.append(f"box probe could not run: {e}")
    return lines


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources if os.path.exists(s))


@pytest.fixture(scope="session")
def oracle():
    from oracle import orb_oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def emu_lib():
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc"))]
    srcs += [os.path.join(ROOT, "include", "orbhip.h"), os.path.join(ROOT, "tests", "emu", "include", "hip", "hip_runtime.h")]
    override = os.environ.get("ORBHIP_EMU_LIB")                      # e.g. the AddressSanitizer build (make -C orb_slam2_amd/csrc emu_asan; tools/asan_suite.sh)
    if override:
        override = os.path.abspath(override)
        if not os.path.exists(override):
            pytest.fail(f"ORBHIP_EMU_LIB={override} does not exist")
        return override
    if not _newer(EMU_LIB, srcs):
        from oracle.orbslam_ref import _locked_make
        _locked_make(["-C", CSRC, "-s", "emu"])
    return EMU_LIB


@pytest.fixture(scope="session")
def gpu_lib():
    if not os.path.exists(GPU_LIB):
        pytest.fail(f"{GPU_LIB} missing: run `python -c 'import __graft_entry__ as g; g.build()'` — no CPU fallback exists")
    return GPU_LIB


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    """Path of the library the parity test drives through the C ABI."""
    if request.param == "emu":
        return request.getfixturevalue("emu_lib")
    return request.getfixturevalue("gpu_lib")


@pytest.fixture(params=["lds", "device_memory"])
def select_tables(request, monkeypatch):
    """Where k_match_select / k_proj_select keep their per-feature tables: LDS (every frame of up to ~8000 features), or device memory - the form frames
    with more features take (orbhip_match_select_big / orbhip_proj_select_big), forced here at the tests' small sizes by ORBHIP_SELECT_BIG=1 (read per call)."""
    if request.param == "device_memory":
        monkeypatch.setenv("ORBHIP_SELECT_BIG", "1")
    else:
        monkeypatch.delenv("ORBHIP_SELECT_BIG", raising=False)
    return request.param
