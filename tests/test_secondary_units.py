"""tools/secondary_units.py (bench.py's matcher_calls / config5 / config4 objects) at tiny shapes on the CPU emulation of the kernels: the tool runs end to end,
every member's outputs agree between the all-reference build and the binding, the back end's loops agree between the reference's per-call loops and the batch
forms, the brute-force query agrees with the CPU scan and the rig's frames with the oracle."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_secondary_units_small(emu_lib):
    from oracle import orbslam_ref as S
    if not (S.build() and S.build_dropin()):
        import pytest
        pytest.skip("reference sources not mounted")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "secondary_units.py"), "--small"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    mc = out["matcher_calls"]
    assert mc["all_parity"] and len(mc["members"]) >= 13, {k: v["parity"] for k, v in mc["members"].items()}
    assert mc["back_end_loops"]["parity"] and "error" not in mc["batch_entries"]
    assert out["config5"]["parity_sample"]["equal"] and out["config4"]["parity_sample"]["mismatches"] == 0
    assert out["host_cpu"]["hardware_threads"] >= 1
