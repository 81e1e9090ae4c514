#!/usr/bin/env python3
"""Randomised extractor + matcher sweep (tests/test_fuzz_gpu.run_case) on the CPU emulation of the kernel sources, LDS poisoned per workgroup.
usage: tools/fuzz_emu.py <seed> <ncases>      (run several seeds side by side: one process each)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HIPEMU_POISON_LDS", "1")
from oracle import orb_oracle as O  # noqa: E402
import test_fuzz_gpu as F  # noqa: E402

lib = os.environ.get("ORBHIP_EMU_LIB", os.path.join(ROOT, "tests", "emu", "liborbhip_emu.so"))
seed, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
bad = ran = 0
for c in range(n):
    tag, ok = F.run_case(rng, O, lib)
    if ok is False:
        print("MISMATCH", tag, flush=True); bad += 1
    ran += ok is True
print("seed", seed, "ran", ran, "mismatches", bad)
sys.exit(1 if bad else 0)
