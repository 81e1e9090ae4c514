#!/usr/bin/env python3
"""Turn the per-pass rocprofv3 counter CSVs written by tools/gpu_pmc.sh into the files kept under profiles/:
profiles/<round>_pmc_pass<i>.csv (the raw rocprofv3 rows of each pass) and profiles/<round>_pmc_summary.json (per-kernel
derived figures bench.py and DESIGN.md quote).

usage: tools/pmc_summarize.py gpurun_out/<tag> [round-prefix, default r01]

Units (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KB as reported; on gfx950
FETCH_SIZE tallies a 128-byte request as 64 bytes, so readers double `fetch_MB_per_dispatch_raw` before comparing it
with a byte count (bench.py does); WRITE_SIZE is uncalibrated and taken as is.
"""
import collections
import csv
import glob
import json
import shutil
import os
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    return agg, {k: len(v) for k, v in disp.items()}


def main():
    src = sys.argv[1]
    prefix = sys.argv[2] if len(sys.argv) > 2 else "r01"
    suffix = sys.argv[3] if len(sys.argv) > 3 else ""         # e.g. "_single": profiles/<prefix>_pmc<suffix>_summary.json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    passes = sorted(glob.glob(os.path.join(src, "p*_counters.csv")))
    if not passes:
        sys.exit(f"no p*_counters.csv under {src}")
    merged = collections.defaultdict(dict)
    ndisp = {}
    for i, p in enumerate(passes, 1):
        agg, disp = load(p)
        shutil.copyfile(p, os.path.join(root, "profiles", f"{prefix}_pmc{suffix}_pass{i}.csv"))      # raw rocprofv3 rows, as collected
        for k, d in agg.items():
            merged[k].update(d)
            ndisp[k] = disp[k]
    out = []
    for k, d in merged.items():
        if not k.startswith("k_"):
            continue
        n = max(ndisp[k], 1)
        w = max(d.get("SQ_WAVES", 0), 1)
        wc = max(d.get("SQ_WAVE_CYCLES", 0), 1)
        row = {
            "kernel": k, "dispatches": n, "waves_per_dispatch": int(w / n),
            "valu_per_wave": round(d.get("SQ_INSTS_VALU", 0) / w), "salu_per_wave": round(d.get("SQ_INSTS_SALU", 0) / w),
            "lds_per_wave": round(d.get("SQ_INSTS_LDS", 0) / w), "vmem_rd_per_wave": round(d.get("SQ_INSTS_VMEM_RD", 0) / w, 1),
            "wave_cycles": round(4 * wc / w),               # SQ_WAVE_CYCLES counts quad-cycles
            "frac_wait_any": round(d.get("SQ_WAIT_ANY", 0) / wc, 2), "frac_wait_inst": round(d.get("SQ_WAIT_INST_ANY", 0) / wc, 2),
            "frac_active": round(d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 2),
            "fetch_MB_per_dispatch_raw": round(d.get("FETCH_SIZE", 0) / 1024 / n), "write_MB_per_dispatch_raw": round(d.get("WRITE_SIZE", 0) / 1024 / n),
            "fetch_KB_per_dispatch_raw": round(d.get("FETCH_SIZE", 0) / n, 1), "write_KB_per_dispatch_raw": round(d.get("WRITE_SIZE", 0) / n, 1),
            "flat_per_wave": round(d.get("SQ_INSTS_FLAT", 0) / w, 1), "smem_per_wave": round(d.get("SQ_INSTS_SMEM", 0) / w, 1),
            "frac_wait_inst_lds": round(d.get("SQ_WAIT_INST_LDS", 0) / wc, 2), "frac_active_valu": round(d.get("SQ_ACTIVE_INST_VALU", 0) / wc, 2),
            "tcc_hit": round(d.get("TCC_HIT_sum", 0) / max(d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0), 1), 2),
            "lds_bank_conflict_frac": round(d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_LDS_IDX_ACTIVE", 0), 1), 2),
        }
        if d.get("SQ_INSTS_MFMA", 0) > 0:                   # the matrix-core pass (kernels that issue MFMA only)
            row.update({"mfma_per_wave": round(d.get("SQ_INSTS_MFMA", 0) / w, 1), "mfma_f6f4_insts_per_dispatch": round(d.get("SQ_INSTS_VALU_MFMA_F6F4", 0) / n),
                        "mfma_i8_insts_per_dispatch": round(d.get("SQ_INSTS_VALU_MFMA_I8", 0) / n),
                        "mfma_mops_f6f4_per_dispatch": round(d.get("SQ_INSTS_VALU_MFMA_MOPS_F6F4", 0) / n), "mfma_mops_i8_per_dispatch": round(d.get("SQ_INSTS_VALU_MFMA_MOPS_I8", 0) / n),
                        "mfma_busy_cycles_per_dispatch": round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n), "busy_cu_cycles_per_dispatch": round(d.get("SQ_BUSY_CU_CYCLES", 0) / n),
                        "mfma_busy_over_busy_cu_cycles": round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(d.get("SQ_BUSY_CU_CYCLES", 0), 1), 3)})
        out.append(row)
    out.sort(key=lambda r: r["kernel"])
    with open(os.path.join(root, "profiles", f"{prefix}_pmc{suffix}_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
    for r in out:
        print(r)


if __name__ == "__main__":
    main()
