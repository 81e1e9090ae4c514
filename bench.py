#!/usr/bin/env python3
"""bench.py — frames/s of ORB extract + match on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the hot path over one batch of B synthetic 1241x376 8-bit frames (one per camera slot,
already resident in HBM): ORBextractor::operator() with 2000 features / 8 levels / scale 1.2 / FAST 20-7 on every
frame PLUS ORBmatcher(0.9,true).SearchForInitialization(F_{t-1}, F_t, window 100) of every slot against the frame the
slot saw in the previous step (the unit of work).  value = frames of all ranks / max-over-ranks time.

No framework is imported at any N: the frames are made resident through the library's own HIP runtime (orbhip_device_alloc), the one
the tests, smoke() and a real ORB_SLAM2 binary run on.  N > 1: one process per GPU (torch.distributed.run is only the launcher); camera
slots are independent, so ranks share nothing on the data path (no collective, north_star: "no RCCL collective needed") -> weak scaling;
the barrier and the max-over-ranks of the timings go over a Unix socket among the node's ranks (orb_slam2_amd.sharding.NodeRendezvous).
`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) re-executes itself under torch.distributed.run with N ranks;
it refuses to run when fewer than N GPUs are visible instead of reporting a 1-GPU number.

The timed region is `--repeats` (default 5) back-to-back measurements of EXACTLY `--steps` steps each, every one bracketed by
barrier + synchronize; `value` / `ms_per_step` are the MEDIAN repeat, the spread is reported in `repeats`.

Extra objects in the JSON line:
  parity        key points, descriptors and matches12 of one camera slot per DISTINCT scene (64) of the LAST timed step compared bit for bit with the
                oracle, and every replica slot of a scene with the first slot of that scene: the whole batch; a mismatch marks the line INVALID
  runtime       the HIP runtime the library ran on (versions, file), every libamdhip64 mapped into the process (must be one)
  roofline      dominant kernel (by HIP-event time measured inside the timed region on the library's stream):
                algorithmic bytes per launch / average launch duration vs the 8 TB/s HBM peak; `traffic` = HBM bytes per launch from
                two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; gfx950 correction) over a short child run of the same workload;
                `bound` = what the evidence says limits the kernel
  roofline_valu the same kernel against its VALU-issue floor: instructions per wave x waves x 4 cycles / (1024 SIMDs x 2.4 GHz), and against the
                clock it actually ran at (GRBM_GUI_ACTIVE / dispatch time in a fourth --pmc child pass): clock_GHz_effective, frac_at_effective_clock
  dropin_loop   the drop-in as ORB_SLAM2 drives it: >= 20 stereo frames through the reference's own Frame.cc / ORBmatcher.cc (stereo Frame
                constructor, ComputeStereoMatches, SearchByProjection(Current, Last), SearchByProjection(Frame, MapPoints)), ms per frame with
                the drop-in on the GPU and with the reference on the host, bit-exact per frame (KITTI shape; EuRoC shape under "euroc")
  matcher_calls / config5 / config4   every secondary unit with the reference beside it (tools/secondary_units.py): per ORBmatcher / Frame member
                gpu_ms, ref_ms and a parity flag through the reference's own callers; the brute-force DB query vs the CPU scan on all host threads;
                the 8-camera 1080p rig through the pool vs the reference extractor on all host threads
  cpu_baseline  the CPU path timed on this box's host cores on a bounded sample of the same workload (rank 0, N=1 only): the
                reference's own Frame / ORBextractor / ORBmatcher sources built into oracle/_ref (kind "reference"; the four
                OpenCV image primitives they call are the oracle's restatements), or the oracle's restatement (kind "port")
                where that build did not travel with the repo
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
W, H, NFEAT, NLEVELS, SCALE, INI_TH, MIN_TH = 1241, 376, 2000, 8, 1.2, 20, 7
WINDOW, NNRATIO = 100, 0.9
PMC_SUMMARIES = (("r06_pmc_summary.json", 512), ("r05_pmc_summary.json", 512), ("r04_pmc_summary.json", 256), ("r03_pmc_summary.json", 256), ("r02_pmc_summary.json", 256), ("r01_pmc_summary.json", 256))
NSCENES = 64               # distinct seeded scenes replicated over the camera slots (data-dependent kernels see 64 workloads x tsteps)


def make_frames(batch, tsteps, pitch, rank):
    """[tsteps][batch][H][pitch] uint8: `nscenes` distinct seeded scenes (replicated over the slots), consecutive time
    steps of a slot are consecutive frames of its sequence (translated by (3,1) px + fresh noise)."""
    from orb_slam2_amd import synth
    nscenes = min(batch, NSCENES)
    out = np.zeros((tsteps, batch, H, pitch), np.uint8)
    for s in range(nscenes):
        sc = synth.scene(W, H, seed=100 * rank + s)
        for t in range(tsteps):
            fr = synth.frame_from_scene(sc, W, H, t=t, seed=100 * rank + s)
            for b in range(s, batch, nscenes):
                out[t, b, :, :W] = fr
    return out


def _cpu_worker(args):
    """One CPU instance per process: extract + match consecutive frames of one slot until the time budget is spent.
    kind "reference": the reference's own Frame constructor (ORBextractor::operator()) + ORBmatcher::SearchForInitialization from
    oracle/_ref/liborbslam_ref_fast.so; kind "port": the oracle's restatement."""
    frames_slot, budget_s, kind, blur = args
    done, prev, t = 0, None, 0
    t0 = time.perf_counter()
    if kind == "reference":
        from oracle import orbslam_ref as S
        S.use_fast_build(True)
        while time.perf_counter() - t0 < budget_s:
            f = S.RefFrame(frames_slot[t % len(frames_slot)], nfeatures=NFEAT, scale=SCALE, nlevels=NLEVELS, ini_th=INI_TH, min_th=MIN_TH)
            if prev is not None:
                S.search_for_initialization(prev, f, window=WINDOW, nnratio=NNRATIO, check_ori=True)
                prev.close()
            prev = f
            done += 1
            t += 1
        return done, time.perf_counter() - t0
    from oracle import orb_oracle as O
    ex = O.OracleExtractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, fast=True, blur_round_mode=blur)
    while time.perf_counter() - t0 < budget_s:
        k, d = ex.extract(frames_slot[t % len(frames_slot)])
        if prev is not None:
            O.search_for_initialization(prev[0], prev[1], k, d, W, H, window=WINDOW, nnratio=NNRATIO, check_ori=True, fast=True)
        prev = (k, d)
        done += 1
        t += 1
    return done, time.perf_counter() - t0


def cpu_baseline(frames, budget_s=8.0, blur_round_mode=0):
    """The CPU path on this box's host cores, one frame stream per core on all cores (SURVEY.md §8d iii) plus the single-thread
    rate.  If the build of the reference's own sources travelled with the repo (oracle/_ref/liborbslam_ref_fast.so: src/Frame.cc,
    src/ORBextractor.cc, src/ORBmatcher.cc at -O3, with the four OpenCV image primitives they call supplied by the oracle's
    restatements) that is what is timed (kind "reference"); otherwise the oracle's restatement (kind "port")."""
    import multiprocessing as mp
    from oracle import orb_oracle as O
    from oracle import orbslam_ref as S
    O.build()
    kind = "reference" if os.path.exists(S.FAST_PATH) and os.path.exists(os.path.join(ROOT, "oracle", "liborb_oracle_fast.so")) else "port"
    tsteps, batch = frames.shape[0], frames.shape[1]
    slots = [[np.ascontiguousarray(frames[t, b, :, :W]) for t in range(tsteps)] for b in range(min(batch, 16))]
    ncores = os.cpu_count() or 1
    ctx = mp.get_context("fork")
    with ctx.Pool(1) as pool:                                   # the single-thread rate in a child too: the library choice is per process
        one_done, one_dt = pool.map(_cpu_worker, [(slots[0], min(budget_s, 6.0), kind, blur_round_mode)])[0]
        port_done, port_dt = pool.map(_cpu_worker, [(slots[0], 3.0, "port", blur_round_mode)])[0] if kind == "reference" else (one_done, one_dt)
    with ctx.Pool(2) as pool:                                   # the reference's own stereo concurrency: two extractor threads (Frame.cc:78-81)
        two = pool.map(_cpu_worker, [(slots[i % len(slots)], 4.0, kind, blur_round_mode) for i in range(2)])
    t0 = time.perf_counter()
    with ctx.Pool(ncores) as pool:
        res = pool.map(_cpu_worker, [(slots[i % len(slots)], budget_s, kind, blur_round_mode) for i in range(ncores)])
    wall = time.perf_counter() - t0
    total = sum(r[0] for r in res)
    rate = sum(r[0] / r[1] for r in res)
    what = ("the reference's own Frame constructor + ORBmatcher::SearchForInitialization (oracle/_ref, -O3 -march=x86-64-v3; its four OpenCV image "
            "primitives are the oracle's restatements)") if kind == "reference" else "oracle -O3 -march=x86-64-v3 build"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from secondary_units import host_cpu
    hc = host_cpu()
    return {"value": round(rate, 1), "unit": "frames/s", "cores": ncores, "cores_note": "`cores` = worker processes run at once = hardware threads of the box",
            "physical_cores": hc["physical_cores"], "hardware_threads": hc["hardware_threads"], "cpu_model": hc["model"], "kind": kind,
            "sample": f"{total} frames 1241x376 (extract + SearchForInitialization vs previous frame) over {ncores} processes x {budget_s:.0f} s (wall {wall:.1f} s), {what}",
            "single_thread_value": round(one_done / one_dt, 2), "two_thread_value": round(sum(r[0] / r[1] for r in two), 2),
            "oracle_port_single_thread_value": round(port_done / port_dt, 2)}


def host_io(ex_resident, frames, device, blur_round_mode, budget_s=3.0):
    """The drop-in boundary as every real caller sees it (Frame::ExtractORB, Frame.cc:247-253): host images in, host key points +
    descriptors out, PCIe both ways, through the pipelined host path (orbhip_submit / orbhip_collect, two batches in flight).
    Extraction only; never the bench `value`.  Measured from pinned caller buffers (DMA straight from / into them) and from pageable
    ones (staged through the library's pinned ring by a few copy threads), plus the single-frame call latency of the drop-in class."""
    import orb_slam2_amd
    Bh = 256
    ex = orb_slam2_amd.ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, W, H, max_batch=Bh, device=device, blur_round_mode=blur_round_mode)
    cap = ex.capacity
    src_pageable = np.ascontiguousarray(frames[0, :Bh, :, :W]) if frames.shape[1] >= Bh else np.ascontiguousarray(np.resize(frames[0, :, :, :W], (Bh, H, W)))
    out = {"batch": Bh, "bytes_per_frame_h2d": W * H, "bytes_per_frame_d2h": cap * 60 + 4}
    pin_src = orb_slam2_amd.pinned_array((Bh, H, W), np.uint8); pin_src[:] = src_pageable
    sets = {"pinned": (pin_src, [(orb_slam2_amd.pinned_array((Bh, cap), orb_slam2_amd.KEYPOINT_DTYPE), orb_slam2_amd.pinned_array((Bh, cap, 32), np.uint8), np.zeros(Bh, np.int32)) for _ in range(3)]),
            "pageable": (src_pageable, [(np.zeros((Bh, cap), orb_slam2_amd.KEYPOINT_DTYPE), np.zeros((Bh, cap, 32), np.uint8), np.zeros(Bh, np.int32)) for _ in range(3)])}
    t = ex.submit(pin_src, out=sets["pinned"][1][0]); ex.collect(t)        # warm-up: allocates the ring
    # the two kinds alternate, three rounds each, and the MEDIAN round of a kind is reported (every round is listed beside it): the link is shared with
    # whatever else the host does, and single rounds of 0.75 s had the kinds trade places from run to run (98 k / 115 k frames/s, then 107 k / 105 k)
    for rnd in range(3):
        for kind in ("pinned", "pageable"):
            imgs, bufs = sets[kind]                                        # one [B, H, W] array: the binding builds the pointer table arithmetically
            t = ex.submit(imgs, out=bufs[0]); ex.collect(t)
            sub, done, t0 = 0, 0, time.perf_counter()
            pending = []
            for _ in range(2):                                             # the ring holds three batches: two are queued ahead of the one being collected,
                pending.append(ex.submit(imgs, out=bufs[sub % 3])); sub += 1   # so the upload engine never waits for the host to come back from a collect
            while time.perf_counter() - t0 < budget_s / 4:
                pending.append(ex.submit(imgs, out=bufs[sub % 3])); sub += 1    # result buffers named at submit: pinned ones are filled by DMA, no host copy
                nout = ex.collect(pending.pop(0))
                done += 1
            while pending:
                nout = ex.collect(pending.pop(0)); done += 1
            dt = time.perf_counter() - t0
            rate = round(done * Bh / dt, 1)
            rounds = (out[kind]["rounds"] if kind in out else []) + [rate]
            med = sorted(rounds)[len(rounds) // 2]
            out[kind] = {"frames_per_s": med, "pcie_GBps": round(med * (W * H + cap * 60 + 4) / 1e9, 2), "keypoints_per_frame": int(nout.mean()), "rounds": rounds}
    one = orb_slam2_amd.ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, W, H, max_batch=1, device=device, blur_round_mode=blur_round_mode)
    one(src_pageable[0])
    t1 = time.perf_counter()
    for i in range(100):
        one(src_pageable[i % Bh])
    out["single_frame_call_ms"] = round((time.perf_counter() - t1) / 100 * 1e3, 4)
    ex.close(); one.close()
    out["value"] = out["pinned"]["frames_per_s"]; out["unit"] = "frames/s"
    out["note"] = "extract only, PCIe both ways, host buffers in / out; value = pinned caller buffers; measured in a fresh process (the drop-in's situation), the median of three alternating rounds per kind (all rounds listed)"
    return out


def host_io_subprocess(device, blur_round_mode):
    """host_io() in a fresh interpreter (python bench.py --host-io-only): one JSON object on its stdout."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--host-io-only", "--blur-round-mode", str(blur_round_mode), "--device", str(device)],
                       capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"host_io subprocess failed (rc {r.returncode}): {r.stderr[-300:]}"}
    return json.loads(lines[-1])


def valu_roofline(dom, launch_ms, B, measured=None):
    """The instruction roofline of the dominant kernel (the HBM object beside it is what north_star demands; this is what bounds the kernel):
    floor_ms = VALU instructions per wave x waves per launch x cycles per wave64 VALU instruction / (SIMDs x clock).  Instruction and wave counts are SQ
    counters (SQ_INSTS_VALU, SQ_WAVES) - measured IN THIS RUN by a rocprofv3 --pmc child pass of the same workload at the same batch (`measured`), else
    taken from the newest committed PMC summary, waves scaled to this run's batch; 4 cycles per instruction is the measured issue rate of the kernel's
    packed-u16 / v_perm / v_alignbyte mix (profiles/r02_valu_issue_rates_ubench.txt); 1024 SIMDs and 2.4 GHz are the chip's (MI355X_MICROARCH.md)."""
    simds, clock, cpi = 1024, 2.4, 4

    def obj(valu, waves, source, extra=None):
        floor_ms = valu * waves * cpi / (simds * clock * 1e9) * 1e3
        o = {"kernel": dom, "valu_insts_per_wave": valu, "waves_per_launch": waves, "cycles_per_inst": cpi, "simds": simds, "clock_GHz": clock,
             "floor_ms": round(floor_ms, 4), "launch_ms": round(launch_ms, 4), "frac": round(floor_ms / launch_ms, 4) if launch_ms > 0 else None, "source": source}
        o.update(extra or {})
        return o
    if measured:
        extra = None
        ck = measured.get("clock")
        if ck and ck.get("GHz"):
            # the floor against the clock the kernel actually ran at in the counter pass (the guide reports 1.9-2.3 GHz effective under load, and profiled
            # passes a little lower than plain ones): the nominal-clock figures stay beside it
            eff_floor = measured["valu_per_wave"] * measured["waves_per_launch"] * cpi / (simds * ck["GHz"] * 1e9) * 1e3
            extra = {"clock_GHz_effective": ck["GHz"], "floor_ms_at_effective_clock": round(eff_floor, 4), "frac_at_effective_clock": round(eff_floor / launch_ms, 4) if launch_ms > 0 else None,
                     "clock_measurement": ck}
        elif ck:
            extra = {"clock_measurement": ck}
        return obj(measured["valu_per_wave"], measured["waves_per_launch"], "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES over a short child run of this workload at this batch, in this run; launch_ms by HIP events in the timed region", extra)
    for name, pmc_batch in PMC_SUMMARIES:                          # (file, the batch its passes ran at)
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            rows = json.load(f)
        for row in rows:
            if row["kernel"].split("<")[0] == dom:
                return obj(row["valu_per_wave"], int(row["waves_per_dispatch"] * B / pmc_batch),
                           f"profiles/{name} (SQ_INSTS_VALU / SQ_WAVES per dispatch at B = {pmc_batch}" + (f", waves scaled to B = {B}" if B != pmc_batch else "") + "; launch_ms measured in this run)",
                           {"lds_bank_conflict_frac": row.get("lds_bank_conflict_frac"), "frac_wave_cycles_waiting": row.get("frac_wait_any")})
    return None


def effective_clock(counter_csv, out_dir, dom):
    """Effective shader clock of the dominant kernel under load = GRBM_GUI_ACTIVE / the dispatch's wall time (MI355X_MICROARCH.md, DVFS give-back), from
    a rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace pass.  The dispatch's begin / end come from the counter file itself where it carries timestamps,
    else from the kernel trace of the same pass, joined on the dispatch id.  -> {"GHz", "cycles_per_launch", "ns_per_launch", "launches"} or {"error"}."""
    import csv, glob
    try:
        rows = [r for r in csv.DictReader(open(counter_csv)) if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0].strip() == dom]
        if not rows:
            return {"error": f"no GRBM_GUI_ACTIVE rows for {dom}"}
        cyc = {}
        for r in rows:                                                       # one row per (dispatch, dimension instance): sum them per dispatch
            cyc[r.get("Dispatch_Id", r.get("Correlation_Id"))] = cyc.get(r.get("Dispatch_Id", r.get("Correlation_Id")), 0.0) + float(r["Counter_Value"])
        span = {}
        if "Start_Timestamp" in rows[0] and "End_Timestamp" in rows[0]:
            for r in rows:
                span[r.get("Dispatch_Id", r.get("Correlation_Id"))] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        else:
            for f in glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Dispatch_Id") in cyc:
                        span[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        both = [k for k in cyc if k in span and span[k] > 0]
        if not both:
            return {"error": "no timestamps for the profiled dispatches"}
        c = sum(cyc[k] for k in both) / len(both); ns = sum(span[k] for k in both) / len(both)
        ghz = c / ns
        note = "GRBM_GUI_ACTIVE / dispatch wall time in the counter pass"
        if ghz > 3.0:                                                        # the counter came back summed over the eight XCDs' GRBMs
            ghz /= 8.0; c /= 8.0; note += "; counter summed over 8 XCDs, divided by 8"
        return {"GHz": round(ghz, 3), "cycles_per_launch": int(c), "ns_per_launch": int(ns), "launches": len(both), "note": note}
    except Exception as e:                                                   # noqa: BLE001
        return {"error": str(e)[:200]}


def hbm_traffic_subprocess(dom, B, args):
    """HBM bytes the dominant kernel moves per launch, from the L2's memory-side counters, collected as MI355X_MICROARCH.md (HBM / rocprofv3)
    prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (kernel trace only), KB as reported, FETCH_SIZE doubled (gfx950 tallies
    a 128-byte request as 64 bytes), WRITE_SIZE as is.  Each pass profiles a short child run of this same workload at this batch."""
    import glob, csv, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    per = {}
    clock = None
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU SQ_WAVES", "GRBM_GUI_ACTIVE"):
        d = tempfile.mkdtemp(prefix="orb_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc"] + counter.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
               "--batch", str(B), "--steps", "3", "--warmup", "1", "--repeats", "1", "--streams", str(args.streams), "--blur-round-mode", str(args.blur_round_mode),
               "--no-cpu-baseline", "--no-host-io", "--no-dropin-loop", "--no-traffic", "--no-secondary", "--parity-slots", "0"]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=300)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode}): {r.stderr[-200:]}"
            if counter == "GRBM_GUI_ACTIVE":
                clock = effective_clock(files[0], d, dom)                  # a secondary figure: a failure here must not cost the traffic object
                continue
            for cname in counter.split():
                tot, disp = 0.0, set()
                for row in csv.DictReader(open(files[0])):
                    if row["Counter_Name"] == cname and row["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0].strip() == dom:
                        tot += float(row["Counter_Value"]); disp.add(row.get("Dispatch_Id", row.get("Correlation_Id")))
                if not disp:
                    return None, f"no {cname} rows for {dom}"
                per[cname] = (tot * (1024.0 if cname.endswith("_SIZE") else 1.0) / len(disp), len(disp))
        except Exception as e:                                            # noqa: BLE001 - a failed profiler pass must not cost the bench line
            return None, f"rocprofv3 --pmc {counter} pass: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"bytes_per_launch": int(2 * per["FETCH_SIZE"][0] + per["WRITE_SIZE"][0]), "fetch_bytes_raw": int(per["FETCH_SIZE"][0]), "write_bytes_raw": int(per["WRITE_SIZE"][0]),
            "valu_per_wave": round(per["SQ_INSTS_VALU"][0] / max(per["SQ_WAVES"][0], 1.0), 1), "waves_per_launch": int(per["SQ_WAVES"][0]),
            "launches_profiled": per["FETCH_SIZE"][1], "clock": clock, "method": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, one pass each over a 4-step child run of this workload at this batch; 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction)"}, None


def dropin_loop_subprocess(blur_round_mode):
    """The drop-in as ORB_SLAM2 drives it (tools/dropin_loop_rate.py, tests/test_dropin_loop.py): Tracking's per-frame stereo sequence through
    the reference's own Frame.cc / ORBmatcher.cc, ms per frame with this repository's extractor / stereo matcher / projection matchers on the
    GPU and with the reference's on the host cores, every frame of both compared bit for bit.  Fresh interpreter, like a maintainer's binary.
    -> the KITTI-shape object (the metric's shape) with the EuRoC-shape object (BASELINE.json configs[2]) under "euroc"."""
    import subprocess
    env = dict(os.environ, ORB_REF_BLUR_ROUND_MODE=str(blur_round_mode))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dropin_loop_rate.py"), "kitti", "euroc", "mono", "rgbd"], capture_output=True, text=True, timeout=900, env=env)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or len(lines) != 4:
        return {"error": f"dropin_loop subprocess failed (rc {r.returncode}): {r.stderr[-300:]}"}
    out = lines[0]
    out["euroc"] = lines[1]
    out["mono"] = lines[2]          # configs[0]'s shape: the monocular sequence (initialisation on 2 x nFeatures, TrackReferenceKeyFrame, relocalisation)
    out["rgbd"] = lines[3]          # ... and the RGB-D one (distorted camera)
    return out


def secondary_units_subprocess(blur_round_mode):
    """tools/secondary_units.py in a fresh interpreter: every secondary unit with the reference beside it -
    matcher_calls (M1-M5, S1, ComputeBoW: gpu_ms / ref_ms / parity per member, through the reference's own callers), config5 (brute-force DB query vs the
    CPU scan on all host threads), config4 (8 x 1920x1080 x 4000 rig through the pool vs the reference extractor on all host threads)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "secondary_units.py"), "--blur-round-mode", str(blur_round_mode)], capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"secondary_units subprocess failed (rc {r.returncode}): {r.stderr[-400:]}"}
    return json.loads(lines[-1])


def parity_of_last_step(host_frames, kps, descs, m12, last_step, T, blur_round_mode, nslots):
    """Bit-for-bit check of what the LAST timed step left in the context (the oracle is the checker, never the thing measured):
    key points, descriptors and matches12 of `nslots` camera slots with distinct scenes against oracle.OracleExtractor /
    search_for_initialization on the same frames, plus — size-independent, over the whole batch — every replica of a scene must equal
    the first slot that holds that scene."""
    from oracle import orb_oracle as O
    O.build()
    B = host_frames.shape[1]
    nscenes = min(B, NSCENES)
    t_cur, t_prev = last_step % T, (last_step - 1) % T
    ora = O.OracleExtractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, blur_round_mode=blur_round_mode)
    slots = [int(v) for v in np.linspace(0, nscenes - 1, min(nslots, nscenes)).round()]
    mism, detail = 0, []
    for b in slots:
        k2, d2 = ora.extract(np.ascontiguousarray(host_frames[t_cur, b, :, :W]))
        bad = []
        if kps[b].tobytes() != k2.tobytes():
            bad.append("keypoints")
        if not np.array_equal(descs[b], d2):
            bad.append("descriptors")
        if m12[b] is not None and last_step > 0:
            k1, d1 = ora.extract(np.ascontiguousarray(host_frames[t_prev, b, :, :W]))
            n_o, m_o, _ = O.search_for_initialization(k1, d1, k2, d2, W, H, window=WINDOW, nnratio=NNRATIO, check_ori=True)
            if not np.array_equal(m12[b], m_o):
                bad.append("matches12")
        if bad:
            mism += 1
            detail.append({"slot": b, "differs": bad})
    replica_mismatch = 0
    for b in range(nscenes, B):
        a = b % nscenes
        same = kps[b].tobytes() == kps[a].tobytes() and np.array_equal(descs[b], descs[a]) and (m12[b] is None or np.array_equal(m12[b], m12[a]))
        replica_mismatch += 0 if same else 1
    out = {"slots": len(slots), "slot_ids": slots, "mismatches": mism, "compared": "keypoints (28-byte records), descriptors, matches12 — bit for bit vs oracle/ on the last timed step",
           "replica_slots_checked": max(B - nscenes, 0), "replica_mismatches": replica_mismatch}
    if detail:
        out["detail"] = detail
    return out


def pool_mode(args):
    """`--pool`: the product-side multi-GPU owner (orbhip_pool_*, what north_star describes: ONE process, one host thread + context +
    pinned staging ring per GPU, camera c on devices[c mod G], no collective) timed at the drop-in boundary — host frames in, host key
    points + descriptors out, PCIe both ways, rounds pipelined two deep.  Extraction only; its own JSON line (never the bench `value` of the
    N-process mode).  G = --gpus (must be visible), cameras = G x --pool-cameras-per-gpu."""
    import orb_slam2_amd
    from oracle import orb_oracle as O                                  # the checker of the last round only
    g = orb_slam2_amd.device_count()
    if g < args.gpus:
        raise SystemExit(f"bench.py --pool --gpus {args.gpus}: only {g} GPU(s) visible — refusing to report a {args.gpus}-GPU number")
    G, per = args.gpus, args.pool_cameras_per_gpu
    ncam = G * per
    pool = orb_slam2_amd.MultiGpuExtractor(list(range(G)), ncam, NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, W, H, blur_round_mode=args.blur_round_mode)
    nscenes = min(ncam, NSCENES)
    rounds = []
    for t in range(2):
        src = orb_slam2_amd.pinned_array((ncam, H, W), np.uint8)
        for s_ in range(nscenes):
            from orb_slam2_amd import synth
            fr = synth.frame_from_scene(synth.scene(W, H, seed=s_), W, H, t=t, seed=s_)
            for c in range(s_, ncam, nscenes):
                src[c] = fr
        rounds.append(src)
    cap = pool.capacity
    bufs = [(orb_slam2_amd.pinned_array((ncam, cap), orb_slam2_amd.KEYPOINT_DTYPE), orb_slam2_amd.pinned_array((ncam, cap, 32), np.uint8), np.zeros(ncam, np.int32)) for _ in range(2)]
    imgs = [[r[c] for c in range(ncam)] for r in rounds]
    for i in range(max(args.warmup, 1)):
        pool.collect(pool.submit(imgs[i & 1]), out=bufs[i & 1])
    times = []
    for rep in range(max(args.repeats, 1)):
        t0 = time.perf_counter()
        pending = [pool.submit(imgs[0])]
        for i in range(1, args.steps):
            pending.append(pool.submit(imgs[i & 1]))                   # round i uploads while round i-1 computes / downloads
            pool.collect(pending.pop(0), out=bufs[(i - 1) & 1])
        nout = pool.collect(pending.pop(0), out=bufs[(args.steps - 1) & 1])
        times.append(time.perf_counter() - t0)
    elapsed = sorted(times)[len(times) // 2]
    last = (args.steps - 1) & 1
    ora = O.OracleExtractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, blur_round_mode=args.blur_round_mode)
    mism = 0
    slots = [int(v) for v in np.linspace(0, nscenes - 1, min(args.parity_slots, nscenes)).round()] if args.parity_slots > 0 else []
    for c in slots:
        ko, do = ora.extract(np.ascontiguousarray(rounds[last][c]))
        n = int(nout[c])
        mism += 0 if (n == len(ko) and bufs[last][0][c, :n].tobytes() == ko.tobytes() and np.array_equal(bufs[last][1][c, :n], do)) else 1
    frames_total = ncam * args.steps
    out = {"metric": "frames/s ORB extract through the one-process GPU pool, host buffers in / out (PCIe both ways), 1241x376 gray, 2000 kpts, 8 lvls",
           "mode": "pool", "value": round(frames_total / elapsed, 1), "unit": "frames/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "KITTI-shaped 1241x376 u8, 2000 features, 8 levels, extract only", "cameras": ncam, "cameras_per_gpu": per,
                      "parallelism": f"one process, {G} pool worker thread(s), camera c -> device c mod {G}, no collective", "buffers": "pinned caller buffers, two rounds in flight"},
           "repeats": {"n": len(times), "frames_per_s_min": round(frames_total / max(times), 1), "frames_per_s_max": round(frames_total / min(times), 1)},
           "parity": {"slots": len(slots), "mismatches": mism, "compared": "key points + descriptors of the last round vs oracle/"},
           "runtime": {"library": orb_slam2_amd.runtime_info(), "mapped": orb_slam2_amd.mapped_hip_runtimes(), "framework_imported": "torch" in sys.modules}}
    pool.close()
    print(json.dumps(out), flush=True)


def multi_gpu_extras(args, G, share):
    """What `--gpus N` adds for N > 1 after the resident measurement, on rank 0 while the other ranks wait (their contexts closed): the two multi-GPU curves that
    CAN bend - the device-resident rate moves nothing over PCIe and is linear by construction.
      host_io_pool     ONE process feeding all G devices from host buffers through the pool (orbhip_pool_*: a worker thread + context + pinned ring per device, camera
                       c on device c mod G), pinned and pageable caller buffers, rounds pipelined two deep: frames/s, per-device share, the NUMA placement of every worker
      config5_sharded  BASELINE.json configs[4] over G devices: the descriptor database split by contiguous row range (orbhip_pool_db_load), every device answers every
                       query over its shard, the host merges G partial answers per query with the matcher's tie rule; query_ms = the whole call, merge_ms = that merge
                       alone (the numpy mirror orb_slam2_amd.sharding.merge_nn on the G per-shard answers), parity = equal to one device scanning the whole database
    `share` (ORB_BENCH_SHARE_GPU=1, the test aid): all G workers on device 0."""
    import orb_slam2_amd
    from orb_slam2_amd import sharding, synth
    devices = [0] * G if share else list(range(G))
    out = {}
    # ---- the pool from host buffers
    per = args.extras_cameras_per_gpu
    ncam = G * per
    pool = orb_slam2_amd.MultiGpuExtractor(devices, ncam, NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, W, H, blur_round_mode=args.blur_round_mode)
    nscenes = min(ncam, 16)
    scenes = [synth.frame_from_scene(synth.scene(W, H, seed=s_), W, H, t=0, seed=s_) for s_ in range(nscenes)]
    cap = pool.capacity
    res = {"cameras": ncam, "cameras_per_gpu": per, "devices": devices,
           "numa_nodes": [{"worker": r, "device": devices[r], "numa_node": pool.numa_node(r)[0], "bound": pool.numa_node(r)[1]} for r in range(G)]}
    for kind in ("pinned", "pageable"):
        alloc = (lambda shape, dt: orb_slam2_amd.pinned_array(shape, dt)) if kind == "pinned" else (lambda shape, dt: np.zeros(shape, dt))
        src = alloc((ncam, H, W), np.uint8)
        for c in range(ncam):
            src[c] = scenes[c % nscenes]
        bufs = [(alloc((ncam, cap), orb_slam2_amd.KEYPOINT_DTYPE), alloc((ncam, cap, 32), np.uint8), np.zeros(ncam, np.int32)) for _ in range(2)]
        imgs = [src[c] for c in range(ncam)]
        pool.collect(pool.submit(imgs), out=bufs[0])
        rounds = max(args.extras_rounds, 2)
        t0 = time.perf_counter()
        pending = [pool.submit(imgs)]
        for i in range(1, rounds):
            pending.append(pool.submit(imgs))
            pool.collect(pending.pop(0), out=bufs[(i - 1) & 1])
        nout = pool.collect(pending.pop(0), out=bufs[(rounds - 1) & 1])
        dt = time.perf_counter() - t0
        rate = ncam * rounds / dt
        res[kind] = {"frames_per_s": round(rate, 1), "per_device_frames_per_s": round(rate / G, 1), "pcie_GBps_total": round(rate * (W * H + cap * 60 + 4) / 1e9, 2),
                     "keypoints_per_frame": int(nout.mean()), "rounds": rounds}
    res["note"] = "extract only, host buffers in / out, PCIe both ways, ONE process for all devices; the single-device figure of the same path is host_io in the N = 1 line"
    out["host_io_pool"] = res
    # ---- config 5 sharded by row range
    nkf, perkf, nq = args.extras_db_keyframes, 2000, 2000
    rng = np.random.default_rng(7)
    db = rng.integers(0, 256, (nkf * perkf, 32), dtype=np.uint8)
    q = db[rng.integers(0, len(db), nq)].copy()
    q[::2, 0] ^= 0x5A
    pool.db_load(db)
    got = pool.db_query(q)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); got = pool.db_query(q); ts.append(time.perf_counter() - t0)
    shards = [pool.db_shard(r) for r in range(G)]
    pool.close()
    parts = [orb_slam2_amd.hamming_nn(q, db[lo:hi], device=devices[r], index_base=lo) for r, (lo, hi) in enumerate(shards)]
    tm = []
    for _ in range(5):
        t0 = time.perf_counter(); merged = sharding.merge_nn(parts); tm.append(time.perf_counter() - t0)
    whole = orb_slam2_amd.hamming_nn(q, db, device=devices[0])
    out["config5_sharded"] = {"shards": G, "rows": len(db), "rows_per_shard": [hi - lo for lo, hi in shards], "queries": nq,
                              "query_ms": round(float(np.median(ts)) * 1e3, 3), "merge_ms": round(float(np.median(tm)) * 1e3, 3),
                              "merge_note": "merge_ms = orb_slam2_amd.sharding.merge_nn (numpy mirror of the library's host merge) over the G per-shard answers; query_ms = orbhip_pool_db_query, "
                                            "host queries in, host answers out: upload to every device, the G scans side by side, download, merge",
                              "parity": bool(all(np.array_equal(a, b) for a, b in zip(got, whole)) and all(np.array_equal(a, b) for a, b in zip(merged, whole)))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="measurements of --steps steps each; value = the median one")
    ap.add_argument("--no-host-io", action="store_true", help="skip the PCIe-inclusive host-buffer measurement (host_io object)")
    ap.add_argument("--batch", type=int, default=512, help="camera slots (frames) per step per GPU")
    ap.add_argument("--tsteps", type=int, default=4, help="distinct resident time steps cycled through")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the library splits each batch over")
    ap.add_argument("--blur-round-mode", type=int, default=1, help="cv::GaussianBlur rounding the extractor reproduces: 1 = the SSE2 column filter of x86-64 OpenCV builds "
                    "(what the reference computes on this x86 box; default), 0 = OpenCV's generic C++ path (DESIGN.md H2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic")
    ap.add_argument("--no-dropin-loop", action="store_true", help="skip the front-end loop through the reference's own callers (dropin_loop object)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary units with the reference beside them (matcher_calls, config5, config4 objects)")
    ap.add_argument("--host-io-only", action="store_true", help="internal: measure the host_io object in this (fresh) process and print it")
    ap.add_argument("--device", type=int, default=0, help="with --host-io-only: the GPU to use")
    ap.add_argument("--pool", action="store_true", help="time the one-process multi-GPU pool (orbhip_pool_*: one host thread per GPU) at the host-buffer boundary instead of "
                    "the one-process-per-GPU device-resident benchmark; prints its own JSON line (mode: pool)")
    ap.add_argument("--pool-cameras-per-gpu", type=int, default=128)
    ap.add_argument("--parity-slots", type=int, default=NSCENES, help="camera slots of the last timed step compared bit for bit with the oracle: by default one per distinct scene, "
                    "i.e. with the replica check every frame of the step (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="N > 1: skip the host-fed pool and the sharded descriptor database that rank 0 measures after the resident run")
    ap.add_argument("--extras-cameras-per-gpu", type=int, default=64)
    ap.add_argument("--extras-rounds", type=int, default=12)
    ap.add_argument("--extras-db-keyframes", type=int, default=10000, help="key frames (x 2000 descriptors) of the sharded database of config5_sharded")
    ap.add_argument("--extract-only", action="store_true", help="diagnostic: skip the matcher (NOT the metric's workload; the JSON line says so)")
    args = ap.parse_args()
    if args.host_io_only:
        from orb_slam2_amd import synth
        frames = np.zeros((1, 256, H, W), np.uint8)
        for s in range(64):
            fr = synth.frame_from_scene(synth.scene(W, H, seed=s), W, H, t=0, seed=s)
            for b in range(s, 256, 64):
                frames[0, b] = fr
        print(json.dumps(host_io(None, frames, args.device, args.blur_round_mode)), flush=True)
        return
    if args.pool:
        if args.steps == 150:
            args.steps = 20
        return pool_mode(args)

    # No framework in this process, at any N: the library runs on the HIP runtime its RUNPATH names (/opt/rocm — the one the tests, smoke()
    # and a real ORB_SLAM2 binary use), frames are made resident through the library (orbhip_device_alloc), and the ranks of an N > 1 run
    # (started by torch.distributed.run, which only spawns them and sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) rendezvous over a
    # Unix socket.  Importing torch here would map its bundled ROCm-7.0 libamdhip64 beside the system one (its libraries NEED the
    # unversioned file name under RPATH $ORIGIN, so the loader does not reuse the mapped libamdhip64.so.7): measured, profiles/r03_*.
    # `runtime.mapped` in the JSON line lists every libamdhip64 /proc/self/maps shows — it must be one file.
    import orb_slam2_amd
    share = os.environ.get("ORB_BENCH_SHARE_GPU") == "1"
    ngpu = orb_slam2_amd.device_count()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # launched bare: become the launcher.  One rank per GPU; never a silent 1-GPU run labelled N.
        if ngpu < args.gpus and not share:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {ngpu} GPU(s) visible on this box — refusing to report a {args.gpus}-GPU number")
        import socket
        import subprocess
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if ngpu < 1:
        raise SystemExit("bench.py needs a GPU: the ORB front-end has no CPU fallback")
    # ORB_BENCH_SHARE_GPU=1 (test aid): all ranks use GPU 0 and rendezvous over gloo, to exercise the N>1 code path on a 1-GPU box
    if not share and ngpu < world:
        raise SystemExit(f"bench.py: {world} ranks but only {ngpu} GPU(s) visible — refusing to share a GPU between ranks")
    if share:
        local_rank = 0
    runtime_line = orb_slam2_amd.runtime_info()                      # first device touch: on the library's runtime
    from orb_slam2_amd.sharding import NodeRendezvous
    rdzv = NodeRendezvous(rank, world)                                # N > 1: barrier + max-reduce of the timings over a Unix socket; no data-path collective exists
    control = "none (single process)" if world == 1 else "unix-socket rendezvous among the node's ranks: barrier + max-reduce of the timings only (no framework, no collective)"

    B, T = args.batch, max(args.tsteps, 2)
    pitch = (W + 63) // 64 * 64
    host_frames = make_frames(B, T, pitch, rank)
    d_frames = orb_slam2_amd.DeviceBuffer.from_array(host_frames, device=local_rank)          # inputs resident in HBM before timing
    ex = orb_slam2_amd.ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, W, H, max_batch=B, device=local_rank, num_streams=args.streams, blur_round_mode=args.blur_round_mode)
    frame_stride, step_stride = H * pitch, B * H * pitch
    base = d_frames.ptr

    def step(i):
        ex.extract_device(base + (i % T) * step_stride, B, frame_stride, pitch, match_prev=not args.extract_only, window=WINDOW, nnratio=NNRATIO, check_ori=True)

    def barrier():
        orb_slam2_amd.device_synchronize(local_rank)                 # hipDeviceSynchronize: every stream of this rank's GPU
        rdzv.barrier()

    for i in range(args.warmup):
        step(i)
    ex.sync()
    ex.profile_enable(True)
    ex.profile_reset()
    times = []
    for r in range(max(args.repeats, 1)):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + r * args.steps + i)
        ex.sync()
        barrier()
        times.append(time.perf_counter() - t0)
    prof = ex.profile()
    ex.profile_enable(False)
    last_step = args.warmup + max(args.repeats, 1) * args.steps - 1  # the step whose results are still in the context
    mine = sorted(times)[len(times) // 2]                            # this rank's own median repeat
    spread = rdzv.allreduce_max([mine, -mine])                       # slowest and fastest rank (max of the negatives = the minimum)
    times = rdzv.allreduce_max(times)                                # per repeat: the slowest rank
    elapsed = sorted(times)[len(times) // 2]                         # the median repeat is the reported one

    # the timed work's results (not timed): what the last step left in the context
    kps, descs = ex.fetch(B)
    m12, nm = ([None] * B, np.zeros(B, np.int32)) if args.extract_only else ex.fetch_matches(B)
    nkp = [len(k) for k in kps]
    parity = parity_of_last_step(host_frames, kps, descs, m12, last_step, T, args.blur_round_mode, args.parity_slots) if rank == 0 and args.parity_slots > 0 else None

    alg_bytes_per_frame = ex.algorithmic_bytes_per_frame()
    extras = None
    if world > 1 and not args.no_extras:
        ex.close(); d_frames.free()                                   # every rank leaves its GPU; rank 0 then drives all of them from one process
        rdzv.barrier()
        if rank == 0:
            try:
                extras = multi_gpu_extras(args, world, share)
            except Exception as e:                                    # noqa: BLE001 - a secondary figure must not cost the line
                extras = {"error": str(e)[:300]}
        rdzv.barrier()
    if rank == 0:
        frames_total = B * args.steps * world
        value = frames_total / elapsed
        kern = {k: {"ms_per_launch": (v["total_ms"] / v["launches"]) if v["launches"] else 0.0, "launches": v["launches"],
                    "alg_bytes_per_frame": v["alg_bytes_per_frame"]} for k, v in prof.items()}
        total_kernel_ms = sum(v["ms_per_launch"] for v in kern.values())
        dom = max(kern, key=lambda k: kern[k]["ms_per_launch"])
        # roofline of the dominant kernel; kernels outside the byte formula (quadtree, matcher: latency-bound control
        # work on a few KB) get the bytes they actually stream (candidate / keypoint records), see DESIGN.md §4
        alg = kern[dom]["alg_bytes_per_frame"]
        note = None
        if alg == 0:
            alg = {"k_quadtree": 16 * 12 * 1024, "k_match_grid": NFEAT * 32, "k_match_candidates": 434 * 40 * 36, "k_match_select": 434 * 40 * 4}.get(dom, 0)
            note = "dominant kernel is latency-bound list/control work outside B(W,H,N); bytes = records it streams (DESIGN.md §4)"
        # roofline.traffic is the HBM bytes counted IN THIS RUN: PMC counters need rocprofv3 around the process, so a plain run reports
        # null; the committed PMC passes of the same workload (separate rocprofv3 --pmc runs, guide's x2 FETCH_SIZE correction) are quoted
        # beside it as traffic_profiled, with their source.
        traffic = None
        traffic_profiled = None
        try:
            for name, pmc_batch in PMC_SUMMARIES:
                pmc = os.path.join(ROOT, "profiles", name)
                if not os.path.exists(pmc):
                    continue
                with open(pmc) as f:
                    for row in json.load(f):
                        if row["kernel"].split("<")[0] == dom:          # (k_fast_cells is a template: "k_fast_cells<48, 40>" in the profiler's rows)
                            raw_kb = 2 * row["fetch_KB_per_dispatch_raw"] + row["write_KB_per_dispatch_raw"] if "fetch_KB_per_dispatch_raw" in row else \
                                (2 * row["fetch_MB_per_dispatch_raw"] + row["write_MB_per_dispatch_raw"]) * 1024
                            traffic_profiled = {"bytes_per_launch": int(raw_kb * 1024 * B / pmc_batch),
                                                "source": f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE at B = {pmc_batch}" + (f", scaled to B = {B}" if B != pmc_batch else "") + "; not measured in this run)"}
                break
        except Exception:
            traffic_profiled = None
        dur_s = kern[dom]["ms_per_launch"] * 1e-3
        achieved = (alg * B / dur_s) / 1e9 if dur_s > 0 else 0.0
        rv = valu_roofline(dom, kern[dom]["ms_per_launch"], B)
        # what bounds the dominant kernel, by the evidence in this line: the VALU-issue floor when the kernel runs within 25 % of it, HBM otherwise
        bound = "valu" if rv and rv["frac"] and rv["frac"] >= 0.75 and achieved / HBM_PEAK_GBS < 0.5 else "hbm"
        out = {
            "metric": "frames/s ORB extract+match, 1241x376 gray, 2000 kpts, 8 lvls" if not args.extract_only else "DIAGNOSTIC extract only (not the metric)",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "KITTI-shaped 1241x376 u8, 2000 features, 8 levels, scale 1.2, FAST 20/7, extract + SearchForInitialization(win 100, nnratio 0.9)",
                       "inputs": "hbm-resident, results stay on device (the PCIe-inclusive host-buffer rate is the host_io object)",
                       "frames_per_step_per_gpu": B, "resident_time_steps": T, "distinct_scenes": min(B, NSCENES), "blur_round_mode": args.blur_round_mode, "row_pitch": pitch, "streams_per_gpu": args.streams, "parallelism": f"frames sharded over {world} GPU(s), no collective",
                       "blur_kernel": "k_blur (VALU)" if os.environ.get("ORBHIP_BLUR") == "valu" else "k_blur_mfma (i8 matrix cores)"},
            "per_rank": {"frames_per_s_slowest_rank": round(B * args.steps / spread[0], 1), "frames_per_s_fastest_rank": round(B * args.steps / -spread[1], 1),
                         "note": "each rank's own median repeat (its B x steps frames / its time); value divides all ranks' frames by the slowest rank's time per repeat"},
            "repeats": {"n": len(times), "steps_each": args.steps, "timed_region_s": round(sum(times), 3), "frames_per_s_median": round(frames_total / elapsed, 1),
                        "frames_per_s_min": round(frames_total / max(times), 1), "frames_per_s_max": round(frames_total / min(times), 1)},
            "roofline": {"bound": bound, "bound_note": "achieved / peak / frac are the HBM figures north_star asks for; `bound` names what the evidence says limits the kernel (roofline_valu.frac)",
                         "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_profiled": traffic_profiled,
                         "alg_bytes_per_launch": alg * B, "launch_ms": round(kern[dom]["ms_per_launch"], 4), "note": note},
            "roofline_valu": rv,
            "pipeline_roofline": {"alg_bytes_per_frame": alg_bytes_per_frame,
                                  "achieved_GBps": round(alg_bytes_per_frame * value / world / 1e9, 2),
                                  "frac_of_hbm_peak": round(alg_bytes_per_frame * value / world / 1e9 / HBM_PEAK_GBS, 5)},
            "kernels_ms_per_launch": {k: round(v["ms_per_launch"], 4) for k, v in kern.items()},
            "kernel_time_fraction_of_step": round(total_kernel_ms / (elapsed / args.steps * 1e3), 3),
            "check": {"keypoints_per_frame_min_max": [int(min(nkp)), int(max(nkp))], "matches_per_frame_min_max": [int(nm.min()), int(nm.max())]},
            "parity": parity,
            "runtime": {"library": runtime_line, "mapped": orb_slam2_amd.mapped_hip_runtimes(), "framework_imported": "torch" in sys.modules, "control_plane": control},
        }
        closed = world > 1 and not args.no_extras
        if extras is not None:
            out.update(extras)
        if parity and (parity["mismatches"] or parity["replica_mismatches"]):
            out["INVALID"] = "results of the timed region differ from the oracle: the throughput above does not count"
        if world == 1 and not args.no_traffic:
            ex.close(); d_frames.free(); closed = True
            t, why = hbm_traffic_subprocess(dom, B, args)
            if t:
                out["roofline"]["traffic"] = t["bytes_per_launch"]
                out["roofline"]["traffic_measured"] = t
                out["roofline"]["traffic_over_algorithmic"] = round(t["bytes_per_launch"] / max(alg * B, 1), 3)
                out["roofline_valu"] = valu_roofline(dom, kern[dom]["ms_per_launch"], B, measured=t)       # the instruction count of THIS run replaces the committed summary's
            else:
                out["roofline"]["traffic_note"] = why
        if world == 1 and not args.no_host_io:
            # measured in a process of its own: in a process that holds (or has held) a large resident context the DMA path from / into the caller's
            # pinned buffers delivers 93-100 k frames/s instead of the 112-117 k of a fresh process, on every box tried, while the pageable path is
            # unaffected (profiles/r03_exp_host_io_pinned_vs_pageable_by_box.jsonl; cause not found) - and the drop-in's callers are such fresh processes
            if not closed:
                ex.close(); d_frames.free(); closed = True
            out["host_io"] = host_io_subprocess(local_rank, args.blur_round_mode)
            try:        # the same measurement in THIS process, which has held the 5 GB resident context: the pinned path used to come out 10-15 % slower here (open defect)
                inproc = host_io(None, host_frames, local_rank, args.blur_round_mode, budget_s=1.5)
                out["host_io"]["in_this_process"] = {k: inproc[k]["frames_per_s"] for k in ("pinned", "pageable")}
                out["host_io"]["in_this_process"]["single_frame_call_ms"] = inproc["single_frame_call_ms"]
            except Exception as e:                                       # noqa: BLE001 - a secondary figure must not cost the line
                out["host_io"]["in_this_process"] = {"error": str(e)[:200]}
        if world == 1 and not args.no_dropin_loop:
            if not closed:
                ex.close(); d_frames.free(); closed = True               # the loop runs beside nothing, like the host_io object
            out["dropin_loop"] = dropin_loop_subprocess(args.blur_round_mode)
        if world == 1 and not args.no_secondary:
            if not closed:
                ex.close(); d_frames.free(); closed = True
            sec = secondary_units_subprocess(args.blur_round_mode)
            for key in ("matcher_calls", "config5", "config4"):
                out[key] = sec.get(key, {"error": sec.get("error", "missing")})
        if world == 1 and not args.no_cpu_baseline:
            os.environ["ORB_REF_BLUR_ROUND_MODE"] = str(args.blur_round_mode)      # the reference build's GaussianBlur stand-in follows the same rounding
            out["cpu_baseline"] = cpu_baseline(host_frames, blur_round_mode=args.blur_round_mode)
            out["cpu_baseline"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    rdzv.barrier()
    rdzv.close()


if __name__ == "__main__":
    main()
