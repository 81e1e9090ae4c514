"""Multi-GPU path on CPU: world_size-2 gloo processes exercise the sharding logic the node-level run uses
(camera slots round-robin over ranks; descriptor-DB row shards answered per rank, merged on the host with the matcher's
strict-'<' / lowest-index rule).  There is no data-path collective to test — that is the design."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, emu_lib, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import orb_slam2_amd
    from orb_slam2_amd import sharding, synth
    # (1) descriptor DB sharded by rows: every rank answers all queries over its shard (kernel logic via the emulation build)
    db = synth.descriptor_db(5, 500, seed=7)
    qd = synth.descriptor_query(db, 64, seed=7)
    lo, hi = sharding.db_shard(len(db), rank, world)
    part = orb_slam2_amd.hamming_nn(qd, db[lo:hi], index_base=lo, library=emu_lib)
    gathered = [None] * world
    dist.all_gather_object(gathered, part)            # host-side gather of 64 x 16 B per rank (control plane, not a data-path collective)
    merged = sharding.merge_nn(gathered)
    # (2) camera slots: rank r owns slots r, r+world, ...; each extracts its own frames independently
    slots = sharding.camera_slots(5, rank, world)
    ex = orb_slam2_amd.ORBextractor(200, 1.2, 8, 20, 7, 320, 240, max_batch=max(len(slots), 1), library=emu_lib)
    ks, ds = ex.extract_batch([synth.frame(320, 240, seed=40 + s) for s in slots])
    mine = {s: (ks[i].tobytes(), ds[i].tobytes()) for i, s in enumerate(slots)}
    allres = [None] * world
    dist.all_gather_object(allres, mine)
    # barrier + max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((merged, allres, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_gloo(oracle, emu_lib):
    import torch.multiprocessing as mp
    from orb_slam2_amd import sharding, synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, emu_lib, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, allres, tmax = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert tmax == 2.0
    db = synth.descriptor_db(5, 500, seed=7)
    qd = synth.descriptor_query(db, 64, seed=7)
    want = oracle.bf_nn(qd, db)
    assert all(np.array_equal(a, b) for a, b in zip(merged, want))          # sharded answer == single left-to-right scan
    got = {}
    for d in allres:
        got.update(d)
    assert sorted(got) == [0, 1, 2, 3, 4]
    ora = oracle.OracleExtractor(200, 1.2, 8, 20, 7)
    for s in range(5):
        k, d = ora.extract(synth.frame(320, 240, seed=40 + s))
        assert got[s] == (k.tobytes(), d.tobytes())


def test_shard_helpers():
    from orb_slam2_amd import sharding
    for n in (0, 1, 7, 20000000):
        for w in (1, 2, 3, 8):
            spans = [sharding.db_shard(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    assert sharding.camera_slots(8, 3, 8) == [3] and sharding.camera_slots(10, 1, 4) == [1, 5, 9]
    # merge rule: ties keep the earlier shard, second = second smallest of the union
    a = (np.array([5]), np.array([10], np.int32), np.array([12], np.int32))
    b = (np.array([900]), np.array([10], np.int32), np.array([11], np.int32))
    idx, best, second = sharding.merge_nn([a, b])
    assert (idx[0], best[0], second[0]) == (5, 10, 10)
    idx, best, second = sharding.merge_nn([b, a])
    assert (idx[0], best[0], second[0]) == (900, 10, 10)
    c = (np.array([7]), np.array([3], np.int32), np.array([40], np.int32))
    idx, best, second = sharding.merge_nn([a, c])
    assert (idx[0], best[0], second[0]) == (7, 3, 10)


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` launched bare becomes its own launcher — and must fail loudly, not report a 1-GPU number, when the box
    has fewer than N GPUs (here: none)."""
    import subprocess
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("ORB_BENCH_SHARE_GPU", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "refusing" in (r.stderr + r.stdout) and '"n_gpus"' not in r.stdout
    # a launcher that started a different number of ranks than --gpus says is an error too
    env["WORLD_SIZE"] = "2"; env["RANK"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_gpus_flag_on_this_box():
    """On the GPU box: `python bench.py --gpus N` with N > the visible GPUs fails loudly; with N <= visible it launches N ranks itself and
    rank 0 reports n_gpus = N."""
    import json
    import subprocess
    import orb_slam2_amd
    g = orb_slam2_amd.device_count()                  # the library's own count: torch (and its bundled HIP runtime) stays out of the test process
    assert g >= 1
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("ORB_BENCH_SHARE_GPU", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(g + 1), "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout) and '"n_gpus"' not in r.stdout
    if g >= 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "1", "--batch", "32", "--no-cpu-baseline", "--no-host-io"],
                           env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        assert json.loads(line)["n_gpus"] == 2


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_gpu():
    """The 8-GPU command the driver runs at round end, first time right: `bench.py --gpus 8` becomes its own launcher, eight ranks rendezvous,
    every rank builds its resident batch and context, the timed region is bracketed by the 8-way barrier, rank 0 prints ONE JSON line with
    n_gpus = 8 and a bit-exact parity object.  No 8-GPU node is available to the builder, so the eight ranks share this box's GPU
    (ORB_BENCH_SHARE_GPU=1: a test aid bench.py names in its config line; the number it prints is not a scaling figure)."""
    import json
    import subprocess
    import time
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env["ORB_BENCH_SHARE_GPU"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--batch", "32", "--steps", "3", "--warmup", "1", "--repeats", "2",
                        "--no-cpu-baseline", "--no-host-io", "--no-traffic", "--no-dropin-loop",
                        "--extras-cameras-per-gpu", "4", "--extras-rounds", "3", "--extras-db-keyframes", "40"], env=env, capture_output=True, text=True, timeout=1200)
    wall = time.time() - t0
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["steps"] == 3 and d["unit"] == "frames/s" and d["value"] > 0
    assert d["config"]["frames_per_step_per_gpu"] == 32 and "8 GPU" in d["config"]["parallelism"]
    assert d["parity"]["mismatches"] == 0 and d["parity"]["replica_mismatches"] == 0 and "INVALID" not in d
    assert "unix-socket rendezvous" in d["runtime"]["control_plane"] and d["runtime"]["framework_imported"] is False
    # N > 1 also carries the two curves that can bend: the one-process pool from host buffers and the row-sharded descriptor database
    hp, c5 = d["host_io_pool"], d["config5_sharded"]
    assert hp["cameras"] == 32 and len(hp["numa_nodes"]) == 8 and hp["pinned"]["frames_per_s"] > 0 and hp["pageable"]["frames_per_s"] > 0 and hp["pinned"]["keypoints_per_frame"] > 1500
    assert c5["shards"] == 8 and sum(c5["rows_per_shard"]) == c5["rows"] == 80000 and c5["parity"] is True and c5["query_ms"] > 0 and c5["merge_ms"] >= 0
    assert d["per_rank"]["frames_per_s_fastest_rank"] >= d["per_rank"]["frames_per_s_slowest_rank"] > 0
    print(f"\n[8 ranks on one GPU] launcher + 8 ranks + rendezvous + run: {wall:.1f} s wall; {d['value']:.0f} frames/s over the shared GPU (not a scaling figure)")


def _rdzv_worker(rank, world, key, q):
    sys.path.insert(0, ROOT)
    from orb_slam2_amd.sharding import NodeRendezvous
    r = NodeRendezvous(rank, world, key=key, timeout_s=60)
    r.barrier()
    a = r.allreduce_max([float(rank), 10.0 - rank, 3.5])
    b = r.allreduce_max([1.0 / (rank + 1)])
    r.barrier()
    r.close()
    q.put((rank, a, b))


@pytest.mark.parametrize("world", [2, 8])
def test_node_rendezvous_barrier_and_max(world):
    """bench.py's N > 1 control plane (framework-free): element-wise max over ranks, identical on every rank, repeated exchanges."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = f"test_{os.getpid()}_{world}"
    ps = [ctx.Process(target=_rdzv_worker, args=(r, world, key, q)) for r in range(world)]
    for p in ps[::-1]:                                   # the listener (rank 0) starts last: clients must wait for it
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, a, b in got:
        assert a == [float(world - 1), 10.0, 3.5] and b == [1.0]
    from orb_slam2_amd.sharding import NodeRendezvous
    solo = NodeRendezvous(0, 1)
    assert solo.allreduce_max([2.0, 1.0]) == [2.0, 1.0]
    solo.barrier(); solo.close()
