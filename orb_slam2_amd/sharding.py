"""Sharding of the ORB front-end across the GPUs of one node: independent units, no collective.

* frames / cameras: camera slot c belongs to rank c mod G; every rank runs its own context on its own GPU.
* descriptor DB (BASELINE.json config 5): the DB is split by contiguous row range, each rank answers every query
  over its shard, and the G partial answers per query are merged on the host with exactly the matcher's rule
  (strict '<', lowest global index wins ties; second = second smallest of the union) — 2000 x G x 16 B, far too
  small to be worth an RCCL all-gather over xGMI.
"""
import numpy as np

INT_MAX = 2 ** 31 - 1


def camera_slots(num_cameras, rank, world_size):
    """Camera slots owned by `rank` (round-robin, like `c mod G`)."""
    return list(range(rank, num_cameras, world_size))


def db_shard(num_rows, rank, world_size):
    """Contiguous row range [lo, hi) of the descriptor DB owned by `rank` (sizes differ by at most one row)."""
    base, rem = divmod(num_rows, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def merge_nn(parts):
    """Merge per-shard brute-force answers given in ascending shard (= ascending global index) order.

    parts: list of (best_idx int64[nq] (global, -1 = none), best_dist int32[nq], second_dist int32[nq]).
    Reproduces a single left-to-right scan: `if d < best: second = best; best = d; idx = i  elif d < second: second = d`
    (ORBmatcher.cc:447-456 idiom).
    """
    idx, best, second = (np.array(a, copy=True) for a in parts[0])
    idx = idx.astype(np.int64)
    best = best.astype(np.int64)
    second = second.astype(np.int64)
    for pi, pb, ps in parts[1:]:
        pi = np.asarray(pi, np.int64)
        pb = np.asarray(pb, np.int64)
        ps = np.asarray(ps, np.int64)
        take = pb < best                                   # strict: an equal distance keeps the earlier shard's index
        new_second = np.where(take, np.minimum(best, ps), np.minimum(second, pb))
        idx = np.where(take, pi, idx)
        best = np.where(take, pb, best)
        second = new_second
    return idx, best.astype(np.int32), second.astype(np.int32)


class NodeRendezvous:
    """Barrier + max-reduce among the ranks of ONE node, without a framework in the process.

    The data path has no collective (ranks own whole cameras / DB row ranges), so the only thing ranks ever exchange is a handful of
    floats around a timed region.  Importing a framework for that would map its bundled HIP runtime into the process next to the system
    runtime the library runs on; a Unix-domain socket does the same job: rank 0 listens on a path derived from the launcher's
    MASTER_PORT (and elastic run id), every other rank connects, and `allreduce_max(values)` = send, element-wise max on rank 0, reply.
    `barrier()` is the same exchange with one dummy value.  One node only (the launcher contract here is --nnodes=1).
    """

    def __init__(self, rank, world, key=None, timeout_s=300.0):
        import os
        import time
        from multiprocessing.connection import Client, Listener
        self.rank, self.world, self.conns = int(rank), int(world), []
        if self.world <= 1:
            return
        key = key or f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}"
        self.path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"orbhip_rdzv_{os.getuid()}_{key}.sock")
        auth = key.encode()
        if self.rank == 0:
            try:
                os.unlink(self.path)
            except OSError:
                pass
            self.listener = Listener(self.path, family="AF_UNIX", authkey=auth)
            peers = {}
            self.listener._listener._socket.settimeout(timeout_s)
            while len(peers) < self.world - 1:
                c = self.listener.accept()
                peers[int(c.recv())] = c
            self.conns = [peers[r] for r in sorted(peers)]
        else:
            t0 = time.time()
            while True:
                try:
                    c = Client(self.path, family="AF_UNIX", authkey=auth)
                    break
                except (FileNotFoundError, ConnectionRefusedError):
                    if time.time() - t0 > timeout_s:
                        raise TimeoutError(f"rank {self.rank}: no rendezvous listener at {self.path} after {timeout_s:.0f} s")
                    time.sleep(0.02)
            c.send(self.rank)
            self.conns = [c]

    def allreduce_max(self, values):
        """Element-wise max over all ranks of a list of floats; returns the reduced list on every rank (also a barrier)."""
        values = [float(v) for v in values]
        if self.world <= 1:
            return values
        if self.rank == 0:
            got = [values] + [c.recv() for c in self.conns]
            out = [max(col) for col in zip(*got)]
            for c in self.conns:
                c.send(out)
            return out
        self.conns[0].send(values)
        return self.conns[0].recv()

    def barrier(self):
        self.allreduce_max([0.0])

    def close(self):
        import os
        for c in self.conns:
            try:
                c.close()
            except OSError:
                pass
        self.conns = []
        if self.world > 1 and self.rank == 0:
            try:
                self.listener.close()
            except OSError:
                pass
            try:
                os.unlink(self.path)
            except OSError:
                pass
