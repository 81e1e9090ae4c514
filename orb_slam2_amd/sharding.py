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
