"""N>1 path on CPU: world_size-2 gloo processes exercise the interval sharding (LPT partition + one ragged
all_gather) and the bench timing reduction.  The per-rank DP worker here is the CPU oracle (test infrastructure);
on the GPU box the worker is Context.dp_batch (tests/test_gpu_align.py::test_dp_sharded_single_process)."""
import os
import socket

import numpy as np
import pytest

from mauvealigner_amd import parallel, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _intervals(seed=7, n=23, nseq=3):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        base = rng.integers(0, 4, int(rng.integers(1, 60)), dtype=np.uint8)
        out.append([synth.mutate(base, 0.15, rng, indel_frac=0.3) if rng.random() > 0.1 else np.zeros(0, np.uint8)
                    for _ in range(nseq)])
    return out


def _oracle_dp(ivs):
    from oracle import pyoracle as O
    cols, scores = [], []
    for iv in ivs:
        c, s = O.align_interval(iv)
        cols.append(c)
        scores.append(s)
    return cols, np.array(scores, dtype=np.int64)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ivs = _intervals()
        cols, score = parallel.dp_sharded(_oracle_dp, ivs, dist)
        t, bp = parallel.reduce_throughput(1.0 + rank, 1000 * (rank + 1), dist)
        q.put((rank, [c.tolist() for c in cols], score.tolist(), t, bp))
    finally:
        dist.destroy_process_group()


class _StandInContext:
    """The begin / dp / finish protocol of mauvealigner_amd._lib.Context (mauve_align_begin / _dp / _finish) with the
    CPU oracle behind it: lets align_sharded itself run under gloo without a GPU."""

    def __init__(self, intervals):
        self.ivs = intervals
        self.finished = None

    def align_begin(self, params=None):
        cost = np.array([parallel.interval_cost(iv) for iv in self.ivs], np.int64)
        cap = np.array([sum(len(s) for s in iv) for iv in self.ivs], np.int64)
        return len(self.ivs), cost, cap

    def align_dp(self, idx, cap):
        cols, score = _oracle_dp([self.ivs[i] for i in np.asarray(idx).tolist()])
        return [np.asarray(c, np.uint32) for c in cols], score, int(sum(parallel.interval_cost(self.ivs[i]) for i in np.asarray(idx).tolist()))

    def align_finish(self, cols_list, scores, cells, fetch=True, names=None, want_xmfa=False):
        assert all(c is not None for c in cols_list)
        self.finished = {"cols": [np.asarray(c).tolist() for c in cols_list], "score": np.asarray(scores).tolist(), "cells": int(cells)}
        return self.finished


def _worker_align(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = parallel.align_sharded(_StandInContext(_intervals(seed=11, n=31)), None, dist)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_align_sharded_world2_gloo():
    """align_sharded end to end with two ranks (begin, LPT split, one ragged all_gather of uint32 columns, finish):
    every rank assembles the whole result, equal to the single-rank run."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_align, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ivs = _intervals(seed=11, n=31)
    want = parallel.align_sharded(_StandInContext(ivs), None, None)
    assert want["cells"] == sum(parallel.interval_cost(iv) for iv in ivs)
    for rank, out in res:
        assert out == want


def test_lpt_partition_deterministic_and_balanced():
    costs = [5, 1, 9, 9, 2, 7, 3, 3, 0, 12]
    parts = parallel.lpt_partition(costs, 3)
    assert sorted(np.concatenate(parts).tolist()) == list(range(len(costs)))
    loads = [int(np.take(costs, p).sum()) for p in parts]
    assert max(loads) - min(loads) <= max(costs)
    again = parallel.lpt_partition(costs, 3)
    assert all(np.array_equal(a, b) for a, b in zip(parts, again))
    assert [len(p) for p in parallel.lpt_partition([], 4)] == [0, 0, 0, 0]
    assert parallel.interval_cost([np.zeros(3), np.zeros(0), np.zeros(4), np.zeros(2)]) == 12 + 14


def test_dp_sharded_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_cols, want_score = _oracle_dp(_intervals())
    for rank, cols, score, t, bp in res:
        assert score == want_score.tolist()
        assert cols == [c.tolist() for c in want_cols]
        assert t == 2.0 and bp == 3000.0          # MAX over ranks, SUM over ranks


def _allgather_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ag = parallel.make_allgather(dist)
        out = []
        for rnd in range(3):                                   # ragged payloads, an empty one among them
            mine = bytes([rank + 1]) * (0 if (rnd == 1 and rank == 0) else 5 + 7 * rank + rnd)
            parts = ag(mine)
            out.append([bytes(p) for p in parts])
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_shard_allgather_world2_gloo():
    """The collective behind mauve_set_shard (parallel.make_allgather): every rank gets every rank's bytes, in rank order, for
    payloads of different and of zero length -- the protocol the C side relies on (include/mauve_hip.h: mauve_allgather_fn)."""
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    ps = [ctxm.Process(target=_allgather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert got[0] == got[1]
    for rnd in range(3):
        exp = [bytes([r + 1]) * (0 if (rnd == 1 and r == 0) else 5 + 7 * r + rnd) for r in range(2)]
        assert got[0][rnd] == exp


def test_shard_lpt_is_deterministic():
    """parallel.lpt_partition (the Python twin of the C side's shard_lpt): heaviest first, ties to the lower index / rank."""
    parts = parallel.lpt_partition([5, 5, 3, 3, 1], 2)
    assert [p.tolist() for p in parts] == [[0, 2, 4], [1, 3]]
