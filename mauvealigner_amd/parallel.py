"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU).

The hot path shards at the level the reference itself names (mauveAligner.cpp:130-131 `--realign-lcb`, "for
parallelization of LCB alignment"): inter-anchor intervals are independent units.  Two shapes are provided:

* weak scaling (bench.py): every rank aligns its own genome set, no data-path collective; only the timing is
  reduced (MAX over ranks) and the aligned base pairs are summed.
* interval sharding: the DP intervals of one alignment are LPT-partitioned over the ranks by estimated cells,
  each rank runs its share through the GappedAligner seam (mauve_dp_batch) and the ragged results are
  exchanged with ONE all_gather (messages are small, latency-bound; no ring all-reduce anywhere).
"""
import numpy as np


def lpt_partition(costs, world):
    """Longest-processing-time-first packing of `costs` into `world` bins.  Deterministic: ties go to the lower
    index / lower bin.  Returns a list of sorted index arrays, one per rank."""
    costs = np.asarray(costs, dtype=np.int64)
    order = np.lexsort((np.arange(len(costs)), -costs))
    load = np.zeros(world, dtype=np.int64)
    bins = [[] for _ in range(world)]
    for i in order.tolist():
        b = int(np.argmin(load))
        bins[b].append(i)
        load[b] += int(costs[i])
    return [np.array(sorted(b), dtype=np.int64) for b in bins]


def interval_cost(interval):
    """DP cells of the progressive alignment of one interval (same count as the kernel reports)."""
    m, cells = 0, 0
    for s in interval:
        n = len(s)
        if n == 0:
            continue
        if m == 0:
            m = n
            continue
        cells += m * n
        m += n           # upper bound of the merged profile length
    return cells


def all_gather_ragged(arr, dist, group=None):
    """all_gather of one 1-D numpy array per rank with different lengths: one size exchange + one padded
    all_gather.  The payload travels in its own width (uint32 column masks as int32 bit patterns: no widening), and
    on RCCL through ONE device buffer per rank (H2D of the rank's share, the collective, D2H of everybody's).
    Returns the list of arrays by rank."""
    import torch
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    arr = np.ascontiguousarray(arr)
    wire = {np.dtype(np.uint32): np.int32, np.dtype(np.uint64): np.int64}.get(arr.dtype)
    payload = arr.view(wire) if wire is not None else arr
    n = torch.tensor([payload.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = torch.zeros(mx, dtype=torch.from_numpy(payload[:0].copy()).dtype, device=dev)
    if payload.size:
        buf[:payload.size] = torch.from_numpy(payload.reshape(-1)).to(dev, non_blocking=True)
    out = torch.empty(world * mx, dtype=buf.dtype, device=dev)
    dist.all_gather_into_tensor(out, buf, group=group)
    host = out.cpu().numpy()
    return [host[r * mx:r * mx + s].view(arr.dtype).copy() for r, s in enumerate(sizes)]


def dp_sharded(dp_fn, intervals, dist=None, group=None):
    """Align `intervals` (list of lists of code arrays) with the per-rank worker `dp_fn(list) -> (cols list,
    scores)`, sharded over the ranks of `dist`.  Every rank returns the full (cols list, scores) in input order."""
    n = len(intervals)
    if dist is None or dist.get_world_size(group) == 1:
        return dp_fn(intervals)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts = lpt_partition([interval_cost(iv) for iv in intervals], world)
    mine = parts[rank]
    cols, score = dp_fn([intervals[i] for i in mine.tolist()]) if len(mine) else ([], np.zeros(0, np.int64))
    lens = np.array([len(c) for c in cols], dtype=np.int64)
    flat = np.concatenate(cols).astype(np.uint32) if len(cols) and lens.sum() else np.zeros(0, np.uint32)
    g_lens = all_gather_ragged(lens, dist, group)
    g_flat = all_gather_ragged(flat, dist, group)
    g_score = all_gather_ragged(np.asarray(score, dtype=np.int64), dist, group)
    out_cols = [None] * n
    out_score = np.zeros(n, dtype=np.int64)
    for r in range(world):
        off = 0
        for k, i in enumerate(parts[r].tolist()):
            ln = int(g_lens[r][k])
            out_cols[i] = g_flat[r][off:off + ln].astype(np.uint32)
            out_score[i] = g_score[r][k]
            off += ln
    return out_cols, out_score


def reduce_throughput(elapsed_s, base_pairs, dist=None, group=None):
    """bench.py contract: MAX of the elapsed time over ranks, SUM of the aligned base pairs."""
    if dist is None or dist.get_world_size(group) == 1:
        return elapsed_s, float(base_pairs)
    import torch
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    b = torch.tensor([float(base_pairs)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()), float(b.item())


def make_allgather(dist, group=None):
    """The collective mauve_set_shard asks for, on torch.distributed: one size exchange + one padded all_gather_into_tensor of
    the byte payload (gloo: host tensors; nccl = RCCL: through one device buffer per rank).  Returns f(bytes) -> [bytes per rank]."""
    import torch
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"

    def allgather(payload):
        n = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
        sizes = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(sizes, n, group=group)
        sizes = [int(x) for x in sizes.cpu().tolist()]
        mx = max(max(sizes), 1)
        buf = torch.zeros(mx, dtype=torch.uint8, device=dev)
        if len(payload):
            buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
        out = torch.empty(world * mx, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(out, buf, group=group)
        host = out.cpu().numpy()
        return [host[r * mx:r * mx + s].tobytes() for r, s in enumerate(sizes)]
    return allgather


def attach_shard(ctx, dist, group=None):
    """One alignment over the ranks of `dist` (mauve_set_shard): after this, ctx.align / ctx.progressive_align / ctx.guide_tree on
    every rank (same genomes, same calls) deal out the guide tree's pairwise finder passes, the gaps of every recursion level and
    the gapped-alignment intervals of the guide-tree nodes, and every rank ends with the whole result."""
    world = 1 if dist is None else dist.get_world_size(group)
    if world <= 1:
        ctx.set_shard(0, 1, None)
        return
    ctx.set_shard(dist.get_rank(group), world, make_allgather(dist, group))


def progressive_align_sharded(ctx, params=None, dist=None, group=None, **kw):
    """mauve_progressive_align with the work of one alignment dealt over the ranks (attach_shard)."""
    attach_shard(ctx, dist, group)
    try:
        return ctx.progressive_align(params, **kw)
    finally:
        ctx.set_shard(0, 1, None)


def align_sharded(ctx, params=None, dist=None, group=None, names=None, want_xmfa=False, fetch=True, out=None):
    """One alignment, its gapped-alignment intervals sharded over the ranks (LCB sharding).

    Every rank holds the same genomes (ctx.set_genomes) and runs the deterministic front of the path
    (mauve_align_begin: seed pass, chaining, recursive anchoring).  The DP intervals are LPT-partitioned by cells,
    each rank aligns its share on its own GPU (mauve_align_dp), the ragged column lists are exchanged with one
    all_gather and every rank assembles the full result (mauve_align_finish).
    """
    n_dp, cost, cap = ctx.align_begin(params)
    world = 1 if dist is None else dist.get_world_size(group)
    rank = 0 if dist is None else dist.get_rank(group)
    parts = lpt_partition(cost, world) if n_dp else [np.zeros(0, np.int64) for _ in range(world)]
    mine = parts[rank]
    cols, score, cells = ctx.align_dp(mine, cap) if len(mine) else ([], np.zeros(0, np.int64), 0)
    if world == 1:
        all_cols, all_score, all_cells = [None] * n_dp, np.zeros(n_dp, np.int64), cells
        for k, i in enumerate(mine.tolist()):
            all_cols[i] = cols[k]
            all_score[i] = score[k]
    else:
        lens = np.array([len(c) for c in cols], dtype=np.int64)
        flat = np.concatenate(cols).astype(np.uint32) if len(cols) and lens.sum() else np.zeros(0, np.uint32)
        g_lens = all_gather_ragged(lens, dist, group)
        g_flat = all_gather_ragged(flat, dist, group)
        g_score = all_gather_ragged(np.asarray(score, dtype=np.int64), dist, group)
        g_cells = all_gather_ragged(np.array([cells], dtype=np.int64), dist, group)
        all_cols, all_score = [None] * n_dp, np.zeros(n_dp, np.int64)
        for r in range(world):
            off = 0
            for k, i in enumerate(parts[r].tolist()):
                ln = int(g_lens[r][k])
                all_cols[i] = g_flat[r][off:off + ln].astype(np.uint32)
                all_score[i] = g_score[r][k]
                off += ln
        all_cells = int(sum(int(x[0]) for x in g_cells))
    kw = {} if out is None else {"out": out}
    return ctx.align_finish(all_cols, all_score, all_cells, fetch=fetch, names=names, want_xmfa=want_xmfa, **kw)


# ---- RCCL inside the library (mauve_set_shard_rccl): what a C++ caller -- which is what the reference is -- uses.  From Python the communicator is
# made through ctypes on the RCCL the process resolves (librccl.so.1), not through torch.distributed, whose communicator handle is not exposed. ----
class RcclComm:
    """one rank's ncclComm_t, made with ncclCommInitRank on the current device; unique_id: bytes of rank 0's ncclGetUniqueId, carried to the
    other ranks by the caller (a file, a socket, torch.distributed.broadcast_object_list ...)"""
    UNIQUE_ID_BYTES = 128

    @staticmethod
    def new_unique_id(lib_path="librccl.so.1"):
        """rank 0's ncclGetUniqueId as bytes (to be carried to every rank before any of them makes its communicator)"""
        import ctypes as C
        lib = C.CDLL(lib_path, mode=C.RTLD_GLOBAL)
        uid = C.create_string_buffer(RcclComm.UNIQUE_ID_BYTES)
        r = lib.ncclGetUniqueId(uid)
        if r != 0:
            raise RuntimeError("ncclGetUniqueId failed (%d)" % r)
        return uid.raw

    def __init__(self, rank, world, unique_id=None, lib_path="librccl.so.1"):
        import ctypes as C
        self.C = C
        self.lib = C.CDLL(lib_path, mode=C.RTLD_GLOBAL)           # global: libmauve_hip resolves ncclAllGather from what the process has loaded
        if unique_id is None:
            assert world == 1, "several ranks: make the id on rank 0 (new_unique_id) and hand it to all of them"
            unique_id = self.new_unique_id(lib_path)
        self.unique_id = unique_id

        class _Uid(C.Structure):
            _fields_ = [("internal", C.c_char * self.UNIQUE_ID_BYTES)]
        u = _Uid(); C.memmove(C.byref(u), unique_id, self.UNIQUE_ID_BYTES)
        self.comm = C.c_void_p()
        self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _Uid, C.c_int]
        self._chk(self.lib.ncclCommInitRank(C.byref(self.comm), world, u, rank), "ncclCommInitRank")
        self.rank, self.world = rank, world

    def _chk(self, r, what):
        if r != 0:
            raise RuntimeError("%s failed (%d)" % (what, r))

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy.argtypes = [self.C.c_void_p]
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


def attach_shard_rccl(ctx, comm):
    """mauve_set_shard_rccl: the library runs the exchanges itself (ncclAllGather on its stream, device buffers)"""
    import ctypes as C
    ctx.L.mauve_set_shard_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    rc = ctx.L.mauve_set_shard_rccl(ctx.h, comm.rank, comm.world, comm.comm)
    if rc:
        raise RuntimeError("mauve_set_shard_rccl failed (%d): %s" % (rc, ctx.last_error() if hasattr(ctx, "last_error") else ""))


def shard_stats(ctx):
    import ctypes as C

    class S(C.Structure):
        _fields_ = [("exchanges", C.c_int64), ("bytes_sent", C.c_int64), ("bytes_received", C.c_int64), ("ms", C.c_double)]
    s = S()
    ctx.L.mauve_shard_get_stats.argtypes = [C.c_void_p, C.POINTER(S)]
    ctx.L.mauve_shard_get_stats(ctx.h, C.byref(s))
    return {"exchanges": int(s.exchanges), "bytes_sent": int(s.bytes_sent), "bytes_received": int(s.bytes_received), "ms": float(s.ms)}
