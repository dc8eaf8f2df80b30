"""ctypes loader for libmauve_hip.so (the C-ABI declared in include/mauve_hip.h).

Plumbing only: tests, bench.py and __graft_entry__ call the product through this module.  There is no
CPU fallback -- `Context()` raises if the library or a GPU is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmauve_hip.so")

MODE_MEM, MODE_UNIQUE, MODE_PAIRWISE = 0, 1, 2
CODING_SEED, SOLID_SEED = 3, 0x7FFFFFFF
K_EXTRACT, K_SORT_HIST, K_SORT_SCAN, K_SORT_SCATTER, K_JOIN, K_EXTEND, K_DP, K_RUNS = range(8)
KERNEL_NAMES = ["seed_extract", "rs_hist", "rs_rowscan", "rs_scatter", "mum_join", "mum_extend", "dp_step", "mum_runs", "canon_sort", "misc_sort"]

# every symbol include/mauve_hip.h declares (checked by tests/test_abi.py without a GPU)
EXPORTS = [
    "mauve_ctx_create", "mauve_ctx_destroy", "mauve_last_error", "mauve_device_name", "mauve_synchronize",
    "mauve_host_alloc", "mauve_host_free", "mauve_set_shard", "mauve_set_shard_rccl", "mauve_shard_get_stats",
    "mauve_get_seed", "mauve_seed_length", "mauve_seed_weight", "mauve_default_seed_weight", "mauve_default_scoring",
    "mauve_default_params", "mauve_default_progressive_params", "mauve_packed_words", "mauve_pack_ascii", "mauve_pack_codes", "mauve_set_genomes", "mauve_set_genomes_contigs",
    "mauve_ambiguity_bitmap",
    "mauve_sorted_mer_list", "mauve_seed_mums", "mauve_get_matches", "mauve_extend_hits", "mauve_seed_match_enumerate",
    "mauve_eliminate_overlaps", "mauve_lcb_chain", "mauve_dp_batch", "mauve_dp_batch_banded", "mauve_match_sp_scores", "mauve_align", "mauve_align_fetch", "mauve_align_fetch_compact",
    "mauve_align_matches", "mauve_align_lcbs", "mauve_align_begin", "mauve_align_begin_matches", "mauve_align_dp_anchors", "mauve_align_dp_cost", "mauve_align_dp", "mauve_align_finish",
    "mauve_guide_tree", "mauve_breakpoint_counts", "mauve_hmm_params_from", "mauve_apply_homology", "mauve_apply_homology_alignment", "mauve_progressive_align", "mauve_progressive_align_tree",
    "mauve_backbone", "mauve_backbone_alignment", "mauve_backbone_fetch", "mauve_merge_matches",
    "mauve_write_xmfa", "mauve_profile_enable", "mauve_profile_reset", "mauve_profile_get", "mauve_last_stage_times",
]


class Scoring(C.Structure):
    _fields_ = [("gap_open", C.c_int32), ("gap_extend", C.c_int32), ("matrix", (C.c_int32 * 4) * 4)]


class Params(C.Structure):
    _fields_ = [("seed_pattern", C.c_uint64), ("seed_weight", C.c_int32), ("seed_rank", C.c_int32),
                ("mode", C.c_int32), ("lcb_weight", C.c_int64), ("collinear", C.c_int32),
                ("recursive", C.c_int32), ("gapped", C.c_int32), ("add_unaligned", C.c_int32),
                ("extend_lcbs", C.c_int32), ("max_extension_iters", C.c_int32),
                ("min_recursive_gap", C.c_int64), ("max_gapped_len", C.c_int64), ("scoring", Scoring),
                ("max_banded_len", C.c_int64), ("lcb_scoring", C.c_int32), ("weight_scaling", C.c_int32),
                ("conservation_scale_ppm", C.c_int32), ("seed_family", C.c_int32), ("min_scaled_penalty", C.c_int64),
                ("refine_rounds", C.c_int32), ("bp_dist_scale_ppm", C.c_int32), ("bp_dist_min_score", C.c_int64)]


class HmmParams(C.Structure):
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("gap", C.c_int32), ("go_homologous", C.c_int32), ("go_unrelated", C.c_int32)]


class AlignSizes(C.Structure):
    _fields_ = [("n_mums", C.c_int64), ("n_lcb", C.c_int64), ("n_anchor", C.c_int64), ("n_iv", C.c_int64),
                ("n_cols", C.c_int64), ("n_gap_dp", C.c_int64), ("n_dp_cells", C.c_int64)]


class StageTimes(C.Structure):
    _fields_ = [("seed_ms", C.c_double), ("chain_ms", C.c_double), ("recurse_ms", C.c_double),
                ("dp_ms", C.c_double), ("assemble_ms", C.c_double), ("total_ms", C.c_double), ("tree_ms", C.c_double)]


_lib = None


def load():
    """dlopen the product library; raises OSError with a clear message when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("libmauve_hip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(make -C mauvealigner_amd/csrc); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    L.mauve_last_error.restype = C.c_char_p
    L.mauve_last_error.argtypes = [C.c_void_p]
    L.mauve_get_seed.restype = C.c_uint64
    L.mauve_get_seed.argtypes = [C.c_int, C.c_int]
    L.mauve_seed_length.argtypes = [C.c_uint64]
    L.mauve_seed_weight.argtypes = [C.c_uint64]
    L.mauve_default_seed_weight.argtypes = [C.c_int64]
    L.mauve_packed_words.restype = C.c_size_t
    L.mauve_packed_words.argtypes = [C.c_int64]
    L.mauve_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.mauve_ctx_destroy.argtypes = [C.c_void_p]
    L.mauve_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    L.mauve_host_free.argtypes = [C.c_void_p]
    L.mauve_host_free.restype = None
    _lib = L
    return L


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# int (*mauve_allgather_fn)(void *user, const void *send, int64_t send_bytes, const void **recv, int64_t *recv_bytes)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64))


class _PinnedBlock:
    """one mauve_host_alloc allocation, released when the last numpy view of it goes away"""

    def __init__(self, nbytes):
        self.L = load()
        p = C.c_void_p()
        rc = self.L.mauve_host_alloc(C.c_size_t(max(int(nbytes), 64)), C.byref(p))
        if rc or not p.value:
            raise MemoryError("mauve_host_alloc(%d) failed (%d)" % (nbytes, rc))
        self.p = p
        self.buf = (C.c_uint8 * max(int(nbytes), 64)).from_address(p.value)

    def __del__(self):
        try:
            if self.p:
                self.L.mauve_host_free(self.p)
                self.p = None
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """numpy array in page-locked host memory (mauve_host_alloc): what a caller hands to set_genomes / fetch for one-DMA transfers"""
    dt = np.dtype(dtype)
    shape = (shape,) if np.isscalar(shape) else tuple(shape)
    n = int(np.prod(shape)) if shape else 1
    blk = _PinnedBlock(n * dt.itemsize)
    a = np.frombuffer(blk.buf, dtype=dt, count=n).reshape(shape)
    blk.buf._mauve_block = blk          # the ctypes buffer is the base of every view: it keeps the allocation alive
    return a


class ResultBuffers:
    """Caller-owned, reusable result arrays for Context.align(..., out=...): page-locked, grown on demand, so that a fetch is
    one DMA per bulk array and no allocation.  The arrays returned by a fetch are views that the next fetch overwrites."""

    def __init__(self):
        self._a = {}

    def get(self, name, shape, dtype):
        shape = (shape,) if np.isscalar(shape) else tuple(shape)
        n = int(np.prod(shape)) if shape else 1
        cur = self._a.get(name)
        if cur is None or cur.size < n or cur.dtype != np.dtype(dtype):
            cur = pinned_empty(max(n + n // 8, 16), dtype)
            self._a[name] = cur
        return cur[:n].reshape(shape)


def get_seed(weight, rank=0):
    return int(load().mauve_get_seed(weight, rank))


def seed_length(p):
    return int(load().mauve_seed_length(C.c_uint64(p)))


def seed_weight(p):
    return int(load().mauve_seed_weight(C.c_uint64(p)))


def default_seed_weight(avg_len):
    return int(load().mauve_default_seed_weight(C.c_int64(int(avg_len))))


def default_params(**kw):
    p = Params()
    load().mauve_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def default_progressive_params(**kw):
    """the progressiveMauve call site's option set (SP scoring, weight scaling 0.5 / 0.5, refinement on)"""
    p = Params()
    load().mauve_default_progressive_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def merge_matches(len_a, st_a, len_b, st_b):
    """mauve_merge_matches (host entry, no context): a, then the matches of b no match of a contains; canonical order."""
    la = np.ascontiguousarray(len_a, np.int64); sa = np.ascontiguousarray(st_a, np.int64)
    lb = np.ascontiguousarray(len_b, np.int64); sb = np.ascontiguousarray(st_b, np.int64)
    N = sa.shape[1] if sa.ndim == 2 and sa.shape[0] else sb.shape[1]
    cap = C.c_int64(len(la) + len(lb))
    lo = np.zeros(max(cap.value, 1), np.int64); so = np.zeros((max(cap.value, 1), N), np.int64)
    rc = load().mauve_merge_matches(N, C.c_int64(len(la)), _p(la, C.c_int64), _p(sa, C.c_int64), C.c_int64(len(lb)), _p(lb, C.c_int64),
                                    _p(sb, C.c_int64), C.byref(cap), _p(lo, C.c_int64), _p(so, C.c_int64))
    if rc:
        raise RuntimeError("mauve_merge_matches failed (%d)" % rc)
    return lo[:cap.value].copy(), so[:cap.value].copy()


def default_scoring():
    s = Scoring()
    load().mauve_default_scoring(C.byref(s))
    return s


def pack_codes(codes):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    L = load()
    words = np.zeros(L.mauve_packed_words(len(codes)), dtype=np.uint64)
    L.mauve_pack_codes(_p(codes, C.c_uint8), C.c_int64(len(codes)), _p(words, C.c_uint64))
    return words


def pack_ascii(text):
    b = text if isinstance(text, (bytes, bytearray)) else text.encode()
    L = load()
    words = np.zeros(L.mauve_packed_words(len(b)), dtype=np.uint64)
    L.mauve_pack_ascii(C.c_char_p(bytes(b)), C.c_int64(len(b)), _p(words, C.c_uint64))
    return words


def eliminate_overlaps(length, start):
    length = np.array(length, dtype=np.int64, copy=True)
    start = np.array(start, dtype=np.int64, copy=True)
    n = C.c_int64(len(length))
    N = start.shape[1]
    rc = load().mauve_eliminate_overlaps(N, C.byref(n), _p(length, C.c_int64), _p(start, C.c_int64))
    if rc:
        raise RuntimeError("mauve_eliminate_overlaps: %d" % rc)
    return length[:n.value].copy(), start[:n.value].copy()


def lcb_chain(length, start, min_weight, collinear=False):
    length = np.ascontiguousarray(length, dtype=np.int64)
    start = np.ascontiguousarray(start, dtype=np.int64)
    n, N = len(length), start.shape[1]
    ml = np.zeros(max(n, 1), np.int64)
    le = np.zeros((max(n, 1), N), np.int64)
    re = np.zeros((max(n, 1), N), np.int64)
    wt = np.zeros(max(n, 1), np.int64)
    la = np.zeros((max(n, 1), N), np.int64)
    ra = np.zeros((max(n, 1), N), np.int64)
    k = C.c_int64()
    rc = load().mauve_lcb_chain(N, C.c_int64(n), _p(length, C.c_int64), _p(start, C.c_int64), C.c_int64(min_weight),
                                int(collinear), _p(ml, C.c_int64), C.byref(k), _p(le, C.c_int64), _p(re, C.c_int64),
                                _p(wt, C.c_int64), _p(la, C.c_int64), _p(ra, C.c_int64))
    if rc:
        raise RuntimeError("mauve_lcb_chain: %d" % rc)
    K = k.value
    return {"n_lcb": K, "match_lcb": ml[:n].copy(), "left_end": le[:K].copy(), "right_end": re[:K].copy(),
            "weight": wt[:K].copy(), "left_adj": la[:K].copy(), "right_adj": ra[:K].copy()}


class Context:
    """One mauve_ctx = one GPU + one stream.  Fails loudly when the extension or the GPU is absent."""

    def __init__(self, device=0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.mauve_ctx_create(device, C.byref(h))
        if rc:
            raise RuntimeError("mauve_ctx_create failed (%d): %s" % (rc, self.L.mauve_last_error(None).decode()))
        self.h = h
        self.nseq = 0
        self._keep = None

    def close(self):
        if self.h:
            self.L.mauve_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.L.mauve_last_error(self.h).decode()))

    def device_name(self):
        buf = C.create_string_buffer(256)
        self.L.mauve_device_name(self.h, buf, 256)
        return buf.value.decode()

    def set_shard(self, rank, world, allgather=None):
        """mauve_set_shard: `allgather(payload: bytes-like) -> list of per-rank bytes-like` is the caller's collective (e.g.
        parallel.make_allgather(dist)); None / world <= 1 switches the sharding off.  The callback object is kept alive here."""
        if allgather is None or world <= 1:
            self._shard_cb = None
            self._chk(self.L.mauve_set_shard(self.h, 0, 1, C.cast(None, ALLGATHER_FN), None), "mauve_set_shard")
            return
        state = {"keep": None}

        def cb(user, send, nbytes, recv_out, sizes_out):
            try:
                mine = (C.c_char * nbytes).from_address(send) if nbytes else b""
                parts = allgather(bytes(mine))
                if len(parts) != world:
                    return 2
                blob = b"".join(bytes(x) for x in parts)
                buf = C.create_string_buffer(blob, max(len(blob), 1))
                state["keep"] = buf                          # valid until the next call
                recv_out[0] = C.cast(buf, C.c_void_p).value
                for r in range(world):
                    sizes_out[r] = len(parts[r])
                return 0
            except Exception as ex:                          # never let an exception cross the C boundary
                state["error"] = repr(ex)
                return 1
        self._shard_cb = ALLGATHER_FN(cb)
        self._shard_state = state
        self._chk(self.L.mauve_set_shard(self.h, int(rank), int(world), self._shard_cb, None), "mauve_set_shard")

    def synchronize(self):
        self._chk(self.L.mauve_synchronize(self.h), "mauve_synchronize")

    def set_genomes(self, codes_list, contig_starts=None, invalid=None):
        """codes_list: list of uint8 arrays of 0..3 codes (packed here with mauve_pack_codes).  contig_starts: per genome
        the 0-based starts of its contigs (first 0); invalid: per genome a boolean array, True = ambiguous base."""
        packed = [pack_codes(c) for c in codes_list]
        n = len(packed)
        arr = (C.POINTER(C.c_uint64) * n)(*[_p(w, C.c_uint64) for w in packed])
        lens = (C.c_int64 * n)(*[len(c) for c in codes_list])
        if contig_starts is None and invalid is None:
            self._chk(self.L.mauve_set_genomes(self.h, n, arr, lens), "mauve_set_genomes")
        else:
            cs = contig_starts or [[0]] * n
            ncont = (C.c_int64 * n)(*[len(x) for x in cs])
            flat = np.array([int(v) for x in cs for v in x] + [0], dtype=np.int64)
            bits = []
            for g in range(n):
                L = len(codes_list[g])
                w = np.zeros(L // 64 + 1, np.uint64)
                if invalid is not None and invalid[g] is not None:
                    b = np.zeros((L // 64 + 1) * 64, np.uint8)
                    b[:L] = np.asarray(invalid[g], dtype=np.uint8)
                    w = np.packbits(b, bitorder="little").view(np.uint64).copy()
                bits.append(w)
            inv = (C.POINTER(C.c_uint64) * n)(*[_p(w, C.c_uint64) for w in bits])
            self._chk(self.L.mauve_set_genomes_contigs(self.h, n, arr, lens, ncont, _p(flat, C.c_int64), inv), "mauve_set_genomes_contigs")
        self.nseq = n
        self.lens = [len(c) for c in codes_list]

    def set_genomes_packed(self, packed, lens):
        """genomes already in the boundary's 2-bit packing (pack_codes): the upload alone (arrays from pinned_empty go up
        in one DMA each, without a staging copy)"""
        n = len(packed)
        arr = (C.POINTER(C.c_uint64) * n)(*[_p(w, C.c_uint64) for w in packed])
        self._chk(self.L.mauve_set_genomes(self.h, n, arr, (C.c_int64 * n)(*[int(x) for x in lens])), "mauve_set_genomes")
        self.nseq = n
        self.lens = [int(x) for x in lens]

    def seed_mums(self, pattern, mode=MODE_MEM, mask=0, extend=True, fetch=True):
        n = C.c_int64()
        self._chk(self.L.mauve_seed_mums(self.h, C.c_uint64(pattern), mode, C.c_uint64(mask), int(bool(extend)),
                                         C.byref(n)), "mauve_seed_mums")
        if not fetch:
            return n.value
        ln = np.zeros(n.value, np.int64)
        st = np.zeros((n.value, self.nseq), np.int64)
        self._chk(self.L.mauve_get_matches(self.h, _p(ln, C.c_int64), _p(st, C.c_int64)), "mauve_get_matches")
        return ln, st

    def extend_hits(self, pattern, mask, pos, strand, extend=True):
        """seed hits enumerated on the host (the MatchFinder callback path) -> extended matches in canonical order"""
        mask = np.ascontiguousarray(mask, dtype=np.uint32)
        pos = np.ascontiguousarray(pos, dtype=np.int64)
        strand = np.ascontiguousarray(strand, dtype=np.uint8)
        n = C.c_int64()
        self._chk(self.L.mauve_extend_hits(self.h, C.c_uint64(pattern), C.c_int64(len(mask)), _p(mask, C.c_uint32), _p(pos, C.c_int64),
                                           _p(strand, C.c_uint8), int(bool(extend)), C.byref(n)), "mauve_extend_hits")
        ln = np.zeros(n.value, np.int64)
        st = np.zeros((n.value, self.nseq), np.int64)
        self._chk(self.L.mauve_get_matches(self.h, _p(ln, C.c_int64), _p(st, C.c_int64)), "mauve_get_matches")
        return ln, st

    def align_matches(self, params, length, start, fetch=True, names=None, want_xmfa=False):
        """Aligner::align on the caller's match list (no seed pass)"""
        p = params or default_params()
        length = np.ascontiguousarray(length, dtype=np.int64)
        start = np.ascontiguousarray(start, dtype=np.int64)
        sz = AlignSizes()
        self._chk(self.L.mauve_align_matches(self.h, C.byref(p), C.c_int64(len(length)), _p(length, C.c_int64), _p(start, C.c_int64),
                                             C.byref(sz)), "mauve_align_matches")
        if not fetch:
            return {k: int(getattr(sz, k)) for k, _ in AlignSizes._fields_}
        return self._fetch(sz, names, want_xmfa)

    def align_lcbs(self, params, length, start, lcb, fetch=True, names=None, want_xmfa=False):
        """Aligner::align resumed from the caller's LCBs (mauve_align_lcbs): recursion + gapped alignment only"""
        p = params or default_params()
        length = np.ascontiguousarray(length, dtype=np.int64)
        start = np.ascontiguousarray(start, dtype=np.int64)
        lcb = np.ascontiguousarray(lcb, dtype=np.int64)
        sz = AlignSizes()
        self._chk(self.L.mauve_align_lcbs(self.h, C.byref(p), C.c_int64(len(length)), _p(length, C.c_int64), _p(start, C.c_int64),
                                          _p(lcb, C.c_int64), C.byref(sz)), "mauve_align_lcbs")
        if not fetch:
            return {k: int(getattr(sz, k)) for k, _ in AlignSizes._fields_}
        return self._fetch(sz, names, want_xmfa)

    def align_dp_anchors(self, n_dp):
        N = self.nseq
        left = np.zeros((max(n_dp, 1), 1 + N), np.int64)
        right = np.zeros((max(n_dp, 1), 1 + N), np.int64)
        self._chk(self.L.mauve_align_dp_anchors(self.h, _p(left, C.c_int64), _p(right, C.c_int64)), "mauve_align_dp_anchors")
        return left[:n_dp], right[:n_dp]

    def sorted_mer_list(self, seq, pattern):
        n = max(0, self.lens[seq] - seed_length(pattern) + 1)
        mer = np.zeros(n, np.uint64)
        pos = np.zeros(n, np.int64)
        k = C.c_int64()
        self._chk(self.L.mauve_sorted_mer_list(self.h, seq, C.c_uint64(pattern), _p(mer, C.c_uint64),
                                               _p(pos, C.c_int64), C.byref(k)), "mauve_sorted_mer_list")
        return mer[:k.value], pos[:k.value]

    def seed_match_enumerate(self, seq, pattern, min_multi=2, max_multi=1000, direct_only=False):
        n, ns = C.c_int64(), C.c_int64()
        args = (self.h, seq, C.c_uint64(pattern), C.c_int64(min_multi), C.c_int64(max_multi), int(direct_only))
        self._chk(self.L.mauve_seed_match_enumerate(*args, C.byref(n), C.byref(ns), None, None, None), "seed_match_enumerate")
        mult = np.zeros(n.value, np.int64)
        off = np.zeros(n.value + 1, np.int64)
        st = np.zeros(ns.value, np.int64)
        self._chk(self.L.mauve_seed_match_enumerate(*args, C.byref(n), C.byref(ns), _p(mult, C.c_int64), _p(off, C.c_int64),
                                                    _p(st, C.c_int64)), "seed_match_enumerate")
        return mult, off, st

    def dp_batch(self, intervals, scoring=None, band_from=None):
        """intervals: list of lists of code arrays (nseq each).  -> (cols list, scores).  band_from: intervals whose
        longest sequence is above it run the banded DP (mauve_dp_batch_banded)."""
        sc = scoring or default_scoring()
        n_iv = len(intervals)
        nseq = len(intervals[0]) if n_iv else 1
        if any(len(iv) != nseq for iv in intervals):
            raise ValueError("dp_batch: every interval must hold the same number of sequences")
        flat, off = [], [0]
        for iv in intervals:
            for s in iv:
                flat.append(np.asarray(s, dtype=np.uint8))
                off.append(off[-1] + len(s))
        codes = np.concatenate(flat) if flat and off[-1] else np.zeros(1, np.uint8)
        off = np.array(off, dtype=np.int64)
        cols = np.zeros(max(int(off[-1]), 1), np.uint32)
        col_off = np.zeros(n_iv + 1, np.int64)
        score = np.zeros(max(n_iv, 1), np.int64)
        if band_from is None:
            self._chk(self.L.mauve_dp_batch(self.h, nseq, C.c_int64(n_iv), _p(codes, C.c_uint8), _p(off, C.c_int64),
                                            C.byref(sc), _p(cols, C.c_uint32), _p(col_off, C.c_int64), _p(score, C.c_int64)),
                      "mauve_dp_batch")
        else:
            self._chk(self.L.mauve_dp_batch_banded(self.h, nseq, C.c_int64(n_iv), _p(codes, C.c_uint8), _p(off, C.c_int64),
                                                   C.byref(sc), C.c_int64(int(band_from)), _p(cols, C.c_uint32),
                                                   _p(col_off, C.c_int64), _p(score, C.c_int64)), "mauve_dp_batch_banded")
        return [cols[col_off[i]:col_off[i + 1]].copy() for i in range(n_iv)], score[:n_iv].copy()

    def match_sp_scores(self, length, start, scoring=None):
        """extant sum-of-pairs scores of ungapped matches on the resident genomes (mauve_match_sp_scores)"""
        sc = scoring or default_scoring()
        length = np.ascontiguousarray(length, dtype=np.int64)
        start = np.ascontiguousarray(start, dtype=np.int64).reshape(len(length), self.nseq)
        out = np.zeros(max(len(length), 1), np.int64)
        self._chk(self.L.mauve_match_sp_scores(self.h, C.c_int64(len(length)), _p(length, C.c_int64), _p(start, C.c_int64),
                                               C.byref(sc), _p(out, C.c_int64)), "mauve_match_sp_scores")
        return out[:len(length)].copy()

    def align(self, params=None, fetch=True, names=None, want_xmfa=False, out=None, compact=False):
        """out: a ResultBuffers the result arrays are fetched into (reused from call to call; views)"""
        p = params or default_params()
        sz = AlignSizes()
        self._chk(self.L.mauve_align(self.h, C.byref(p), C.byref(sz)), "mauve_align")
        res = {k: int(getattr(sz, k)) for k, _ in AlignSizes._fields_}
        if not fetch:
            return res
        return self._fetch(sz, names, want_xmfa, out, compact)

    def _fetch_compact(self, sz, bufs=None):
        """mauve_align_fetch_compact: columns in 1 / 2 / 4 bytes by the genome count, match and anchor tables as int32"""
        N = self.nseq
        out = {k: int(getattr(sz, k)) for k, _ in AlignSizes._fields_}
        cb = 1 if N <= 8 else (2 if N <= 16 else 4)
        cdt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[cb]
        shapes = {
            "mum_length": (sz.n_mums, np.int32), "mum_start": ((sz.n_mums, N), np.int32),
            "lcb_left": ((sz.n_lcb, N), np.int64), "lcb_right": ((sz.n_lcb, N), np.int64), "lcb_weight": (sz.n_lcb, np.int64),
            "anchor_length": (sz.n_anchor, np.int32), "anchor_start": ((sz.n_anchor, N), np.int32), "anchor_lcb": (sz.n_anchor, np.int32),
            "left": ((sz.n_iv, N), np.int64), "right": ((sz.n_iv, N), np.int64), "reverse": ((sz.n_iv, N), np.int8),
            "col_off": (sz.n_iv + 1, np.int64), "cols": (sz.n_cols, cdt), "dp_score": (sz.n_iv, np.int64),
        }
        a = {k: (np.zeros(sh, dt) if bufs is None else bufs.get("c_" + k, sh, dt)) for k, (sh, dt) in shapes.items()}
        self.L.mauve_align_fetch_compact.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 14
        vp = lambda x: x.ctypes.data_as(C.c_void_p)
        self._chk(self.L.mauve_align_fetch_compact(self.h, cb, vp(a["mum_length"]), vp(a["mum_start"]), vp(a["lcb_left"]), vp(a["lcb_right"]), vp(a["lcb_weight"]),
                                                   vp(a["anchor_length"]), vp(a["anchor_start"]), vp(a["anchor_lcb"]), vp(a["left"]), vp(a["right"]), vp(a["reverse"]),
                                                   vp(a["col_off"]), vp(a["cols"]), vp(a["dp_score"])), "mauve_align_fetch_compact")
        out.update(a)
        out["col_bytes"] = cb
        return out

    def _fetch(self, sz, names=None, want_xmfa=False, bufs=None, compact=False):
        if compact and not want_xmfa:
            return self._fetch_compact(sz, bufs)
        N = self.nseq
        out = {k: int(getattr(sz, k)) for k, _ in AlignSizes._fields_}
        shapes = {
            "mum_length": (sz.n_mums, np.int64), "mum_start": ((sz.n_mums, N), np.int64),
            "lcb_left": ((sz.n_lcb, N), np.int64), "lcb_right": ((sz.n_lcb, N), np.int64),
            "lcb_weight": (sz.n_lcb, np.int64),
            "anchor_length": (sz.n_anchor, np.int64), "anchor_start": ((sz.n_anchor, N), np.int64),
            "anchor_lcb": (sz.n_anchor, np.int64),
            "left": ((sz.n_iv, N), np.int64), "right": ((sz.n_iv, N), np.int64),
            "reverse": ((sz.n_iv, N), np.int8), "col_off": (sz.n_iv + 1, np.int64),
            "cols": (sz.n_cols, np.uint32), "dp_score": (sz.n_iv, np.int64),
        }
        if bufs is None:
            a = {k: np.zeros(sh, dt) for k, (sh, dt) in shapes.items()}
        else:
            a = {k: bufs.get(k, sh, dt) for k, (sh, dt) in shapes.items()}
        self._chk(self.L.mauve_align_fetch(
            self.h, _p(a["mum_length"], C.c_int64), _p(a["mum_start"], C.c_int64), _p(a["lcb_left"], C.c_int64),
            _p(a["lcb_right"], C.c_int64), _p(a["lcb_weight"], C.c_int64), _p(a["anchor_length"], C.c_int64),
            _p(a["anchor_start"], C.c_int64), _p(a["anchor_lcb"], C.c_int64), _p(a["left"], C.c_int64),
            _p(a["right"], C.c_int64), _p(a["reverse"], C.c_int8), _p(a["col_off"], C.c_int64),
            _p(a["cols"], C.c_uint32), _p(a["dp_score"], C.c_int64)), "mauve_align_fetch")
        out.update(a)
        if want_xmfa:
            nm = names or ["seq%d" % i for i in range(N)]
            narr = (C.c_char_p * N)(*[s.encode() for s in nm])
            ln = C.c_int64()
            self._chk(self.L.mauve_write_xmfa(self.h, narr, None, C.byref(ln)), "mauve_write_xmfa")
            buf = C.create_string_buffer(ln.value)
            self._chk(self.L.mauve_write_xmfa(self.h, narr, buf, C.byref(ln)), "mauve_write_xmfa")
            out["xmfa"] = buf.value.decode()
        return out

    # ---- sharded form: begin / dp on a subset / finish (include/mauve_hip.h) ----
    def align_begin(self, params=None):
        p = params or default_params()
        n_dp, n_codes = C.c_int64(), C.c_int64()
        self._chk(self.L.mauve_align_begin(self.h, C.byref(p), C.byref(n_dp), C.byref(n_codes)), "mauve_align_begin")
        cost = np.zeros(max(n_dp.value, 1), np.int64)
        cap = np.zeros(max(n_dp.value, 1), np.int64)
        self._chk(self.L.mauve_align_dp_cost(self.h, _p(cost, C.c_int64), _p(cap, C.c_int64)), "mauve_align_dp_cost")
        return n_dp.value, cost[:n_dp.value], cap[:n_dp.value]

    def align_dp(self, idx, cap):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        n = len(idx)
        cols = np.zeros(max(int(cap[idx].sum()) if n else 0, 1), np.uint32)
        col_off = np.zeros(n + 1, np.int64)
        score = np.zeros(max(n, 1), np.int64)
        cells = C.c_int64()
        self._chk(self.L.mauve_align_dp(self.h, _p(idx, C.c_int64), C.c_int64(n), _p(cols, C.c_uint32), _p(col_off, C.c_int64),
                                        _p(score, C.c_int64), C.byref(cells)), "mauve_align_dp")
        return [cols[col_off[i]:col_off[i + 1]] for i in range(n)], score[:n], cells.value

    def align_finish(self, cols_list, scores, cells, fetch=True, names=None, want_xmfa=False, out=None):
        n = len(cols_list)
        col_off = np.zeros(n + 1, np.int64)
        for i, c in enumerate(cols_list):
            col_off[i + 1] = col_off[i] + len(c)
        flat = np.concatenate(cols_list).astype(np.uint32) if n and col_off[n] else np.zeros(1, np.uint32)
        scores = np.ascontiguousarray(scores, dtype=np.int64) if n else np.zeros(1, np.int64)
        sz = AlignSizes()
        self._chk(self.L.mauve_align_finish(self.h, _p(flat, C.c_uint32), _p(col_off, C.c_int64), _p(scores, C.c_int64),
                                            C.c_int64(cells), C.byref(sz)), "mauve_align_finish")
        if not fetch:
            return {k: int(getattr(sz, k)) for k, _ in AlignSizes._fields_}
        return self._fetch(sz, names, want_xmfa, out)

    def guide_tree(self, pattern):
        N = self.nseq
        dist = np.zeros((N, N), np.int64)
        left = np.zeros(2 * N - 1, np.int32)
        right = np.zeros(2 * N - 1, np.int32)
        self._chk(self.L.mauve_guide_tree(self.h, C.c_uint64(pattern), _p(dist, C.c_int64), _p(left, C.c_int32),
                                          _p(right, C.c_int32)), "mauve_guide_tree")
        return dist, left, right

    def breakpoint_counts(self, pattern, min_len):
        """mauve_breakpoint_counts: broken adjacencies between the pairwise matches (length >= min_len) of every genome pair."""
        N = self.nseq
        bp = np.zeros((N, N), np.int64)
        self.L.mauve_breakpoint_counts.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.POINTER(C.c_int64)]
        self._chk(self.L.mauve_breakpoint_counts(self.h, C.c_uint64(pattern), C.c_int64(min_len), _p(bp, C.c_int64)), "mauve_breakpoint_counts")
        return bp

    def progressive_align(self, params=None, fetch=True, names=None, want_xmfa=False, tree=None, out=None, compact=False):
        """tree=(left, right): align along the caller's guide tree (mauve_progressive_align_tree).  out: ResultBuffers."""
        p = params or default_params()
        N = self.nseq
        sz = AlignSizes()
        dist = np.zeros((N, N), np.int64)
        if tree is None:
            left = np.zeros(2 * N - 1, np.int32)
            right = np.zeros(2 * N - 1, np.int32)
            self._chk(self.L.mauve_progressive_align(self.h, C.byref(p), C.byref(sz), _p(left, C.c_int32), _p(right, C.c_int32),
                                                     _p(dist, C.c_int64)), "mauve_progressive_align")
        else:
            left = np.ascontiguousarray(tree[0], np.int32)
            right = np.ascontiguousarray(tree[1], np.int32)
            if len(left) != 2 * N - 1 or len(right) != 2 * N - 1:
                raise ValueError("guide tree must have 2*nseq-1 nodes")
            self._chk(self.L.mauve_progressive_align_tree(self.h, C.byref(p), C.byref(sz), _p(left, C.c_int32),
                                                          _p(right, C.c_int32)), "mauve_progressive_align_tree")
        res = self._fetch(sz, names, want_xmfa, out, compact) if fetch else {k: int(getattr(sz, k)) for k, _ in AlignSizes._fields_}
        res["tree"] = (left, right)
        res["dist"] = dist
        return res

    def _backbone_fetch(self, n_seg, n_isl):
        N = self._bb_nseq
        out = {"seg_iv": np.zeros(n_seg, np.int64), "seg_col": np.zeros(n_seg, np.int64), "seg_len": np.zeros(n_seg, np.int64),
               "seg_mask": np.zeros(n_seg, np.uint32), "seg_left": np.zeros((n_seg, N), np.int64),
               "seg_right": np.zeros((n_seg, N), np.int64), "islands": np.zeros((n_isl, 8), np.int64)}
        self._chk(self.L.mauve_backbone_fetch(self.h, _p(out["seg_iv"], C.c_int64), _p(out["seg_col"], C.c_int64), _p(out["seg_len"], C.c_int64),
                                              _p(out["seg_mask"], C.c_uint32), _p(out["seg_left"], C.c_int64), _p(out["seg_right"], C.c_int64),
                                              _p(out["islands"], C.c_int64)), "mauve_backbone_fetch")
        return out

    def hmm_params(self, identity=0.7, pgh=1e-5, pgu=1e-9, **kw):
        """mauve_hmm_params_from: the call site's knobs (progressiveMauve.cpp:319-322) as integer scores; kw overrides fields."""
        h = HmmParams()
        self.L.mauve_hmm_params_from.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(HmmParams)]
        self.L.mauve_hmm_params_from.restype = None
        self.L.mauve_hmm_params_from(identity, pgh, pgu, C.byref(h))
        for k, v in kw.items():
            setattr(h, k, v)
        return h

    def apply_homology(self, hmm=None, fetch=True, names=None, want_xmfa=False):
        """mauve_apply_homology: un-align what the pair HMM classes as unrelated in the alignment this context holds.
        -> the rewritten result (or its sizes) with 'n_moved'."""
        h = hmm or self.hmm_params()
        sz = AlignSizes()
        moved = C.c_int64(0)
        self.L.mauve_apply_homology.argtypes = [C.c_void_p, C.POINTER(HmmParams), C.POINTER(AlignSizes), C.POINTER(C.c_int64)]
        self._chk(self.L.mauve_apply_homology(self.h, C.byref(h), C.byref(sz), C.byref(moved)), "mauve_apply_homology")
        res = self._fetch(sz, names, want_xmfa) if fetch else {k: int(getattr(sz, k)) for k, _ in AlignSizes._fields_}
        res["n_moved"] = int(moved.value)
        return res

    def backbone(self, island_gap=20, nseq=None):
        """Backbone segments and islands of the alignment this context holds (mauve_backbone)."""
        ns, ni = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.mauve_backbone(self.h, C.c_int64(island_gap), C.byref(ns), C.byref(ni)), "mauve_backbone")
        self._bb_nseq = nseq or self.nseq
        return self._backbone_fetch(ns.value, ni.value)

    def backbone_alignment(self, left, right, reverse, col_off, cols, island_gap=20):
        """... of the caller's alignment (mauve_backbone_alignment): left/right/reverse [n_iv, nseq], col_off [n_iv+1], cols."""
        left = np.ascontiguousarray(left, np.int64)
        right = np.ascontiguousarray(right, np.int64)
        reverse = np.ascontiguousarray(reverse, np.int8)
        col_off = np.ascontiguousarray(col_off, np.int64)
        cols = np.ascontiguousarray(cols, np.uint32)
        n_iv, N = left.shape
        ns, ni = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.mauve_backbone_alignment(self.h, N, C.c_int64(n_iv), _p(left, C.c_int64), _p(right, C.c_int64), _p(reverse, C.c_int8),
                                                  _p(col_off, C.c_int64), _p(cols if len(cols) else np.zeros(1, np.uint32), C.c_uint32),
                                                  C.c_int64(island_gap), C.byref(ns), C.byref(ni)), "mauve_backbone_alignment")
        self._bb_nseq = N
        return self._backbone_fetch(ns.value, ni.value)

    def apply_homology_alignment(self, left, right, reverse, col_off, cols, hmm=None):
        """mauve_apply_homology_alignment: the homology pass on the caller's alignment (its genomes are the context's) -> (col_off, cols, n_moved)"""
        left = np.ascontiguousarray(left, np.int64)
        right = np.ascontiguousarray(right, np.int64)
        reverse = np.ascontiguousarray(reverse, np.int8)
        col_off = np.ascontiguousarray(col_off, np.int64)
        cols = np.ascontiguousarray(cols, np.uint32)
        n_iv, N = left.shape
        h = hmm or self.hmm_params()
        residues = int(np.sum(np.where(left != 0, right - left + 1, 0)))
        ncols = np.zeros(max(residues, len(cols)) + 1, np.uint32)
        noff = np.zeros(n_iv + 1, np.int64)
        moved = C.c_int64(0)
        self.L.mauve_apply_homology_alignment.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int8), C.POINTER(C.c_int64),
                                                          C.POINTER(C.c_uint32), C.POINTER(HmmParams), C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.POINTER(C.c_int64)]
        self._chk(self.L.mauve_apply_homology_alignment(self.h, N, n_iv, _p(left, C.c_int64), _p(right, C.c_int64), _p(reverse, C.c_int8), _p(col_off, C.c_int64),
                                                        _p(cols if len(cols) else np.zeros(1, np.uint32), C.c_uint32), C.byref(h), _p(noff, C.c_int64), _p(ncols, C.c_uint32),
                                                        C.byref(moved)), "mauve_apply_homology_alignment")
        return noff, ncols[:noff[-1]].copy(), int(moved.value)

    def stage_times(self):
        t = StageTimes()
        self.L.mauve_last_stage_times(self.h, C.byref(t))
        return {k: getattr(t, k) for k, _ in StageTimes._fields_}

    def profile(self, on=True):
        self.L.mauve_profile_enable(self.h, int(on))

    def profile_reset(self):
        self.L.mauve_profile_reset(self.h)

    def profile_get(self):
        out = {}
        for i, nm in enumerate(KERNEL_NAMES):
            ms, ln, un = C.c_double(), C.c_int64(), C.c_int64()
            self.L.mauve_profile_get(self.h, i, C.byref(ms), C.byref(ln), C.byref(un))
            out[nm] = {"ms": ms.value, "launches": ln.value, "units": un.value}
        return out
