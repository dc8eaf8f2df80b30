"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (see oracle/mauve_oracle.h).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product
package.  PARITY UNPINNED: the oracle is this repository's CPU restatement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODE_MEM = 0
MODE_UNIQUE = 1
MODE_PAIRWISE = 2
CODING_SEED = 3
SOLID_SEED = 0x7FFFFFFF
MAX_SEQ = 32


class Scoring(C.Structure):
    _fields_ = [("gap_open", C.c_int32), ("gap_extend", C.c_int32), ("matrix", (C.c_int32 * 4) * 4)]


class Matches(C.Structure):
    _fields_ = [("n", C.c_int64), ("nseq", C.c_int32), ("length", C.POINTER(C.c_int64)),
                ("start", C.POINTER(C.c_int64))]


class Lcbs(C.Structure):
    _fields_ = [("n_lcb", C.c_int64), ("nseq", C.c_int32), ("match_lcb", C.POINTER(C.c_int64)),
                ("left_end", C.POINTER(C.c_int64)), ("right_end", C.POINTER(C.c_int64)),
                ("weight", C.POINTER(C.c_int64)), ("left_adj", C.POINTER(C.c_int64)),
                ("right_adj", C.POINTER(C.c_int64))]


class Alignment(C.Structure):
    _fields_ = [("n_iv", C.c_int64), ("nseq", C.c_int32), ("left", C.POINTER(C.c_int64)),
                ("right", C.POINTER(C.c_int64)), ("reverse", C.POINTER(C.c_int8)),
                ("col_off", C.POINTER(C.c_int64)), ("cols", C.POINTER(C.c_uint32)),
                ("dp_score", C.POINTER(C.c_int64)), ("n_anchor", C.c_int64),
                ("anchor_length", C.POINTER(C.c_int64)), ("anchor_start", C.POINTER(C.c_int64)),
                ("anchor_lcb", C.POINTER(C.c_int64)), ("n_gap_dp", C.c_int64), ("n_dp_cells", C.c_int64)]


class Params(C.Structure):
    _fields_ = [("seed_pattern", C.c_uint64), ("seed_weight", C.c_int32), ("seed_rank", C.c_int32),
                ("mode", C.c_int32), ("lcb_weight", C.c_int64), ("collinear", C.c_int32),
                ("recursive", C.c_int32), ("gapped", C.c_int32), ("add_unaligned", C.c_int32),
                ("extend_lcbs", C.c_int32), ("max_extension_iters", C.c_int32),
                ("min_recursive_gap", C.c_int64), ("max_gapped_len", C.c_int64), ("scoring", Scoring),
                ("max_banded_len", C.c_int64), ("lcb_scoring", C.c_int32), ("weight_scaling", C.c_int32),
                ("conservation_scale_ppm", C.c_int32), ("seed_family", C.c_int32), ("min_scaled_penalty", C.c_int64),
                ("refine_rounds", C.c_int32), ("bp_dist_scale_ppm", C.c_int32), ("bp_dist_min_score", C.c_int64)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("mauve_oracle.c", "mauve_oracle.h", "seed_table.inc")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_get_seed.restype = C.c_uint64
        L.orc_get_seed.argtypes = [C.c_int, C.c_int]
        L.orc_seed_length.argtypes = [C.c_uint64]
        L.orc_seed_weight.argtypes = [C.c_uint64]
        L.orc_default_seed_weight.argtypes = [C.c_int64]
        L.orc_mers.restype = C.c_int64
        L.orc_sorted_mer_list.restype = C.c_int64
        L.orc_align_interval.restype = C.c_int64
        L.orc_profile_dp.restype = C.c_int64
        L.orc_profile_dp_band.restype = C.c_int64
        L.orc_align_interval_band.restype = C.c_int64
        L.orc_band_cells.restype = C.c_int64
        L.orc_band_cells.argtypes = [C.c_int64, C.c_int64, C.c_int]
        L.orc_write_xmfa.restype = C.c_void_p
        L.orc_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _seq_args(codes):
    codes = [np.ascontiguousarray(c, dtype=np.uint8) for c in codes]
    n = len(codes)
    arr = (C.POINTER(C.c_uint8) * n)(*[_u8p(c) for c in codes])
    lens = (C.c_int64 * n)(*[len(c) for c in codes])
    return codes, arr, lens


def get_seed(weight, rank=0):
    return int(lib().orc_get_seed(weight, rank))


def seed_length(p):
    return int(lib().orc_seed_length(C.c_uint64(p)))


def seed_weight(p):
    return int(lib().orc_seed_weight(C.c_uint64(p)))


def default_seed_weight(avg_len):
    return int(lib().orc_default_seed_weight(C.c_int64(int(avg_len))))


def default_scoring():
    s = Scoring()
    lib().orc_default_scoring(C.byref(s))
    return s


def default_params(**kw):
    p = Params()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def default_progressive_params(**kw):
    p = Params()
    lib().orc_default_progressive_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def encode(ascii_bytes):
    a = np.frombuffer(ascii_bytes if isinstance(ascii_bytes, (bytes, bytearray)) else ascii_bytes.encode(), dtype=np.uint8)
    out = np.empty(len(a), dtype=np.uint8)
    lib().orc_encode(a.ctypes.data_as(C.c_char_p), C.c_int64(len(a)), _u8p(out))
    return out


def pack2bit(codes):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    words = np.zeros((len(codes) + 15) // 16, dtype=np.uint32)
    lib().orc_pack2bit(_u8p(codes), C.c_int64(len(codes)), words.ctypes.data_as(C.POINTER(C.c_uint32)))
    return words


def mers(codes, pattern):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    np_ = max(0, len(codes) - seed_length(pattern) + 1)
    canon = np.zeros(np_, dtype=np.uint64)
    strand = np.zeros(np_, dtype=np.uint8)
    lib().orc_mers(_u8p(codes), C.c_int64(len(codes)), C.c_uint64(pattern),
                   canon.ctypes.data_as(C.POINTER(C.c_uint64)), _u8p(strand))
    return canon, strand


def sorted_mer_list(codes, pattern):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    np_ = max(0, len(codes) - seed_length(pattern) + 1)
    mer = np.zeros(np_, dtype=np.uint64)
    pos = np.zeros(np_, dtype=np.int64)
    lib().orc_sorted_mer_list(_u8p(codes), C.c_int64(len(codes)), C.c_uint64(pattern),
                              mer.ctypes.data_as(C.POINTER(C.c_uint64)), pos.ctypes.data_as(C.POINTER(C.c_int64)))
    return mer, pos


def _matches_to_np(m):
    n, N = int(m.n), int(m.nseq)
    if n == 0 or not m.length:
        return np.zeros(0, np.int64), np.zeros((0, N), np.int64)
    length = np.ctypeslib.as_array(m.length, shape=(max(n, 1),))[:n].copy()
    start = np.ctypeslib.as_array(m.start, shape=(max(n, 1) * N,))[:n * N].copy().reshape(n, N)
    return length, start


def _np_to_matches(length, start):
    length = np.ascontiguousarray(length, dtype=np.int64)
    start = np.ascontiguousarray(start, dtype=np.int64)
    m = Matches()
    m.n = len(length)
    m.nseq = start.shape[1] if start.ndim == 2 else 0
    m.length = length.ctypes.data_as(C.POINTER(C.c_int64))
    m.start = start.ctypes.data_as(C.POINTER(C.c_int64))
    return m, (length, start)


def find_matches(codes, pattern, mode=MODE_MEM, mask=0, extend=True):
    """-> (length[n], start[n, N]) in canonical order."""
    codes, arr, lens = _seq_args(codes)
    m = Matches()
    rc = lib().orc_find_matches(len(codes), arr, lens, C.c_uint64(pattern), mode, C.c_uint64(mask),
                                int(bool(extend)), C.byref(m))
    if rc:
        raise RuntimeError("orc_find_matches failed: %d" % rc)
    out = _matches_to_np(m)
    lib().orc_free_matches(C.byref(m))
    return out


def find_matches_masked(codes, pattern, valid, mode=MODE_MEM, mask=0, extend=True):
    """valid: per genome a list of (lo, hi) 1-based inclusive intervals (sorted, disjoint)."""
    codes, arr, lens = _seq_args(codes)
    off = np.zeros(len(codes) + 1, np.int64)
    lo, hi = [], []
    for g, ivs in enumerate(valid):
        off[g + 1] = off[g] + len(ivs)
        lo += [a for a, _ in ivs]
        hi += [b for _, b in ivs]
    lo = np.array(lo + [0], np.int64)
    hi = np.array(hi + [0], np.int64)
    m = Matches()
    rc = lib().orc_find_matches_masked(len(codes), arr, lens, C.c_uint64(pattern), mode, C.c_uint64(mask), int(bool(extend)),
                                       off.ctypes.data_as(C.POINTER(C.c_int64)), lo.ctypes.data_as(C.POINTER(C.c_int64)),
                                       hi.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(m))
    if rc:
        raise RuntimeError("orc_find_matches_masked failed: %d" % rc)
    out = _matches_to_np(m)
    lib().orc_free_matches(C.byref(m))
    return out


def seed_match_enumerate(codes, pattern, min_multi=2, max_multi=1000, direct_only=False):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n = C.c_int64()
    mult = C.POINTER(C.c_int64)()
    off = C.POINTER(C.c_int64)()
    st = C.POINTER(C.c_int64)()
    lib().orc_seed_match_enumerate(_u8p(codes), C.c_int64(len(codes)), C.c_uint64(pattern), C.c_int64(min_multi),
                                   C.c_int64(max_multi), int(direct_only), C.byref(n), C.byref(mult),
                                   C.byref(off), C.byref(st))
    k = n.value
    if not mult:
        return np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros(0, np.int64)
    m = np.ctypeslib.as_array(mult, shape=(max(k, 1),))[:k].copy()
    o = np.ctypeslib.as_array(off, shape=(k + 1,)).copy()
    s = np.ctypeslib.as_array(st, shape=(max(int(o[k]), 1),))[:int(o[k])].copy()
    for p in (mult, off, st):
        lib().orc_free(C.cast(p, C.c_void_p))
    return m, o, s


def multiplicity_filter(length, start, mult):
    m, keep = _np_to_matches(length, start)
    out = Matches()
    lib().orc_multiplicity_filter(C.byref(m), mult, C.byref(out))
    res = _matches_to_np(out)
    lib().orc_free_matches(C.byref(out))
    return res


def eliminate_overlaps(length, start):
    length = np.array(length, dtype=np.int64, copy=True)
    start = np.array(start, dtype=np.int64, copy=True)
    m, keep = _np_to_matches(length, start)
    lib().orc_eliminate_overlaps(C.byref(m))
    n = int(m.n)
    return length[:n].copy(), start[:n].copy()


def _lcbs_to_dict(l, nmatch):
    K, N = int(l.n_lcb), int(l.nseq)

    def arr(p, n):
        return np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].copy() if p else np.zeros(0, np.int64)
    return {
        "n_lcb": K,
        "match_lcb": arr(l.match_lcb, nmatch),
        "left_end": arr(l.left_end, K * N).reshape(K, N),
        "right_end": arr(l.right_end, K * N).reshape(K, N),
        "weight": arr(l.weight, K),
        "left_adj": arr(l.left_adj, K * N).reshape(K, N),
        "right_adj": arr(l.right_adj, K * N).reshape(K, N),
    }


def compute_lcbs(length, start, min_weight, collinear=False):
    m, keep = _np_to_matches(length, start)
    out = Lcbs()
    lib().orc_compute_lcbs(C.byref(m), C.c_int64(min_weight), int(collinear), C.byref(out))
    d = _lcbs_to_dict(out, len(keep[0]))
    lib().orc_free_lcbs(C.byref(out))
    return d


def match_sp_scores(seqs, length, start, scoring=None):
    """extant sum-of-pairs score of every (ungapped) match, DESIGN.md S11"""
    sc = scoring or default_scoring()
    seqs, arr, lens = _seq_args(seqs)
    m, keep = _np_to_matches(length, start)
    out = np.zeros(max(len(keep[0]), 1), np.int64)
    lib().orc_match_sp_scores(len(seqs), arr, C.byref(m), C.byref(sc), out.ctypes.data_as(C.POINTER(C.c_int64)))
    return out[:len(keep[0])].copy()


def profile_dp(cnt, k_rows, seq, scoring=None, banded=False):
    sc = scoring or default_scoring()
    cnt = np.ascontiguousarray(cnt, dtype=np.uint8).reshape(-1, 4)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    m, n = len(cnt), len(seq)
    ops = np.zeros(m + n + 1, dtype=np.uint8)
    score = C.c_int64()
    L = lib().orc_profile_dp_band(C.c_int64(m), _u8p(cnt), k_rows, C.c_int64(n), _u8p(seq), C.byref(sc), _u8p(ops),
                                  C.byref(score), int(bool(banded)))
    return ops[:L].copy(), int(score.value)


def align_interval(seqs, scoring=None, banded=False, want_cells=False):
    sc = scoring or default_scoring()
    seqs, arr, lens = _seq_args(seqs)
    total = sum(len(s) for s in seqs)
    cols = np.zeros(max(total, 1), dtype=np.uint32)
    score = C.c_int64()
    cells = C.c_int64()
    nc = lib().orc_align_interval_band(len(seqs), arr, lens, C.byref(sc), cols.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       C.byref(score), C.byref(cells), int(bool(banded)))
    if want_cells:
        return cols[:nc].copy(), int(score.value), int(cells.value)
    return cols[:nc].copy(), int(score.value)


def sp_score_cols(seqs, cols, scoring=None):
    """sum-of-pairs score of the columns of one interval."""
    sc = scoring or default_scoring()
    seqs, arr, lens = _seq_args(seqs)
    cols = np.ascontiguousarray(cols, np.uint32)
    lib().orc_sp_score_cols.restype = C.c_int64
    return int(lib().orc_sp_score_cols(len(seqs), arr, lens, cols.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int64(len(cols)), C.byref(sc)))


def align_interval_refined(seqs, rounds, scoring=None):
    """progressive alignment + rotated orders, best sum-of-pairs score.  -> (cols, dp score, cells)"""
    sc = scoring or default_scoring()
    seqs, arr, lens = _seq_args(seqs)
    total = sum(len(s) for s in seqs)
    cols = np.zeros(max(total, 1), dtype=np.uint32)
    score, cells = C.c_int64(), C.c_int64()
    lib().orc_align_interval_refined.restype = C.c_int64
    nc = lib().orc_align_interval_refined(len(seqs), arr, lens, C.byref(sc), int(rounds), cols.ctypes.data_as(C.POINTER(C.c_uint32)),
                                          C.byref(score), C.byref(cells))
    return cols[:nc].copy(), int(score.value), int(cells.value)


def breakpoint_counts(codes, pattern, min_len):
    """broken adjacencies between the pairwise matches of every genome pair.  -> [N, N] int64"""
    codes, arr, lens = _seq_args(codes)
    N = len(codes)
    bp = np.zeros((N, N), np.int64)
    rc = lib().orc_breakpoint_counts(N, arr, lens, C.c_uint64(pattern), C.c_int64(min_len), bp.ctypes.data_as(C.POINTER(C.c_int64)))
    if rc:
        raise RuntimeError("orc_breakpoint_counts failed: %d" % rc)
    return bp


def align(codes, params=None, names=None, want_xmfa=False):
    """Whole path.  -> dict(mums=(len,start), lcbs={...}, aln={...}, xmfa=str|None)"""
    p = params or default_params()
    codes, arr, lens = _seq_args(codes)
    N = len(codes)
    mm, lc, al = Matches(), Lcbs(), Alignment()
    rc = lib().orc_align(N, arr, lens, C.byref(p), C.byref(mm), C.byref(lc), C.byref(al))
    if rc:
        raise RuntimeError("orc_align failed: %d" % rc)
    mums = _matches_to_np(mm)
    nway = int((np.count_nonzero(mums[1], axis=1) == N).sum()) if len(mums[0]) else 0
    niv, na = int(al.n_iv), int(al.n_anchor)

    def arr64(ptr, n):
        return np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].copy()
    col_off = np.ctypeslib.as_array(al.col_off, shape=(niv + 1,)).copy()
    ncol = int(col_off[niv])
    aln = {
        "n_iv": niv,
        "left": arr64(al.left, niv * N).reshape(niv, N),
        "right": arr64(al.right, niv * N).reshape(niv, N),
        "reverse": np.ctypeslib.as_array(al.reverse, shape=(max(niv * N, 1),))[:niv * N].copy().reshape(niv, N),
        "col_off": col_off,
        "cols": np.ctypeslib.as_array(al.cols, shape=(max(ncol, 1),))[:ncol].copy(),
        "dp_score": arr64(al.dp_score, niv),
        "anchor_length": arr64(al.anchor_length, na),
        "anchor_start": arr64(al.anchor_start, na * N).reshape(na, N),
        "anchor_lcb": arr64(al.anchor_lcb, na),
        "n_gap_dp": int(al.n_gap_dp),
        "n_dp_cells": int(al.n_dp_cells),
    }
    xmfa = None
    if want_xmfa:
        nm = names or ["seq%d" % i for i in range(N)]
        narr = (C.c_char_p * N)(*[s.encode() for s in nm])
        tl = C.c_int64()
        ptr = lib().orc_write_xmfa(N, arr, lens, narr, C.byref(al), C.byref(tl))
        xmfa = C.string_at(ptr, tl.value).decode()
        lib().orc_free(C.c_void_p(ptr))
    # LCB struct's match_lcb refers to the overlap-eliminated N-way list; report its length via anchors
    lcbs = _lcbs_to_dict(lc, 0)
    lib().orc_free_matches(C.byref(mm))
    lib().orc_free_lcbs(C.byref(lc))
    lib().orc_free_alignment(C.byref(al))
    return {"mums": mums, "n_nway": nway, "lcbs": lcbs, "aln": aln, "xmfa": xmfa}


def guide_tree(codes, pattern):
    codes, arr, lens = _seq_args(codes)
    N = len(codes)
    dist = np.zeros((N, N), np.int64)
    left = np.zeros(2 * N - 1, np.int32)
    right = np.zeros(2 * N - 1, np.int32)
    rc = lib().orc_guide_tree(N, arr, lens, C.c_uint64(pattern), dist.ctypes.data_as(C.POINTER(C.c_int64)),
                              left.ctypes.data_as(C.POINTER(C.c_int32)), right.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc:
        raise RuntimeError("orc_guide_tree failed: %d" % rc)
    return dist, left, right


def check_tree(nseq, left, right):
    left = np.ascontiguousarray(left, np.int32)
    right = np.ascontiguousarray(right, np.int32)
    if len(left) != 2 * nseq - 1 or len(right) != 2 * nseq - 1:
        return False
    return lib().orc_check_tree(nseq, left.ctypes.data_as(C.POINTER(C.c_int32)), right.ctypes.data_as(C.POINTER(C.c_int32))) == 0


def progressive_align(codes, params=None, names=None, want_xmfa=False, tree=None):
    """Guide-tree recursive anchoring.  -> dict(aln={...}, tree=(left,right), dist, xmfa)
    tree=(left, right): align along the caller's guide tree (--input-guide-tree) instead of the UPGMA one."""
    p = params or default_params()
    codes, arr, lens = _seq_args(codes)
    N = len(codes)
    al = Alignment()
    dist = np.zeros((N, N), np.int64)
    if tree is None:
        left = np.zeros(2 * N - 1, np.int32)
        right = np.zeros(2 * N - 1, np.int32)
        rc = lib().orc_progressive_align(N, arr, lens, C.byref(p), left.ctypes.data_as(C.POINTER(C.c_int32)),
                                         right.ctypes.data_as(C.POINTER(C.c_int32)), dist.ctypes.data_as(C.POINTER(C.c_int64)),
                                         C.byref(al))
    else:
        left = np.ascontiguousarray(tree[0], np.int32)
        right = np.ascontiguousarray(tree[1], np.int32)
        if len(left) != 2 * N - 1 or len(right) != 2 * N - 1:
            raise ValueError("guide tree must have 2*nseq-1 nodes")
        rc = lib().orc_progressive_align_tree(N, arr, lens, C.byref(p), left.ctypes.data_as(C.POINTER(C.c_int32)),
                                              right.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(al))
    if rc:
        raise RuntimeError("orc_progressive_align failed: %d" % rc)
    niv = int(al.n_iv)

    def arr64(ptr, n):
        return np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].copy()
    col_off = np.ctypeslib.as_array(al.col_off, shape=(niv + 1,)).copy()
    ncol = int(col_off[niv])
    aln = {
        "n_iv": niv,
        "left": arr64(al.left, niv * N).reshape(niv, N),
        "right": arr64(al.right, niv * N).reshape(niv, N),
        "reverse": np.ctypeslib.as_array(al.reverse, shape=(max(niv * N, 1),))[:niv * N].copy().reshape(niv, N),
        "col_off": col_off,
        "cols": np.ctypeslib.as_array(al.cols, shape=(max(ncol, 1),))[:ncol].copy(),
        "dp_score": arr64(al.dp_score, niv),
        "n_gap_dp": int(al.n_gap_dp),
        "n_dp_cells": int(al.n_dp_cells),
    }
    xmfa = None
    if want_xmfa:
        nm = names or ["seq%d" % i for i in range(N)]
        narr = (C.c_char_p * N)(*[s.encode() for s in nm])
        tl = C.c_int64()
        ptr = lib().orc_write_xmfa(N, arr, lens, narr, C.byref(al), C.byref(tl))
        xmfa = C.string_at(ptr, tl.value).decode()
        lib().orc_free(C.c_void_p(ptr))
    lib().orc_free_alignment(C.byref(al))
    return {"aln": aln, "tree": (left, right), "dist": dist, "xmfa": xmfa}


class Backbone(C.Structure):
    _fields_ = [("nseq", C.c_int32), ("n_seg", C.c_int64), ("seg_iv", C.POINTER(C.c_int64)), ("seg_col", C.POINTER(C.c_int64)),
                ("seg_len", C.POINTER(C.c_int64)), ("seg_mask", C.POINTER(C.c_uint32)), ("seg_left", C.POINTER(C.c_int64)),
                ("seg_right", C.POINTER(C.c_int64)), ("n_isl", C.c_int64), ("isl", C.POINTER(C.c_int64))]


def backbone(left, right, reverse, col_off, cols, island_gap=20):
    """Backbone segments and pairwise islands of an alignment.  left/right/reverse: [n_iv, nseq].
    -> dict(seg_iv, seg_col, seg_len, seg_mask, seg_left, seg_right, islands[n, 8])"""
    left = np.ascontiguousarray(left, np.int64)
    right = np.ascontiguousarray(right, np.int64)
    reverse = np.ascontiguousarray(reverse, np.int8)
    col_off = np.ascontiguousarray(col_off, np.int64)
    cols = np.ascontiguousarray(cols, np.uint32)
    niv, N = left.shape
    if len(cols) == 0:
        cols = np.zeros(1, np.uint32)
    bb = Backbone()
    rc = lib().orc_backbone_detect(N, C.c_int64(niv), left.ctypes.data_as(C.POINTER(C.c_int64)), right.ctypes.data_as(C.POINTER(C.c_int64)),
                                   reverse.ctypes.data_as(C.POINTER(C.c_int8)), col_off.ctypes.data_as(C.POINTER(C.c_int64)),
                                   cols.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int64(island_gap), C.byref(bb))
    if rc:
        raise RuntimeError("orc_backbone_detect failed: %d" % rc)
    n, ni = int(bb.n_seg), int(bb.n_isl)

    def take(ptr, k, dt):
        return np.ctypeslib.as_array(ptr, shape=(k,)).astype(dt).copy() if k else np.zeros(0, dt)
    out = {
        "seg_iv": take(bb.seg_iv, n, np.int64), "seg_col": take(bb.seg_col, n, np.int64), "seg_len": take(bb.seg_len, n, np.int64),
        "seg_mask": take(bb.seg_mask, n, np.uint32), "seg_left": take(bb.seg_left, n * N, np.int64).reshape(n, N),
        "seg_right": take(bb.seg_right, n * N, np.int64).reshape(n, N), "islands": take(bb.isl, ni * 8, np.int64).reshape(ni, 8),
    }
    lib().orc_free_backbone(C.byref(bb))
    return out


class HmmParams(C.Structure):
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("gap", C.c_int32), ("go_homologous", C.c_int32), ("go_unrelated", C.c_int32)]


def hmm_params(identity=0.7, pgh=1e-5, pgu=1e-9, **kw):
    """the call site's knobs (progressiveMauve.cpp:319-322) as the integer scores of DESIGN.md S12b"""
    h = HmmParams()
    lib().orc_hmm_params_from(C.c_double(identity), C.c_double(pgh), C.c_double(pgu), C.byref(h))
    for k, v in kw.items():
        setattr(h, k, v)
    return h


def homology_apply(codes, left, right, reverse, col_off, cols, hmm=None):
    """un-align what the two-state pair HMM classes as unrelated.  -> (col_off, cols, residues moved)"""
    h = hmm or hmm_params()
    codes, arr, lens = _seq_args(codes)
    left = np.ascontiguousarray(left, np.int64); right = np.ascontiguousarray(right, np.int64)
    reverse = np.ascontiguousarray(reverse, np.int8); col_off = np.ascontiguousarray(col_off, np.int64)
    cols = np.ascontiguousarray(cols, np.uint32)
    niv, N = left.shape
    nres = int(np.unpackbits(cols.view(np.uint8)).sum()) if len(cols) else 0
    out = np.zeros(max(nres, 1), np.uint32)
    off = np.zeros(niv + 1, np.int64)
    lib().orc_homology_apply.restype = C.c_int64
    moved = lib().orc_homology_apply(N, arr, C.c_int64(niv), left.ctypes.data_as(C.POINTER(C.c_int64)), right.ctypes.data_as(C.POINTER(C.c_int64)),
                                     reverse.ctypes.data_as(C.POINTER(C.c_int8)), col_off.ctypes.data_as(C.POINTER(C.c_int64)),
                                     (cols if len(cols) else np.zeros(1, np.uint32)).ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(h),
                                     off.ctypes.data_as(C.POINTER(C.c_int64)), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    if moved < 0:
        raise RuntimeError("orc_homology_apply failed: %d" % moved)
    return off, out[:int(off[niv])].copy(), int(moved)


def merge_matches(len_a, st_a, len_b, st_b):
    """Seed-family union: a, then the matches of b that no match of a contains; canonical order."""
    la = np.ascontiguousarray(len_a, np.int64); sa = np.ascontiguousarray(st_a, np.int64)
    lb = np.ascontiguousarray(len_b, np.int64); sb = np.ascontiguousarray(st_b, np.int64)
    N = sa.shape[1] if sa.ndim == 2 and sa.shape[0] else sb.shape[1]
    A = Matches(); B = Matches(); out = Matches()
    keep = [la, sa, lb, sb]
    for M, l, s_ in ((A, la, sa), (B, lb, sb)):
        M.n = len(l); M.nseq = N
        M.length = l.ctypes.data_as(C.POINTER(C.c_int64)); M.start = s_.ctypes.data_as(C.POINTER(C.c_int64))
    rc = lib().orc_merge_matches(N, C.byref(A), C.byref(B), C.byref(out))
    if rc:
        raise RuntimeError("orc_merge_matches failed")
    n = int(out.n)
    ln = np.ctypeslib.as_array(out.length, shape=(max(n, 1),))[:n].copy()
    st = np.ctypeslib.as_array(out.start, shape=(max(n, 1) * N,))[:n * N].copy().reshape(n, N)
    lib().orc_free_matches(C.byref(out))
    del keep
    return ln, st
