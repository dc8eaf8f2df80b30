"""Deterministic synthetic genomes for the BASELINE.json configs.

Ancestor: iid uniform ACGT.  Star phylogeny: every genome is an independently mutated descendant at
per-lineage rate d/2 (pairwise divergence ~ d); events are 90 % substitutions (uniform over the three
alternatives) and 10 % indels (half insertions, half deletions; geometric length, mean 3, cap 50).
Inversions reverse-complement non-overlapping segments with log-uniform lengths.

PRNG: xoshiro256** seeded through splitmix64, base seed 0x4D41555645 + config id.  A stream is named by a tuple of keys
(config id, then e.g. the genome index): seed = BASE_SEED; for every key: seed = splitmix64(seed + key); the four state words are the next four
splitmix64 outputs.  Draws (class Xoshiro, the handful of numpy-Generator methods this module uses, in a fixed order): random() = (x >> 11) * 2^-53;
integers(lo, hi) = lo + (((x >> 32) * (hi - lo)) >> 32) for ranges below 2^32; uniform base codes = x >> 62; geometric(p): round by round one draw
per still-unfinished sample in index order, success when x < floor(p * 2^64); uniform(a, b) = a + (b - a) * random().  The raw stream comes from
csrc/synth_rng.c (libmauve_synth.so, plain C; compiled on the spot with gcc if it is missing), so a C or C++ caller reproduces the workloads from the
same twenty lines.  Bases are returned as uint8 codes A,C,G,T = 0..3.  (Rounds 1-3 used numpy's PCG64 for the same distributions.)
"""
import ctypes
import os
import subprocess

import numpy as np

BASE_SEED = 0x4D41555645
_M64 = (1 << 64) - 1


def _splitmix64(x):
    """-> (next state, output)"""
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return x, z ^ (z >> 31)


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(here, "libmauve_synth.so")
        if not os.path.exists(so):
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(here, "csrc", "synth_rng.c")])
        L = ctypes.CDLL(so)
        L.xo_fill.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        L.xo_fill_bases.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        _LIB = L
    return _LIB


class Xoshiro:
    """xoshiro256** with the numpy-Generator methods synth.py uses (module docstring: how each draw is made)"""

    def __init__(self, *keys):
        seed = BASE_SEED
        for k in keys:
            _, seed = _splitmix64((seed + int(k)) & _M64)
        st, w = seed, []
        for _ in range(4):
            st, o = _splitmix64(st)
            w.append(o)
        self.s = np.array(w, dtype=np.uint64)

    def raw(self, n):
        out = np.empty(int(n), dtype=np.uint64)
        if n:
            _lib().xo_fill(self.s.ctypes.data, out.ctypes.data, int(n))
        return out

    def random(self, size=None):
        x = (self.raw(1 if size is None else size) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return float(x[0]) if size is None else x

    def uniform(self, low, high):
        return low + (high - low) * self.random()

    def integers(self, low, high=None, size=None, dtype=np.int64):
        if high is None:
            low, high = 0, low
        rng_ = int(high) - int(low)
        if rng_ <= 0 or rng_ > (1 << 32):
            raise ValueError("Xoshiro.integers: range must be in 1 .. 2^32")
        n = 1 if size is None else int(size)
        if rng_ == 4 and int(low) == 0 and np.dtype(dtype) == np.uint8:          # base codes: the top two bits
            out = np.empty(n, dtype=np.uint8)
            if n:
                _lib().xo_fill_bases(self.s.ctypes.data, out.ctypes.data, n)
            return out[0] if size is None else out
        x = ((self.raw(n) >> np.uint64(32)) * np.uint64(rng_)) >> np.uint64(32)
        out = (x.astype(np.int64) + int(low)).astype(dtype)
        return out[0] if size is None else out

    def geometric(self, p, size=None):
        n = 1 if size is None else int(size)
        thr = min(int(p * 18446744073709551616.0), _M64)
        out = np.zeros(n, dtype=np.int64)
        todo = np.arange(n)
        k = 0
        while len(todo):
            k += 1
            x = self.raw(len(todo))
            hit = x < np.uint64(thr)
            out[todo[hit]] = k
            todo = todo[~hit]
            if k >= 4096:
                out[todo] = k
                break
        return int(out[0]) if size is None else out


def _rng(*keys):
    return Xoshiro(*keys)


def random_genome(length, rng):
    return rng.integers(0, 4, size=int(length), dtype=np.uint8)


def mutate(codes, rate, rng, indel_frac=0.10, mean_indel=3.0, cap=50, origin=None):
    """Point-mutate `codes` at per-base event rate `rate`.  With `origin` (signed 1-based ancestor coordinate per
    base, 0 = no ancestor) the same draws are made and (codes, origin) of the descendant is returned: the truth
    used by accuracy.py."""
    L = len(codes)
    if rate <= 0 or L == 0:
        return codes.copy() if origin is None else (codes.copy(), origin.copy())
    ev = rng.random(L) < rate
    pos = np.flatnonzero(ev)
    kind = rng.random(len(pos))
    out = codes.copy()
    sub = pos[kind >= indel_frac]
    out[sub] = (out[sub] + rng.integers(1, 4, size=len(sub), dtype=np.uint8)) & 3
    ind = pos[kind < indel_frac]
    if len(ind) == 0:
        return out if origin is None else (out, origin.copy())
    lens = np.minimum(rng.geometric(1.0 / mean_indel, size=len(ind)), cap)
    is_ins = rng.random(len(ind)) < 0.5
    pieces, opieces = [], []
    cur = 0
    for p, ln, ins in zip(ind.tolist(), lens.tolist(), is_ins.tolist()):
        if p < cur:
            continue
        pieces.append(out[cur:p])
        if origin is not None:
            opieces.append(origin[cur:p])
        if ins:
            pieces.append(rng.integers(0, 4, size=ln, dtype=np.uint8))
            if origin is not None:
                opieces.append(np.zeros(ln, dtype=np.int64))
            cur = p
        else:
            cur = min(L, p + ln)
    pieces.append(out[cur:])
    if origin is None:
        return np.concatenate(pieces)
    opieces.append(origin[cur:])
    return np.concatenate(pieces), np.concatenate(opieces)


def revcomp(codes):
    return (3 - codes[::-1]).astype(np.uint8)


def invert_segments(codes, n_inv, rng, min_len, max_len, origin=None):
    """Reverse-complement n_inv non-overlapping segments (log-uniform lengths).  `origin` (if given) is updated in
    place on a copy: an inverted base keeps its ancestor coordinate with the sign flipped."""
    L = len(codes)
    out = codes.copy()
    if origin is not None:
        origin[:] = origin          # caller passes a private copy
    if n_inv <= 0:
        return out, []
    slot = L // n_inv
    segs = []
    for i in range(n_inv):
        ln = int(np.exp(rng.uniform(np.log(min_len), np.log(max_len))))
        ln = max(1, min(ln, slot - 2))
        st = i * slot + int(rng.integers(0, max(1, slot - ln)))
        out[st:st + ln] = revcomp(out[st:st + ln])
        if origin is not None:
            origin[st:st + ln] = -origin[st:st + ln][::-1]
        segs.append((st, ln))
    return out, segs


def star_genomes(n, length, divergence, seed_key, inversions=0, inv_min=5000, inv_max=500000, track=False):
    """n descendants of one random ancestor; `inversions` total, spread over genomes 1..n-1.  track=True also
    returns, per genome, the signed ancestor coordinate of every base (the truth alignment)."""
    anc = random_genome(length, _rng(seed_key, 0))
    genomes, origins = [], []
    per = [0] * n
    for i in range(inversions):
        per[1 + i % (n - 1)] += 1
    for g in range(n):
        rng = _rng(seed_key, 1 + g)
        x = anc
        o = np.arange(1, length + 1, dtype=np.int64) if track else None
        if per[g]:
            x, _ = invert_segments(x, per[g], rng, min(inv_min, max(50, length // 400)),
                                   min(inv_max, max(100, length // (2 * max(1, per[g])))), origin=o)
        if track:
            x, o = mutate(x, divergence / 2.0, rng, origin=o)
            genomes.append(x)
            origins.append(o)
        else:
            genomes.append(mutate(x, divergence / 2.0, rng))
    return (genomes, origins) if track else genomes


def tree_genomes(n_leaves, length, branch_div, seed_key, inv_per_branch=2, insert_per_branch=1, insert_len=(2000, 12000)):
    """Balanced binary tree (BASELINE config C4): every branch mutates at `branch_div`, adds `inv_per_branch`
    inversions and `insert_per_branch` clade-specific insertions, so sister genomes share sequence the rest lacks."""
    def grow(seq, depth, key):
        rng = _rng(seed_key, key)
        x = mutate(seq, branch_div, rng)
        if inv_per_branch:
            x, _ = invert_segments(x, inv_per_branch, rng, max(200, length // 200), max(400, length // 20))
        for _ in range(insert_per_branch):
            ln = int(rng.integers(min(insert_len[0], max(50, length // 40)), min(insert_len[1], max(100, length // 10))))
            at = int(rng.integers(0, len(x)))
            x = np.concatenate([x[:at], rng.integers(0, 4, size=ln, dtype=np.uint8), x[at:]])
        if depth == 0:
            return [x]
        return grow(x, depth - 1, key * 2) + grow(x, depth - 1, key * 2 + 1)
    depth = int(np.ceil(np.log2(n_leaves)))
    anc = random_genome(length, _rng(seed_key, 0))
    leaves = grow(anc, depth - 1, 2) + grow(anc, depth - 1, 3)
    return leaves[:n_leaves]


def make_config(name, scale=1.0):
    """BASELINE.json configs.  `scale` < 1 shrinks genome lengths (parity tests, bounded CPU baselines)."""
    name = name.upper()
    if name == "C1":
        return star_genomes(2, int(200_000 * scale), 0.03, 1)
    if name == "C2":
        return star_genomes(3, int(5_000_000 * scale), 0.03, 2)
    if name == "C3":
        L = int(5_000_000 * scale)
        return star_genomes(5, L, 0.03, 3, inversions=max(1, int(round(50 * min(1.0, scale * 4)))) if scale < 0.25 else 50)
    if name == "C4":
        return tree_genomes(8, int(2_000_000 * scale), 0.01, 4)
    if name == "C5":
        L = int(100_000_000 * scale)
        gs = star_genomes(2, L, 0.03, 5)
        out = []
        for g, x in enumerate(gs):
            rng = _rng(5, 100 + g)
            n_ins = max(1, int(2000 * scale))
            pos = np.sort(rng.integers(0, len(x), size=n_ins))
            pieces, cur = [], 0
            for p in pos.tolist():
                pieces.append(x[cur:p])
                pieces.append(rng.integers(0, 4, size=int(rng.integers(1000, 10000)), dtype=np.uint8))
                cur = p
            pieces.append(x[cur:])
            y = np.concatenate(pieces)
            n_hd = max(1, int(200 * scale))
            for _ in range(n_hd):
                ln = int(rng.integers(2000, 20000))
                st = int(rng.integers(0, max(1, len(y) - ln)))
                y[st:st + ln] = mutate(y[st:st + ln], 0.25, rng, indel_frac=0.0)[:ln]
            out.append(y)
        return out
    raise ValueError("unknown config " + name)


def to_ascii(codes):
    return np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes()
