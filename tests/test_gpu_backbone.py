"""GPU parity tests of the backbone / island stage (mauve_backbone, mauve_backbone_alignment) against the
CPU oracle's column-by-column restatement.  Integer work: every segment, genome set, column range and coordinate must
match exactly."""
import numpy as np
import pytest

from mauvealigner_amd import synth
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu
KEYS = ("seg_iv", "seg_col", "seg_len", "seg_mask", "seg_left", "seg_right", "islands")


@pytest.fixture(scope="module")
def ctx():
    from mauvealigner_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _random_alignment(rng, N, n_iv, length, long_runs=False):
    """intervals over random genome subsets: stretches where a random subset is present (the rest gapped), with
    per-column dropouts; lengths chosen so that runs cross the 64-column words and the 4096-column chunks"""
    left, right, rev, col_off, cols = [], [], [], [0], []
    for _ in range(n_iv):
        k = int(rng.integers(1, N + 1))
        S = np.sort(rng.choice(N, k, replace=False))
        full = 0
        for g in S:
            full |= 1 << int(g)
        out = []
        L = int(rng.integers(1, length))
        while len(out) < L:
            kind = rng.random()
            if kind < 0.45:                                   # everybody present, a few dropouts
                n = int(rng.integers(1, 200))
                m = np.full(n, full, np.uint32)
                drop = rng.random(n) < 0.05
                m[drop] &= ~np.uint32(1 << int(rng.choice(S)))
            elif kind < 0.9:                                  # a subset present: short or long run
                n = int(rng.integers(1, 9000 if long_runs and rng.random() < 0.2 else 70))
                sub = 0
                for g in S:
                    if rng.random() < 0.5:
                        sub |= 1 << int(g)
                m = np.full(n, sub, np.uint32)
            else:                                             # alternating single genomes
                n = int(rng.integers(1, 60))
                m = np.array([1 << int(rng.choice(S)) for _ in range(n)], np.uint32)
            out.append(m[m != 0])
        m = np.concatenate(out) if out else np.zeros(0, np.uint32)
        for g in S:                                           # every genome of the interval has a residue
            if not np.any(m >> np.uint32(g) & 1):
                m = np.concatenate([m, np.array([1 << int(g)], np.uint32)])
        lo = np.zeros(N, np.int64)
        hi = np.zeros(N, np.int64)
        rv = np.zeros(N, np.int8)
        for g in S:
            cnt = int(np.count_nonzero(m >> np.uint32(g) & 1))
            lo[g] = int(rng.integers(1, 1 << 20))
            hi[g] = lo[g] + cnt - 1
            rv[g] = int(rng.random() < 0.4)
        left.append(lo)
        right.append(hi)
        rev.append(rv)
        cols.append(m)
        col_off.append(col_off[-1] + len(m))
    return np.array(left), np.array(right), np.array(rev), np.array(col_off, np.int64), np.concatenate(cols)


def _same(r, e):
    for k in KEYS:
        assert r[k].shape == e[k].shape, (k, r[k].shape, e[k].shape)
        assert np.array_equal(r[k], e[k]), k


def test_backbone_hand_case(ctx):
    """genome 1 lacks five columns the other two share: with island_gap 3 they are an island of 0 and of 2 against 1,
    the backbone is {0,1,2} / {0,2} / {0,1,2}; with island_gap 5 the gap is small and one segment covers everything"""
    cols = np.array([7] * 10 + [5] * 5 + [7] * 5, np.uint32)
    left, right, rev = np.array([[1, 101, 201]]), np.array([[20, 115, 220]]), np.array([[0, 0, 1]], np.int8)
    r = ctx.backbone_alignment(left, right, rev, [0, 20], cols, island_gap=3)
    assert r["seg_mask"].tolist() == [7, 5, 7] and r["seg_col"].tolist() == [0, 10, 15] and r["seg_len"].tolist() == [10, 5, 5]
    assert r["seg_left"].tolist() == [[1, 101, -211], [11, 0, -206], [16, 111, -201]]
    assert r["seg_right"].tolist() == [[10, 110, -220], [15, 0, -210], [20, 115, -205]]
    assert r["islands"].tolist() == [[0, 0, 1, 0, 10, 14, 11, 15], [0, 1, 2, 2, 10, 14, -206, -210]]
    _same(r, O.backbone(left, right, rev, [0, 20], cols, island_gap=3))
    r = ctx.backbone_alignment(left, right, rev, [0, 20], cols, island_gap=5)
    assert r["seg_mask"].tolist() == [7] and r["seg_len"].tolist() == [20] and len(r["islands"]) == 0
    # nothing to do: single-genome intervals only, and no interval at all
    r = ctx.backbone_alignment(np.array([[1, 0]]), np.array([[5, 0]]), np.zeros((1, 2), np.int8), [0, 5], np.full(5, 1, np.uint32))
    assert len(r["seg_iv"]) == 0 and len(r["islands"]) == 0
    r = ctx.backbone_alignment(np.zeros((0, 3), np.int64), np.zeros((0, 3), np.int64), np.zeros((0, 3), np.int8), [0], np.zeros(0, np.uint32))
    assert len(r["seg_iv"]) == 0


@pytest.mark.parametrize("seed", range(6))
def test_backbone_random_alignments(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    N = [2, 3, 5, 8, 17, 32][seed]
    left, right, rev, col_off, cols = _random_alignment(rng, N, n_iv=12, length=[3000, 9000, 20000][seed % 3], long_runs=seed >= 2)
    for gap in (0, 3, 20, 500):
        r = ctx.backbone_alignment(left, right, rev, col_off, cols, island_gap=gap)
        e = O.backbone(left, right, rev, col_off, cols, island_gap=gap)
        _same(r, e)
    assert len(e["seg_iv"]) > 0


def test_backbone_record_list_regrows(ctx):
    """more open regions than the record list holds at first: the launch is repeated with room"""
    rng = np.random.default_rng(5)
    n = 400000
    cols = np.where(np.arange(n) % 3 == 0, 3, np.where(np.arange(n) % 3 == 1, 1, 3)).astype(np.uint32)     # an island of one every third column
    left, right, rev = np.array([[1, 1]]), np.array([[n, int(np.count_nonzero(cols & 2))]]), np.zeros((1, 2), np.int8)
    r = ctx.backbone_alignment(left, right, rev, [0, n], cols, island_gap=0)
    e = O.backbone(left, right, rev, [0, n], cols, island_gap=0)
    assert len(e["islands"]) > 100000
    _same(r, e)
    del rng


def _partition_checks(r, a, N):
    """every genome's segments lie inside their interval and do not overlap one another"""
    for g in range(N):
        rows = np.flatnonzero(r["seg_mask"] >> np.uint32(g) & 1)
        lo = np.abs(r["seg_left"][rows, g])
        hi = np.abs(r["seg_right"][rows, g])
        lo, hi = np.minimum(lo, hi), np.maximum(lo, hi)
        iv = r["seg_iv"][rows]
        assert np.all(lo >= a["left"][iv, g]) and np.all(hi <= a["right"][iv, g])
        o = np.argsort(lo)
        assert np.all(lo[o][1:] > hi[o][:-1])


def test_backbone_of_the_resident_alignment(ctx):
    """mauve_backbone on the columns the alignment left in HBM (device tail) and on a host-assembled result
    (progressive path) = the oracle on the fetched alignment"""
    from mauvealigner_amd import _lib
    import os
    gs = synth.make_config("C3", scale=0.03)
    N = len(gs)
    ctx.set_genomes(gs)
    old = os.environ.get("MAUVE_CANON_DEVICE_MIN")
    os.environ["MAUVE_CANON_DEVICE_MIN"] = "1"               # small lists take the device tail as well
    try:
        sz = ctx.align(_lib.default_params(), fetch=False)
        r = ctx.backbone(island_gap=20)                      # before any fetch: the columns are only in HBM
        a = ctx.align(_lib.default_params())
    finally:
        if old is None:
            del os.environ["MAUVE_CANON_DEVICE_MIN"]
        else:
            os.environ["MAUVE_CANON_DEVICE_MIN"] = old
    assert sz["n_iv"] == a["n_iv"]
    e = O.backbone(a["left"], a["right"], a["reverse"], a["col_off"], a["cols"], island_gap=20)
    _same(r, e)
    assert len(r["seg_iv"]) >= a["n_lcb"] and np.any(r["seg_mask"] == (1 << N) - 1)
    _partition_checks(r, a, N)
    _same(ctx.backbone(island_gap=20), e)                    # after the fetch as well
    # the progressive path assembles on the host: the columns are uploaded
    gs4 = synth.make_config("C4", scale=0.02)
    ctx.set_genomes(gs4)
    p = ctx.progressive_align(_lib.default_params())
    r = ctx.backbone(island_gap=20)
    e = O.backbone(p["left"], p["right"], p["reverse"], p["col_off"], p["cols"], island_gap=20)
    _same(r, e)
    assert len(set(r["seg_mask"].tolist())) > 1              # clade-specific backbone next to the all-genome one
    _partition_checks(r, p, len(gs4))
    # an alignment that kept nothing (no LCB reaches the weight, leftovers not listed) has an empty backbone
    ctx.set_genomes(gs)
    z = ctx.align(_lib.default_params(lcb_weight=10 ** 9, add_unaligned=0))
    assert z["n_iv"] == 0
    r = ctx.backbone()
    assert len(r["seg_iv"]) == 0 and len(r["islands"]) == 0
    # error behaviour: nothing aligned yet in a fresh context, gap out of range
    c2 = _lib.Context(0)
    with pytest.raises(RuntimeError):
        c2.backbone()
    c2.close()
    with pytest.raises(RuntimeError):
        ctx.backbone(island_gap=-1)


def test_inconsistent_columns_are_refused(ctx):
    """A caller's alignment whose columns do not hold the residues its ends announce (the homology pass would turn the surplus into
    base addresses outside the interval -- below 0 on the reverse strand) is refused with MAUVE_ERR_ARG by both _alignment entry
    points, before any kernel reads a genome through it; the consistent array goes through and equals the oracle's."""
    gs = synth.star_genomes(3, 20_000, 0.05, 7)
    ctx.set_genomes(gs)
    e = O.align(gs, O.default_params())["aln"]
    left, right, rev, off, cols = e["left"], e["right"], e["reverse"], e["col_off"], e["cols"]
    nz = np.count_nonzero(left, axis=1)
    assert (nz >= 2).any() and (nz == 1).any()           # the cases below need an interval of each kind
    noff, ncols, moved = ctx.apply_homology_alignment(left, right, rev, off, cols)
    eoff, ecols, emoved = O.homology_apply(gs, left, right, rev, off, cols)
    assert moved == emoved and np.array_equal(noff, eoff) and np.array_equal(ncols, ecols)
    multi = int(np.argmax(np.count_nonzero(left, axis=1) >= 2))
    # (1) one more residue of genome 0 than its ends allow; (2) a residue of a genome the interval does not have; (3) a bit above nseq;
    # (4) a reverse-strand interval that starts at base 1 with surplus residues: the unguarded address would be negative
    bad = []
    c1 = cols.copy(); k = off[multi] + int(np.argmax((c1[off[multi]:off[multi + 1]] & 1) == 0)); c1[k] |= 1; bad.append((left, right, rev, c1))
    single = int(np.argmax(np.count_nonzero(left, axis=1) == 1))
    other = int(np.argmax(left[single] == 0))
    c2 = cols.copy(); c2[off[single]] |= np.uint32(1 << other); bad.append((left, right, rev, c2))
    c3 = cols.copy(); c3[off[multi]] |= np.uint32(1 << 7); bad.append((left, right, rev, c3))
    l4, r4, v4 = left.copy(), right.copy(), rev.copy()
    n0 = int(right[multi, 0] - left[multi, 0] + 1)
    l4[multi, 0], r4[multi, 0], v4[multi, 0] = 1, n0 - 5, 1
    bad.append((l4, r4, v4, cols))
    for l, r, v, c in bad:
        assert not (np.array_equal(l, left) and np.array_equal(c, cols))
        with pytest.raises(RuntimeError, match=r"\(-1\)"):
            ctx.apply_homology_alignment(l, r, v, off, c)
        with pytest.raises(RuntimeError, match=r"\(-1\)"):
            ctx.backbone_alignment(l, r, v, off, c)
    _same(ctx.backbone_alignment(left, right, rev, off, cols), O.backbone(left, right, rev, off, cols, island_gap=20))
