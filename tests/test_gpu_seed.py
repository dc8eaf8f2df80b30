"""GPU parity tests of the seed pass (extract -> radix sort -> join -> extension), through the C-ABI,
bit-exact against the CPU oracle and the committed golden fixtures."""
import os

import numpy as np
import pytest

from mauvealigner_amd import synth
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    from mauvealigner_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _same(ctx, gs, pat, mode=0, mask=0, extend=True):
    ctx.set_genomes(gs)
    ln, st = ctx.seed_mums(pat, mode=mode, mask=mask, extend=extend)
    eln, est = O.find_matches(gs, pat, mode=mode, mask=mask, extend=extend)
    assert len(ln) == len(eln)
    assert np.array_equal(ln, eln)
    assert np.array_equal(st, est)
    return ln, st


def test_host_helpers_match_oracle():
    from mauvealigner_amd import _lib
    for w in range(3, 32):
        for r in (0, 1, 2, _lib.CODING_SEED, _lib.SOLID_SEED):
            assert _lib.get_seed(w, r) == O.get_seed(w, r)
    for L in (1, 50, 200, 1000, 20000, 200000, 5000000, 100000000):
        assert _lib.default_seed_weight(L) == O.default_seed_weight(L)
    rng = np.random.default_rng(0)
    c = rng.integers(0, 4, 1000, dtype=np.uint8)
    w64 = _lib.pack_codes(c)
    w32 = O.pack2bit(c)
    assert np.array_equal(w64[:len(w32) // 2 + 1].view(np.uint32)[:len(w32)], w32)
    assert np.array_equal(_lib.pack_ascii(synth.to_ascii(c)), w64)


@pytest.mark.parametrize("name", ["g2x2k", "g3x5k_inv", "g5x3k_unique"])
def test_golden_mums(ctx, name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    gs = [z["genome%d" % g] for g in range(int(z["nseq"]))]
    ctx.set_genomes(gs)
    ln, st = ctx.seed_mums(int(z["pattern"]), mode=int(z["mode"]))
    assert np.array_equal(ln, z["mum_length"]) and np.array_equal(st, z["mum_start"])


@pytest.mark.parametrize("cfg,scale,w", [("C1", 0.25, 13), ("C2", 0.02, 11), ("C3", 0.02, 11), ("C4", 0.02, 9)])
def test_mums_equal_oracle_configs(ctx, cfg, scale, w):
    gs = synth.make_config(cfg, scale=scale)
    for rank in (0, 1):
        pat = O.get_seed(w, rank)
        _same(ctx, gs, pat, mode=0)
    pat = O.get_seed(w, 0)
    _same(ctx, gs, pat, mode=1)
    _same(ctx, gs, pat, mode=0, mask=(1 << len(gs)) - 1)
    _same(ctx, gs, pat, mode=1, extend=False)


def test_mums_64bit_keys(ctx):
    gs = synth.make_config("C1", scale=0.1)
    for w in (17, 21, 31):
        _same(ctx, gs, O.get_seed(w, 0))
    _same(ctx, gs, O.get_seed(19, O.SOLID_SEED))
    _same(ctx, gs, O.get_seed(12, O.CODING_SEED))


def test_edge_cases(ctx):
    pat = O.get_seed(11, 0)
    rng = np.random.default_rng(9)
    g = rng.integers(0, 4, 5000, dtype=np.uint8)
    # identical genomes: one match, maximal length (long wave-cooperative walk)
    ln, st = _same(ctx, [g, g.copy()], pat)
    assert ln.tolist() == [5000] and st.tolist() == [[1, 1]]
    ln, st = _same(ctx, [g, synth.revcomp(g)], pat)
    assert ln.tolist() == [5000] and st.tolist() == [[1, -1]]
    # genomes shorter than the seed, empty genome, ragged lengths
    _same(ctx, [g[:5], g[:7]], pat)
    _same(ctx, [g, np.zeros(0, np.uint8), g[100:900].copy()], pat)
    _same(ctx, [g[:15], g[:15].copy()], pat)          # exactly one window
    # repeats: MEM kills the seed, UNIQUE drops only the repeated genome
    a = rng.integers(0, 4, 300, dtype=np.uint8)
    rep = np.concatenate([a, rng.integers(0, 4, 50, dtype=np.uint8), a])
    _same(ctx, [a, a.copy(), rep], pat, mode=0)
    _same(ctx, [a, a.copy(), rep], pat, mode=1)
    # low-complexity sequence: very long runs of one mer
    poly = np.zeros(3000, np.uint8)
    _same(ctx, [poly, poly.copy()], pat, mode=1)
    _same(ctx, [np.concatenate([poly, g]), np.concatenate([g, poly])], pat, mode=0)
    # tile-boundary sizes of the radix sort (4096 keys per workgroup)
    for L in (4096 + 14, 8192 + 14, 8193 + 14):
        x = rng.integers(0, 4, L, dtype=np.uint8)
        _same(ctx, [x, synth.mutate(x, 0.02, rng)], pat)


def test_oversize_buckets_fall_back(ctx):
    """Genome sets large enough for the partial sort + LDS hash join (join_hash), with repeat families and
    low-complexity blocks that blow single buckets far beyond the hash table: those slices must come back through
    the full sort + serial join with the same result (MEM and UNIQUE rules, with and without the N-way mask)."""
    rng = np.random.default_rng(77)
    anc = rng.integers(0, 4, 60000, dtype=np.uint8)
    unit = rng.integers(0, 4, 37, dtype=np.uint8)
    gs = []
    for g in range(3):
        x = synth.mutate(anc, 0.02, rng)
        x = np.concatenate([x[:20000], np.zeros(9000, np.uint8), x[20000:40000], np.tile(unit, 250),
                            x[40000:], np.tile(np.array([0, 1], np.uint8), 3000 + 500 * g)])
        gs.append(x)
    for w in (11, 15):
        pat = O.get_seed(w, 0)
        _same(ctx, gs, pat, mode=0)
        _same(ctx, gs, pat, mode=1)
        _same(ctx, gs, pat, mode=0, mask=7)
        _same(ctx, gs, pat, mode=1, mask=3)
    # 64-bit keys, and one bucket holding nearly everything (every range oversize: the whole list goes back)
    _same(ctx, gs, O.get_seed(19, 0), mode=1)
    poly = [np.zeros(40000, np.uint8), np.zeros(30000, np.uint8)]
    _same(ctx, poly, O.get_seed(11, 0), mode=1)


def test_many_genomes(ctx):
    rng = np.random.default_rng(4)
    anc = rng.integers(0, 4, 3000, dtype=np.uint8)
    gs = [synth.mutate(anc, 0.02, rng) for _ in range(17)]
    pat = O.get_seed(9, 0)
    _same(ctx, gs, pat, mode=1)
    _same(ctx, gs, pat, mode=0, mask=(1 << 17) - 1)
    # more than eight components per hit WITH components on the other strand (the extension builds the base streams of eight components at a
    # time, the reversed ones from the complemented far end), matches that run into the genomes' ends included
    gs = synth.star_genomes(12, 5000, 0.03, 9, inversions=14, inv_min=300, inv_max=1500)
    gs[3] = gs[3][::-1].copy() ^ 3                       # one genome as its reverse complement altogether
    for w in (9, 11):
        _same(ctx, gs, O.get_seed(w, 0), mode=1)
        _same(ctx, gs, O.get_seed(w, 0), mode=0, mask=(1 << 12) - 1)


def test_sorted_mer_list(ctx):
    gs = synth.make_config("C1", scale=0.05)
    ctx.set_genomes(gs)
    for w in (11, 19):
        pat = O.get_seed(w, 0)
        for s in (0, 1):
            mer, pos = ctx.sorted_mer_list(s, pat)
            emer, epos = O.sorted_mer_list(gs[s], pat)
            assert np.array_equal(mer, emer) and np.array_equal(pos, epos)


def test_seed_match_enumerator(ctx):
    rng = np.random.default_rng(11)
    unit = rng.integers(0, 4, 400, dtype=np.uint8)
    g = np.concatenate([rng.integers(0, 4, 1000, dtype=np.uint8), unit, rng.integers(0, 4, 777, dtype=np.uint8),
                        synth.revcomp(unit), rng.integers(0, 4, 600, dtype=np.uint8), unit])
    ctx.set_genomes([g])
    pat = O.get_seed(9, 0)
    for args in ((2, 1000, False), (2, 1000, True), (3, 3, False)):
        m, o, s = ctx.seed_match_enumerate(0, pat, *args)
        em, eo, es = O.seed_match_enumerate(g, pat, *args)
        assert np.array_equal(m, em) and np.array_equal(o, eo) and np.array_equal(s, es)


def _host_unique_hits(ctx, gs, pat):
    """The host callback path in miniature: merge the per-genome sorted mer lists, and for every mer apply the rule
    of UniqueMatchFinder::EnumerateMatches (UniqueMatchFinder.cpp:36-60: ids seen exactly once; >= 2 of them)."""
    keys, gen, pos, strand = [], [], [], []
    for g in range(len(gs)):
        mer, p = ctx.sorted_mer_list(g, pat)
        keys.append(mer >> np.uint64(1)); gen.append(np.full(len(mer), g)); pos.append(p); strand.append((mer & np.uint64(1)).astype(np.uint8))
    keys, gen, pos, strand = map(np.concatenate, (keys, gen, pos, strand))
    order = np.lexsort((gen, keys))
    keys, gen, pos, strand = keys[order], gen[order], pos[order], strand[order]
    cut = np.flatnonzero(np.concatenate([[True], keys[1:] != keys[:-1], [True]]))
    N = len(gs)
    masks, hp, hs = [], [], []
    for a, b in zip(cut[:-1], cut[1:]):
        if b - a < 2:
            continue
        ids, cnt = np.unique(gen[a:b], return_counts=True)
        uniq = ids[cnt == 1]
        if len(uniq) < 2:
            continue
        m = 0
        prow, srow = np.zeros(N, np.int64), np.zeros(N, np.uint8)
        for i in range(a, b):
            if gen[i] in uniq:
                m |= 1 << int(gen[i]); prow[gen[i]] = pos[i]; srow[gen[i]] = strand[i]
        masks.append(m); hp.append(prow); hs.append(srow)
    return np.array(masks, np.uint32), np.array(hp, np.int64).reshape(-1, N), np.array(hs, np.uint8).reshape(-1, N)


def test_extend_hits_host_callback_path(ctx):
    """mauve_extend_hits: hits enumerated on the host by the UniqueMatchFinder rule, extended on the device, equal
    the in-kernel UNIQUE finder and the oracle; the strand flags and window starts are validated."""
    rng = np.random.default_rng(5)
    anc = rng.integers(0, 4, 6000, dtype=np.uint8)
    gs = [synth.mutate(anc, 0.03, rng) for _ in range(4)]
    gs[2] = synth.revcomp(gs[2])
    gs[3] = np.concatenate([gs[3], gs[3][1000:1400]])          # a repeat: genome 3 drops out of those seeds
    pat = O.get_seed(9, 0)
    ctx.set_genomes(gs)
    m, p, s = _host_unique_hits(ctx, gs, pat)
    assert len(m) > 100
    for extend in (True, False):
        ln, st = ctx.extend_hits(pat, m, p, s, extend=extend)
        eln, est = O.find_matches(gs, pat, mode=1, extend=extend)
        assert np.array_equal(ln, eln) and np.array_equal(st, est)
    bad = p.copy(); bad[0, int(np.flatnonzero([m[0] >> g & 1 for g in range(4)])[0])] = 10 ** 7
    with pytest.raises(RuntimeError, match=r"\(-1\).*outside"):
        ctx.extend_hits(pat, m, bad, s)
    with pytest.raises(RuntimeError, match=r"\(-1\)"):
        ctx.extend_hits(pat, np.array([1], np.uint32), p[:1], s[:1])          # one component is not a match


def _valid_intervals(L, contig_starts, invalid):
    """1-based inclusive intervals a seed window must lie in: the contigs, cut at every ambiguous base"""
    out = []
    edges = list(contig_starts) + [L]
    for a, b in zip(edges[:-1], edges[1:]):
        cur = a
        bad = np.flatnonzero(invalid[a:b]) + a if invalid is not None else []
        for x in list(bad) + [b]:
            if x > cur:
                out.append((cur + 1, int(x)))
            cur = x + 1
    return out


def test_contigs_and_ambiguous_bases(ctx):
    """mauve_set_genomes_contigs (RepeatHashCat.h:19-20 concat_contig_start; multi-record FastA, runs of N): no seed
    window and no ungapped extension crosses a contig join or touches an ambiguous base -- bit-exact against the
    oracle's interval-restricted finder -- while matches still pair windows of different contigs."""
    from mauvealigner_amd import _lib
    rng = np.random.default_rng(42)
    anc = rng.integers(0, 4, 20000, dtype=np.uint8)
    gs = [synth.mutate(anc, 0.02, rng) for _ in range(3)]
    contigs = [[0, 5000, 5003, 12000], [0], [0, 9000]]
    invalid = [np.zeros(len(g), bool) for g in gs]
    invalid[0][7000:7100] = True                      # a run of N (packed as A: would seed poly-A matches)
    invalid[1][300] = True
    invalid[2][15000:15040] = True
    for g in range(3):
        gs[g] = gs[g].copy(); gs[g][invalid[g]] = 0
    gs[1][2000:2200] = 0; gs[0][1500:1700] = 0        # real poly-A stays seedable
    valid = [_valid_intervals(len(gs[g]), contigs[g], invalid[g]) for g in range(3)]
    ctx.set_genomes(gs, contig_starts=contigs, invalid=invalid)
    for w, mode, mask in ((11, 0, 0), (11, 1, 0), (9, 0, 7), (15, 1, 0)):
        pat = O.get_seed(w, 0)
        ln, st = ctx.seed_mums(pat, mode=mode, mask=mask)
        eln, est = O.find_matches_masked(gs, pat, valid, mode=mode, mask=mask)
        assert len(ln) > 20 and np.array_equal(ln, eln) and np.array_equal(st, est), (w, mode, mask)
        # no match of genome 0 covers an ambiguous base (windows on the two sides of a contig join may still abut
        # into one match when the join is collinear in the other genomes: no single window crosses it)
        a = np.abs(st[:, 0]); has = a > 0
        assert not np.any(has & (a <= 7100) & (a + ln - 1 >= 7001))
    # the whole path: alignment still partitions the genomes; XMFA prints N at the ambiguous bases
    ctx.set_genomes(gs, contig_starts=contigs, invalid=invalid)
    r = ctx.align(_lib.default_params(), names=["a", "b", "c"], want_xmfa=True)
    assert r["n_anchor"] > 10
    rows = "".join(l for l in r["xmfa"].splitlines() if l and l[0] not in "#>=")
    assert rows.count("N") == int(sum(x.sum() for x in invalid))
    with pytest.raises(RuntimeError, match=r"\(-1\).*ascend"):
        ctx.set_genomes(gs, contig_starts=[[0, 50, 40], [0], [0]])


def test_full_size_properties(ctx):
    """BASELINE config C2 at full size (3 x 5 Mbp, weight 15): too big for the oracle in a unit test, so check
    size-independent properties: determinism, canonical order, reverse-complement symmetry of the input."""
    gs = synth.make_config("C2", scale=1.0)
    pat = O.get_seed(15, 0)
    ctx.set_genomes(gs)
    ln, st = ctx.seed_mums(pat, mask=7)
    ln2, st2 = ctx.seed_mums(pat, mask=7)
    assert np.array_equal(ln, ln2) and np.array_equal(st, st2)
    assert len(ln) > 10000
    assert np.all(st[:, 0] > 0)
    assert np.all(np.diff(st[:, 0]) >= 0)
    assert np.all(ln >= O.seed_length(pat))
    # reverse-complementing genome 2 must mirror its coordinates and flip its strand, nothing else
    L2 = len(gs[2])
    ctx.set_genomes([gs[0], gs[1], synth.revcomp(gs[2])])
    ln3, st3 = ctx.seed_mums(pat, mask=7)
    assert np.array_equal(ln, ln3) and np.array_equal(st[:, :2], st3[:, :2])
    expect = -(L2 - (np.abs(st[:, 2]) + ln - 1) + 1) * np.sign(st[:, 2])
    assert np.array_equal(st3[:, 2], expect)


def test_pairwise_match_finder(ctx):
    """PairwiseMatchFinder (progressiveMauve.cpp:496-501): MemHash on every genome pair, sorted mer list built once."""
    from mauvealigner_amd import _lib
    for cfg, scale, w in (("C3", 0.02, 11), ("C4", 0.02, 9), ("C1", 0.2, 13)):
        gs = synth.make_config(cfg, scale=scale)
        pat = O.get_seed(w, 0)
        ln, st = _same(ctx, gs, pat, mode=_lib.MODE_PAIRWISE)
        assert np.all(np.count_nonzero(st, axis=1) == 2)
        _same(ctx, gs, pat, mode=_lib.MODE_PAIRWISE, extend=False)


def test_error_behaviour(ctx):
    """Status codes and messages at the boundary (libMems: boolean / gnException; here MAUVE_ERR_* + last error):
    the context stays usable after every refused call."""
    from mauvealigner_amd import _lib
    rng = np.random.default_rng(3)
    g = rng.integers(0, 4, 2000, dtype=np.uint8)
    fresh = _lib.Context(0)
    try:
        with pytest.raises(RuntimeError, match=r"\(-5\)"):            # MAUVE_ERR_STATE: no genomes yet
            fresh.seed_mums(O.get_seed(11, 0))
        fresh.set_genomes([g])
        with pytest.raises(RuntimeError, match=r"\(-5\).*two genomes"):
            fresh.align(_lib.default_params())
    finally:
        fresh.close()
    ctx.set_genomes([g, synth.mutate(g, 0.02, rng)])
    with pytest.raises(RuntimeError, match=r"\(-1\).*palindromic"):    # MAUVE_ERR_ARG
        ctx.seed_mums(0b1101)
    with pytest.raises((RuntimeError, IndexError)):
        ctx.sorted_mer_list(5, O.get_seed(11, 0))                       # no such sequence
    with pytest.raises(RuntimeError, match=r"\(-1\)"):
        ctx.set_genomes([g] * 33)                                        # MAUVE_MAX_SEQ = 32
    with pytest.raises(ValueError):
        ctx.dp_batch([[g[:5], g[:5]], [g[:5]]])
    # still alive, and the refused set_genomes did not replace the genomes
    ln, st = ctx.seed_mums(O.get_seed(11, 0))
    assert len(ln) > 0 and st.shape[1] == 2


def test_pairwise_passes_in_groups(ctx):
    """PairwiseMatchFinder on the device runs its N (N - 1) / 2 passes in groups of pairs with different lower genomes (one join, one
    run detection, one candidate list per group).  Without extension every hit is a candidate, so a group of near-identical genomes
    has about as many candidates as windows -- more than the P / 2 a single pass can have (the candidate list was sized for that
    once; found by the randomised sweep as a history-dependent failure).  Six genomes, with and without extension, against the oracle."""
    rng = np.random.default_rng(431)
    anc = rng.integers(0, 4, 5300, dtype=np.uint8)
    gs = [np.ascontiguousarray(synth.mutate(anc, 0.01, rng, indel_frac=0.1)) for _ in range(6)]
    ctx.set_genomes(gs)
    for w in (17, 11):
        pat = O.get_seed(w, 0)
        for ext in (False, True):
            ln, st = ctx.seed_mums(pat, mode=2, extend=ext)
            eln, est = O.find_matches(gs, pat, mode=2, extend=ext)
            assert np.array_equal(ln, eln) and np.array_equal(st, est), (w, ext)
        if w == 17:
            assert len(ln) > 0

