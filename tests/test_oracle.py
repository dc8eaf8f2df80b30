"""CPU tests of the oracle (the checker itself): brute-force cross-checks on tiny inputs, DP known-answer
tests, invariants of the LCB stage, XMFA format, and the committed golden fixtures."""
import os
import re

import numpy as np
import pytest

from mauvealigner_amd import synth
from oracle import pyoracle as O
from tests import bruteforce as B

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
HOXD = [[91, -114, -31, -123], [-114, 100, -125, -31], [-31, -125, 100, -114], [-123, -31, -114, 91]]


def _asc(c):
    return synth.to_ascii(np.asarray(c, dtype=np.uint8)).decode()


# ------------------------------------------------------------------------------------------ seeds
def test_seed_table_contract():
    for w in range(3, 32):
        spans = []
        for rank in range(3):
            p = O.get_seed(w, rank)
            s = bin(p)[2:]
            assert s == s[::-1], "palindromic"
            assert s[0] == "1" and s[-1] == "1"
            assert O.seed_weight(p) == w
            spans.append(O.seed_length(p))
        assert spans == sorted(spans)
        assert O.get_seed(w, O.SOLID_SEED) == (1 << w) - 1
        c = bin(O.get_seed(w, O.CODING_SEED))[2:]
        assert c == c[::-1] and c.count("1") == w
    assert O.get_seed(2, 0) == 0 and O.get_seed(32, 0) == 0
    # BASELINE.json config 2 quotes seed weight 15 for 5 Mbp genomes
    assert O.default_seed_weight(5_000_000) == 15
    assert O.default_seed_weight(200_000) == 13
    assert O.default_seed_weight(1) == 5


def test_pack2bit_layout():
    codes = np.array([0, 1, 2, 3] * 9, dtype=np.uint8)
    w = O.pack2bit(codes)
    assert len(w) == 3
    for i, c in enumerate(codes):
        assert (int(w[i // 16]) >> (2 * (i % 16))) & 3 == c
    assert list(O.encode(b"ACGTNacgtx")) == [0, 1, 2, 3, 0, 0, 1, 2, 3, 0]


def test_mers_against_strings():
    rng = np.random.default_rng(5)
    codes = rng.integers(0, 4, 300, dtype=np.uint8)
    s = _asc(codes)
    for w in (5, 8, 11, 16, 21):
        pat = O.get_seed(w, 0)
        offs, span = B.pattern_offsets(pat)
        canon, strand = O.mers(codes, pat)
        assert len(canon) == len(s) - span + 1
        val = {"A": 0, "C": 1, "G": 2, "T": 3}
        for p in range(len(canon)):
            f = B.masked(s, p, offs)
            r = B.rc(f)
            c = min(f, r)
            k = 0
            for ch in c:
                k = k * 4 + val[ch]
            assert int(canon[p]) == k
            assert int(strand[p]) == (1 if r < f else 0)
        # revcomp symmetry: the mer list of the reverse complement is the mirrored list, strands flipped
        c2, s2 = O.mers(synth.revcomp(codes), pat)
        assert np.array_equal(c2, canon[::-1])
        assert np.array_equal(s2, 1 - strand[::-1]) or w % 2 == 0


def test_sorted_mer_list():
    rng = np.random.default_rng(6)
    codes = rng.integers(0, 4, 2000, dtype=np.uint8)
    pat = O.get_seed(7, 0)
    mer, pos = O.sorted_mer_list(codes, pat)
    canon, strand = O.mers(codes, pat)
    assert sorted(pos.tolist()) == list(range(len(canon)))
    key = mer >> np.uint64(1)
    assert np.all(key[1:] >= key[:-1])
    w = 7
    for i in range(len(mer)):
        assert int(mer[i]) >> (64 - 2 * w) == int(canon[pos[i]])
        assert int(mer[i]) & 1 == int(strand[pos[i]])
    same = key[1:] == key[:-1]
    assert np.all(pos[1:][same] > pos[:-1][same])


# ------------------------------------------------------------------------------- MUMs vs brute force
def _tiny_set(seed, n, L, div, inv=False):
    rng = np.random.default_rng(seed)
    anc = rng.integers(0, 4, L, dtype=np.uint8)
    out = []
    for g in range(n):
        x = synth.mutate(anc, div, rng)
        if inv and g == n - 1:
            a, b = L // 3, 2 * L // 3
            x = x.copy()
            x[a:b] = synth.revcomp(x[a:b])
        out.append(x)
    return out


@pytest.mark.parametrize("seed,n,w,mode,inv", [
    (1, 2, 5, "mem", False), (2, 3, 5, "mem", True), (3, 3, 7, "unique", True),
    (4, 4, 5, "unique", False), (5, 2, 6, "mem", True), (6, 3, 8, "mem", False),
])
def test_matches_equal_bruteforce(seed, n, w, mode, inv):
    gs = _tiny_set(seed, n, 260, 0.06, inv)
    pat = O.get_seed(w, 0)
    ln, st = O.find_matches(gs, pat, mode=O.MODE_MEM if mode == "mem" else O.MODE_UNIQUE)
    got = set((int(l), tuple(int(x) for x in s)) for l, s in zip(ln, st))
    assert len(got) == len(ln), "no duplicates"
    want = B.brute_matches([_asc(g) for g in gs], pat, mode=mode)
    assert got == want
    # seeds only (SeedMatchEnumerator-style, no extension)
    ln0, st0 = O.find_matches(gs, pat, mode=O.MODE_MEM if mode == "mem" else O.MODE_UNIQUE, extend=False)
    got0 = set((int(l), tuple(int(x) for x in s)) for l, s in zip(ln0, st0))
    assert got0 == B.brute_matches([_asc(g) for g in gs], pat, mode=mode, extend=False)
    # N-way mask (MaskedMemHash::SetMask(2^N-1), mauveAligner.cpp:525-531)
    full = (1 << n) - 1
    lnm, stm = O.find_matches(gs, pat, mode=O.MODE_MEM if mode == "mem" else O.MODE_UNIQUE, mask=full)
    gotm = set((int(l), tuple(int(x) for x in s)) for l, s in zip(lnm, stm))
    assert gotm == B.brute_matches([_asc(g) for g in gs], pat, mode=mode, mask=full)


def test_pairwise_finder_is_memhash_per_pair():
    """PairwiseMatchFinder = the MemHash search of each pair alone (brute force on the pair's two strings)."""
    gs = _tiny_set(11, 4, 240, 0.05, inv=True)
    pat = O.get_seed(5, 0)
    ln, st = O.find_matches(gs, pat, mode=O.MODE_PAIRWISE)
    got = set((int(l), tuple(int(x) for x in s)) for l, s in zip(ln, st))
    want = set()
    for i in range(4):
        for j in range(i + 1, 4):
            for l, (a, b) in B.brute_matches([_asc(gs[i]), _asc(gs[j])], pat, mode="mem"):
                row = [0, 0, 0, 0]
                row[i], row[j] = a, b
                want.add((l, tuple(row)))
    assert got == want and len(got) == len(ln)


def test_masked_matches_equal_bruteforce():
    """S9: windows touching a base outside the valid intervals are invisible to seeding and extension."""
    gs = _tiny_set(21, 3, 300, 0.04, inv=True)
    pat = O.get_seed(5, 0)
    valid = [[(1, 120), (160, 300)], [(10, 290)], [(1, 99), (101, 180), (200, len(gs[2]))]]
    ln, st = O.find_matches_masked(gs, pat, valid)
    got = set((int(l), tuple(int(x) for x in s)) for l, s in zip(ln, st))
    want = B.brute_matches([_asc(g) for g in gs], pat, mode="mem", valid=valid)
    assert got == want and len(got) == len(ln)
    # no match may touch a masked base
    for l, row in got:
        for g, s0 in enumerate(row):
            if s0:
                le, re = abs(s0), abs(s0) + l - 1
                assert any(lo <= le and re <= hi for lo, hi in valid[g])
    # an all-valid mask changes nothing
    ln2, st2 = O.find_matches_masked(gs, pat, [[(1, len(g))] for g in gs])
    ln3, st3 = O.find_matches(gs, pat)
    assert np.array_equal(ln2, ln3) and np.array_equal(st2, st3)


def test_guide_tree_upgma():
    gs = synth.tree_genomes(4, 20000, 0.02, 77, inv_per_branch=0, insert_per_branch=0)
    dist, left, right = O.guide_tree(gs, O.get_seed(11, 0))
    assert np.array_equal(dist, dist.T) and np.all(np.diag(dist) == 0)
    assert dist[0, 1] < dist[0, 2] and dist[2, 3] < dist[1, 2]
    assert sorted([(int(left[4]), int(right[4])), (int(left[5]), int(right[5]))]) == [(0, 1), (2, 3)]
    assert (int(left[6]), int(right[6])) == (4, 5)
    assert left[:4].tolist() == [-1] * 4


def test_progressive_align_properties():
    """Guide-tree recursive anchoring: every base in exactly one block, rows reproduce the genomes, clade-specific
    sequence is aligned below the root, and with nothing clade-specific the result equals the plain N-way path."""
    rng = np.random.default_rng(5)
    anc = rng.integers(0, 4, 30000, dtype=np.uint8)
    Lc, Rc = synth.mutate(anc, 0.02, rng), synth.mutate(anc, 0.02, rng)
    Lc = np.concatenate([Lc[:10000], rng.integers(0, 4, 3000, dtype=np.uint8), Lc[10000:]])
    Rc = np.concatenate([Rc[:20000], rng.integers(0, 4, 2500, dtype=np.uint8), Rc[20000:]])
    gs = [synth.mutate(Lc, 0.005, rng), synth.mutate(Lc, 0.005, rng), synth.mutate(Rc, 0.005, rng), synth.mutate(Rc, 0.005, rng)]
    r = O.progressive_align(gs, O.default_params(max_gapped_len=1000), want_xmfa=True)
    _check_xmfa(r["xmfa"], gs)
    a = r["aln"]
    sets = [tuple(np.flatnonzero(a["left"][b]).tolist()) for b in range(a["n_iv"])]
    assert (0, 1, 2, 3) in sets and (0, 1) in sets and (2, 3) in sets
    b01 = sets.index((0, 1))
    assert a["right"][b01, 0] - a["left"][b01, 0] + 1 >= 2900          # the clade insert is aligned at node (0,1)
    # two genomes: the tree is trivial and the path is the plain one
    two = gs[:2]
    rp = O.progressive_align(two, want_xmfa=True)
    ra = O.align(two, want_xmfa=True)
    assert rp["xmfa"] == ra["xmfa"]


def test_progressive_align_given_tree():
    """--input-guide-tree: the oracle's own UPGMA tree handed back in reproduces its result, another tree is a different
    but complete alignment, and orc_check_tree refuses what is not a binary tree in merge order."""
    gs = synth.make_config("C4", scale=0.01)[:4]
    N = len(gs)
    r0 = O.progressive_align(gs, want_xmfa=True)
    r1 = O.progressive_align(gs, want_xmfa=True, tree=r0["tree"])
    assert r0["xmfa"] == r1["xmfa"]
    left = np.array([-1, -1, -1, -1, 0, 4, 5], np.int32)     # caterpillar ((0,3),1),2
    right = np.array([-1, -1, -1, -1, 3, 1, 2], np.int32)
    assert O.check_tree(N, left, right)
    r2 = O.progressive_align(gs, want_xmfa=True, tree=(left, right))
    _check_xmfa(r2["xmfa"], gs)
    bad = left.copy(); bad[5] = 0                            # leaf 0 used twice
    assert not O.check_tree(N, bad, right)
    with pytest.raises(RuntimeError):
        O.progressive_align(gs, tree=(bad, right))
    assert not O.check_tree(N, left[:-1], right[:-1])


def test_progressive_weight_scaling_oracle():
    """S11b in the oracle: scale 0 changes nothing, a scaled threshold never exceeds the unscaled one (so it keeps at
    least the blocks a heavier threshold keeps), and the floor is respected."""
    gs = synth.make_config("C4", scale=0.01)[:4]
    base = O.progressive_align(gs, O.default_params(lcb_weight=300), want_xmfa=True)
    zero = O.progressive_align(gs, O.default_params(lcb_weight=300, weight_scaling=1, conservation_scale_ppm=0), want_xmfa=True)
    assert zero["xmfa"] == base["xmfa"]
    full = O.progressive_align(gs, O.default_params(lcb_weight=300, weight_scaling=1, conservation_scale_ppm=1000000), want_xmfa=True)
    _check_xmfa(full["xmfa"], gs)
    multi = lambda r: int(np.count_nonzero(np.count_nonzero(r["aln"]["left"], axis=1) >= 2))
    assert multi(full) >= multi(base)
    floor = O.progressive_align(gs, O.default_params(lcb_weight=300, weight_scaling=1, conservation_scale_ppm=1000000, min_scaled_penalty=10 ** 9))
    assert multi(floor) == 0


def test_seed_family_merge_known_answers_and_host_entry():
    """a later seed's match is dropped only when an earlier one contains it on the same diagonal (forward
    and reverse components, subsets of the components); mauve_merge_matches (host code of the product) agrees with the
    oracle on random lists."""
    from mauvealigner_amd import _lib
    # same diagonal inside -> dropped; off the diagonal, sticking out -> kept
    ln, st = O.merge_matches([100], [[100, 500]], [30, 30, 50], [[120, 520], [120, 521], [180, 580]])
    assert ln.tolist() == [100, 30, 50] and st.tolist() == [[100, 500], [120, 521], [180, 580]]
    # reverse component: column k of the kept match sits at 500 + 99 - k, so x = (120, -550) is inside and (120, -549) is not
    ln, st = O.merge_matches([100], [[100, -500]], [30, 30], [[120, -550], [120, -549]])
    assert st.tolist() == [[100, -500], [120, -549]]
    # a two-component match inside a three-component one is contained; the other way round never
    ln, st = O.merge_matches([100], [[100, 500, 900]], [40, 40], [[0, 510, 910], [110, 510, 0]])
    assert len(ln) == 1
    ln, st = O.merge_matches([40], [[0, 510, 910]], [100], [[100, 500, 900]])
    assert len(ln) == 2
    # x found with its first component forward, inside a match that reads that component on the reverse strand
    ln, st = O.merge_matches([100], [[100, -500, 900]], [30], [[0, 550, -920]])       # flipped view of columns 20..49
    assert len(ln) == 1, st
    ln, st = O.merge_matches([100], [[100, -500, 900]], [30], [[0, 550, -921]])
    assert len(ln) == 2
    rng = np.random.default_rng(11)
    for trial in range(60):
        N = int(rng.integers(2, 5))
        na, nb = int(rng.integers(0, 60)), int(rng.integers(0, 60))

        def rand_list(n, base=None):
            ln_, st_ = [], []
            for _ in range(n):
                if base is not None and len(base[0]) and rng.random() < 0.5:      # cut out of a kept match: contained, or nearly
                    k = int(rng.integers(0, len(base[0])))
                    L, s = int(base[0][k]), base[1][k]
                    l2 = int(rng.integers(1, L + 1)); d = int(rng.integers(0, L - l2 + 1))
                    row = []
                    for g in range(N):
                        if s[g] == 0 or rng.random() < 0.15:
                            row.append(0)
                        elif s[g] > 0:
                            row.append(int(s[g]) + d)
                        else:
                            row.append(-(abs(int(s[g])) + L - l2 - d))
                    if rng.random() < 0.3 and any(row):
                        g = int(rng.choice([g for g in range(N) if row[g]])); row[g] += 1 if row[g] > 0 else -1     # off the diagonal
                    nz = [g for g in range(N) if row[g]]
                    if len(nz) < 2:
                        continue
                    if row[nz[0]] < 0:                                            # canonical form: first component forward
                        row = [(-(abs(v)) if v > 0 else abs(v)) if v else 0 for v in row]
                        # (flipping the view turns forward into reverse and keeps the left ends)
                    ln_.append(l2); st_.append(row)
                else:
                    L = int(rng.integers(5, 300))
                    row = [int(rng.integers(1, 5000)) * int(rng.choice([1, 1, -1])) if rng.random() < 0.8 else 0 for _ in range(N)]
                    nz = [g for g in range(N) if row[g]]
                    if len(nz) < 2:
                        continue
                    row[nz[0]] = abs(row[nz[0]])
                    ln_.append(L); st_.append(row)
            return np.array(ln_, np.int64), np.array(st_, np.int64).reshape(len(ln_), N)
        a = rand_list(na)
        b = rand_list(nb, a)
        if len(a[0]) + len(b[0]) == 0:
            continue
        # canonical order of the inputs (the oracle's merge of a list with the empty list sorts it)
        a = O.merge_matches(np.zeros(0, np.int64), np.zeros((0, N), np.int64), a[0], a[1]) if len(a[0]) else a
        want = O.merge_matches(a[0], a[1], b[0], b[1])
        got = _lib.merge_matches(a[0], a[1], b[0], b[1])
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), trial


def _cols_of(rows):
    cols = np.zeros(len(rows[0]), np.uint32)
    for g, r in enumerate(rows):
        cols |= (np.frombuffer(r.encode(), np.uint8) != ord("-")).astype(np.uint32) << np.uint32(g)
    return cols


def test_backbone_oracle_known_answers():
    """S12 on hand cases: islands, open regions at the ends, transitive components, reverse strands."""
    # genome 1 lacks five columns: islands of 0 and of 2 against 1 when the gap limit is 3, none when it is 5
    cols = _cols_of(["x" * 20, "x" * 10 + "-" * 5 + "x" * 5, "x" * 20])
    left, right, rev = np.array([[1, 101, 201]]), np.array([[20, 115, 220]]), np.array([[0, 0, 1]], np.int8)
    r = O.backbone(left, right, rev, [0, 20], cols, island_gap=3)
    assert r["seg_mask"].tolist() == [7, 5, 7] and r["seg_col"].tolist() == [0, 10, 15] and r["seg_len"].tolist() == [10, 5, 5]
    assert r["seg_left"].tolist() == [[1, 101, -211], [11, 0, -206], [16, 111, -201]]
    assert r["seg_right"].tolist() == [[10, 110, -220], [15, 0, -210], [20, 115, -205]]
    assert r["islands"].tolist() == [[0, 0, 1, 0, 10, 14, 11, 15], [0, 1, 2, 2, 10, 14, -206, -210]]
    r = O.backbone(left, right, rev, [0, 20], cols, island_gap=5)
    assert r["seg_mask"].tolist() == [7] and r["seg_len"].tolist() == [20] and len(r["islands"]) == 0
    # ends: a pair is not joined before its first / after its last common column, however short the overhang
    cols = _cols_of(["xxxxxxxx--", "--xxxxxxxx"])
    r = O.backbone(np.array([[1, 1]]), np.array([[8, 8]]), np.zeros((1, 2), np.int8), [0, 10], cols, island_gap=20)
    assert r["seg_col"].tolist() == [2] and r["seg_len"].tolist() == [6] and r["seg_left"].tolist() == [[3, 1]] and r["seg_right"].tolist() == [[8, 6]]
    # two short one-sided runs of different genomes in a row are two runs, not one: no island at gap 3, both at gap 2
    cols = _cols_of(["xx" + "xxx" + "---" + "xx", "xx" + "---" + "xxx" + "xx"])
    lr = (np.array([[1, 1]]), np.array([[7, 7]]), np.zeros((1, 2), np.int8), [0, 10], cols)
    assert O.backbone(*lr, island_gap=3)["seg_len"].tolist() == [10]
    r = O.backbone(*lr, island_gap=2)
    assert r["seg_col"].tolist() == [0, 8] and r["islands"][:, 3:6].tolist() == [[0, 2, 4], [1, 5, 7]]
    # 0-1 share the first half, 1-2 the second, 0-2 never a column: two components, one after the other
    cols = _cols_of(["xxxx----", "xxxxxxxx", "----xxxx"])
    r = O.backbone(np.array([[1, 1, 1]]), np.array([[4, 8, 4]]), np.zeros((1, 3), np.int8), [0, 8], cols, island_gap=20)
    assert r["seg_mask"].tolist() == [3, 6] and r["seg_col"].tolist() == [0, 4]
    # columns neither genome of a pair has are skipped: the run of 0 against 1 is 4 + 4 columns around genome 2's insert
    cols = _cols_of(["xx" + "xxxx" + "---" + "xxxx" + "xx", "xx" + "----" + "---" + "----" + "xx", "xx" + "----" + "xxx" + "----" + "xx"])
    r = O.backbone(np.array([[1, 1, 1]]), np.array([[12, 4, 7]]), np.zeros((1, 3), np.int8), [0, 15], cols, island_gap=7)
    assert [row[1:6] for row in r["islands"].tolist() if row[1:3] == [0, 1]] == [[0, 1, 0, 2, 12]]


def test_matches_canonical_order_and_content():
    gs = synth.make_config("C1", scale=0.05)
    pat = O.get_seed(11, 0)
    ln, st = O.find_matches(gs, pat)
    assert len(ln) > 10
    # every match: masked windows agree at its first and last window; first component forward
    offs, span = B.pattern_offsets(pat)
    s = [_asc(g) for g in gs]
    for l, row in zip(ln, st):
        assert row[0] > 0 or row[0] == 0
        comps = [g for g in range(2) if row[g] != 0]
        for off in (0, l - span):
            ws = []
            for g in comps:
                if row[g] > 0:
                    ws.append(B.masked(s[g], row[g] - 1 + off, offs))
                else:
                    le = -row[g] - 1
                    ws.append(B.rc(B.masked(s[g], le + (l - span - off), offs)))
            assert len(set(ws)) == 1
    keyed = [(abs(int(r[0])), tuple(int(x) for x in r)) for r in st]
    assert keyed == sorted(keyed)


def test_edge_cases_empty_and_short():
    pat = O.get_seed(11, 0)
    ln, st = O.find_matches([np.zeros(0, np.uint8), np.zeros(5, np.uint8)], pat)
    assert len(ln) == 0
    # identical genomes: one match covering everything
    rng = np.random.default_rng(9)
    g = rng.integers(0, 4, 500, dtype=np.uint8)
    ln, st = O.find_matches([g, g.copy()], pat)
    assert list(ln) == [500] and st.tolist() == [[1, 1]]
    # reverse complement: one reverse match
    ln, st = O.find_matches([g, synth.revcomp(g)], pat)
    assert list(ln) == [500] and st.tolist() == [[1, -1]]
    # a mer repeated inside one genome: MEM drops the seed, UNIQUE keeps the other genomes
    a = rng.integers(0, 4, 200, dtype=np.uint8)
    rep = np.concatenate([a, rng.integers(0, 4, 50, dtype=np.uint8), a])
    ln_m, st_m = O.find_matches([a, a.copy(), rep], pat, mode=O.MODE_MEM)
    ln_u, st_u = O.find_matches([a, a.copy(), rep], pat, mode=O.MODE_UNIQUE)
    assert all(r[2] == 0 for r in st_m.tolist()) or len(ln_m) == 0
    assert [200, [1, 1, 0]] in [[int(l), r] for l, r in zip(ln_u, st_u.tolist())]


def test_seed_match_enumerator():
    rng = np.random.default_rng(11)
    unit = rng.integers(0, 4, 40, dtype=np.uint8)
    g = np.concatenate([rng.integers(0, 4, 100, dtype=np.uint8), unit, rng.integers(0, 4, 77, dtype=np.uint8),
                        synth.revcomp(unit), rng.integers(0, 4, 60, dtype=np.uint8), unit])
    pat = O.get_seed(9, 0)
    span = O.seed_length(pat)
    mult, off, st = O.seed_match_enumerate(g, pat, 2, 1000, False)
    assert len(mult) >= 40 - span + 1
    trip = [st[off[i]:off[i + 1]].tolist() for i in range(len(mult)) if mult[i] == 3]
    assert 40 - span + 1 <= len(trip) <= 40 - span + 4
    for t in trip:
        assert t[0] > 0 and t[1] < 0 and t[2] > 0          # SetDirection parity (SeedMatchEnumerator.h:127-141)
        assert abs(t[0]) < abs(t[1]) < abs(t[2])           # sorted by position (:73)
    mult_d, off_d, st_d = O.seed_match_enumerate(g, pat, 2, 1000, True)
    for i in range(len(mult_d)):
        assert all(x > 0 for x in st_d[off_d[i]:off_d[i + 1]])
    assert sum(1 for i in range(len(mult_d)) if mult_d[i] == 2) >= 40 - span + 1
    m2, _, _ = O.seed_match_enumerate(g, pat, 3, 3, False)
    assert all(x == 3 for x in m2)


# ------------------------------------------------------------------------------------------- LCBs
def test_eliminate_overlaps_properties():
    gs = synth.make_config("C3", scale=0.01)
    N = len(gs)
    pat = O.get_seed(11, 0)
    ln, st = O.find_matches(gs, pat)
    ln, st = O.multiplicity_filter(ln, st, N)
    l2, s2 = O.eliminate_overlaps(ln, st)
    assert len(l2) <= len(ln) and len(l2) > 0
    for g in range(N):
        le = np.abs(s2[:, g])
        order = np.argsort(le, kind="stable")
        re = le[order] + l2[order] - 1
        assert np.all(le[order][1:] > re[:-1]), "no overlap left in genome %d" % g
    # idempotent
    l3, s3 = O.eliminate_overlaps(l2, s2)
    assert np.array_equal(l2, l3) and np.array_equal(s2, s3)


def test_lcbs_recover_inversions():
    rng = np.random.default_rng(3)
    L = 60000
    anc = rng.integers(0, 4, L, dtype=np.uint8)
    g1 = anc.copy()
    g1[20000:30000] = synth.revcomp(g1[20000:30000])
    g1[45000:50000] = synth.revcomp(g1[45000:50000])
    gs = [synth.mutate(anc, 0.01, rng), synth.mutate(g1, 0.01, rng)]
    r = O.align(gs, O.default_params(recursive=0, gapped=0))
    lc = r["lcbs"]
    assert lc["n_lcb"] == 5
    orient = [int(x < 0) for x in lc["left_end"][:, 1]]
    assert orient == [0, 1, 0, 1, 0]
    assert np.all(lc["weight"] >= 3 * 13 * 2)
    # adjacency: genome 0 order is the LCB numbering
    assert lc["left_adj"][:, 0].tolist() == [-1, 0, 1, 2, 3]
    assert lc["right_adj"][:, 0].tolist() == [1, 2, 3, 4, -1]
    assert sorted(lc["left_adj"][:, 1].tolist()) == [-1, 0, 1, 2, 3]
    # collinear flag (mauveAligner.cpp:665-666) forces a single LCB
    r2 = O.align(gs, O.default_params(recursive=0, gapped=0, collinear=1))
    assert r2["lcbs"]["n_lcb"] == 1


def test_lcb_greedy_removes_small_blocks():
    # hand-made 2-way matches: two long collinear blocks with a small reversed block between them
    length = np.array([100, 100, 10, 100, 100], dtype=np.int64)
    start = np.array([[1, 1], [201, 201], [401, -5000], [601, 601], [801, 801]], dtype=np.int64)
    d = O.compute_lcbs(length, start, 50)
    assert d["n_lcb"] == 1 and d["match_lcb"].tolist() == [0, 0, -1, 0, 0] and d["weight"].tolist() == [800]
    d = O.compute_lcbs(length, start, 10)
    assert d["n_lcb"] == 3 and d["match_lcb"].tolist() == [0, 0, 1, 2, 2]
    assert d["left_end"].tolist() == [[1, 1], [401, -5000], [601, 601]]
    assert d["right_end"].tolist() == [[300, 300], [410, -5009], [900, 900]]


# --------------------------------------------------------------------------------------------- DP
def test_dp_known_answers():
    sc = O.default_scoring()
    A, Cc, G, T = 0, 1, 2, 3

    def cnt_of(seq):
        c = np.zeros((len(seq), 4), dtype=np.uint8)
        for i, b in enumerate(seq):
            c[i, b] = 1
        return c
    # identical 8-mers: all aligned, score = sum of diagonal
    s = [A, Cc, G, T, A, Cc, G, T]
    ops, score = O.profile_dp(cnt_of(s), 1, s)
    assert ops.tolist() == [3] * 8 and score == 2 * (91 + 100 + 100 + 91)
    # one deleted base: a single gap column of cost -400
    t = s[:3] + s[4:]
    ops, score = O.profile_dp(cnt_of(s), 1, t)
    assert sorted(ops.tolist()) == [1] + [3] * 7 and score == 2 * (91 + 100 + 100 + 91) - 91 - 400
    # two adjacent deleted bases: open + extend
    u = s[:3] + s[5:]
    ops, score = O.profile_dp(cnt_of(s), 1, u)
    assert ops.tolist().count(1) == 2 and score == 2 * (91 + 100 + 100 + 91) - 91 - 91 - 430
    # empty against non-empty
    ops, score = O.profile_dp(np.zeros((0, 4), np.uint8), 1, s)
    assert ops.tolist() == [2] * 8 and score == -400 - 7 * 30
    ops, score = O.profile_dp(cnt_of(s), 1, [])
    assert ops.tolist() == [1] * 8 and score == -400 - 7 * 30
    # transition mismatch preferred over gaps: A vs G = -31
    ops, score = O.profile_dp(cnt_of([A, A, A]), 1, [A, G, A])
    assert ops.tolist() == [3, 3, 3] and score == 91 - 31 + 91


@pytest.mark.parametrize("seed", range(12))
def test_dp_optimal_vs_exhaustive(seed):
    rng = np.random.default_rng(100 + seed)
    m, n = int(rng.integers(0, 5)), int(rng.integers(0, 5))
    k = int(rng.integers(1, 4))
    cnt = np.zeros((m, 4), dtype=np.uint8)
    for i in range(m):
        r = int(rng.integers(1, k + 1))
        for _ in range(r):
            cnt[i, int(rng.integers(0, 4))] += 1
    seq = rng.integers(0, 4, n, dtype=np.uint8)
    ops, score = O.profile_dp(cnt, k, seq)
    best = B.brute_best_score(cnt.tolist(), k, seq.tolist(), HOXD, -400, -30)
    if m == 0 and n == 0:
        assert score == 0 and len(ops) == 0
        return
    assert score == best
    assert B.score_path(ops.tolist(), cnt.tolist(), k, seq.tolist(), HOXD, -400, -30) == score


def test_align_interval_progressive():
    rng = np.random.default_rng(7)
    base = rng.integers(0, 4, 60, dtype=np.uint8)
    seqs = [base, np.delete(base, [10, 11]), np.insert(base, 30, [1, 2, 3]), np.zeros(0, np.uint8)]
    cols, score = O.align_interval(seqs)
    for g, s in enumerate(seqs):
        assert int(((cols >> g) & 1).sum()) == len(s)
    assert len(cols) == 63
    assert np.all(cols != 0)
    assert int(((cols >> 3) & 1).sum()) == 0


# ------------------------------------------------------------------------------- whole path + XMFA
def _check_xmfa(xmfa, gs):
    assert xmfa.startswith("#FormatVersion Mauve1\n")
    N = len(gs)
    for i in range(N):
        assert "#Sequence%dEntry\t%d\n" % (i + 1, i + 1) in xmfa
        assert "#Sequence%dFormat\tFastA\n" % (i + 1) in xmfa
    body = xmfa[xmfa.index("> "):] if "> " in xmfa else ""
    blocks = [b for b in body.split("=\n") if b.strip()]
    cover = [np.zeros(len(g), dtype=np.int32) for g in gs]
    asc = [_asc(g) for g in gs]
    for b in blocks:
        rows = []
        for ent in b.split("> ")[1:]:
            head, *lines = ent.split("\n")
            mo = re.match(r"(\d+):(\d+)-(\d+) ([+-]) ", head + " ")
            g, le, re_, strand = int(mo.group(1)) - 1, int(mo.group(2)), int(mo.group(3)), mo.group(4)
            assert all(len(x) <= 80 for x in lines)
            assert all(len(x) == 80 for x in lines[:-2])
            text = "".join(lines)
            rows.append(text)
            res = text.replace("-", "")
            seg = asc[g][le - 1:re_]
            assert res == (seg if strand == "+" else B.rc(seg))
            cover[g][le - 1:re_] += 1
        assert len(set(len(r) for r in rows)) == 1
    for c in cover:
        assert np.all(c == 1), "every base appears in exactly one block"
    return blocks


def test_whole_path_c1_xmfa():
    gs = synth.make_config("C1", scale=0.1)
    r = O.align(gs, want_xmfa=True)
    blocks = _check_xmfa(r["xmfa"], gs)
    assert r["lcbs"]["n_lcb"] == 1
    a = r["aln"]
    # anchors are collinear, non-overlapping and inside the LCB
    st, ln = a["anchor_start"], a["anchor_length"]
    assert np.all(st[1:, 0] > st[:-1, 0] + ln[:-1] - 1)
    assert np.all(st[1:, 1] > st[:-1, 1] + ln[:-1] - 1)
    cols = a["cols"][a["col_off"][0]:a["col_off"][1]]
    frac_aligned = np.mean(cols == 3)
    assert frac_aligned > 0.9
    assert len(blocks) == a["n_iv"]


def test_whole_path_inversions_and_recursion():
    gs = synth.make_config("C3", scale=0.01)
    r0 = O.align(gs, O.default_params(recursive=0), want_xmfa=True)
    r1 = O.align(gs, O.default_params(recursive=1), want_xmfa=True)
    _check_xmfa(r0["xmfa"], gs)
    _check_xmfa(r1["xmfa"], gs)
    assert r1["lcbs"]["n_lcb"] == r0["lcbs"]["n_lcb"] >= 3
    assert len(r1["aln"]["anchor_length"]) >= len(r0["aln"]["anchor_length"])
    # a hyper-divergent stretch forces recursion to find extra anchors
    rng = np.random.default_rng(21)
    anc = rng.integers(0, 4, 30000, dtype=np.uint8)
    b = synth.mutate(anc, 0.01, rng)
    b[10000:14000] = synth.mutate(b[10000:14000], 0.35, rng, indel_frac=0.0)[:4000]
    ra = O.align([anc, b], O.default_params(recursive=0))
    rb = O.align([anc, b], O.default_params(recursive=1))
    assert len(rb["aln"]["anchor_length"]) > len(ra["aln"]["anchor_length"])
    assert rb["aln"]["n_dp_cells"] < ra["aln"]["n_dp_cells"]


# ---------------------------------------------------------------------------------------- goldens
@pytest.mark.parametrize("name", ["g2x2k", "g3x5k_inv", "g5x3k_unique"])
def test_golden_fixtures_reproduce(name):
    path = os.path.join(GOLDEN, name + ".npz")
    z = np.load(path)
    N = int(z["nseq"])
    gs = [z["genome%d" % g] for g in range(N)]
    pat = int(z["pattern"])
    ln, st = O.find_matches(gs, pat, mode=int(z["mode"]))
    assert np.array_equal(ln, z["mum_length"]) and np.array_equal(st, z["mum_start"])
    p = O.default_params(seed_pattern=pat, mode=int(z["mode"]))
    r = O.align(gs, p, names=["g%d" % g for g in range(N)], want_xmfa=True)
    assert np.array_equal(r["aln"]["cols"], z["cols"])
    assert np.array_equal(r["aln"]["col_off"], z["col_off"])
    assert np.array_equal(r["aln"]["left"], z["left"]) and np.array_equal(r["aln"]["right"], z["right"])
    assert np.array_equal(r["aln"]["dp_score"], z["dp_score"])
    assert np.array_equal(r["lcbs"]["weight"], z["lcb_weight"])
    assert np.array_equal(r["aln"]["anchor_start"], z["anchor_start"])
    with open(os.path.join(GOLDEN, name + ".xmfa")) as f:
        assert f.read() == r["xmfa"]


def test_golden_round2_fixture_reproduces():
    """seed-family alignment + its backbone / islands, and a progressive alignment along a given tree with scaled node
    weights: the committed fixture is what the oracle computes."""
    z = np.load(os.path.join(GOLDEN, "g3x6k_round2.npz"))
    gs = [z["genome%d" % g] for g in range(3)]
    names = ["g%d" % g for g in range(3)]
    fam = O.align(gs, O.default_params(seed_weight=int(z["seed_weight"]), seed_family=1), names=names, want_xmfa=True)
    assert np.array_equal(fam["mums"][0], z["fam_mum_length"]) and np.array_equal(fam["mums"][1], z["fam_mum_start"])
    assert len(z["fam_mum_length"]) > int(z["one_seed_mums"])
    a = fam["aln"]
    for k in ("anchor_start", "anchor_length", "left", "right", "reverse", "col_off", "cols"):
        assert np.array_equal(a[k], z["fam_" + k]), k
    with open(os.path.join(GOLDEN, "g3x6k_round2.xmfa")) as f:
        assert f.read() == fam["xmfa"]
    bb = O.backbone(a["left"], a["right"], a["reverse"], a["col_off"], a["cols"], island_gap=int(z["bb_island_gap"]))
    for k in ("seg_iv", "seg_col", "seg_len", "seg_mask", "seg_left", "seg_right", "islands"):
        assert np.array_equal(bb[k], z["bb_" + k]), k
    pr = O.progressive_align(gs, O.default_params(seed_weight=int(z["seed_weight"]), weight_scaling=1, conservation_scale_ppm=500000),
                             tree=(z["tree_left"], z["tree_right"]))["aln"]
    for k in ("left", "right", "reverse", "col_off", "cols"):
        assert np.array_equal(pr[k], z["prog_" + k]), k


def test_golden_progressive_fixture_reproduces():
    z = np.load(os.path.join(GOLDEN, "g4x3k_tree.npz"))
    N = int(z["nseq"])
    gs = [z["genome%d" % g] for g in range(N)]
    pat = int(z["pattern"])
    ln, st = O.find_matches(gs, pat, mode=O.MODE_PAIRWISE)
    assert np.array_equal(ln, z["pair_length"]) and np.array_equal(st, z["pair_start"])
    r = O.progressive_align(gs, O.default_params(seed_pattern=pat), names=["g%d" % g for g in range(N)], want_xmfa=True)
    assert np.array_equal(r["tree"][0], z["tree_left"]) and np.array_equal(r["tree"][1], z["tree_right"])
    assert np.array_equal(r["dist"], z["dist"])
    for k in ("left", "right", "reverse", "col_off", "cols", "dp_score"):
        assert np.array_equal(r["aln"][k], z[k]), k
    with open(os.path.join(GOLDEN, "g4x3k_tree.xmfa")) as f:
        assert f.read() == r["xmfa"]


def test_banded_dp_oracle():
    """Banded profile DP: equal to the full DP while the band covers the matrix, never better than it,
    cut by a shift larger than the band, and the cell count is the band's area."""
    rng = np.random.default_rng(21)
    for _ in range(30):
        a = rng.integers(0, 4, int(rng.integers(1, 129)), dtype=np.uint8)
        b = rng.integers(0, 4, int(rng.integers(1, 129)), dtype=np.uint8)
        f, g = O.align_interval([a, b]), O.align_interval([a, b], banded=True)
        assert np.array_equal(f[0], g[0]) and f[1] == g[1]
    L, shift = 4000, 700
    a = rng.integers(0, 4, L, dtype=np.uint8)
    b = np.concatenate([a[:600], rng.integers(0, 4, shift, dtype=np.uint8), a[600:2000], a[2000 + shift:]])
    f = O.align_interval([a, b], want_cells=True)
    g = O.align_interval([a, b], banded=True, want_cells=True)
    assert g[1] < f[1] and f[2] == L * L
    W = 128 + 0 + (2 * L) // 64
    assert g[2] == sum(min(L, (i * L) // L + W) - max(1, (i * L) // L - W) + 1 for i in range(1, L + 1))
    # every sequence is spelled out once, in order, whatever the band did
    for k, s in enumerate((a, b)):
        assert int(((g[0] >> k) & 1).sum()) == len(s)
    # a small shift stays inside the band: same optimum
    b2 = np.concatenate([a[:600], rng.integers(0, 4, 60, dtype=np.uint8), a[600:]])
    assert O.align_interval([a, b2])[1] == O.align_interval([a, b2], banded=True)[1]


def test_sp_scoring_oracle():
    """Extant sum-of-pairs anchor scores: by hand on a tiny case, and through the LCB computation."""
    sc = O.default_scoring()
    M = np.array([[sc.matrix[i][j] for j in range(4)] for i in range(4)], dtype=np.int64)
    A, C_, G, T = 0, 1, 2, 3
    g0 = np.array([A, C_, G, T, A, A], np.uint8)
    g1 = np.array([T, A, C_, G, A, A], np.uint8)                  # g0[0:3] = g1[1:4]
    g2 = np.array([T, T, C_, G, T, G], np.uint8)                  # reverse complement of g2[1:4] = (C, G, A) -> rc = T C G ... checked below
    # match of length 3: g0 at 1, g1 at 2, g2 reverse at 2 (covers g2[1:4] = T C G, read backwards and complemented: C G A)
    s = O.match_sp_scores([g0, g1, g2], [3], [[1, 2, -2]])
    cols = [(g0[0 + c], g1[1 + c], 3 - g2[1 + (2 - c)]) for c in range(3)]
    want = sum(int(M[a][b] + M[a][d] + M[b][d]) for a, b, d in cols)
    assert int(s[0]) == want
    # an absent component drops its pairs
    s2 = O.match_sp_scores([g0, g1, g2], [3], [[1, 2, 0]])
    assert int(s2[0]) == sum(int(M[a][b]) for a, b, _ in cols)
    # whole path: SP scoring keeps fewer or equal LCBs than no threshold and reports score weights
    gs = synth.make_config("C3", scale=0.02)
    e0 = O.align(gs, O.default_params())
    e1 = O.align(gs, O.default_params(lcb_scoring=1))
    assert e1["lcbs"]["n_lcb"] >= 1
    ml, ms = e1["aln"]["anchor_length"], e1["aln"]["anchor_start"]
    assert int(e1["lcbs"]["weight"].sum()) > int(e0["lcbs"]["weight"].sum())       # scores (~95 per pair and column) against lengths


def test_c1_full_size_plumbing():
    """BASELINE config 1 at its full size (2 x 200 kbp, default seed weight, XMFA out) through the CPU restatement: every base of
    both genomes in exactly one interval, the rows of the XMFA text spell the genomes, the text parses back into the same
    intervals (the same checks the GPU suite makes at this size, tests/test_gpu_align.py::test_align_equals_oracle[C1-1.0])."""
    gs = synth.make_config("C1", scale=1.0)
    names = ["c1a.fas", "c1b.fas"]
    r = O.align(gs, O.default_params(), names=names, want_xmfa=True)
    a = r["aln"]
    assert r["lcbs"]["n_lcb"] >= 1 and len(a["anchor_length"]) > 300
    N = 2
    for g in range(N):
        cover = np.zeros(len(gs[g]) + 1, np.int64)
        pres = a["left"][:, g] != 0
        np.add.at(cover, a["left"][pres, g] - 1, 1); np.add.at(cover, a["right"][pres, g], -1)
        assert np.all(np.cumsum(cover)[:-1] == 1)
    # the XMFA text: one block per interval, rows without gaps equal the genome stretch (reverse rows: reverse complement)
    blocks = r["xmfa"].split("=\n")[:-1]
    assert len(blocks) == a["n_iv"]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    for blk in blocks[:50] + blocks[-5:]:
        rows = blk.split(">")[1:]
        for row in rows:
            head, *lines = row.split("\n")
            g = int(head.split(":")[0]) - 1
            lo, hi = [int(x) for x in head.split(":")[1].split()[0].split("-")]
            seq = "".join(lines).replace("-", "")
            want = _asc(gs[g])[lo - 1:hi]
            if " - " in head:
                want = "".join(comp[ch] for ch in reversed(want))
            assert seq == want


def test_sp_score_and_refinement():
    """the sum-of-pairs objective against a plain restatement (pair by pair, column by column), and the
    refined interval alignment: never a lower objective than the progressive one, rows still spell the sequences."""
    rng = np.random.default_rng(31)
    anc = rng.integers(0, 4, 120, dtype=np.uint8)
    sc = O.default_scoring()
    mat = [[sc.matrix[i][j] for j in range(4)] for i in range(4)]
    for trial in range(6):
        seqs = [np.ascontiguousarray(synth.mutate(anc, 0.2, rng, indel_frac=0.4)) for _ in range(int(rng.integers(2, 6)))]
        if trial == 3:
            seqs[1] = seqs[1][:0]                               # an absent sequence takes part in no pair
        cols, _ = O.align_interval(seqs)
        want = 0
        for a in range(len(seqs)):
            for b in range(a + 1, len(seqs)):
                if not len(seqs[a]) or not len(seqs[b]):
                    continue
                pa = pb = 0
                prev = 0
                for m in cols.tolist():
                    ha, hb = m >> a & 1, m >> b & 1
                    if ha and hb:
                        want += mat[seqs[a][pa]][seqs[b][pb]]; prev = 0
                    elif ha:
                        want += sc.gap_extend if prev == 1 else sc.gap_open; prev = 1
                    elif hb:
                        want += sc.gap_extend if prev == 2 else sc.gap_open; prev = 2
                    pa += ha; pb += hb
        assert O.sp_score_cols(seqs, cols) == want
        base = want
        for rounds in (1, 2, 7):
            rc, rs, cells = O.align_interval_refined(seqs, rounds)
            assert O.sp_score_cols(seqs, rc) >= base
            for g, sq in enumerate(seqs):                       # every sequence still runs through the columns once
                assert int(((rc >> g) & 1).sum()) == len(sq)
            assert not (rc == 0).any()
        k = sum(1 for sq in seqs if len(sq))
        if k < 3:
            assert np.array_equal(O.align_interval_refined(seqs, 3)[0], cols)


def test_breakpoint_counts_of_known_rearrangements():
    """identical genomes have no breakpoint; one inversion in the middle breaks two adjacencies; a
    translocated block breaks three."""
    rng = np.random.default_rng(41)
    a = rng.integers(0, 4, 12000, dtype=np.uint8)
    pat = O.get_seed(11, 0)
    cut = lambda x, div: np.ascontiguousarray(synth.mutate(x, div, np.random.default_rng(7), indel_frac=0.0))
    same = cut(a, 0.02)                                         # substitutions only: many matches, all collinear
    assert O.breakpoint_counts([a, same], pat, 22)[0, 1] == 0
    inv = same.copy(); inv[4000:8000] = synth.revcomp(inv[4000:8000])
    assert O.breakpoint_counts([a, inv], pat, 22)[0, 1] == 2
    moved = np.concatenate([same[:2000], same[6000:9000], same[2000:6000], same[9000:]])
    bp = O.breakpoint_counts([a, moved, inv], pat, 22)
    assert bp[0, 1] == 3 and bp[0, 2] == 2 and np.array_equal(bp, bp.T) and not bp.diagonal().any()
    assert O.breakpoint_counts([a, inv], pat, 10 ** 6)[0, 1] == 0          # no match passes the floor


def test_homology_pass_restated():
    """orc_homology_apply against a plain restatement (explicit Viterbi tables per pair, then the split), on
    hand-made intervals with a reverse strand and an absent genome; identical rows stay, a block of mismatches is taken apart
    once it outweighs the two transitions."""
    rng = np.random.default_rng(17)
    h = O.hmm_params(pgh=1e-2, pgu=1e-2)                        # cheap transitions: short stretches already flip
    assert (h.match, h.mismatch, h.gap) == (1030, -916, -500) and h.go_homologous == h.go_unrelated == -4605

    def restate(gs, left, right, reverse, col_off, cols):
        N = len(gs); out_cols = []; out_off = [0]; moved = 0
        for iv in range(len(left)):
            cs = [int(x) for x in cols[col_off[iv]:col_off[iv + 1]]]
            keep = [0] * len(cs)
            for a in range(N):
                for b in range(a + 1, N):
                    if not left[iv][a] or not left[iv][b]:
                        continue
                    ka = kb = 0; vh, vu = -(1 << 60), 0; tb = []
                    for c, m in enumerate(cs):
                        ha, hb = m >> a & 1, m >> b & 1
                        if not (ha or hb):
                            tb.append(None); continue
                        s = h.gap
                        if ha and hb:
                            pa = right[iv][a] - ka if reverse[iv][a] else left[iv][a] + ka
                            pb = right[iv][b] - kb if reverse[iv][b] else left[iv][b] + kb
                            xa = int(gs[a][pa - 1]); xb = int(gs[b][pb - 1])
                            xa = 3 - xa if reverse[iv][a] else xa; xb = 3 - xb if reverse[iv][b] else xb
                            s = h.match if xa == xb else h.mismatch
                        ph, pu = vh >= vu + h.go_homologous, vu >= vh + h.go_unrelated
                        vh, vu = max(s + (vh if ph else vu + h.go_homologous), -(1 << 60)), max(vu if pu else vh + h.go_unrelated, -(1 << 60))
                        tb.append((ph, pu)); ka += ha; kb += hb
                    st = vh >= vu
                    for c in range(len(cs) - 1, -1, -1):
                        if tb[c] is None:
                            continue
                        if st and cs[c] >> a & 1 and cs[c] >> b & 1:
                            keep[c] |= 1 << a | 1 << b
                        st = tb[c][0] if st else not tb[c][1]
            for c, m in enumerate(cs):
                k = keep[c] & m
                if k:
                    out_cols.append(k)
                for g in range(N):
                    if (m & ~k) >> g & 1:
                        out_cols.append(1 << g); moved += bool(m & (m - 1))
            out_off.append(len(out_cols))
        return np.array(out_off, np.int64), np.array(out_cols, np.uint32), moved

    anc = rng.integers(0, 4, 400, dtype=np.uint8)
    g0 = anc.copy(); g1 = anc.copy(); g1[150:230] = (g1[150:230] + 1 + rng.integers(0, 3, 80)) % 4       # 80 mismatching columns in the middle
    g2 = synth.revcomp(anc[100:300])
    gs = [g0, g1.astype(np.uint8), np.ascontiguousarray(g2), rng.integers(0, 4, 50, dtype=np.uint8)]
    left = np.array([[1, 1, 0, 0], [101, 101, 1, 0]], np.int64); right = np.array([[100, 100, 0, 0], [400, 400, 200, 0]], np.int64)
    reverse = np.array([[0, 0, 0, 0], [0, 0, 1, 0]], np.int8)
    cols = np.concatenate([np.full(100, 0b011, np.uint32), np.full(200, 0b111, np.uint32), np.full(100, 0b011, np.uint32)])
    col_off = np.array([0, 100, 400], np.int64)
    off, oc, moved = O.homology_apply(gs, left, right, reverse, col_off, cols, h)
    eoff, ec, emoved = restate(gs, left, right, reverse, col_off, cols)
    assert np.array_equal(off, eoff) and np.array_equal(oc, ec) and moved == emoved
    assert moved > 0 and np.array_equal(oc[:100], cols[:100])                 # the clean interval stays; genome 1 leaves its mismatching stretch
    assert int(((oc >> 1) & 1).sum()) == 400 and int(((oc >> 2) & 1).sum()) == 200
    same = [g0, g0.copy()]
    off, oc, moved = O.homology_apply(same, left[:1, :2] * 0 + [[1, 1]], np.array([[400, 400]]), np.zeros((1, 2), np.int8), np.array([0, 400]), np.full(400, 3, np.uint32), h)
    assert moved == 0 and np.array_equal(oc, np.full(400, 3, np.uint32))


def test_workload_generator_is_xoshiro256starstar():
    """SURVEY 8(d) names xoshiro256** seeded through splitmix64 for the synthetic workloads.  Known answers of the two published generators
    (xoshiro256** from the state {1, 2, 3, 4}; splitmix64 from the seed 1234567 -- the vectors the reference implementations' test suites
    carry), and the C stream of mauvealigner_amd/csrc/synth_rng.c against a restatement of the recurrence written out here."""
    M = (1 << 64) - 1
    x = synth.Xoshiro(0)
    x.s = np.array([1, 2, 3, 4], dtype=np.uint64)
    assert x.raw(10).tolist() == [11520, 0, 1509978240, 1215971899390074240, 1216172134540287360, 607988272756665600,
                                  16172922978634559625, 8476171486693032832, 10595114339597558777, 2904607092377533576]
    st, out = 1234567, []
    for _ in range(5):
        st, v = synth._splitmix64(st)
        out.append(v)
    assert out == [6457827717110365317, 3203168211198807973, 9817491932198370423, 4593380528125082431, 16408922859458223821]

    def rotl(v, k):
        return ((v << k) | (v >> (64 - k))) & M
    g = synth.Xoshiro(3, 7)                               # the way a config seeds its streams: splitmix64 over the keys, four more for the state
    s = [int(v) for v in g.s]
    want = []
    for _ in range(1000):
        want.append((rotl((s[1] * 5) & M, 7) * 9) & M)
        t = (s[1] << 17) & M
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45)
    assert g.raw(1000).tolist() == want
    # the draws synth.py makes from the stream (module docstring): base codes are the top two bits, integers the multiply-shift of the top 32
    g2 = synth.Xoshiro(3, 7)
    assert g2.integers(0, 4, 1000, dtype=np.uint8).tolist() == [w >> 62 for w in want]
    g3 = synth.Xoshiro(3, 7)
    assert g3.integers(10, 1010, 1000).tolist() == [10 + (((w >> 32) * 1000) >> 32) for w in want]
