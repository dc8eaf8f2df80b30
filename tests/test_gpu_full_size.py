"""Every BASELINE.json config at FULL size through the HIP path (C-ABI), on the GPU box.

C2 3 x 5 Mbp lives in test_gpu_align.py::test_full_size_c2_properties (and bench.py compares it with the oracle).
Here: C3 (5 x 5 Mbp, 50 inversions) and C4 (8 x 2 Mbp, progressive path) bit-exact against the oracle at full
size, C5 (2 x 100 Mbp, 64-bit keys, default weight 19) through size-independent properties at full size and
bit-exact against the oracle at the scale the oracle finishes in seconds (still the 64-bit-key path).
"""
import numpy as np
import pytest

from mauvealigner_amd import synth
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu

KEYS = ("anchor_start", "anchor_length", "left", "right", "reverse", "col_off", "cols", "dp_score")


@pytest.fixture(scope="module")
def ctx():
    from mauvealigner_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def check_partition(gs, r):
    """every base of every genome lies in exactly one interval and every interval row holds exactly its bases"""
    cols, off = r["cols"], r["col_off"]
    N = len(gs)
    # residues per (interval, genome) from the column masks, in one pass per genome
    for g in range(N):
        bit = ((cols >> np.uint32(g)) & np.uint32(1)).astype(np.int64)
        csum = np.concatenate([[0], np.cumsum(bit)])
        n_res = csum[off[1:]] - csum[off[:-1]]
        le, re = r["left"][:, g], r["right"][:, g]
        present = le != 0
        assert np.all(n_res[~present] == 0)
        assert np.array_equal(n_res[present], (re - le + 1)[present])
        cover = np.zeros(len(gs[g]) + 1, np.int64)
        np.add.at(cover, le[present] - 1, 1)
        np.add.at(cover, re[present], -1)
        assert np.all(np.cumsum(cover)[:-1] == 1)


def test_c3_full_size_equals_oracle(ctx):
    """BASELINE config 3: 5 x 5 Mbp, ~50 inversions, weight 15 -- the workload the 50 Mbp/s target is quoted on."""
    from mauvealigner_amd import _lib
    gs = synth.make_config("C3", scale=1.0)
    ctx.set_genomes(gs)
    r = ctx.align(_lib.default_params(seed_weight=15))
    full = O.align(gs, O.default_params(seed_weight=15))
    e = full["aln"]
    for k in KEYS:
        assert np.array_equal(r[k], e[k]), k
    assert r["n_lcb"] == full["lcbs"]["n_lcb"] and 50 <= r["n_lcb"] <= 101      # 50 inversions -> at most 101 LCBs
    assert r["n_gap_dp"] == e["n_gap_dp"] and r["n_dp_cells"] == e["n_dp_cells"]
    check_partition(gs, r)
    r2 = ctx.align(_lib.default_params(seed_weight=15))
    for k in KEYS:
        assert np.array_equal(r[k], r2[k]), k
    # the backbone stage on the columns this alignment left in HBM, at full size against the oracle
    import time
    ctx.align(_lib.default_params(seed_weight=15), fetch=False)
    ctx.backbone(island_gap=20)                               # first call sizes its buffers
    t0 = time.perf_counter()
    b = ctx.backbone(island_gap=20)
    dt = time.perf_counter() - t0
    eb = O.backbone(e["left"], e["right"], e["reverse"], e["col_off"], e["cols"], island_gap=20)
    for k in ("seg_iv", "seg_col", "seg_len", "seg_mask", "seg_left", "seg_right", "islands"):
        assert np.array_equal(b[k], eb[k]), k
    assert len(b["seg_iv"]) >= r["n_lcb"] and len(b["islands"]) > 10
    print("backbone at C3: %d segments, %d islands, %.2f ms" % (len(b["seg_iv"]), len(b["islands"]), dt * 1e3))


def test_c4_full_size_progressive_equals_oracle(ctx):
    """BASELINE config 4: 8 x 2 Mbp on a balanced tree, progressive path (guide tree + guide-tree anchoring), at the progressiveMauve
    call site's defaults (progressiveMauve.cpp:578-579,624-637: extant sum-of-pairs LCB scoring, weight scaling on with both scales
    0.5, refinement on) -- mauve_default_progressive_params."""
    from mauvealigner_amd import _lib
    gs = synth.make_config("C4", scale=1.0)
    ctx.set_genomes(gs)
    p = _lib.default_progressive_params()
    assert (p.lcb_scoring, p.weight_scaling, p.conservation_scale_ppm, p.bp_dist_scale_ppm, p.refine_rounds) == (1, 1, 500000, 500000, 2)
    r = ctx.progressive_align(p)
    e = O.progressive_align(gs, O.default_progressive_params())
    assert np.array_equal(r["dist"], e["dist"])
    assert np.array_equal(r["tree"][0], e["tree"][0]) and np.array_equal(r["tree"][1], e["tree"][1])
    a = e["aln"]
    for k in ("left", "right", "reverse", "col_off", "cols", "dp_score"):
        assert np.array_equal(r[k], a[k]), k
    assert r["n_gap_dp"] == a["n_gap_dp"] and r["n_dp_cells"] == a["n_dp_cells"]
    check_partition(gs, r)
    N = len(gs)
    left, right = r["tree"]
    cherries = sorted(tuple(sorted((int(left[k]), int(right[k])))) for k in range(N, 2 * N - 1) if left[k] < N and right[k] < N)
    assert cherries == [(0, 1), (2, 3), (4, 5), (6, 7)]


def test_c5_scaled_equals_oracle_64bit_keys(ctx):
    """BASELINE config 5 at 1/10 (2 x 11 Mbp, default weight 17 -> 34-bit mers: the 64-bit-key path), long
    insertions and hyper-divergent segments included, bit-exact against the oracle."""
    from mauvealigner_amd import _lib
    gs = synth.make_config("C5", scale=0.1)
    assert 2 * _lib.default_seed_weight(sum(len(g) for g in gs) // 2) > 32
    ctx.set_genomes(gs)
    r = ctx.align(_lib.default_params())
    e = O.align(gs, O.default_params())["aln"]
    for k in KEYS:
        assert np.array_equal(r[k], e[k]), k
    assert r["n_dp_cells"] == e["n_dp_cells"]
    check_partition(gs, r)


def test_c5_full_size_properties(ctx):
    """BASELINE config 5 at full size (2 x ~111 Mbp, weight 19, 64-bit keys): partition, determinism, and the
    bulk of both genomes aligned in few LCBs."""
    from mauvealigner_amd import _lib
    gs = synth.make_config("C5", scale=1.0)
    assert _lib.default_seed_weight(sum(len(g) for g in gs) // 2) == 19
    ctx.set_genomes(gs)
    p = _lib.default_params()
    r = ctx.align(p)
    check_partition(gs, r)
    both = np.count_nonzero(r["cols"] == 3)
    assert both > 0.8 * 100_000_000                     # the shared ancestor is 100 Mbp at 3 % divergence
    assert r["n_lcb"] >= 1 and r["n_gap_dp"] > 10_000
    sizes2 = ctx.align(p)
    for k in KEYS:
        assert np.array_equal(r[k], sizes2[k]), k


def test_banded_long_intervals(ctx):
    """Long gaps through the banded DP at lengths the full DP has no traceback memory for: a 60 kb pair
    bit-exact against the oracle's banded DP, a 150 kb pair through properties (every base spelled once and in
    order; the score is at least that of the gap-free diagonal walk plus the end gap, which lies inside the band)."""
    import time
    rng = np.random.default_rng(31)

    def pair(L):
        a = rng.integers(0, 4, L, dtype=np.uint8)
        b = synth.mutate(a, 0.08, rng, indel_frac=0.2)
        return [a, b]
    iv = pair(60000)
    t0 = time.time()
    cols, score = ctx.dp_batch([iv], band_from=10000)
    t1 = time.time()
    ec, es = O.align_interval(iv, banded=True)
    assert int(score[0]) == es and np.array_equal(cols[0], ec)
    big = pair(150000)
    t2 = time.time()
    cols, score = ctx.dp_batch([big], band_from=10000)
    t3 = time.time()
    c = cols[0]
    for k, s in enumerate(big):
        assert int(((c >> np.uint32(k)) & np.uint32(1)).sum()) == len(s)
    assert np.all(c != 0)
    sc = O.default_scoring()
    m, n = len(big[0]), len(big[1])
    k = min(m, n)
    mat = np.array([[sc.matrix[i][j] for j in range(4)] for i in range(4)], dtype=np.int64)
    diag = int(mat[big[0][:k], big[1][:k]].sum()) + (sc.gap_open + (abs(m - n) - 1) * sc.gap_extend if m != n else 0)
    assert int(score[0]) >= diag
    print("banded 60k: %.3f s, 150k: %.3f s" % (t1 - t0, t3 - t2))
