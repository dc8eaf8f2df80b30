"""GPU parity tests of the batched DP kernel and of the whole path (mauve_align) against the CPU oracle and the
golden fixtures.  Integer scores and alignment columns must match bit-exactly."""
import os

import ctypes as C
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


def _rand_interval(rng, nseq, L, div, empty_prob=0.15):
    base = rng.integers(0, 4, L, dtype=np.uint8)
    out = []
    for _ in range(nseq):
        if rng.random() < empty_prob:
            out.append(np.zeros(0, np.uint8))
        else:
            out.append(synth.mutate(base, div, rng, indel_frac=0.3))
    return out


def _check_dp(ctx, intervals):
    cols, score = ctx.dp_batch(intervals)
    for iv, c, s in zip(intervals, cols, score):
        ec, es = O.align_interval(iv)
        assert len(c) == len(ec)
        assert np.array_equal(c, ec)
        assert int(s) == es


def test_dp_small_intervals(ctx):
    rng = np.random.default_rng(1)
    for nseq in (2, 3, 5):
        ivs = [_rand_interval(rng, nseq, int(rng.integers(1, 40)), 0.15) for _ in range(300)]
        _check_dp(ctx, ivs)


def test_dp_known_answers(ctx):
    A, C_, G, T = 0, 1, 2, 3
    s = np.array([A, C_, G, T, A, C_, G, T], dtype=np.uint8)
    cols, score = ctx.dp_batch([[s, s.copy()], [s, np.delete(s, 3)], [s, np.delete(s, [3, 4])],
                                [np.zeros(0, np.uint8), s], [s, np.zeros(0, np.uint8)],
                                [np.array([A, A, A], np.uint8), np.array([A, G, A], np.uint8)]])
    assert score.tolist() == [764, 764 - 91 - 400, 764 - 182 - 430, 0, 0, 91 - 31 + 91]
    assert cols[0].tolist() == [3] * 8
    assert sorted(cols[1].tolist()) == [1] + [3] * 7
    assert cols[3].tolist() == [2] * 8 and cols[4].tolist() == [1] * 8
    assert cols[5].tolist() == [3, 3, 3]


def test_dp_multi_stripe_and_ragged(ctx):
    rng = np.random.default_rng(2)
    ivs, ivs4 = [], []
    for L in (63, 64, 65, 128, 129, 200, 500):
        ivs.append(_rand_interval(rng, 2, L, 0.1, empty_prob=0.0))
        ivs4.append(_rand_interval(rng, 4, L, 0.2, empty_prob=0.1))
    _check_dp(ctx, ivs4)
    # very unequal lengths, single bases, all-empty interval
    ivs.append([rng.integers(0, 4, 300, dtype=np.uint8), rng.integers(0, 4, 3, dtype=np.uint8)])
    ivs.append([rng.integers(0, 4, 2, dtype=np.uint8), rng.integers(0, 4, 400, dtype=np.uint8)])
    ivs.append([np.array([1], np.uint8), np.array([2], np.uint8)])
    _check_dp(ctx, ivs)
    ivs3 = [[np.zeros(0, np.uint8)] * 3, _rand_interval(rng, 3, 20, 0.1, 0.0), [np.zeros(0, np.uint8), np.array([3], np.uint8), np.zeros(0, np.uint8)]]
    _check_dp(ctx, ivs3)


def test_dp_many_sequences(ctx):
    rng = np.random.default_rng(3)
    ivs = [_rand_interval(rng, 12, int(rng.integers(5, 90)), 0.1) for _ in range(40)]
    _check_dp(ctx, ivs)


def test_dp_long_interval(ctx):
    rng = np.random.default_rng(4)
    _check_dp(ctx, [_rand_interval(rng, 2, 3000, 0.2, 0.0)])
    _check_dp(ctx, [_rand_interval(rng, 3, 1500, 0.25, 0.0), _rand_interval(rng, 3, 700, 0.1, 0.0)])


def test_dp_workgroup_pipeline(ctx):
    """Tail intervals run as a 16-wave stripe pipeline (dp_step_big); same bytes as the one-wave path and the
    oracle.  Shapes around the pipeline's edges: fewer stripes than waves, more stripes than waves, n just above
    the 256-column threshold (waves wait for their predecessor to finish), a short last stripe, and big + small
    intervals in one launch (both kernels run side by side)."""
    rng = np.random.default_rng(7)
    def pair(m, n, related=True):
        a = rng.integers(0, 4, m, dtype=np.uint8)
        if related and n <= m:
            b = synth.mutate(a, 0.15, rng, indel_frac=0.3)[:n]
            if len(b) < n:
                b = np.concatenate([b, rng.integers(0, 4, n - len(b), dtype=np.uint8)])
        else:
            b = rng.integers(0, 4, n, dtype=np.uint8)
        return [a, b]
    ivs = [pair(2500, 2300), pair(129, 256), pair(1100, 257), pair(300, 4000, related=False), pair(64 * 17 + 1, 700),
           pair(64 * 40, 300), pair(4000, 4000, related=False)]
    ivs += [_rand_interval(rng, 2, int(rng.integers(1, 60)), 0.1, 0.0) for _ in range(200)]
    _check_dp(ctx, ivs)
    # three sequences: the second step's profile is the merged first two
    _check_dp(ctx, [_rand_interval(rng, 3, 1200, 0.2, 0.0)] + [_rand_interval(rng, 3, 30, 0.1, 0.0) for _ in range(50)])


def test_dp_subwave_groups(ctx):
    """Small intervals share a wave (four per wave when every profile fits 16 rows, two when it fits 32): boundary
    sizes of the classes, unequal neighbours in one wave, empty members (a different first sequence per group), a
    count that does not fill the last wave, and long second sequences up to the LDS slice."""
    rng = np.random.default_rng(11)
    def iv(lens, div=0.2):
        base = rng.integers(0, 4, max(max(lens), 1), dtype=np.uint8)
        out = []
        for L in lens:
            if L == 0:
                out.append(np.zeros(0, np.uint8)); continue
            x = synth.mutate(base, div, rng, indel_frac=0.3)[:L]
            if len(x) < L:
                x = np.concatenate([x, rng.integers(0, 4, L - len(x), dtype=np.uint8)])
            out.append(x)
        return out
    two = [iv(l) for l in ([16, 16], [17, 3], [16, 176], [15, 177], [32, 160], [33, 100], [1, 1], [2, 190], [190, 2],
                           [31, 161], [8, 8], [16, 1], [1, 16], [5, 0], [0, 7], [12, 13], [3, 150], [9, 9], [4, 4])]
    _check_dp(ctx, two)
    three = [iv(l) for l in ([5, 6, 4], [8, 8, 8], [9, 8, 30], [0, 7, 9], [7, 0, 9], [7, 9, 0], [0, 0, 5], [16, 0, 100],
                             [10, 7, 170], [16, 16, 16], [2, 2, 2], [1, 40, 1], [20, 12, 150], [6, 6, 180])]
    _check_dp(ctx, three)
    many = [iv([int(rng.integers(0, 12)) for _ in range(5)], div=0.1) for _ in range(403)]
    _check_dp(ctx, many)


ROUNDS_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from mauvealigner_amd import _lib, synth
from oracle import pyoracle as O
rng = np.random.default_rng(8)
ivs = []
for k in range(40):                      # long, ragged intervals: 40 x (300 .. 900)^2 cells of traceback
    base = rng.integers(0, 4, int(rng.integers(300, 900)), dtype=np.uint8)
    ivs.append([synth.mutate(base, 0.1, rng, indel_frac=0.3) for _ in range(3)])
ctx = _lib.Context(0)
cols, score = ctx.dp_batch(ivs)
for iv, c, s in zip(ivs, cols, score):
    ec, es = O.align_interval(iv)
    assert np.array_equal(c, ec) and int(s) == es
gs = synth.make_config("C3", scale=0.02)
ctx.set_genomes(gs)
r = ctx.align(_lib.default_params())
e = O.align(gs, O.default_params())["aln"]
for k in ("cols", "col_off", "dp_score", "anchor_start"):
    assert np.array_equal(r[k], e[k]), k
print("OK")
"""


def test_dp_rounds_under_a_traceback_budget():
    """A batch whose traceback exceeds the budget runs in rounds over one buffer (both front ends: mauve_dp_batch and
    mauve_align), with the same results; one interval beyond the budget is refused with MAUVE_ERR_LIMIT."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MAUVE_DP_TB_BUDGET=str(3 << 20), MAUVE_TRACE="1")
    r = subprocess.run([sys.executable, "-c", ROUNDS_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr[-3000:]
    rounds = [int(l.split(" round(s)")[0].rsplit(" ", 1)[1]) for l in r.stderr.splitlines() if " round(s)" in l]
    assert rounds and max(rounds) > 1, r.stderr[-2000:]
    env["MAUVE_DP_TB_BUDGET"] = str(1 << 16)
    r = subprocess.run([sys.executable, "-c", ROUNDS_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "(-4)" in r.stderr and "MAUVE_DP_TB_BUDGET" in r.stderr


TAIL_SCRIPT = r"""
import sys, json, hashlib, numpy as np
sys.path.insert(0, %(root)r)
from mauvealigner_amd import _lib, synth
from oracle import pyoracle as O
ctx = _lib.Context(0)
out = {}
cases = [("C3", 0.04, dict(recursive=0)), ("C3", 0.04, dict(recursive=0, add_unaligned=0)), ("C2", 0.03, dict(recursive=0, gapped=0)),
         ("C3", 0.04, dict()), ("C1", 0.3, dict(recursive=0, max_gapped_len=40))]
def long_gaps():
    rng = np.random.default_rng(5)
    base = synth.random_genome(30000, rng)
    gs = []
    for g in range(3):
        x = synth.mutate(base, 0.04, rng, indel_frac=0.1)
        p = 9000 + 3000 * g
        gs.append(np.concatenate([x[:p], rng.integers(0, 4, 1500, dtype=np.uint8), x[p + 1200:]]).astype(np.uint8))
    return gs
cases.append(("long gaps", 0, dict()))                 # the recursion has work: the chains go back to the host
for cfg, scale, kw in cases:
    gs = long_gaps() if cfg == "long gaps" else synth.make_config(cfg, scale=scale)
    ctx.set_genomes(gs)
    names = ["g%%d" %% i for i in range(len(gs))]
    n0 = ctx.align(_lib.default_params(**kw), fetch=False)          # sizes alone, nothing fetched
    r = ctx.align(_lib.default_params(**kw), names=names, want_xmfa=True)
    e = O.align(gs, O.default_params(**kw), names=names, want_xmfa=True)
    a = e["aln"]
    assert n0["n_cols"] == r["n_cols"] == len(a["cols"]) and n0["n_iv"] == a["n_iv"] and n0["n_anchor"] == len(a["anchor_length"])
    for k in ("anchor_length", "anchor_start", "anchor_lcb", "left", "right", "reverse", "col_off", "cols", "dp_score"):
        assert np.array_equal(r[k], a[k]), (cfg, kw, k)
    assert np.array_equal(r["lcb_weight"], e["lcbs"]["weight"]) and r["n_dp_cells"] == a["n_dp_cells"] and r["xmfa"] == e["xmfa"]
print("OK")
"""


def test_device_tail_equals_host_tail_and_oracle():
    """mauve_align keeps the chains, the DP results and the assembly on the device when no gap needs the recursion
    (chain_order_device / assemble_dev.hip): same bytes as the oracle, with and without islands, without DP, with gaps
    emitted unaligned, and sizes available before anything is fetched.  The trace shows which tail ran; MAUVE_HOST_TAIL
    forces the host tail on the same inputs."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MAUVE_TRACE="1", MAUVE_CANON_DEVICE_MIN="1")      # small lists, too, are sorted and chained on the device
    r = subprocess.run([sys.executable, "-c", TAIL_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr[-3000:]
    stay = [l for l in r.stderr.splitlines() if "stay there" in l]
    back = [l for l in r.stderr.splitlines() if "copy back" in l or "anchors to the host" in l]
    assert len(stay) >= 8 and back, r.stderr[-2000:]          # recursion-free cases stay; the one with long gaps goes back
    env["MAUVE_HOST_TAIL"] = "1"
    r = subprocess.run([sys.executable, "-c", TAIL_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr[-3000:]
    assert not [l for l in r.stderr.splitlines() if "stay there" in l]


CLASS_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from mauvealigner_amd import _lib, synth
from oracle import pyoracle as O
rng = np.random.default_rng(12)
ctx = _lib.Context(0)
for nseq in (2, 3, 5):
    ivs = []
    for k in range(400):
        base = rng.integers(0, 4, int(rng.integers(1, 60)), dtype=np.uint8)
        ivs.append([synth.mutate(base, 0.2, rng, indel_frac=0.5) if rng.random() > 0.1 else np.zeros(0, np.uint8) for _ in range(nseq)])
    cols, score = ctx.dp_batch(ivs)
    for iv, c, s in zip(ivs, cols, score):
        ec, es = O.align_interval(iv)
        assert np.array_equal(c, ec) and int(s) == es
gs = synth.make_config("C3", scale=0.03)
ctx.set_genomes(gs)
r = ctx.align(_lib.default_params())
e = O.align(gs, O.default_params())["aln"]
for k in ("cols", "col_off", "dp_score"):
    assert np.array_equal(r[k], e[k]), k
assert r["n_dp_cells"] == e["n_dp_cells"]
print("OK")
"""


def test_dp_class_estimate_and_fallback():
    """Intervals are put into the sub-wave classes by an estimate of their profile lengths; a group whose profile outgrows
    its rows hands the interval to the one-wave path.  MAUVE_DP_CLASS=wild underestimates on purpose (half the longest
    sequence), so most groups fall back; =bound is the safe worst-case classing.  Same results in all three."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("wild", "bound", "estimate"):
        env = dict(os.environ, MAUVE_DP_CLASS=mode)
        r = subprocess.run([sys.executable, "-c", CLASS_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), mode + "\n" + r.stdout + r.stderr[-3000:]


def _check_dp_banded(ctx, intervals, band_from):
    widths = sorted({len(iv) for iv in intervals})
    if len(widths) > 1:                                        # one launch per number of sequences
        for w in widths:
            _check_dp_banded(ctx, [iv for iv in intervals if len(iv) == w], band_from)
        return
    cols, score = ctx.dp_batch(intervals, band_from=band_from)
    for iv, c, s in zip(intervals, cols, score):
        banded = max(len(x) for x in iv) > band_from
        ec, es = O.align_interval(iv, banded=banded)
        assert len(c) == len(ec)
        assert np.array_equal(c, ec)
        assert int(s) == es


def _long_gap_pair(rng, L, shift, div=0.05):
    """Two related sequences whose optimal alignment leaves the band: `shift` extra bases early in b, `shift` bases
    dropped later (the lengths stay close, so the band does not widen with them)."""
    a = rng.integers(0, 4, L, dtype=np.uint8)
    b = a.copy()
    mut = rng.random(L) < div
    b[mut] = (b[mut] + 1) % 4
    b = np.concatenate([b[:L // 6], rng.integers(0, 4, shift, dtype=np.uint8), b[L // 6:L // 2], b[L // 2 + shift:]])
    return [a, b]


def test_dp_banded(ctx):
    """Banded steps against the oracle's banded DP: a path that the band cuts (the result differs from the
    full DP, so the band is what is being tested), unequal lengths, three sequences, the smallest shapes with every
    interval banded, and banded + full intervals in one launch."""
    rng = np.random.default_rng(11)
    cut = _long_gap_pair(rng, 6000, 900)
    full = O.align_interval(cut)
    band = O.align_interval(cut, banded=True)
    assert band[1] < full[1]                                   # the band is live on this input
    _check_dp_banded(ctx, [cut], 1000)
    ivs = [_long_gap_pair(rng, 3000, 500), [rng.integers(0, 4, 500, dtype=np.uint8), rng.integers(0, 4, 5000, dtype=np.uint8)],
           [rng.integers(0, 4, 5000, dtype=np.uint8), rng.integers(0, 4, 300, dtype=np.uint8)],
           _rand_interval(rng, 3, 2500, 0.2, 0.0), _rand_interval(rng, 4, 1300, 0.15, 0.2),
           _long_gap_pair(rng, 64 * 30, 400) + [rng.integers(0, 4, 700, dtype=np.uint8)]]
    ivs += [_rand_interval(rng, 2, int(rng.integers(1, 400)), 0.1, 0.1) for _ in range(60)]
    _check_dp_banded(ctx, ivs, 1000)                           # long ones banded, the rest in full, one launch
    _check_dp_banded(ctx, ivs[1:], 0)                          # every interval banded, down to single bases
    edge = [[np.array([1], np.uint8), np.array([2], np.uint8)], [np.zeros(0, np.uint8), rng.integers(0, 4, 9, dtype=np.uint8)],
            [rng.integers(0, 4, 64, dtype=np.uint8), rng.integers(0, 4, 65, dtype=np.uint8)],
            [rng.integers(0, 4, 129, dtype=np.uint8), rng.integers(0, 4, 1, dtype=np.uint8)]]
    _check_dp_banded(ctx, edge, 0)


def _genomes_with_long_gaps(seed=5):
    rng = np.random.default_rng(seed)
    base = synth.random_genome(40000, rng)
    gs = []
    for g in range(3):
        x = synth.mutate(base, 0.04, rng, indel_frac=0.1)
        # a divergent stretch (no anchors inside) and a genome-specific insertion
        p = 9000 + 3000 * g
        x = np.concatenate([x[:p], rng.integers(0, 4, 1500 + 400 * g, dtype=np.uint8), x[p + 1200:]])
        gs.append(x.astype(np.uint8))
    return gs


def test_align_banded_long_gaps(ctx):
    """max_banded_len: inter-anchor intervals above max_gapped_len are aligned by the banded DP instead of staying
    unaligned -- whole path against the oracle, device front end and the host one (sharded phases)."""
    from mauvealigner_amd import _lib, parallel
    gs = _genomes_with_long_gaps()
    _same_align(ctx, gs, max_gapped_len=400, max_banded_len=20000)
    # without recursive anchoring the three 1.5 - 2.3 kb stretches stay whole: banded intervals well above the limit
    off = _same_align(ctx, gs, max_gapped_len=400, recursive=0)
    on = _same_align(ctx, gs, max_gapped_len=400, max_banded_len=20000, recursive=0)
    assert on["n_gap_dp"] > off["n_gap_dp"] and on["n_cols"] < off["n_cols"] and on["n_dp_cells"] > 10 * off["n_dp_cells"]
    names = ["g%d" % i for i in range(len(gs))]
    sh = parallel.align_sharded(ctx, _lib.default_params(max_gapped_len=400, max_banded_len=20000, recursive=0), None, names=names, want_xmfa=True)
    for k in ("cols", "col_off", "left", "right", "reverse", "dp_score"):
        assert np.array_equal(on[k], sh[k]), k
    assert on["xmfa"] == sh["xmfa"] and on["n_dp_cells"] == sh["n_dp_cells"]


def test_sp_lcb_scoring(ctx):
    """lcb_scoring = sum-of-pairs (ProgressiveAligner::setLcbScoringScheme): the device's anchor scores equal
    the oracle's, and mauve_align / mauve_progressive_align with score-weighted LCBs equal the oracle's results."""
    from mauvealigner_amd import _lib
    gs = synth.make_config("C3", scale=0.03)
    ctx.set_genomes(gs)
    r = ctx.align(_lib.default_params())
    sc = ctx.match_sp_scores(r["anchor_length"], r["anchor_start"])
    assert np.array_equal(sc, O.match_sp_scores(gs, r["anchor_length"], r["anchor_start"]))
    assert sc.min() > 0
    _same_align(ctx, gs, lcb_scoring=1)
    _same_align(ctx, gs, lcb_scoring=1, lcb_weight=200000)        # a breakpoint penalty as a score
    gs4 = synth.make_config("C4", scale=0.01)
    _same_progressive(ctx, gs4, lcb_scoring=1)
    # score-weighted LCBs are not extended (the extension rule counts columns): the flag changes nothing
    a = ctx.align(_lib.default_params(lcb_scoring=1, extend_lcbs=1)); b = ctx.align(_lib.default_params(lcb_scoring=1, extend_lcbs=0))
    assert np.array_equal(a["cols"], b["cols"]) and np.array_equal(a["anchor_start"], b["anchor_start"])


def test_lcb_extension(ctx):
    """S10 (lcb_extension): masked re-search of the regions outside every LCB with lighter seeds; bit-exact against
    the oracle, and it only ever adds anchored columns."""
    from mauvealigner_amd import _lib
    for cfg, scale, kw in (("C3", 0.02, {}), ("C4", 0.05, {}), ("C3", 0.03, {"seed_weight": 13, "max_extension_iters": 2}),
                           ("C1", 0.3, {}), ("C3", 0.02, {"mode": 1})):
        gs = synth.make_config(cfg, scale=scale)
        r1 = _same_align(ctx, gs, extend_lcbs=1, **kw)
        r0 = ctx.align(_lib.default_params(extend_lcbs=0, **kw))
        assert int(r1["anchor_length"].sum()) >= int(r0["anchor_length"].sum())
    # genomes that are one LCB end to end: nothing outside, nothing changes
    rng = np.random.default_rng(5)
    g = rng.integers(0, 4, 20000, dtype=np.uint8)
    gs = [g, synth.mutate(g, 0.02, rng)]
    ctx.set_genomes(gs)
    a = ctx.align(_lib.default_params(extend_lcbs=1)); b = ctx.align(_lib.default_params(extend_lcbs=0))
    assert np.array_equal(a["cols"], b["cols"]) and np.array_equal(a["anchor_start"], b["anchor_start"])


def test_golden_progressive(ctx):
    from mauvealigner_amd import _lib
    z = np.load(os.path.join(GOLDEN, "g4x3k_tree.npz"))
    N = int(z["nseq"])
    gs = [z["genome%d" % g] for g in range(N)]
    pat = int(z["pattern"])
    ctx.set_genomes(gs)
    ln, st = ctx.seed_mums(pat, mode=_lib.MODE_PAIRWISE)
    assert np.array_equal(ln, z["pair_length"]) and np.array_equal(st, z["pair_start"])
    r = ctx.progressive_align(_lib.default_params(seed_pattern=pat), names=["g%d" % g for g in range(N)], want_xmfa=True)
    assert np.array_equal(r["tree"][0], z["tree_left"]) and np.array_equal(r["tree"][1], z["tree_right"])
    for k in ("left", "right", "reverse", "col_off", "cols", "dp_score"):
        assert np.array_equal(r[k], z[k]), k
    with open(os.path.join(GOLDEN, "g4x3k_tree.xmfa")) as f:
        assert f.read() == r["xmfa"]


def test_golden_round2(ctx):
    """The round-2 fixture through the C-ABI: seed-family alignment, its backbone and islands, progressive alignment
    along the given tree with scaled node weights."""
    from mauvealigner_amd import _lib
    z = np.load(os.path.join(GOLDEN, "g3x6k_round2.npz"))
    gs = [z["genome%d" % g] for g in range(3)]
    names = ["g%d" % g for g in range(3)]
    ctx.set_genomes(gs)
    r = ctx.align(_lib.default_params(seed_weight=int(z["seed_weight"]), seed_family=1), names=names, want_xmfa=True)
    for k in ("anchor_start", "anchor_length", "left", "right", "reverse", "col_off", "cols"):
        assert np.array_equal(r[k], z["fam_" + k]), k
    with open(os.path.join(GOLDEN, "g3x6k_round2.xmfa")) as f:
        assert f.read() == r["xmfa"]
    bb = ctx.backbone(island_gap=int(z["bb_island_gap"]))
    for k in ("seg_iv", "seg_col", "seg_len", "seg_mask", "seg_left", "seg_right", "islands"):
        assert np.array_equal(bb[k], z["bb_" + k]), k
    assert len(bb["islands"]) >= 1 and len(bb["seg_iv"]) >= 1
    pr = ctx.progressive_align(_lib.default_params(seed_weight=int(z["seed_weight"]), weight_scaling=1, conservation_scale_ppm=500000),
                               tree=(z["tree_left"], z["tree_right"]))
    for k in ("left", "right", "reverse", "col_off", "cols"):
        assert np.array_equal(pr[k], z["prog_" + k]), k


def _same_align(ctx, gs, **kw):
    from mauvealigner_amd import _lib
    ctx.set_genomes(gs)
    names = ["g%d" % i for i in range(len(gs))]
    r = ctx.align(_lib.default_params(**kw), names=names, want_xmfa=True)
    e = O.align(gs, O.default_params(**kw), names=names, want_xmfa=True)
    N = len(gs)
    eml, ems = O.multiplicity_filter(e["mums"][0], e["mums"][1], N)
    assert np.array_equal(r["mum_length"], eml) and np.array_equal(r["mum_start"], ems)
    assert r["n_lcb"] == e["lcbs"]["n_lcb"]
    assert np.array_equal(r["lcb_weight"], e["lcbs"]["weight"])
    a = e["aln"]
    assert np.array_equal(r["anchor_length"], a["anchor_length"])
    assert np.array_equal(r["anchor_start"], a["anchor_start"])
    assert np.array_equal(r["anchor_lcb"], a["anchor_lcb"])
    assert r["n_iv"] == a["n_iv"]
    assert np.array_equal(r["left"], a["left"]) and np.array_equal(r["right"], a["right"])
    assert np.array_equal(r["reverse"], a["reverse"])
    assert np.array_equal(r["col_off"], a["col_off"])
    assert np.array_equal(r["cols"], a["cols"])
    assert np.array_equal(r["dp_score"], a["dp_score"])
    assert r["n_gap_dp"] == a["n_gap_dp"] and r["n_dp_cells"] == a["n_dp_cells"]
    assert r["xmfa"] == e["xmfa"]
    return r


@pytest.mark.parametrize("name", ["g2x2k", "g3x5k_inv", "g5x3k_unique"])
def test_golden_alignment(ctx, name):
    from mauvealigner_amd import _lib
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    N = int(z["nseq"])
    gs = [z["genome%d" % g] for g in range(N)]
    ctx.set_genomes(gs)
    r = ctx.align(_lib.default_params(seed_pattern=int(z["pattern"]), mode=int(z["mode"])),
                  names=["g%d" % g for g in range(N)], want_xmfa=True)
    assert np.array_equal(r["cols"], z["cols"]) and np.array_equal(r["col_off"], z["col_off"])
    assert np.array_equal(r["left"], z["left"]) and np.array_equal(r["right"], z["right"])
    assert np.array_equal(r["dp_score"], z["dp_score"])
    assert np.array_equal(r["lcb_weight"], z["lcb_weight"])
    assert np.array_equal(r["anchor_start"], z["anchor_start"])
    with open(os.path.join(GOLDEN, name + ".xmfa")) as f:
        assert f.read() == r["xmfa"]


@pytest.mark.parametrize("cfg,scale", [("C1", 1.0), ("C2", 0.02), ("C3", 0.03), ("C4", 0.02)])     # C1: BASELINE config 1 at its full size (2 x 200 kbp), XMFA text included
def test_align_equals_oracle(ctx, cfg, scale):
    gs = synth.make_config(cfg, scale=scale)
    _same_align(ctx, gs, recursive=0)
    _same_align(ctx, gs, recursive=1)


def test_align_options(ctx):
    gs = synth.make_config("C3", scale=0.02)
    _same_align(ctx, gs, collinear=1)
    _same_align(ctx, gs, gapped=0)
    _same_align(ctx, gs, add_unaligned=0)
    _same_align(ctx, gs, seed_weight=11, lcb_weight=500)
    _same_align(ctx, gs, mode=1)
    _same_align(ctx, gs, max_gapped_len=30)
    _same_align(ctx, gs, seed_rank=1)


def test_align_recursion_hyperdivergent(ctx):
    rng = np.random.default_rng(21)
    anc = rng.integers(0, 4, 60000, dtype=np.uint8)
    gs = []
    for g in range(3):
        b = synth.mutate(anc, 0.01, rng)
        if g:
            b[10000:14000] = synth.mutate(b[10000:14000], 0.33, rng, indel_frac=0.0)[:4000]
            b[30000:42000] = synth.mutate(b[30000:42000], 0.30, rng, indel_frac=0.0)[:12000]
        gs.append(b)
    gs[2] = gs[2].copy()
    gs[2][25000:50000] = synth.revcomp(gs[2][25000:50000])
    r1 = _same_align(ctx, gs, recursive=1)
    r0 = _same_align(ctx, gs, recursive=0)
    assert r1["n_anchor"] > r0["n_anchor"]


GAPCHAIN_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from mauvealigner_amd import _lib, synth
from oracle import pyoracle as O
ctx = _lib.Context(0)
rng = np.random.default_rng(21)
anc = rng.integers(0, 4, 60000, dtype=np.uint8)
hyper = []
for g in range(3):
    b = synth.mutate(anc, 0.01, rng)
    if g:
        b[10000:14000] = synth.mutate(b[10000:14000], 0.33, rng, indel_frac=0.0)[:4000]
        b[30000:42000] = synth.mutate(b[30000:42000], 0.30, rng, indel_frac=0.0)[:12000]
    hyper.append(b)
sets = [hyper, synth.make_config("C5", scale=0.004), synth.make_config("C3", scale=0.02)]
base = rng.integers(0, 4, 40000, dtype=np.uint8)                   # a repeat family inside a divergent stretch: overlapping matches in a gap
rep = rng.integers(0, 4, 300, dtype=np.uint8)
fam = []
for g in range(3):
    x = synth.mutate(base, 0.02, rng)
    mid = np.concatenate([synth.mutate(rep, 0.03, rng) for _ in range(6)] + [rng.integers(0, 4, 200, dtype=np.uint8)])
    fam.append(np.concatenate([x[:15000], mid, x[15000:]]).astype(np.uint8))
sets.append(fam)
for gs in sets:
    ctx.set_genomes(gs)
    r = ctx.align(_lib.default_params())
    e = O.align(gs, O.default_params())["aln"]
    for k in ("anchor_length", "anchor_start", "anchor_lcb", "cols", "col_off", "dp_score"):
        assert np.array_equal(r[k], e[k]), k
print("OK")
"""


def test_recursion_gaps_chained_on_the_device():
    """The N-way matches of a whole recursion batch are overlap-eliminated and reduced to one collinear chain per gap on
    the device (chain_device_gaps), with the per-gap greedy step on the compact graph; MAUVE_HOST_GAP_CHAIN keeps the
    host loop.  MAUVE_CANON_DEVICE_MIN=1 sends small lists down the device path too."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MAUVE_TRACE="1", MAUVE_CANON_DEVICE_MIN="1")
    r = subprocess.run([sys.executable, "-c", GAPCHAIN_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr[-3000:]
    dev = [l for l in r.stderr.splitlines() if "per-gap chaining" in l and "(device" in l]
    assert len(dev) >= 4, r.stderr[-3000:]
    # overlap clusters beyond the per-thread limit: the global-memory kernel in the recursion batches, the host chain for the
    # main list -- with the limit lowered to 2 nearly every cluster takes those paths
    env["MAUVE_CH_CL_MAX"] = "2"
    r = subprocess.run([sys.executable, "-c", GAPCHAIN_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr[-3000:]
    assert [l for l in r.stderr.splitlines() if "per-gap chaining" in l and "(device" in l]
    del env["MAUVE_CH_CL_MAX"]
    env["MAUVE_HOST_GAP_CHAIN"] = "1"
    r = subprocess.run([sys.executable, "-c", GAPCHAIN_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr[-3000:]
    assert not [l for l in r.stderr.splitlines() if "per-gap chaining" in l and "(device" in l]


def test_align_matches_given_list(ctx):
    """Aligner::align(MatchList&, ...) (mauveAligner.cpp:698): the caller's match list is chained, not re-found.
    The finder's own N-way list reproduces mauve_align; order, subset matches and duplicates-free shuffles do not
    matter; a list with some matches removed gives an alignment anchored on the rest only."""
    from mauvealigner_amd import _lib
    gs = synth.make_config("C3", scale=0.03)
    ctx.set_genomes(gs)
    N = len(gs)
    p = _lib.default_params()
    whole = ctx.align(p)
    w = _lib.default_seed_weight(sum(len(g) for g in gs) // N)
    pat = _lib.get_seed(w, 0)
    ln, st = ctx.seed_mums(pat, mask=(1 << N) - 1)
    assert np.array_equal(ln, whole["mum_length"])
    keys = ("cols", "col_off", "left", "right", "reverse", "dp_score", "anchor_start", "anchor_length", "lcb_weight", "mum_start")
    same = ctx.align_matches(p, ln, st)
    for k in keys:
        assert np.array_equal(whole[k], same[k]), k
    # shuffled, with subset matches of the UNIQUE finder mixed in (ignored: not N-way)
    sl, ss = ctx.seed_mums(pat, mode=1)
    sub = np.count_nonzero(ss, axis=1) < N
    rng = np.random.default_rng(1)
    al, as_ = np.concatenate([ln, sl[sub]]), np.concatenate([st, ss[sub]])
    perm = rng.permutation(len(al))
    mixed = ctx.align_matches(p, al[perm], as_[perm])
    for k in keys:
        assert np.array_equal(whole[k], mixed[k]), k
    # a thinned list: every anchor of the result comes from a kept match
    keep = rng.random(len(ln)) < 0.5
    thin = ctx.align_matches(_lib.default_params(recursive=0, extend_lcbs=0), ln[keep], st[keep])      # (neither stage may add anchors of its own)
    kept = {tuple(r) for r in np.abs(st[keep]).tolist()}
    a0 = thin["anchor_start"][:, 0]
    assert thin["n_mums"] == int(keep.sum()) and len(a0) > 0
    lo, hi = np.abs(st[keep][:, 0]), np.abs(st[keep][:, 0]) + ln[keep] - 1
    order = np.argsort(lo)
    idx = np.searchsorted(lo[order], a0, side="right") - 1
    assert np.all(idx >= 0) and np.all(a0 <= hi[order][idx])
    with pytest.raises(RuntimeError, match=r"\(-1\)"):
        ctx.align_matches(p, np.array([50], np.int64), np.array([[10 ** 9] * N], np.int64))


def test_full_size_c2_properties(ctx):
    """BASELINE config C2 (3 x 5 Mbp, weight 15) through the whole path: size-independent properties --
    every base of every genome appears in exactly one interval, rows reproduce the genomes, determinism."""
    from mauvealigner_amd import _lib
    gs = synth.make_config("C2", scale=1.0)
    ctx.set_genomes(gs)
    p = _lib.default_params(seed_weight=15)
    r = ctx.align(p)
    r2 = ctx.align(p)
    for k in ("cols", "col_off", "left", "right", "anchor_start", "dp_score"):
        assert np.array_equal(r[k], r2[k])
    N = len(gs)
    assert r["n_lcb"] == 1
    cols, off = r["cols"], r["col_off"]
    for g in range(N):
        cover = np.zeros(len(gs[g]), np.int32)
        for iv in range(r["n_iv"]):
            le, re = r["left"][iv, g], r["right"][iv, g]
            n_res = int(((cols[off[iv]:off[iv + 1]] >> g) & 1).sum())
            if le == 0:
                assert n_res == 0
                continue
            assert n_res == re - le + 1
            cover[le - 1:re] += 1
        assert np.all(cover == 1)
    lcb_cols = cols[off[0]:off[1]]
    assert np.mean(lcb_cols == 7) > 0.9
    assert r["n_dp_cells"] > 0


def test_sharded_align_equals_whole(ctx):
    """The begin / dp / finish phases (LCB-sharded form of mauve_align): a single rank and a simulated two-rank
    split of the DP intervals both reproduce mauve_align bit for bit."""
    from mauvealigner_amd import _lib, parallel
    gs = synth.make_config("C3", scale=0.03)
    ctx.set_genomes(gs)
    names = ["g%d" % i for i in range(len(gs))]
    whole = ctx.align(_lib.default_params(), names=names, want_xmfa=True)
    one = parallel.align_sharded(ctx, _lib.default_params(), None, names=names, want_xmfa=True)
    keys = ("cols", "col_off", "left", "right", "reverse", "dp_score", "anchor_start", "anchor_length", "lcb_weight")
    for k in keys:
        assert np.array_equal(whole[k], one[k]), k
    assert whole["xmfa"] == one["xmfa"] and whole["n_dp_cells"] == one["n_dp_cells"]
    # two simulated ranks on one GPU: each aligns its LPT share, the shares are merged, every rank finishes
    n_dp, cost, cap = ctx.align_begin(_lib.default_params())
    assert n_dp == whole["n_gap_dp"] and int(cost.sum()) >= whole["n_dp_cells"]
    parts = parallel.lpt_partition(cost, 2)
    assert abs(int(cost[parts[0]].sum()) - int(cost[parts[1]].sum())) <= int(cost.max())
    all_cols, all_score, cells = [None] * n_dp, np.zeros(n_dp, np.int64), 0
    for part in parts:
        cols, score, c = ctx.align_dp(part, cap)
        cells += c
        for k, i in enumerate(part.tolist()):
            all_cols[i] = cols[k].copy()
            all_score[i] = score[k]
    two = ctx.align_finish(all_cols, all_score, cells, names=names, want_xmfa=True)
    for k in keys:
        assert np.array_equal(whole[k], two[k]), k
    assert whole["xmfa"] == two["xmfa"] and whole["n_dp_cells"] == two["n_dp_cells"]
    # phase order is enforced
    with pytest.raises(RuntimeError):
        ctx.align_dp(np.array([0], np.int64), cap)


def _same_progressive(ctx, gs, tree=None, **kw):
    from mauvealigner_amd import _lib
    ctx.set_genomes(gs)
    names = ["g%d" % i for i in range(len(gs))]
    r = ctx.progressive_align(_lib.default_params(**kw), names=names, want_xmfa=True, tree=tree)
    e = O.progressive_align(gs, O.default_params(**kw), names=names, want_xmfa=True, tree=tree)
    if tree is None:
        assert np.array_equal(r["dist"], e["dist"])
    assert np.array_equal(r["tree"][0], e["tree"][0]) and np.array_equal(r["tree"][1], e["tree"][1])
    a = e["aln"]
    assert r["n_iv"] == a["n_iv"]
    for k in ("left", "right", "reverse", "col_off", "cols", "dp_score"):
        assert np.array_equal(r[k], a[k]), k
    assert r["n_gap_dp"] == a["n_gap_dp"] and r["n_dp_cells"] == a["n_dp_cells"]
    assert r["xmfa"] == e["xmfa"]
    return r


def _random_tree(N, rng):
    """a random binary tree over N leaves in merge order (children before parents, root last)"""
    left = np.full(2 * N - 1, -1, np.int32)
    right = np.full(2 * N - 1, -1, np.int32)
    roots = list(range(N))
    for k in range(N, 2 * N - 1):
        a, b = rng.choice(len(roots), 2, replace=False)
        left[k], right[k] = roots[a], roots[b]
        roots = [x for i, x in enumerate(roots) if i not in (a, b)] + [k]
    return left, right


def test_progressive_align_along_a_given_tree(ctx):
    """--input-guide-tree (progressiveMauve.cpp:689-690; mauve_progressive_align_tree): the UPGMA tree handed back in
    reproduces the default result; any other binary tree gives the oracle's result for that tree; what is not a
    binary tree in merge order is refused."""
    from mauvealigner_amd import _lib
    gs = synth.make_config("C4", scale=0.02)
    N = len(gs)
    r0 = _same_progressive(ctx, gs)
    r1 = _same_progressive(ctx, gs, tree=r0["tree"])
    for k in ("left", "right", "reverse", "col_off", "cols", "dp_score"):
        assert np.array_equal(r0[k], r1[k]), k
    assert r0["xmfa"] == r1["xmfa"]
    rng = np.random.default_rng(77)
    differs = 0
    for _ in range(3):
        t = _random_tree(N, rng)
        r = _same_progressive(ctx, gs, tree=t)
        differs += r["xmfa"] != r0["xmfa"]
        for g in range(N):                                   # still a partition of every genome
            cover = np.zeros(len(gs[g]), np.int32)
            for iv in range(r["n_iv"]):
                if r["left"][iv, g]:
                    cover[r["left"][iv, g] - 1:r["right"][iv, g]] += 1
            assert np.all(cover == 1)
    assert differs                                           # the tree does order the alignment
    _same_progressive(ctx, gs, tree=_random_tree(N, rng), lcb_scoring=1)
    gs3 = synth.make_config("C3", scale=0.01)
    _same_progressive(ctx, gs3, tree=_random_tree(len(gs3), rng))
    # refused: a leaf with children, a child used twice, a forward reference, a self pair, the wrong size
    ctx.set_genomes(gs)
    good = r0["tree"]
    for mut in ("leaf", "twice", "forward", "self"):
        left, right = good[0].copy(), good[1].copy()
        if mut == "leaf":
            left[0], right[0] = 1, 2
        elif mut == "twice":
            left[N + 1] = left[N]
        elif mut == "forward":
            left[N] = 2 * N - 2
        else:
            right[N] = left[N]
        assert not O.check_tree(N, left, right)
        with pytest.raises(RuntimeError):
            ctx.progressive_align(_lib.default_params(), tree=(left, right))
    with pytest.raises(ValueError):
        ctx.progressive_align(_lib.default_params(), tree=(good[0][:-1], good[1][:-1]))
    assert O.check_tree(N, good[0], good[1])


def test_seed_family(ctx):
    """DESIGN.md S3b (progressiveMauve.cpp:502-546 --seed-family; setUseSeedFamilies :604-605): the three seeds of the
    weight searched longest first and merged -- bit-exact against the oracle through mauve_align and at every node of
    the progressive path; the family finds at least the anchors of its first seed's search, and more on divergent input."""
    from mauvealigner_amd import _lib, accuracy
    for cfg, scale in (("C1", 0.1), ("C3", 0.02), ("C2", 0.01)):
        _same_align(ctx, synth.make_config(cfg, scale=scale), seed_family=1)
    gs = synth.make_config("C3", scale=0.02)
    _same_align(ctx, gs, seed_family=1, seed_weight=11, recursive=0)
    _same_align(ctx, gs, seed_family=1, extend_lcbs=1)
    _same_align(ctx, gs, seed_family=1, lcb_scoring=1)
    _same_progressive(ctx, synth.make_config("C4", scale=0.02), seed_family=1)
    _same_progressive(ctx, synth.make_config("C4", scale=0.02), seed_family=1, weight_scaling=1, conservation_scale_ppm=500000, lcb_scoring=1)
    # sensitivity on a divergent pair: the family anchors more
    g2, org = synth.star_genomes(2, 60000, 0.16, 77, inversions=2, track=True)
    ctx.set_genomes(g2)
    one = ctx.align(_lib.default_params(recursive=0, gapped=0))
    fam = ctx.align(_lib.default_params(recursive=0, gapped=0, seed_family=1))
    assert fam["n_mums"] > one["n_mums"] and fam["anchor_length"].sum() > one["anchor_length"].sum()
    assert accuracy.score_alignment(fam, org)["tp"] > accuracy.score_alignment(one, org)["tp"]
    # an explicit pattern cannot name a family
    with pytest.raises(RuntimeError):
        ctx.align(_lib.default_params(seed_family=1, seed_pattern=O.get_seed(11, 0)))
    with pytest.raises(RuntimeError):
        ctx.progressive_align(_lib.default_params(seed_family=1, seed_pattern=O.get_seed(11, 0)))


def test_progressive_weight_scaling(ctx):
    """DESIGN.md S11b (ProgressiveAligner::setUseLcbWeightScaling / setConservationDistanceScale /
    setMinimumBreakpointPenalty, progressiveMauve.cpp:626-652): every node's minimum LCB weight scaled by the
    conservation distance of its subtrees -- bit-exact against the oracle with length and sum-of-pairs weights, along
    the UPGMA and along a given tree; scale 0 is the unscaled result; a floor above every weight empties the blocks."""
    from mauvealigner_amd import _lib
    gs = synth.make_config("C4", scale=0.02)
    N = len(gs)
    plain = _same_progressive(ctx, gs)
    zero = _same_progressive(ctx, gs, weight_scaling=1, conservation_scale_ppm=0)
    assert zero["xmfa"] == plain["xmfa"]
    half = _same_progressive(ctx, gs, weight_scaling=1, conservation_scale_ppm=500000)
    full = _same_progressive(ctx, gs, weight_scaling=1, conservation_scale_ppm=1000000, lcb_weight=400)
    unscaled = _same_progressive(ctx, gs, lcb_weight=400)
    assert full["n_lcb"] >= unscaled["n_lcb"]                  # lighter thresholds keep at least as many blocks
    assert half["n_lcb"] >= plain["n_lcb"]
    _same_progressive(ctx, gs, weight_scaling=1, conservation_scale_ppm=500000, lcb_scoring=1)
    _same_progressive(ctx, gs, weight_scaling=1, conservation_scale_ppm=700000, min_scaled_penalty=300)
    rng = np.random.default_rng(9)
    _same_progressive(ctx, gs, tree=_random_tree(N, rng), weight_scaling=1, conservation_scale_ppm=500000)
    none = _same_progressive(ctx, gs, weight_scaling=1, conservation_scale_ppm=500000, min_scaled_penalty=10 ** 9)
    assert none["n_lcb"] == 0
    _same_progressive(ctx, synth.make_config("C3", scale=0.02), weight_scaling=1, conservation_scale_ppm=500000)


def test_breakpoint_distance_scaling(ctx):
    """DESIGN.md S11c (ProgressiveAligner::setBreakpointDistanceScale / setBpDistEstimateMinScore, progressiveMauve.cpp:628-642):
    the pairwise breakpoint estimate -- broken adjacencies between every pair's matches, counted on the device from the
    guide tree's pairwise records (two sorts of (pair, position) keys, ranks, one adjacency kernel) -- equals the oracle's for
    several length floors, grows with the rearrangements of the inputs, and the node weights scaled by it give the oracle's
    alignment along the UPGMA and along a given tree."""
    from mauvealigner_amd import _lib
    gs = synth.make_config("C4", scale=0.02)
    N = len(gs)
    pat = O.get_seed(11, 0)
    ctx.set_genomes(gs)
    for min_len in (0, 22, 40):
        bp = ctx.breakpoint_counts(pat, min_len)
        assert np.array_equal(bp, O.breakpoint_counts(gs, pat, min_len)), min_len
    assert np.array_equal(bp, bp.T) and not bp.diagonal().any() and bp.max() > 0
    flat = synth.star_genomes(3, 30000, 0.02, 5, inversions=0)
    ctx.set_genomes(flat)
    b0 = ctx.breakpoint_counts(pat, 22)
    assert np.array_equal(b0, O.breakpoint_counts(flat, pat, 22)) and b0.max() == 0        # collinear genomes: no breakpoints
    half = _same_progressive(ctx, gs, weight_scaling=1, conservation_scale_ppm=500000, bp_dist_scale_ppm=500000)
    cons = _same_progressive(ctx, gs, weight_scaling=1, conservation_scale_ppm=500000)
    assert half["xmfa"] != cons["xmfa"]                         # the second factor moves the thresholds of the nodes
    _same_progressive(ctx, gs, weight_scaling=1, conservation_scale_ppm=0, bp_dist_scale_ppm=1000000, bp_dist_min_score=30)
    _same_progressive(ctx, gs, weight_scaling=1, conservation_scale_ppm=500000, bp_dist_scale_ppm=500000, lcb_scoring=1)
    rng = np.random.default_rng(11)
    _same_progressive(ctx, gs, tree=_random_tree(N, rng), weight_scaling=1, conservation_scale_ppm=300000, bp_dist_scale_ppm=700000)
    off = _same_progressive(ctx, gs, bp_dist_scale_ppm=500000)  # without weight_scaling the scale is not looked at
    assert off["xmfa"] == _same_progressive(ctx, gs)["xmfa"]


def test_progressive_refinement(ctx):
    """DESIGN.md S13 (ProgressiveAligner::setRefinement, progressiveMauve.cpp:578-579): every gapped interval of >= 3 sequences is
    also aligned in its rotated orders -- one more batch through the same DP kernels -- scored by the sum-of-pairs kernel
    (dp_sp_scores) and the best alignment kept.  Bit-exact against the oracle for 1, 2 and more rounds than sequences; the
    refinement only ever raises the sum-of-pairs score of an interval's columns; with two genomes it changes nothing."""
    from mauvealigner_amd import _lib
    def family(n, length, div, seed):                           # indel-rich descendants of one ancestor: many intervals with k >= 3
        rng = np.random.default_rng(seed)
        anc = rng.integers(0, 4, length, dtype=np.uint8)
        return [np.ascontiguousarray(synth.mutate(anc, div, rng, indel_frac=0.3)) for _ in range(n)]
    gs = family(5, 60000, 0.06, 21)
    plain = _same_progressive(ctx, gs, seed_weight=11)
    r1 = _same_progressive(ctx, gs, seed_weight=11, refine_rounds=1)
    r2 = _same_progressive(ctx, gs, seed_weight=11, refine_rounds=2)
    r9 = _same_progressive(ctx, gs, seed_weight=11, refine_rounds=9)
    assert r1["n_dp_cells"] > plain["n_dp_cells"] and r2["n_dp_cells"] > r1["n_dp_cells"] and r9["n_dp_cells"] >= r2["n_dp_cells"]
    assert r2["xmfa"] != plain["xmfa"]                          # some interval found a better order
    assert r2["n_iv"] == plain["n_iv"] and np.array_equal(r2["left"], plain["left"]) and np.array_equal(r2["right"], plain["right"])
    _same_progressive(ctx, synth.make_config("C4", scale=0.02), refine_rounds=2, weight_scaling=1, conservation_scale_ppm=500000, bp_dist_scale_ppm=500000)
    two = family(2, 40000, 0.05, 3)
    assert _same_progressive(ctx, two, refine_rounds=2)["xmfa"] == _same_progressive(ctx, two)["xmfa"]


def test_homology_pass_before_the_backbone(ctx):
    """DESIGN.md S12b (detectAndApplyBackbone's homology HMM, progressiveMauve.cpp:226-243,319-322): the two-state Viterbi path per
    interval and genome pair -- one wave per (interval, pair), 64 columns per step as a max-plus matrix scan, the way back on two
    ballot words per step -- and the re-split columns equal the oracle's, bit for bit: on a clean alignment nothing moves; a
    stretch of unrelated sequence that the gapped aligner forced into columns is taken apart again; reverse-strand intervals,
    a progressive alignment with absent genomes, other scores; the backbone afterwards is the oracle's backbone of the new columns."""
    from mauvealigner_amd import _lib
    rng = np.random.default_rng(3)
    anc = rng.integers(0, 4, 30000, dtype=np.uint8)

    def check(gs, progressive=False, hmm_kw=None, **kw):
        ctx.set_genomes(gs)
        if progressive:
            ctx.progressive_align(_lib.default_params(**kw), fetch=False)
            a = O.progressive_align(gs, O.default_params(**kw))["aln"]
        else:
            ctx.align(_lib.default_params(**kw), fetch=False)
            a = O.align(gs, O.default_params(**kw))["aln"]
        hk = hmm_kw or {}
        r = ctx.apply_homology(ctx.hmm_params(**hk))
        off, cols, moved = O.homology_apply(gs, a["left"], a["right"], a["reverse"], a["col_off"], a["cols"], O.hmm_params(**hk))
        assert r["n_moved"] == moved and r["n_cols"] == len(cols)
        assert np.array_equal(r["col_off"], off) and np.array_equal(r["cols"], cols)
        assert np.array_equal(r["left"], a["left"]) and np.array_equal(r["right"], a["right"])
        for g in range(len(gs)):                                # every residue is still in exactly one column
            assert int(((r["cols"] >> g) & 1).sum()) == int(((a["cols"] >> g) & 1).sum())
        b = ctx.backbone(island_gap=20)
        eb = O.backbone(a["left"], a["right"], a["reverse"], off, cols, island_gap=20)
        for k in ("seg_iv", "seg_col", "seg_len", "seg_mask", "seg_left", "seg_right", "islands"):
            assert np.array_equal(b[k], eb[k]), k
        return r, a

    h = ctx.hmm_params()
    oh = O.hmm_params()
    assert (h.match, h.mismatch, h.gap, h.go_homologous, h.go_unrelated) == (oh.match, oh.mismatch, oh.gap, oh.go_homologous, oh.go_unrelated) == (1030, -916, -500, -11513, -20723)
    clean = [np.ascontiguousarray(synth.mutate(anc, 0.03, rng, indel_frac=0.1)) for _ in range(3)]
    r, a = check(clean, seed_weight=11)
    assert r["n_moved"] == 0 and np.array_equal(r["cols"], a["cols"])
    gs = [np.ascontiguousarray(synth.mutate(anc, 0.05, rng, indel_frac=0.2)) for _ in range(3)]
    g1 = gs[1].copy(); g1[12000:12600] = rng.integers(0, 4, 600, dtype=np.uint8); gs[1] = g1        # unrelated sequence of the same length
    r, a = check(gs, seed_weight=11)
    assert r["n_moved"] > 100 and r["n_cols"] > len(a["cols"])
    inv = [g.copy() for g in gs]
    inv[2][5000:20000] = synth.revcomp(inv[2][5000:20000])     # the stretch now lies in a reverse-strand interval of genome 2
    r, _ = check(inv, seed_weight=11)
    assert r["n_moved"] > 100
    check(gs, seed_weight=11, hmm_kw=dict(identity=0.9, pgh=1e-3, pgu=1e-3))
    check(gs, seed_weight=11, hmm_kw=dict(gap=-3000, go_unrelated=-2000))
    c4 = synth.make_config("C4", scale=0.02)
    c4[3] = c4[3].copy(); c4[3][9000:9500] = rng.integers(0, 4, 500, dtype=np.uint8)
    check(c4, progressive=True)
    check(synth.make_config("C3", scale=0.4), seed_weight=15)   # device-assembled result: the columns never left HBM


def test_guide_tree_and_progressive_align(ctx):
    """ProgressiveAligner stand-in: guide tree + guide-tree recursive anchoring, bit-exact vs oracle."""
    gs = synth.make_config("C4", scale=0.04)
    r = _same_progressive(ctx, gs)
    N = len(gs)
    # the UPGMA tree recovers the balanced topology the generator used: the cherries (nodes of two leaves) are the sisters (0,1), (2,3), (4,5), (6,7)
    left, right = r["tree"]
    cherries = sorted(tuple(sorted((int(left[k]), int(right[k])))) for k in range(N, 2 * N - 1) if left[k] < N and right[k] < N)
    assert cherries == [(0, 1), (2, 3), (4, 5), (6, 7)]
    # clade-specific insertions are aligned below the root: some blocks hold a proper subset of >= 2 genomes
    multi = np.count_nonzero(r["left"][:r["n_lcb"]], axis=1)
    assert (multi == N).any() and ((multi >= 2) & (multi < N)).any()
    # every base of every genome appears in exactly one interval
    for g in range(N):
        cover = np.zeros(len(gs[g]), np.int32)
        for iv in range(r["n_iv"]):
            if r["left"][iv, g]:
                cover[r["left"][iv, g] - 1:r["right"][iv, g]] += 1
        assert np.all(cover == 1)
    _same_progressive(ctx, gs, max_gapped_len=500)
    _same_progressive(ctx, gs, recursive=0)
    _same_progressive(ctx, synth.make_config("C3", scale=0.02))
    _same_progressive(ctx, synth.make_config("C1", scale=0.1))


def test_resident_result_survives_other_device_work(ctx):
    """A result left in HBM (align(fetch=False)) keeps its anchor table and match list in buffers later seed passes
    reuse: every entry point that runs such work first brings them to the host, so a fetch after it is still the result."""
    from mauvealigner_amd import _lib
    gs = synth.make_config("C3", scale=0.4)             # > 16 k matches: the chains, the DP front and the assembly stay on the device
    ctx.set_genomes(gs)
    p = _lib.default_params(seed_weight=15)
    ref = ctx.align(p)
    if True:
        for other in ("seed_mums", "sorted_mer_list", "enumerate", "dp_batch", "guide_tree"):
            sz = ctx.align(p, fetch=False)
            assert sz["n_anchor"] == ref["n_anchor"]
            if other == "seed_mums":
                ctx.seed_mums(_lib.get_seed(9, 0), mask=0)
            elif other == "sorted_mer_list":
                ctx.sorted_mer_list(1, _lib.get_seed(9, 0))
            elif other == "enumerate":
                ctx.seed_match_enumerate(0, _lib.get_seed(7, 0))
            elif other == "dp_batch":
                ctx.dp_batch([[gs[0][:50], gs[1][:60]] + [np.zeros(0, np.uint8)] * (len(gs) - 2)])
            else:
                ctx.guide_tree(_lib.get_seed(11, 0))
            sz_t = _lib.AlignSizes(**{k: sz[k] for k, _ in _lib.AlignSizes._fields_})
            r = ctx._fetch(sz_t)
            for k in ("mum_length", "mum_start", "anchor_start", "anchor_length", "anchor_lcb", "left", "right", "reverse", "col_off", "cols", "dp_score"):
                assert np.array_equal(r[k], ref[k]), (other, k)


def test_page_locked_caller_buffers(ctx):
    """mauve_host_alloc buffers on both sides of the pass: genomes uploaded straight from them, results fetched straight into
    them; same result as through pageable buffers, and the XMFA text (which needs the host copy of the genomes) as well."""
    from mauvealigner_amd import _lib
    gs = synth.make_config("C3", scale=0.4)
    ctx.set_genomes(gs)
    p = _lib.default_params(seed_weight=15)
    ref = ctx.align(p, want_xmfa=True)
    packed = []
    for g in gs:
        w = _lib.pack_codes(g)
        pw = _lib.pinned_empty(len(w), np.uint64)
        pw[:] = w
        packed.append(pw)
    bufs = _lib.ResultBuffers()
    for _ in range(2):
        ctx.set_genomes_packed(packed, [len(g) for g in gs])
        r = ctx.align(p, out=bufs, want_xmfa=True)
        for k in ("mum_length", "mum_start", "anchor_start", "anchor_length", "anchor_lcb", "left", "right", "reverse", "col_off", "cols", "dp_score"):
            assert np.array_equal(r[k], ref[k]), k
        assert r["xmfa"] == ref["xmfa"]
    # a second fetch of the same result into pageable memory still works (the columns are still in HBM)
    sz_t = _lib.AlignSizes(**{k: r[k] for k, _ in _lib.AlignSizes._fields_})
    again = ctx._fetch(sz_t)
    assert np.array_equal(again["cols"], ref["cols"])


SCAN_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from mauvealigner_amd import _lib, synth
from oracle import pyoracle as O
rng = np.random.default_rng(21)
ctx = _lib.Context(0)
def seqs(lens, div=0.15):
    base = rng.integers(0, 4, max(max(lens), 1), dtype=np.uint8)
    out = []
    for L in lens:
        if L == 0:
            out.append(np.zeros(0, np.uint8)); continue
        x = synth.mutate(base, div, rng, indel_frac=0.3)[:L]
        if len(x) < L:
            x = np.concatenate([x, rng.integers(0, 4, L - len(x), dtype=np.uint8)])
        out.append(x)
    return out
def check(ivs):
    cols, score = ctx.dp_batch(ivs)
    for iv, c, s in zip(ivs, cols, score):
        ec, es = O.align_interval(iv)
        assert len(c) == len(ec) and np.array_equal(c, ec) and int(s) == es, [len(x) for x in iv]
# two sequences: every combination of one band / several bands of rows and of columns, band edges, tall and flat
shapes2 = [(700, 20), (20, 700), (256, 256), (257, 255), (255, 257), (513, 64), (64, 513), (600, 512), (512, 600), (1030, 300), (300, 1030),
           (120, 117), (117, 120), (65, 64), (1, 900), (900, 1), (2, 2), (300, 299), (1500, 7), (7, 1500), (260, 3), (3, 260)]
check([seqs(list(s)) for s in shapes2])
# unrelated sequences (long gap runs in the traceback) and identical ones (one long diagonal run)
a = rng.integers(0, 4, 800, dtype=np.uint8)
check([[a, rng.integers(0, 4, 30, dtype=np.uint8)], [rng.integers(0, 4, 30, dtype=np.uint8), a], [a, a.copy()], [a[:300], a[:300].copy()],
       [rng.integers(0, 4, 400, dtype=np.uint8), rng.integers(0, 4, 380, dtype=np.uint8)]])
# more sequences: the long one first, in the middle, last; empty members; profiles that grow across the band edge
check([seqs(l) for l in ([1200, 20, 18, 22, 19], [20, 18, 1200, 22, 19], [20, 18, 22, 19, 1200], [250, 260, 255, 0, 258], [0, 130, 0, 140, 135],
                         [90, 95, 100, 105, 110], [300, 10, 0, 310, 12], [5, 5, 600, 600, 5])])
check([seqs([int(rng.integers(0, 140)) for _ in range(4)], div=0.1) for _ in range(150)])
print("OK")
"""


def test_dp_scan_kernels():
    """The one-wave class runs the scan-formulated sweeps (rows on the lanes column by column, or columns on the lanes row by
    row, chosen per step) and the run-length traceback; with the workgroup pipeline switched off every shape goes through them.
    The anti-diagonal sweep of the same class (MAUVE_DP_NOSCAN) must agree as well."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({"MAUVE_DP_ONE_WAVE": "1"}, {"MAUVE_DP_ONE_WAVE": "1", "MAUVE_DP_NOSCAN": "1"}, {"MAUVE_DP_ONE_WAVE": "1", "MAUVE_DP_NO_GROUPS": "1"}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", SCAN_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), str(extra) + "\n" + r.stdout + r.stderr[-3000:]



WIDE_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from mauvealigner_amd import _lib, synth
from oracle import pyoracle as O
rng = np.random.default_rng(33)
ctx = _lib.Context(0)
def seqs(lens, div=0.15):
    base = rng.integers(0, 4, max(max(lens), 1), dtype=np.uint8)
    out = []
    for L in lens:
        if L == 0:
            out.append(np.zeros(0, np.uint8)); continue
        x = synth.mutate(base, div, rng, indel_frac=0.3)[:L]
        if len(x) < L:
            x = np.concatenate([x, rng.integers(0, 4, L - len(x), dtype=np.uint8)])
        out.append(x)
    return out
def check(ivs):
    cols, score = ctx.dp_batch(ivs)
    for iv, c, s in zip(ivs, cols, score):
        ec, es = O.align_interval(iv)
        assert len(c) == len(ec) and np.array_equal(c, ec) and int(s) == es, [len(x) for x in iv]
# two sequences: one band against several, several against several, one and more than one super-band (8 waves x 256) in either dimension,
# band and super-band edges, tall-thin and flat-long
shapes2 = [(6700, 20), (20, 6700), (1600, 1600), (257, 3), (3, 257), (256, 257), (257, 256), (512, 513), (2048, 300), (2049, 300), (300, 2049), (2047, 2050),
           (4100, 130), (130, 4100), (4097, 2), (2, 4097), (1030, 1025), (700, 699), (5000, 900), (900, 5000), (3000, 1)]
check([seqs(list(s)) for s in shapes2])
# unrelated sequences (long gap runs) and identical ones (one diagonal), beside small intervals of the other kernels
a = rng.integers(0, 4, 2600, dtype=np.uint8)
check([[a, rng.integers(0, 4, 40, dtype=np.uint8)], [rng.integers(0, 4, 40, dtype=np.uint8), a], [a, a.copy()], [a[:700], a[:700].copy()],
       [rng.integers(0, 4, 1400, dtype=np.uint8), rng.integers(0, 4, 1380, dtype=np.uint8)]] + [seqs([int(rng.integers(1, 60)), int(rng.integers(1, 60))]) for _ in range(100)])
# more sequences: the long one first, in the middle, last; empty members; a profile that outgrows a band / a super-band while it is built
check([seqs(l) for l in ([2300, 20, 18, 22, 19], [20, 18, 2300, 22, 19], [20, 18, 22, 19, 2300], [250, 260, 255, 0, 258], [0, 1300, 0, 1400, 1350],
                         [700, 750, 800, 850, 900], [300, 10, 0, 2100, 12], [5, 5, 600, 600, 5])])
check([seqs([2040, 2040, 30, 2040]), seqs([30, 2600, 2500, 28])])
check([seqs([int(rng.integers(0, 700)) for _ in range(4)], div=0.1) for _ in range(40)])
print("OK")
"""


def test_dp_wide_sweeps():
    """The workgroup class runs the wide sweep (dp_step_wide: the scan-formulated sweep over all waves of a workgroup, one column / row of a
    whole super-band per step): with MAUVE_DP_WIDE_MIN=1 every interval with a dimension beyond one band goes through it -- both
    orientations, one and several super-bands, the one-wave fallback for steps small in both dimensions, several sequences.  Bit-exact against
    the oracle; the stripe pipeline that banded intervals keep (MAUVE_DP_NO_WIDE) must agree on the same shapes.  The largest intervals of a launch get a
    cluster of up to four workgroups (super-bands pipelined across CUs through parked lines and progress tokens): on by default here, off
    (MAUVE_DP_CLUSTER=0) and with the 16-wave shape as well."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({"MAUVE_DP_WIDE_MIN": "1"}, {"MAUVE_DP_WIDE_MIN": "1", "MAUVE_DP_BIG_MAX": "7"}, {"MAUVE_DP_WIDE_MIN": "1", "MAUVE_DP_NO_WIDE": "1"},
                  {"MAUVE_DP_WIDE_MIN": "1", "MAUVE_DP_CLUSTER": "0"}, {"MAUVE_DP_WIDE_MIN": "1", "MAUVE_DP_WIDE_R2": "1", "MAUVE_DP_CLUSTER": "3"}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", WIDE_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), str(extra) + "\n" + r.stdout + r.stderr[-3000:]



def test_compact_fetch_equals_the_wide_one(ctx):
    """mauve_align_fetch_compact: one byte per column for up to 8 genomes (two up to 16, four beyond), int32 match and anchor tables -- the same
    values as mauve_align_fetch, from page-locked buffers (narrowed on the device, one DMA each) and from pageable ones (converted on the host),
    for a device-assembled result, a host-chained one and a progressive one; a column type too narrow for the genome count is refused."""
    from mauvealigner_amd import _lib
    keys = ("mum_length", "mum_start", "lcb_left", "lcb_right", "lcb_weight", "anchor_length", "anchor_start", "anchor_lcb", "left", "right", "reverse", "col_off", "cols", "dp_score")
    cases = [(synth.make_config("C3", scale=0.4), dict(seed_weight=15), False, 1), (synth.make_config("C3", scale=0.02), {}, False, 1),
             (synth.star_genomes(10, 20_000, 0.04, 3), {}, False, 2), (synth.star_genomes(17, 8_000, 0.04, 4), {}, False, 4),
             (synth.make_config("C4", scale=0.05), {}, True, 1)]
    for gs, kw, prog, cb in cases:
        ctx.set_genomes(gs)
        run = (lambda **k: ctx.progressive_align(_lib.default_progressive_params(**kw), **k)) if prog else (lambda **k: ctx.align(_lib.default_params(**kw), **k))
        wide = run()
        for bufs in (None, _lib.ResultBuffers()):
            r = run(out=bufs, compact=True)
            assert r["col_bytes"] == cb and r["cols"].dtype.itemsize == cb and r["anchor_start"].dtype == np.int32
            for k in keys:
                if prog and k.startswith(("mum_", "anchor_", "lcb_")):
                    continue
                assert np.array_equal(r[k], wide[k]), (len(gs), k)
    gs = synth.star_genomes(9, 5_000, 0.04, 5)
    ctx.set_genomes(gs)
    sz = ctx.align(_lib.default_params(), fetch=False)
    L = ctx.L
    L.mauve_align_fetch_compact.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 14
    assert L.mauve_align_fetch_compact(ctx.h, 1, *([None] * 14)) == -1          # nine genomes do not fit a byte
    assert L.mauve_align_fetch_compact(ctx.h, 3, *([None] * 14)) == -1
    assert L.mauve_align_fetch_compact(ctx.h, 2, *([None] * 14)) == 0


RCCL_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from mauvealigner_amd import _lib, synth, parallel
ctx = _lib.Context(0)                                            # (sets the device the communicator is made on)
comm = parallel.RcclComm(0, 1)
parallel.attach_shard_rccl(ctx, comm)
keys = ("cols", "col_off", "dp_score", "left", "right", "reverse")
for cfg, scale, prog in (("C5", 0.04, False), ("C4", 0.08, True)):
    gs = synth.make_config(cfg, scale=scale)
    ctx.set_genomes(gs)
    r = ctx.progressive_align(_lib.default_progressive_params()) if prog else ctx.align(_lib.default_params())
    st = parallel.shard_stats(ctx)
    ctx.L.mauve_set_shard_rccl(ctx.h, 0, 1, None)                       # sharding off: the plain result
    ref = ctx.progressive_align(_lib.default_progressive_params()) if prog else ctx.align(_lib.default_params())
    parallel.attach_shard_rccl(ctx, comm)
    for k in keys:
        assert np.array_equal(r[k], ref[k]), (cfg, k)
    assert st["exchanges"] >= 2 and st["bytes_received"] == st["bytes_sent"] > 0, st
    print(cfg, st)
comm.close(); ctx.close()
print("OK")
"""


def test_rccl_exchange_inside_the_library():
    """mauve_set_shard_rccl: the library runs its exchanges itself -- ncclAllGather of the sizes, then of the padded payloads, on its own stream
    between device buffers; no callback, no Python in the data path.  One GPU here, so ONE rank rehearses the path (MAUVE_SHARD_SINGLE: the units
    are still dealt out -- all to rank 0 -- and every exchange really goes through RCCL): recursion batches at C5, the guide tree's pairs and the
    nodes' intervals at C4 give the same result as without the collective, and the counters show the traffic."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MAUVE_SHARD_SINGLE="1")
    r = subprocess.run([sys.executable, "-c", RCCL_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr[-3000:]


GUARD_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from mauvealigner_amd import _lib, synth
from oracle import pyoracle as O
what = sys.argv[1]
ctx = _lib.Context(0)
if what == "cand":            # a candidate list deliberately too small: the kernel skips the stores past its end, the host refuses the list
    gs = synth.star_genomes(3, 60_000, 0.03, 9)
    ctx.set_genomes(gs)
    try:
        ctx.seed_mums(O.get_seed(11, 0), mode=0, mask=0)
        print("NO ERROR")
    except RuntimeError as e:
        print("OK" if "(-4)" in str(e) and "candidates for a list" in str(e) else "WRONG: %%s" %% e)
elif what == "runs":          # ... the run list of the pairwise finder
    gs = synth.star_genomes(4, 40_000, 0.03, 10)
    ctx.set_genomes(gs)
    try:
        ctx.guide_tree(O.get_seed(11, 0))
        print("NO ERROR")
    except RuntimeError as e:
        print("OK" if "(-4)" in str(e) and "runs for a list" in str(e) else "WRONG: %%s" %% e)
else:                         # the device extension hands a round back: the host rounds give the oracle's alignment
    gs = synth.star_genomes(4, 1_500_000, 0.05, 78, inversions=12)
    ctx.set_genomes(gs)
    p = _lib.default_params(seed_weight=15)
    r = ctx.align(p); e = O.align(gs, O.default_params(seed_weight=15))["aln"]
    assert r["n_mums"] > 16384
    for k in ("anchor_start", "anchor_length", "anchor_lcb", "left", "right", "reverse", "col_off", "cols", "dp_score"):
        assert np.array_equal(r[k], e[k]), k
    print("OK")
"""


def test_list_capacity_guards_and_extension_fallback():
    """Every compaction store of the seed pass is bounded by the capacity of its list (the counter keeps counting, the host turns a list
    that outgrew its buffer into MAUVE_ERR_LIMIT instead of a wild write): with MAUVE_LIST_CAP the candidate list and the pairwise
    finder's run list are made too small on purpose.  And the device LCB extension, when it meets a round it does not cover, puts everything
    back and the match-level rounds run on the host (MAUVE_EXT_DECLINE forces that hand-back): same alignment as the oracle."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for what, extra in (("cand", {"MAUVE_LIST_CAP": "50"}), ("runs", {"MAUVE_LIST_CAP": "50"}), ("ext", {"MAUVE_EXT_DECLINE": "1"})):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", GUARD_SCRIPT % {"root": root}, what], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), what + "\n" + r.stdout + r.stderr[-3000:]


ORDER_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from mauvealigner_amd import _lib
from oracle import pyoracle as O
z = np.load(%(case)r); gs = [z[k] for k in z.files]
ctx = _lib.Context(0); ctx.set_genomes(gs)
for kw in (dict(seed_family=1, recursive=0, extend_lcbs=0, add_unaligned=0, max_gapped_len=300, lcb_weight=240),
           dict(seed_family=1, recursive=1, extend_lcbs=1, add_unaligned=0, max_gapped_len=300, lcb_weight=240)):
    r = ctx.align(_lib.default_params(**kw)); a = O.align(gs, O.default_params(**kw))["aln"]
    for k in ("anchor_length", "anchor_start", "anchor_lcb", "left", "right", "col_off", "cols", "dp_score"):
        assert np.array_equal(r[k], a[k]), (kw, k)
print("OK")
"""


def test_chain_order_after_the_elimination():
    """With three or more matches overlapping, the crops of one elimination pass can carry a match past a neighbour;
    the anchors of a chain are ordered by where they are afterwards, not by their place in the list.  A merged seed-family list is
    where it happens: the case the randomised sweep found (tests/golden/case_family_chain_order.npz: two anchors of 2 and 1
    columns changed places), on the device route (a caller's / merged list of that size goes there with MAUVE_CANON_DEVICE_MIN=1),
    equals the oracle."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    case = os.path.join(root, "tests", "golden", "case_family_chain_order.npz")
    env = dict(os.environ, MAUVE_CANON_DEVICE_MIN="1")
    r = subprocess.run([sys.executable, "-c", ORDER_SCRIPT % {"root": root, "case": case}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr[-3000:]


def test_given_and_merged_lists_take_the_device_route(ctx):
    """A caller's match list (mauve_align_matches) and the merged list of a seed family (S3b) of more than 16 k matches are put where
    the seed pass would have left its own, so that overlap elimination, LCBs, extension and the tail run on the device: same result
    as the oracle (whose elimination works on the dense merged list on the host) and as the host route (MAUVE_GIVEN_ON_HOST is read once
    per process, so the host route is what the small-list tests cover)."""
    from mauvealigner_amd import _lib
    gs = synth.make_config("C3", scale=0.4)
    r = _same_align(ctx, gs, seed_family=1, seed_weight=15)
    assert r["n_mums"] > 16384
    ctx.set_genomes(gs)
    ln, st = ctx.seed_mums(O.get_seed(15, 0), mode=0, mask=(1 << len(gs)) - 1)
    g = ctx.align_matches(_lib.default_params(seed_weight=15), ln, st)
    w = ctx.align(_lib.default_params(seed_weight=15))
    for k in ("anchor_length", "anchor_start", "anchor_lcb", "left", "right", "col_off", "cols", "dp_score", "lcb_weight"):
        assert np.array_equal(g[k], w[k]), k


def test_lcb_extension_on_the_device(ctx):
    """Lists of more than 16 k matches keep their chains on the device; the extension rounds then work on the LCB table alone
    (extend_dev.hip: pieces outside the LCBs gathered into small virtual genomes, the new matches re-chained against the LCBs
    as units, slipped into the device-resident anchor list).  Same result as the oracle's match-level rule: a clean C3, a
    5 %-divergent set where rounds do add anchors, a smaller seed with two rounds, and a C5-shaped pair whose gaps still need
    the recursion afterwards (the anchors then come back to the host)."""
    from mauvealigner_amd import _lib
    gs = synth.make_config("C3", scale=0.4)
    r1 = _same_align(ctx, gs, extend_lcbs=1, seed_weight=15)
    assert r1["n_mums"] > 16384
    gs = synth.star_genomes(4, 1_500_000, 0.05, 77, inversions=24)
    r1 = _same_align(ctx, gs, extend_lcbs=1, seed_weight=15)
    r0 = ctx.align(_lib.default_params(seed_weight=15, extend_lcbs=0))
    assert r1["n_mums"] > 16384 and int(r1["anchor_length"].sum()) > int(r0["anchor_length"].sum()) and r1["n_anchor"] > r0["n_anchor"]
    _same_align(ctx, gs, extend_lcbs=1, seed_weight=13, max_extension_iters=2)
    gs = synth.make_config("C5", scale=0.04)
    r1 = _same_align(ctx, gs, extend_lcbs=1)
    assert r1["n_mums"] > 16384


def test_collinear_extension_takes_the_host_rounds(ctx):
    """--collinear (mauveAligner.cpp:118) runs the greedy step down to one LCB, so the new matches of an extension round can
    outweigh an old LCB; the unit-level re-chaining of the device rounds assumes old LCBs survive, therefore that option keeps
    the match-level rounds on the host.  (Found by the randomised sweep of round 3: the device rounds then indexed with the id
    of a dead LCB.)  Same result as the oracle on a list long enough for the device-resident chains."""
    gs = synth.star_genomes(4, 1_500_000, 0.05, 78, inversions=12)
    r1 = _same_align(ctx, gs, extend_lcbs=1, seed_weight=15, collinear=1)
    assert r1["n_mums"] > 16384 and r1["n_lcb"] == 1


def test_result_ends_with_its_genomes(ctx):
    """mauve_set_genomes ends the result the context holds: what of it is still on the device is not carried over, and a fetch
    afterwards is refused (MAUVE_ERR_STATE) instead of handing out half a result; the next alignment is whole again."""
    from mauvealigner_amd import _lib
    gs = synth.make_config("C3", scale=0.4)
    ctx.set_genomes(gs)
    p = _lib.default_params(seed_weight=15)
    sz = ctx.align(p, fetch=False)
    ctx.set_genomes(gs)
    sz_t = _lib.AlignSizes(**{k: sz[k] for k, _ in _lib.AlignSizes._fields_})
    with pytest.raises(RuntimeError, match=r"\(-5\)"):
        ctx._fetch(sz_t)
    r = ctx.align(p)
    assert r["n_anchor"] == sz["n_anchor"] and len(r["cols"]) == sz["n_cols"]


def test_sharded_contexts(ctx, tmp_path):
    """mauve_set_shard: two processes (gloo), each with its own context on the GPU, make the same calls on the same genomes; the
    gaps of every recursion level, the guide tree's pairwise passes and the intervals of the guide-tree nodes are dealt out and
    exchanged through the caller's all-gather.  Both ranks end with the whole result, bit-identical to a single context's."""
    import socket, subprocess, sys
    from mauvealigner_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
    outs = [str(tmp_path / ("rank%d.npz" % r)) for r in range(2)]
    logs = [str(tmp_path / ("rank%d.log" % r)) for r in range(2)]          # files, not pipes: a child that fills its pipe while the other is drained would block
    procs = [subprocess.Popen([sys.executable, "-m", "tests.shard_worker", str(r), "2", port, outs[r]], cwd=root,
                              stdout=open(logs[r], "w"), stderr=subprocess.STDOUT) for r in range(2)]
    for r, p in enumerate(procs):
        p.wait(timeout=600)
        assert p.returncode == 0, open(logs[r]).read()[-3000:]
    gs = synth.make_config("C5", scale=0.04)
    ctx.set_genomes(gs)
    one = ctx.align(_lib.default_params())
    gs4 = synth.make_config("C4", scale=0.08)
    ctx.set_genomes(gs4)
    onep = ctx.progressive_align(_lib.default_params())
    for o in outs:
        z = np.load(o)
        for k in ("anchor_start", "anchor_length", "cols", "col_off", "dp_score", "left", "right"):
            assert np.array_equal(z["c5_" + k], one[k]), k
        for k in ("cols", "col_off", "dp_score", "left", "right", "reverse", "dist"):
            assert np.array_equal(z["c4_" + k], onep[k]), k
        assert np.array_equal(z["c4_tree"], np.stack(onep["tree"]))
        # the work really was dealt out: recursion batches at C5; the guide tree's pairs and the nodes' intervals at C4
        assert z["c5_exchanges"][0] >= 2 and z["c4_exchanges"][0] >= 3, (z["c5_exchanges"], z["c4_exchanges"])


def test_align_lcbs_resumes_from_an_lcb_table(ctx):
    """mauve_align_lcbs (--lcb-input / --realign-lcb, mauveAligner.cpp:705-744): from the LCB table -- the anchors of a run without
    recursion and gapped alignment, with their LCB ids -- recursion and gapped alignment give the alignment of the whole call;
    begun from the ORACLE's LCB table they give the oracle's whole alignment.  A list that is not a set of collinear chains is
    refused."""
    from mauvealigner_amd import _lib
    keys = ("anchor_start", "anchor_length", "anchor_lcb", "lcb_weight", "left", "right", "reverse", "col_off", "cols", "dp_score")
    for cfg, scale, kw in (("C3", 0.03, {}), ("C5", 0.02, {}), ("C3", 0.4, {"seed_weight": 15})):
        gs = synth.make_config(cfg, scale=scale)
        ctx.set_genomes(gs)
        whole = ctx.align(_lib.default_params(**kw))
        tab = ctx.align(_lib.default_params(recursive=0, gapped=0, **kw))          # seed pass, chaining and LCB extension only
        res = ctx.align_lcbs(_lib.default_params(**kw), tab["anchor_length"], tab["anchor_start"], tab["anchor_lcb"])
        for k in keys:
            assert np.array_equal(res[k], whole[k]), (cfg, k)
        assert res["n_mums"] == 0 and res["n_dp_cells"] == whole["n_dp_cells"]
        if scale <= 0.03:                                                           # ... and from the oracle's table (its own restart)
            ot = O.align(gs, O.default_params(recursive=0, gapped=0, **kw))["aln"]
            ow = O.align(gs, O.default_params(**kw))["aln"]
            perm = np.random.default_rng(3).permutation(len(ot["anchor_length"]))   # any order
            res = ctx.align_lcbs(_lib.default_params(**kw), ot["anchor_length"][perm], ot["anchor_start"][perm], ot["anchor_lcb"][perm])
            for k in ("anchor_start", "anchor_length", "anchor_lcb", "left", "right", "reverse", "col_off", "cols", "dp_score"):
                assert np.array_equal(res[k], ow[k]), (cfg, k)
    ln, st, lc = tab["anchor_length"].copy(), tab["anchor_start"].copy(), tab["anchor_lcb"].copy()
    st[1], st[0] = st[0].copy(), st[1].copy()                                       # two anchors of a chain swapped in every genome but the first...
    st[0, 0], st[1, 0] = st[1, 0], st[0, 0]
    with pytest.raises(RuntimeError, match=r"\(-1\)"):
        ctx.align_lcbs(_lib.default_params(), ln, st, lc)
