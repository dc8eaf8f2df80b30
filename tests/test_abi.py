"""CPU-side checks of the product library: it loads, exports every symbol include/mauve_hip.h declares, its
host-side entry points (seeds, packing, chaining) agree with the oracle, and -- without a GPU -- it fails loudly
instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from mauvealigner_amd import _lib, synth
from oracle import pyoracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    L = _lib.load()
    with open(os.path.join(ROOT, "include", "mauve_hip.h")) as f:
        hdr = f.read()
    declared = set(re.findall(r"\b(mauve_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mauve_ctx"}
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_struct_layouts_match_oracle_side():
    # the ctypes mirrors of mauve_params / orc_params must stay field-for-field compatible
    assert [f[0] for f in _lib.Params._fields_] == [f[0] for f in O.Params._fields_]
    assert C.sizeof(_lib.Params) == C.sizeof(O.Params)
    for p, q in ((_lib.default_params(), O.default_params()), (_lib.default_progressive_params(), O.default_progressive_params())):
        for name, _ in _lib.Params._fields_:
            if name != "scoring":
                assert getattr(p, name) == getattr(q, name), name
    # the progressiveMauve call site's option set (progressiveMauve.cpp:578-579,624-637)
    p = _lib.default_progressive_params()
    assert (p.lcb_scoring, p.weight_scaling, p.conservation_scale_ppm, p.bp_dist_scale_ppm, p.refine_rounds) == (1, 1, 500000, 500000, 2)
    p, q = _lib.default_params(), O.default_params()
    assert [list(r) for r in p.scoring.matrix] == [list(r) for r in q.scoring.matrix]
    assert (p.scoring.gap_open, p.scoring.gap_extend) == (-400, -30)


def test_hmm_params_conversion_matches_oracle():
    """mauve_hmm_params_from (host code of the product library, no GPU needed): the call site's three knobs (progressiveMauve.cpp:319-322)
    as the integer scores of DESIGN.md S12b -- same numbers as the oracle's conversion, for the defaults and for other settings."""
    L = _lib.load()
    L.mauve_hmm_params_from.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(_lib.HmmParams)]
    L.mauve_hmm_params_from.restype = None
    for ident, pgh, pgu in ((0.7, 1e-5, 1e-9), (0.9, 1e-3, 1e-3), (0.55, 0.5, 1e-12)):
        h = _lib.HmmParams()
        L.mauve_hmm_params_from(ident, pgh, pgu, C.byref(h))
        o = O.hmm_params(ident, pgh, pgu)
        assert (h.match, h.mismatch, h.gap, h.go_homologous, h.go_unrelated) == (o.match, o.mismatch, o.gap, o.go_homologous, o.go_unrelated)
    h = _lib.HmmParams()
    L.mauve_hmm_params_from(0.7, 1e-5, 1e-9, C.byref(h))
    assert (h.match, h.mismatch, h.gap, h.go_homologous, h.go_unrelated) == (1030, -916, -500, -11513, -20723)
    assert C.sizeof(_lib.HmmParams) == C.sizeof(O.HmmParams) == 20
    # knobs outside their domain have no logarithm to round: the sentinel (positive transition scores) is what mauve_apply_homology* refuse
    for ident, pgh, pgu in ((1.0, 1e-5, 1e-9), (0.25, 1e-5, 1e-9), (0.7, 0.0, 1e-9), (0.7, 1e-5, -1.0), (float("nan"), 1e-5, 1e-9), (0.7, 2.0, 1e-9)):
        L.mauve_hmm_params_from(ident, pgh, pgu, C.byref(h))
        assert h.go_homologous > 0 and h.go_unrelated > 0, (ident, pgh, pgu)


def test_seed_helpers_and_packing():
    for w in range(0, 34):
        for r in (0, 1, 2, 3, _lib.SOLID_SEED, 7, -1):
            assert _lib.get_seed(w, r) == O.get_seed(w, r)
    for L in (0, 1, 2, 49, 200, 1000, 123456, 5_000_000, 10**8, 2**40):
        assert _lib.default_seed_weight(L) == O.default_seed_weight(L)
    pat = _lib.get_seed(15, 0)
    assert _lib.seed_length(pat) == 21 and _lib.seed_weight(pat) == 15
    rng = np.random.default_rng(0)
    for n in (0, 1, 31, 32, 33, 1000):
        c = rng.integers(0, 4, n, dtype=np.uint8)
        w64 = _lib.pack_codes(c)
        assert len(w64) == (n + 31) // 32 + 3
        for i in range(n):
            assert (int(w64[i // 32]) >> (2 * (i % 32))) & 3 == c[i]
        assert np.array_equal(_lib.pack_ascii(synth.to_ascii(c)), w64)
    assert np.array_equal(_lib.pack_ascii(b"acgtNnXA"), _lib.pack_codes(np.array([0, 1, 2, 3, 0, 0, 0, 0], np.uint8)))


@pytest.mark.parametrize("cfg,scale,w", [("C3", 0.02, 11), ("C2", 0.04, 11), ("C4", 0.03, 9)])
def test_host_chaining_equals_oracle(cfg, scale, w):
    gs = synth.make_config(cfg, scale=scale)
    N = len(gs)
    ln, st = O.find_matches(gs, O.get_seed(w, 0))
    ln, st = O.multiplicity_filter(ln, st, N)
    a = O.eliminate_overlaps(ln, st)
    b = _lib.eliminate_overlaps(ln, st)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for mw, col in ((0, False), (3 * w * N, False), (2000, False), (10**7, False), (0, True)):
        d1 = O.compute_lcbs(a[0], a[1], mw, col)
        d2 = _lib.lcb_chain(a[0], a[1], mw, col)
        for k in ("n_lcb", "match_lcb", "left_end", "right_end", "weight", "left_adj", "right_adj"):
            assert np.array_equal(d1[k], d2[k]), (mw, col, k)


def test_host_chaining_random_dense_matches():
    """Host overlap elimination and LCB chaining against the oracle on random match lists that are short, dense and
    heavily overlapping -- the regime (light seeds in small pools) where crops leave matches with equal left ends, so
    the (left end, index) order of S5 matters.  Found by the GPU fuzz sweep; this is its CPU-only form."""
    rng = np.random.default_rng(2026)
    for case in range(400):
        N = int(rng.integers(2, 5))
        n = int(rng.integers(2, 140))
        span = int(rng.integers(30, 1500))
        length = rng.integers(1, int(rng.choice([4, 10, 40])), size=n).astype(np.int64)
        start = np.zeros((n, N), np.int64)
        start[:, 0] = rng.integers(1, span, size=n)
        for g in range(1, N):
            near = rng.random(n) < 0.6                                  # mostly collinear, the rest anywhere
            pos = np.where(near, start[:, 0] + rng.integers(-6, 7, size=n), rng.integers(1, span, size=n))
            pos = np.maximum(pos, 1)
            start[:, g] = np.where(rng.random(n) < 0.15, -pos, pos)
        order = np.lexsort((length, *[start[:, g] for g in range(N - 1, 0, -1)], start[:, 0]))   # canonical-ish input order
        length, start = length[order], start[order]
        a = O.eliminate_overlaps(length, start)
        b = _lib.eliminate_overlaps(length, start)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), case
        for mw, col in ((0, False), (3 * 7 * N, False), (0, True)):
            d1 = O.compute_lcbs(a[0], a[1], mw, col)
            d2 = _lib.lcb_chain(a[0], a[1], mw, col)
            for k in ("n_lcb", "match_lcb", "left_end", "right_end", "weight"):
                assert np.array_equal(d1[k], d2[k]), (case, mw, col, k)


def test_host_chaining_handmade_cases():
    length = np.array([100, 100, 10, 100, 100], dtype=np.int64)
    start = np.array([[1, 1], [201, 201], [401, -5000], [601, 601], [801, 801]], dtype=np.int64)
    d = _lib.lcb_chain(length, start, 50)
    assert d["n_lcb"] == 1 and d["match_lcb"].tolist() == [0, 0, -1, 0, 0] and d["weight"].tolist() == [800]
    d = _lib.lcb_chain(length, start, 10)
    assert d["n_lcb"] == 3 and d["left_end"].tolist() == [[1, 1], [401, -5000], [601, 601]]
    # overlap elimination: shorter match gives way, reverse components crop at the other end
    length = np.array([50, 30], dtype=np.int64)
    start = np.array([[1, 1], [41, -100]], dtype=np.int64)
    l2, s2 = _lib.eliminate_overlaps(length, start)
    e = O.eliminate_overlaps(length, start)
    assert np.array_equal(l2, e[0]) and np.array_equal(s2, e[1])
    assert l2.tolist() == [50, 20] and s2.tolist() == [[1, 1], [51, -100]]
    # empty input
    d = _lib.lcb_chain(np.zeros(0, np.int64), np.zeros((0, 3), np.int64), 10)
    assert d["n_lcb"] == 0
    # a component with NO_MATCH is rejected (N-way input required)
    with pytest.raises(RuntimeError):
        _lib.lcb_chain(np.array([5], np.int64), np.array([[1, 0]], np.int64), 1)


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError) as ei:
        _lib.Context(0)
    assert "no CPU fallback" in str(ei.value) or "HIP" in str(ei.value)


def test_product_never_touches_the_oracle():
    """The product tree must not include, link or import anything under oracle/."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "mauvealigner_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".hpp", ".h", "Makefile")):
                with open(os.path.join(dp, fn), errors="ignore") as f:
                    txt = f.read()
                if re.search(r'#include\s*[<"][^>"]*oracle|pyoracle|liboracle|mauve_oracle\.h|^\s*(from|import)\s+oracle|\borc_[a-z_]+\(',
                             txt, re.M):
                    bad.append(fn)
    assert not bad, bad
