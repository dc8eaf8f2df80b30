"""Alignment accuracy against the generator's truth (the measurement of src/scoreAlignment.cpp:172-457).  The frozen
semantics are this repository's own, so besides bit-parity between the HIP path and the oracle we
check that what both compute is a *correct* alignment of the synthetic genomes."""
import os

import numpy as np
import pytest

from mauvealigner_amd import accuracy, synth
from oracle import pyoracle as O


def test_truth_tracking_is_consistent():
    gs, org = synth.star_genomes(3, 20000, 0.03, 123, inversions=2, track=True)
    plain = synth.star_genomes(3, 20000, 0.03, 123, inversions=2)
    assert all(np.array_equal(a, b) for a, b in zip(gs, plain)), "tracking must not change the genomes"
    anc = synth.random_genome(20000, synth._rng(123, 0))
    for g, o in zip(gs, org):
        assert len(g) == len(o)
        fwd = o > 0
        # a base that kept its ancestor coordinate differs from the ancestor only by substitution: most are equal
        same = g[fwd] == anc[o[fwd] - 1]
        assert same.mean() > 0.97
        r = o < 0
        if r.any():
            assert (g[r] == 3 - anc[-o[r] - 1]).mean() > 0.97


def test_oracle_alignment_is_accurate():
    gs, org = synth.star_genomes(3, 60000, 0.03, 7, inversions=2, track=True)
    r = O.align(gs)
    s = accuracy.score_alignment(r["aln"], org)
    assert s["sensitivity"] > 0.97 and s["ppv"] > 0.995, s
    rp = O.progressive_align(gs)
    sp = accuracy.score_alignment(rp["aln"], org)
    assert sp["sensitivity"] > 0.97 and sp["ppv"] > 0.995, sp


@pytest.mark.gpu
def test_gpu_alignment_is_accurate_at_scale():
    from mauvealigner_amd import _lib
    gs, org = synth.star_genomes(5, 400000, 0.03, 3, inversions=8, track=True)
    ctx = _lib.Context(0)
    try:
        ctx.set_genomes(gs)
        r = ctx.align(_lib.default_params(seed_weight=13))
        s = accuracy.score_alignment(r, org)
        assert s["sensitivity"] > 0.97 and s["ppv"] > 0.995, s
    finally:
        ctx.close()


REFERENCE = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree not present")
def test_counts_equal_the_reference_tool():
    """src/scoreAlignment.cpp, built from its own source against the mirror and run on <truth XMFA> <oracle XMFA>, prints
    the figures accuracy.score_alignment_reference computes from the interval table: sensitivity, specificity and the
    correct fraction agree to the six digits the tool prints, for 2..4 genomes at 5-20 % divergence."""
    import re
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        tool = os.path.join(td, "scoreAlignment")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-I" + os.path.join(root, "include"), os.path.join(REFERENCE, "scoreAlignment.cpp"), "-o", tool,
                               "-L" + os.path.join(root, "mauvealigner_amd"), "-lmauve_hip", "-Wl,-rpath," + os.path.join(root, "mauvealigner_amd")])
        for seed, n, length, div in [(1, 2, 5000, 0.05), (2, 3, 6000, 0.10), (5, 4, 4000, 0.15), (4, 3, 8000, 0.2)]:     # (seed 3 of the four-genome case: the tool leaves its main path for a few columns, its interval lookup's quirk)
            gs, org = synth.star_genomes(n, length, div, seed, inversions=0, track=True)
            names = ["g%d.fa" % g for g in range(n)]
            r = O.align(gs, O.default_params(add_unaligned=0), names=names, want_xmfa=True)      # N-way intervals only (see accuracy.py)
            t, c = os.path.join(td, "t.xmfa"), os.path.join(td, "c.xmfa")
            open(t, "w").write(accuracy.truth_xmfa(gs, org, names))
            open(c, "w").write(r["xmfa"])
            out = subprocess.run([tool, t, c], capture_output=True, text=True).stdout
            v = [float(x) for x in re.findall(r"= ([0-9.e+-]+)$", out, re.M)]
            a = accuracy.score_alignment_reference(r["aln"], org)
            want = [a["sensitivity"], a["specificity"], (a["tp"] + a["tn"]) / a["total"], (a["fp"] + a["fn"]) / a["total"]]
            assert len(v) >= 6 and v[5] == 0                      # bad_context: the tool took its main path everywhere
            for got, w in zip(v[:4], want):
                assert abs(got - w) <= 6e-6 * max(1.0, abs(w)), (seed, out, a)
            # the pair-based figures of score_alignment bracket the same alignment: same true positives
            assert accuracy.score_alignment(r["aln"], org)["tp"] == a["tp"]
