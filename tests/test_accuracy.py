"""Alignment accuracy against the generator's truth (the measurement of src/scoreAlignment.cpp:172-457).  The frozen
semantics are this repository's own, so besides bit-parity between the HIP path and the oracle we
check that what both compute is a *correct* alignment of the synthetic genomes."""
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
