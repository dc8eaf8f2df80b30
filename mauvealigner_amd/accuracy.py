"""Accuracy of an alignment against the generator's truth -- the measurement of the reference's hand-run harness
`scoreAlignment <correct xmfa> <calculated xmfa>` (src/scoreAlignment.cpp:99-104,172-457: true/false positive pairs
-> sensitivity and positive predictive value), computed from the interval table instead of two XMFA files.

Truth: synth.star_genomes(..., track=True) gives, for every base of every genome, its signed 1-based ancestor
coordinate (0 = inserted after the split, negative = on the reverse strand of the ancestor).  Two bases of different
genomes are homologous iff they share the ancestor coordinate; an aligned pair is correct iff it is homologous and
the strands agree with the interval's orientation flags.
"""
import numpy as np


def _positions(cols, g, left, right, rev):
    """0-based genome position of genome g's residue in each column of one interval (-1 where absent)."""
    present = ((cols >> np.uint32(g)) & np.uint32(1)).astype(bool)
    k = np.cumsum(present) - 1
    pos = np.where(present, (right - 1 - k) if rev else (left - 1 + k), -1)
    return pos.astype(np.int64)


def score_alignment(aln, origins):
    """aln: dict with left, right, reverse, col_off, cols (Context.align / progressive_align / oracle result).
    Returns dict(tp, fp, fn, sensitivity, ppv) over all genome pairs."""
    N = len(origins)
    left, right, rev = np.asarray(aln["left"]), np.asarray(aln["right"]), np.asarray(aln["reverse"])
    col_off, cols = np.asarray(aln["col_off"]), np.asarray(aln["cols"])
    tp = fp = 0
    for iv in range(left.shape[0]):
        gs = [g for g in range(N) if left[iv, g]]
        if len(gs) < 2:
            continue
        c = cols[col_off[iv]:col_off[iv + 1]]
        pos = {g: _positions(c, g, int(left[iv, g]), int(right[iv, g]), bool(rev[iv, g])) for g in gs}
        for a in range(len(gs)):
            for b in range(a + 1, len(gs)):
                g, h = gs[a], gs[b]
                both = (pos[g] >= 0) & (pos[h] >= 0)
                if not both.any():
                    continue
                og, oh = origins[g][pos[g][both]], origins[h][pos[h][both]]
                flip = bool(rev[iv, g]) != bool(rev[iv, h])
                ok = (og != 0) & (np.abs(og) == np.abs(oh)) & ((np.sign(og) * np.sign(oh) < 0) == flip)
                tp += int(ok.sum())
                fp += int((~ok).sum())
    total = 0
    sets = [np.unique(np.abs(o[o != 0])) for o in origins]
    for g in range(N):
        for h in range(g + 1, N):
            total += len(np.intersect1d(sets[g], sets[h], assume_unique=True))
    fn = total - tp
    return {"tp": tp, "fp": fp, "fn": fn, "sensitivity": tp / max(total, 1), "ppv": tp / max(tp + fp, 1)}
