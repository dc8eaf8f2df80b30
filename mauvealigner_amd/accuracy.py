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


def score_alignment_reference(aln, origins):
    """The four counts of src/scoreAlignment.cpp as its code (not its comments) assigns them (:262-457): every base i of
    the truth is looked up in the calculated alignment against every other sequence j; a truth base-base pair is visited
    once (from the lower sequence index), a truth base-gap pair from the side that has the base.
      truth base-base:  calculated the same base -> TP; another base -> FP; a gap inside an interval -> FP (:431-434);
                        i in no calculated interval shared with j -> FN ("unaligned", :349-353)
      truth base-gap:   calculated a base -> FN (:424-425); a gap or unaligned -> TN
    -> dict(tp, tn, fp, fn, total, sensitivity = TP/(TP+FN), specificity = TN/(TN+FP)), the tool's first two lines.
    tests/test_accuracy.py runs the tool itself (built from its source on the mirror) and compares all four counts; the
    tool's interval lookup misbehaves when an interval lacks a sequence, so it is run on N-way intervals only."""
    N = len(origins)
    left, right, rev = np.asarray(aln["left"]), np.asarray(aln["right"]), np.asarray(aln["reverse"])
    col_off, cols = np.asarray(aln["col_off"]), np.asarray(aln["cols"])
    # calculated partner of every base of i in j (-1: gap or not aligned with j); whether i sits in an interval shared with j
    P = {(i, j): np.full(len(origins[i]), -1, np.int64) for i in range(N) for j in range(N) if i != j}
    inside = {(i, j): np.zeros(len(origins[i]), bool) for i in range(N) for j in range(N) if i != j}
    for iv in range(left.shape[0]):
        gs = [g for g in range(N) if left[iv, g]]
        if len(gs) < 2:
            continue
        c = cols[col_off[iv]:col_off[iv + 1]]
        pos = {g: _positions(c, g, int(left[iv, g]), int(right[iv, g]), bool(rev[iv, g])) for g in gs}
        for g in gs:
            for h in gs:
                if g == h:
                    continue
                both = (pos[g] >= 0) & (pos[h] >= 0)
                P[(g, h)][pos[g][both]] = pos[h][both]
                inside[(g, h)][int(left[iv, g]) - 1:int(right[iv, g])] = True
    tp = tn = fp = fn = 0
    for i in range(N):
        oi = np.abs(origins[i])
        for j in range(N):
            if i == j:
                continue
            oj = np.abs(origins[j])
            where = np.full(int(max(oi.max(), oj.max())) + 1, -1, np.int64)
            nz = np.flatnonzero(oj)
            where[oj[nz]] = nz
            T = np.where(oi > 0, where[oi], -1)                 # truth partner of every base of i in j
            p, ins = P[(i, j)], inside[(i, j)]
            has = T >= 0
            if i < j:
                tp += int(np.count_nonzero(has & (p == T)))
                fp += int(np.count_nonzero(has & (p >= 0) & (p != T))) + int(np.count_nonzero(has & (p < 0) & ins))
                fn += int(np.count_nonzero(has & (p < 0) & ~ins))
            fn += int(np.count_nonzero(~has & (p >= 0)))
            tn += int(np.count_nonzero(~has & (p < 0)))
    total = tp + tn + fp + fn
    return {"tp": tp, "tn": tn, "fp": fp, "fn": fn, "total": total, "sensitivity": tp / max(tp + fn, 1), "specificity": tn / max(tn + fp, 1)}


def truth_xmfa(genomes, origins, names):
    """The generator's truth as the XMFA file scoreAlignment takes as its <correct alignment>: one block, every sequence
    forward from base 1 -- so only for genome sets without rearrangements.  Bases that share an ancestor coordinate
    share a column; bases inserted after the split get columns of their own behind the ancestor column they follow."""
    N = len(genomes)
    if any((np.asarray(o) < 0).any() for o in origins):
        raise ValueError("truth_xmfa: the genomes carry inversions; one forward block cannot hold them")
    keys = []
    for g in range(N):
        o = np.asarray(origins[g], np.int64)
        last = np.maximum.accumulate(np.where(o > 0, o, 0))          # the ancestor column an inserted base follows
        ins = o == 0
        run = np.zeros(len(o), np.int64)                             # 1, 2, .. within a run of inserted bases
        idx = np.arange(len(o))
        start = np.maximum.accumulate(np.where(~ins, idx, -1))
        run[ins] = (idx - start)[ins]
        keys.append(np.stack([last, np.where(ins, 1 + g, 0), run, np.full(len(o), g), idx], axis=1))
    allk = np.concatenate(keys)
    order = np.lexsort((allk[:, 2], allk[:, 1], allk[:, 0]))
    allk = allk[order]
    newcol = np.ones(len(allk), bool)
    newcol[1:] = np.any(allk[1:, :3] != allk[:-1, :3], axis=1)
    col = np.cumsum(newcol) - 1
    ncol = int(col[-1]) + 1
    out = ["#FormatVersion Mauve1\n"]
    for g in range(N):
        row = np.full(ncol, ord("-"), np.uint8)
        sel = allk[:, 3] == g
        row[col[sel]] = np.frombuffer(b"ACGT", np.uint8)[np.asarray(genomes[g])[allk[sel, 4]]]
        text = row.tobytes().decode()
        out.append("> %d:1-%d + %s\n" % (g + 1, len(genomes[g]), names[g]))
        out.extend(text[p:p + 80] + "\n" for p in range(0, ncol, 80))
    out.append("=\n")
    return "".join(out)
