#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz and *.xmfa from the CPU oracle (the reference ships no
fixtures and cannot be built here, so golden vectors come from this repository's own restatement).

Run from the repo root:  python tests/golden/make_golden.py
Each fixture stores the inputs (genomes as 0..3 codes, seed pattern, mode) and the expected outputs
of every stage: MUM list, LCB weights, anchors, interval table, alignment columns, DP scores, XMFA.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mauvealigner_amd import synth  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def one(name, gs, weight, mode):
    pat = O.get_seed(weight, 0)
    ln, st = O.find_matches(gs, pat, mode=mode)
    p = O.default_params(seed_pattern=pat, mode=mode)
    r = O.align(gs, p, names=["g%d" % g for g in range(len(gs))], want_xmfa=True)
    a = r["aln"]
    d = {"nseq": len(gs), "pattern": np.uint64(pat), "mode": mode, "mum_length": ln, "mum_start": st,
         "lcb_weight": r["lcbs"]["weight"], "lcb_left": r["lcbs"]["left_end"], "lcb_right": r["lcbs"]["right_end"],
         "anchor_start": a["anchor_start"], "anchor_length": a["anchor_length"], "anchor_lcb": a["anchor_lcb"],
         "left": a["left"], "right": a["right"], "reverse": a["reverse"], "col_off": a["col_off"],
         "cols": a["cols"], "dp_score": a["dp_score"]}
    for g, x in enumerate(gs):
        d["genome%d" % g] = x
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    with open(os.path.join(OUT, name + ".xmfa"), "w") as f:
        f.write(r["xmfa"])
    print(name, "mums", len(ln), "lcbs", r["lcbs"]["n_lcb"], "anchors", len(a["anchor_length"]), "cols", len(a["cols"]))


def progressive(name, gs, weight):
    """progressiveMauve-shaped fixture: pairwise matches, guide tree, progressive alignment."""
    pat = O.get_seed(weight, 0)
    pln, pst = O.find_matches(gs, pat, mode=O.MODE_PAIRWISE)
    names = ["g%d" % g for g in range(len(gs))]
    r = O.progressive_align(gs, O.default_params(seed_pattern=pat), names=names, want_xmfa=True)
    a = r["aln"]
    d = {"nseq": len(gs), "pattern": np.uint64(pat), "pair_length": pln, "pair_start": pst,
         "tree_left": r["tree"][0], "tree_right": r["tree"][1], "dist": r["dist"],
         "left": a["left"], "right": a["right"], "reverse": a["reverse"], "col_off": a["col_off"],
         "cols": a["cols"], "dp_score": a["dp_score"]}
    for g, x in enumerate(gs):
        d["genome%d" % g] = x
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    with open(os.path.join(OUT, name + ".xmfa"), "w") as f:
        f.write(r["xmfa"])
    print(name, "pairwise matches", len(pln), "blocks", a["n_iv"], "cols", len(a["cols"]))


def main():
    rng = np.random.default_rng(20261003)
    anc = rng.integers(0, 4, 2000, dtype=np.uint8)
    one("g2x2k", [synth.mutate(anc, 0.02, rng), synth.mutate(anc, 0.02, rng)], 9, O.MODE_MEM)
    anc = rng.integers(0, 4, 5000, dtype=np.uint8)
    gs = [synth.mutate(anc, 0.015, rng) for _ in range(3)]
    gs[2] = gs[2].copy()
    gs[2][1500:2700] = synth.revcomp(gs[2][1500:2700])
    one("g3x5k_inv", gs, 9, O.MODE_MEM)
    anc = rng.integers(0, 4, 3000, dtype=np.uint8)
    gs = [synth.mutate(anc, 0.02, rng) for _ in range(5)]
    gs[3] = np.concatenate([gs[3], gs[3][500:900]])     # a duplicated segment: UNIQUE vs MEM differ
    one("g5x3k_unique", gs, 7, O.MODE_UNIQUE)
    # four leaves of a two-level tree with clade-specific inserts and inversions (config C4 in miniature)
    progressive("g4x3k_tree", synth.tree_genomes(4, 3000, 0.02, 77, inv_per_branch=1, insert_per_branch=1, insert_len=(60, 300)), 9)


if __name__ == "__main__":
    main()
