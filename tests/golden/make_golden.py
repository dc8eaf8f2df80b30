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


def round2(name):
    """Fixture of the later stages: a seed-family alignment and its backbone / islands, and a
    progressive alignment along a given tree with the node weights scaled by conservation distance (S9, S11b)."""
    rng = np.random.default_rng(20261004)
    anc = rng.integers(0, 4, 6000, dtype=np.uint8)
    gs = [synth.mutate(anc, 0.06, rng, indel_frac=0.3) for _ in range(3)]
    gs[1] = np.concatenate([gs[1][:2000], rng.integers(0, 4, 90, dtype=np.uint8), gs[1][2000:]])      # an island of genome 1
    gs[2] = gs[2].copy(); gs[2][3500:4600] = synth.revcomp(gs[2][3500:4600])
    names = ["g%d" % g for g in range(3)]
    fam = O.align(gs, O.default_params(seed_weight=9, seed_family=1), names=names, want_xmfa=True)
    one_seed = O.align(gs, O.default_params(seed_weight=9))
    a = fam["aln"]
    bb = O.backbone(a["left"], a["right"], a["reverse"], a["col_off"], a["cols"], island_gap=20)
    tree = (np.array([-1, -1, -1, 0, 3], np.int32), np.array([-1, -1, -1, 2, 1], np.int32))            # ((0,2),1): not the UPGMA tree
    pr = O.progressive_align(gs, O.default_params(seed_weight=9, weight_scaling=1, conservation_scale_ppm=500000), names=names, want_xmfa=True, tree=tree)
    d = {"nseq": 3, "seed_weight": 9, "fam_mum_length": fam["mums"][0], "fam_mum_start": fam["mums"][1], "one_seed_mums": len(one_seed["mums"][0]),
         "fam_anchor_start": a["anchor_start"], "fam_anchor_length": a["anchor_length"], "fam_left": a["left"], "fam_right": a["right"],
         "fam_reverse": a["reverse"], "fam_col_off": a["col_off"], "fam_cols": a["cols"],
         "bb_island_gap": 20, "bb_seg_iv": bb["seg_iv"], "bb_seg_col": bb["seg_col"], "bb_seg_len": bb["seg_len"], "bb_seg_mask": bb["seg_mask"],
         "bb_seg_left": bb["seg_left"], "bb_seg_right": bb["seg_right"], "bb_islands": bb["islands"],
         "tree_left": tree[0], "tree_right": tree[1], "prog_left": pr["aln"]["left"], "prog_right": pr["aln"]["right"],
         "prog_reverse": pr["aln"]["reverse"], "prog_col_off": pr["aln"]["col_off"], "prog_cols": pr["aln"]["cols"]}
    for g, x in enumerate(gs):
        d["genome%d" % g] = x
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    with open(os.path.join(OUT, name + ".xmfa"), "w") as f:
        f.write(fam["xmfa"])
    print(name, "family mums", len(fam["mums"][0]), "single-seed mums", len(one_seed["mums"][0]), "segments", len(bb["seg_iv"]), "islands", len(bb["islands"]),
          "progressive blocks", pr["aln"]["n_iv"])


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "round2":          # only the later fixture (the others stay byte for byte)
        round2("g3x6k_round2")
        return
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
    round2("g3x6k_round2")


if __name__ == "__main__":
    main()
