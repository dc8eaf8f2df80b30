import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mauvealigner_amd import _lib, synth
from oracle import pyoracle as O
gs = synth.make_config("C4", 1.0)
ctx = _lib.Context(0); ctx.set_genomes(gs)
pat = O.get_seed(O.default_seed_weight(sum(len(g) for g in gs) // len(gs)), 0)
for i in range(4):
    ctx.guide_tree(pat)
