"""replay of one saved sweep case around the breakpoint estimate: counts, distances, progressive result"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mauvealigner_amd import _lib
from oracle import pyoracle as O
z = np.load(sys.argv[1]); gs = [z[k] for k in z.files]
kw = eval(sys.argv[2])
ctx = _lib.Context(0); ctx.set_genomes(gs)
w = kw.get('seed_weight') or O.default_seed_weight(sum(len(g) for g in gs) // len(gs))
pat = O.get_seed(w, 0)
for ml in (0, 2 * w, 30):
    a, b = ctx.breakpoint_counts(pat, ml), O.breakpoint_counts(gs, pat, ml)
    print('min_len', ml, 'equal', np.array_equal(a, b)); 
    if not np.array_equal(a, b): print(a); print(b)
d1 = ctx.guide_tree(pat); d2 = O.guide_tree(gs, pat)
print('dist equal', np.array_equal(d1[0], d2[0]), 'tree equal', np.array_equal(d1[1], d2[1]) and np.array_equal(d1[2], d2[2]))
for drop in ([], ['bp_dist_scale_ppm'], ['seed_family'], ['weight_scaling']):
    k2 = {k: v for k, v in kw.items() if k not in drop}
    r = ctx.progressive_align(_lib.default_params(**k2)); e = O.progressive_align(gs, O.default_params(**k2))['aln']
    print('without', drop, 'n_iv', r['n_iv'], e['n_iv'], 'left equal', np.array_equal(r['left'], e['left']))
