"""replay of one saved align case: where do the anchor tables differ"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mauvealigner_amd import _lib
from oracle import pyoracle as O
z = np.load(sys.argv[1]); gs = [z[k] for k in z.files]
kw = eval(sys.argv[2])
ctx = _lib.Context(0); ctx.set_genomes(gs)
r = ctx.align(_lib.default_params(**kw)); e = O.align(gs, O.default_params(**kw)); a = e['aln']
print('n_lcb', r['n_lcb'], e['lcbs']['n_lcb'], 'n_anchor', r['n_anchor'], len(a['anchor_length']))
rs, es = r['anchor_start'], a['anchor_start']
bad = np.flatnonzero(np.any(rs != es, axis=1) | (r['anchor_length'] != a['anchor_length']) | (r['anchor_lcb'] != a['anchor_lcb']))
print('rows that differ', len(bad), bad[:10])
for i in bad[:6]:
    print(i, 'product', r['anchor_length'][i], rs[i].tolist(), 'lcb', r['anchor_lcb'][i], '| oracle', a['anchor_length'][i], es[i].tolist(), 'lcb', a['anchor_lcb'][i])
if len(bad):
    i = bad[0]
    for j in range(max(0, i - 2), min(len(rs), i + 4)):
        print('  ', j, 'P', r['anchor_length'][j], rs[j].tolist(), r['anchor_lcb'][j], ' O', a['anchor_length'][j], es[j].tolist(), a['anchor_lcb'][j])
print('lcb_weight equal', np.array_equal(r['lcb_weight'], e['lcbs']['weight']), 'left equal', np.array_equal(r['left'], a['left']))
