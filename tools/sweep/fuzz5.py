"""Randomised GPU-vs-oracle sweep of this round's late additions: LCB extension (incremental rounds), the backbone /
island stage on the resident alignment (device tail and host tail), progressive alignment along random guide trees.
usage: fuzz5.py <seed> <seconds> [Lmax]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mauvealigner_amd import _lib, synth
from oracle import pyoracle as O

ctx = _lib.Context(0)
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
LMAX = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
t_end = time.time() + budget
it = int(os.environ.get('FUZZ_IT', '0'))
ONLY = 'FUZZ_IT' in os.environ
it_end = it + int(os.environ.get('FUZZ_N', '1000000000'))
ITF = os.environ.get('FUZZ_IT_FILE')
KEYS = ('anchor_start', 'anchor_length', 'anchor_lcb', 'left', 'right', 'reverse', 'col_off', 'cols', 'dp_score', 'lcb_weight')
BK = ('seg_iv', 'seg_col', 'seg_len', 'seg_mask', 'seg_left', 'seg_right', 'islands')


def random_tree(N, rng):
    left = np.full(2 * N - 1, -1, np.int32); right = np.full(2 * N - 1, -1, np.int32)
    roots = list(range(N))
    for k in range(N, 2 * N - 1):
        a, b = rng.choice(len(roots), 2, replace=False)
        left[k], right[k] = roots[a], roots[b]
        roots = [x for i, x in enumerate(roots) if i not in (a, b)] + [k]
    return left, right


n_ext = n_tree = n_bb = n_hom = n_moved = 0
while time.time() < t_end and it < it_end:
    if ITF: open(ITF, 'w').write('%d\n' % it)
    if os.environ.get('FUZZ_IT_FILE'): open(os.environ['FUZZ_IT_FILE'], 'w').write('%d\n' % it)
    rng = np.random.default_rng(seed0 * 100019 + it)
    N = int(rng.integers(2, 6))
    L = int(rng.integers(800, LMAX))
    div = float(rng.choice([0.0, 0.01, 0.03, 0.08]))
    anc = rng.integers(0, 4, L, dtype=np.uint8)
    gs = []
    for g in range(N):
        x = synth.mutate(anc, div, rng, indel_frac=float(rng.choice([0.0, 0.1, 0.4])))
        for _ in range(int(rng.integers(0, 4))):                    # inversions: several LCBs, room between them for the extension
            a = int(rng.integers(0, len(x) - 60)); b = a + int(rng.integers(30, min(max(3000, LMAX // 4), len(x) - a)))
            x = x.copy(); x[a:b] = synth.revcomp(x[a:b])
        if rng.random() < 0.5:                                      # islands: a stretch the others lack, a divergent stretch
            p = int(rng.integers(1, len(x) - 1))
            x = np.concatenate([x[:p], rng.integers(0, 4, int(rng.integers(1, 400)), dtype=np.uint8), x[p:]])
        if rng.random() < 0.4:
            a = int(rng.integers(0, max(1, len(x) - 700))); x = x.copy(); x[a:a + 600] = synth.mutate(x[a:a + 600], 0.3, rng, indel_frac=0.0)[:len(x[a:a + 600])]
        gs.append(np.ascontiguousarray(x))
    mode = rng.random()
    what = ''
    try:
        ctx.set_genomes(gs)
        if mode < 0.3 and N >= 3:
            kw = dict(seed_weight=int(rng.choice([0, 9, 11])), recursive=int(rng.random() < 0.7), lcb_scoring=int(rng.random() < 0.3), seed_family=int(rng.random() < 0.4),
                      weight_scaling=int(rng.random() < 0.4), conservation_scale_ppm=int(rng.choice([0, 300000, 500000, 1000000])),
                      max_gapped_len=int(rng.choice([10000, 300])), refine_rounds=int(rng.choice([0, 0, 1, 2, 5])),
                      bp_dist_scale_ppm=int(rng.choice([0, 500000, 1000000])), bp_dist_min_score=int(rng.choice([-1, 0, 30])))
            t = random_tree(N, rng) if rng.random() < 0.8 else None
            what = 'progressive %s tree %s' % (kw, None if t is None else (t[0].tolist(), t[1].tolist()))
            r = ctx.progressive_align(_lib.default_params(**kw), tree=t)
            a = O.progressive_align(gs, O.default_params(**kw), tree=t)['aln']
            if ONLY:
                print('n_iv', r['n_iv'], a['n_iv'], flush=True)
                m_ = min(len(r['left']), len(a['left']))
                bad = [i for i in range(m_) if not (np.array_equal(r['left'][i], a['left'][i]) and np.array_equal(r['right'][i], a['right'][i]))]
                print('first differing intervals', bad[:4], [(r['left'][i].tolist(), r['right'][i].tolist(), a['left'][i].tolist(), a['right'][i].tolist()) for i in bad[:3]], flush=True)
            for k in ('left', 'right', 'reverse', 'col_off', 'cols', 'dp_score'):
                assert np.array_equal(r[k], a[k]), (what, k)
            n_tree += 1
        else:
            kw = dict(seed_weight=int(rng.choice([0, 9, 11, 13])), recursive=int(rng.random() < 0.7), collinear=int(rng.random() < 0.1),
                      add_unaligned=int(rng.integers(0, 2)), extend_lcbs=int(rng.random() < 0.75), seed_family=int(rng.random() < 0.4), max_extension_iters=int(rng.choice([1, 4, 4, 6])),
                      max_gapped_len=int(rng.choice([10000, 300])))
            if rng.random() < 0.3: kw['lcb_weight'] = int(rng.choice([20, 60, 200, 1000])) * N
            for kv in os.environ.get('FUZZ_KW', '').split():
                k_, v_ = kv.split('='); kw[k_] = int(v_)
            what = 'align %s' % kw
            r = ctx.align(_lib.default_params(**kw), fetch=False)
            gap = int(rng.choice([0, 2, 20, 100]))
            first = rng.random() < 0.5
            if first: b = ctx.backbone(island_gap=gap)              # before the fetch: the columns may still be in HBM only
            r = ctx.align(_lib.default_params(**kw)) if first else ctx._fetch(ctx.last_sizes) if hasattr(ctx, 'last_sizes') else ctx.align(_lib.default_params(**kw))
            if not first: b = ctx.backbone(island_gap=gap)
            e = O.align(gs, O.default_params(**kw))
            a = dict(e['aln']); a['lcb_weight'] = e['lcbs']['weight']
            if ONLY:
                eml, ems = O.multiplicity_filter(e['mums'][0], e['mums'][1], N)
                print('mums', len(eml), len(r['mum_length']), np.array_equal(eml, r['mum_length']) and np.array_equal(ems, r['mum_start']), flush=True)
                if len(eml) == len(r['mum_length']):
                    bad = np.flatnonzero((eml != r['mum_length']) | np.any(ems != r['mum_start'], axis=1))
                    print('first diffs', bad[:5], [(eml[i], ems[i].tolist(), r['mum_length'][i], r['mum_start'][i].tolist()) for i in bad[:3]], flush=True)
                else:
                    so = set(map(tuple, np.column_stack([eml, ems]).tolist())); sp = set(map(tuple, np.column_stack([r['mum_length'], r['mum_start']]).tolist()))
                    print('only oracle', sorted(so - sp)[:5], 'only product', sorted(sp - so)[:5], flush=True)
            if ONLY:
                so = set(map(tuple, np.column_stack([a['anchor_length'], a['anchor_start']]).tolist())); sp = set(map(tuple, np.column_stack([r['anchor_length'], r['anchor_start']]).tolist()))
                print('anchors', len(so), len(sp), 'only oracle', sorted(so - sp)[:6], 'only product', sorted(sp - so)[:6], flush=True)
            for k in KEYS:
                assert np.array_equal(r[k], a[k]), (what, k)
            n_ext += kw['extend_lcbs']
        eb = O.backbone(a['left'], a['right'], a['reverse'], a['col_off'], a['cols'], island_gap=20 if mode < 0.3 and N >= 3 else gap)
        if mode < 0.3 and N >= 3: b = ctx.backbone(island_gap=20)
        for k in BK:
            assert b[k].shape == eb[k].shape and np.array_equal(b[k], eb[k]), (what, 'backbone', k)
        n_bb += 1
        if rng.random() < 0.5:                                      # homology pass (S12b) on the result the context holds, then its backbone
            hk = {} if rng.random() < 0.5 else dict(identity=float(rng.choice([0.6, 0.8, 0.95])), pgh=float(rng.choice([1e-5, 1e-2])), pgu=float(rng.choice([1e-9, 1e-3])))
            if rng.random() < 0.3: hk['gap'] = int(rng.choice([-100, -2000]))
            what += ' + homology %s' % hk
            r2 = ctx.apply_homology(ctx.hmm_params(**hk))
            off, hcols, moved = O.homology_apply(gs, a['left'], a['right'], a['reverse'], a['col_off'], a['cols'], O.hmm_params(**hk))
            assert r2['n_moved'] == moved and np.array_equal(r2['col_off'], off) and np.array_equal(r2['cols'], hcols), (what, 'homology')
            g2 = int(rng.choice([0, 20]))
            b2 = ctx.backbone(island_gap=g2)
            eb2 = O.backbone(a['left'], a['right'], a['reverse'], off, hcols, island_gap=g2)
            for k in BK:
                assert b2[k].shape == eb2[k].shape and np.array_equal(b2[k], eb2[k]), (what, 'backbone after homology', k)
            n_hom += 1; n_moved += moved > 0
    except Exception as ex:
        print('FAIL seed', seed0, 'it', it, what, [len(g) for g in gs], repr(ex)[:400], flush=True)
        np.savez('/tmp/fuzz5_fail_%d_%d.npz' % (seed0, it), *gs)
        sys.exit(1)
    it += 1
    if ONLY: break
print('fuzz5 seed %d: %d cases ok (%d with extension, %d along trees, %d backbones, %d homology passes of which %d moved residues)' % (seed0, it, n_ext, n_tree, n_bb, n_hom, n_moved), flush=True)
