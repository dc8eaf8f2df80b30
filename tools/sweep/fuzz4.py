"""Randomised GPU-vs-oracle sweep of mauve_align on the round-2 paths: run with MAUVE_CANON_DEVICE_MIN=1 so that small lists,
too, are sorted / chained / assembled on the device (device tail, recursion batches chained on the device, hybrid tail);
options drawn at random incl. banded DP and sum-of-pairs LCB scoring.  usage: fuzz4.py <seed> <seconds> [Lmax]"""
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
it = int(os.environ.get('FUZZ_IT0', '0'))      # start iteration (replay of one case: FUZZ_IT0=n FUZZ_N=1)
it_end = it + int(os.environ.get('FUZZ_N', '1000000000'))
ITF = os.environ.get('FUZZ_IT_FILE')
KEYS = ('anchor_start', 'anchor_length', 'anchor_lcb', 'left', 'right', 'reverse', 'col_off', 'cols', 'dp_score', 'lcb_weight')
while time.time() < t_end and it < it_end:
    if ITF: open(ITF, 'w').write('%d\n' % it)
    rng = np.random.default_rng(seed0 * 100003 + it)
    N = int(rng.integers(2, 6))
    L = int(rng.integers(500, LMAX))
    div = float(rng.choice([0.0, 0.01, 0.03, 0.08, 0.15]))
    anc = rng.integers(0, 4, L, dtype=np.uint8)
    gs = []
    for g in range(N):
        x = synth.mutate(anc, div, rng, indel_frac=float(rng.choice([0.0, 0.1, 0.4])))
        r = rng.random()
        if r < 0.25:
            a = int(rng.integers(0, len(x) - 50)); b = a + int(rng.integers(20, min(max(2000, LMAX // 3), len(x) - a)))
            x = x.copy(); x[a:b] = synth.revcomp(x[a:b])
        elif r < 0.35:
            a = int(rng.integers(0, len(x) - 50)); b = a + int(rng.integers(20, min(1500, len(x) - a)))
            x = np.concatenate([x, x[a:b]])
        elif r < 0.55:                      # a stretch the others do not have, and a hyper-divergent one: long gaps
            p = len(x) // 2
            x = np.concatenate([x[:p], rng.integers(0, 4, int(rng.integers(1, max(800, LMAX // 6))), dtype=np.uint8), x[p:]])
            a = int(rng.integers(0, max(1, len(x) - 900))); x = x.copy(); x[a:a + 800] = synth.mutate(x[a:a + 800], 0.35, rng, indel_frac=0.0)[:len(x[a:a + 800])]
        gs.append(np.ascontiguousarray(x))
    kw = dict(seed_weight=int(rng.choice([0, 7, 9, 11, 13])), recursive=int(rng.random() < 0.7), collinear=int(rng.random() < 0.15),
              add_unaligned=int(rng.integers(0, 2)), gapped=int(rng.random() < 0.9), max_gapped_len=int(rng.choice([10000, 300, 60])),
              max_banded_len=int(rng.choice([0, 0, 5000])), lcb_scoring=int(rng.random() < 0.25))
    if rng.random() < 0.2: kw['min_recursive_gap'] = int(rng.choice([30, 80, 400]))
    for kv in os.environ.get('FUZZ_KW', '').split():
        k_, v_ = kv.split('='); kw[k_] = int(v_)
    what = 'align %s' % kw
    try:
        ctx.set_genomes(gs)
        r = ctx.align(_lib.default_params(**kw))
        e = O.align(gs, O.default_params(**kw))
        a = dict(e['aln']); a['lcb_weight'] = e['lcbs']['weight']
        for k in KEYS:
            assert np.array_equal(r[k], a[k]), (what, k)
        assert r['n_dp_cells'] == a['n_dp_cells']
    except Exception as ex:
        print('FAIL seed', seed0, 'it', it, what, [len(g) for g in gs], repr(ex)[:300], flush=True)
        np.savez('/tmp/fuzz4_fail_%d_%d.npz' % (seed0, it), *gs)
        sys.exit(1)
    it += 1
print('fuzz4 seed %d: %d cases ok' % (seed0, it), flush=True)
