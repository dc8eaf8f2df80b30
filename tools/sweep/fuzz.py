"""Randomised GPU-vs-oracle parity sweep (seed pass in all modes, align, progressive align)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mauvealigner_amd import _lib, synth
from oracle import pyoracle as O

ctx = _lib.Context(0)
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
LMAX = int(sys.argv[3]) if len(sys.argv) > 3 else 6000
t_end = time.time() + budget
it = int(os.environ.get('FUZZ_IT0', '0'))      # start iteration (replay of one case: FUZZ_IT0=n FUZZ_N=1)
it_end = it + int(os.environ.get('FUZZ_N', '1000000000'))
ITF = os.environ.get('FUZZ_IT_FILE')
KEYS = ('anchor_start', 'anchor_length', 'left', 'right', 'reverse', 'col_off', 'cols', 'dp_score')
while time.time() < t_end and it < it_end:
    if ITF: open(ITF, 'w').write('%d\n' % it)
    rng = np.random.default_rng(seed0 * 100003 + it)
    N = int(rng.integers(2, 7))
    L = int(rng.integers(150, LMAX))
    div = float(rng.choice([0.0, 0.01, 0.03, 0.08, 0.2]))
    anc = rng.integers(0, 4, L, dtype=np.uint8)
    gs = []
    for g in range(N):
        x = synth.mutate(anc, div, rng, indel_frac=float(rng.choice([0.0, 0.1, 0.4])))
        r = rng.random()
        if r < 0.25 and len(x) > 100:
            a = int(rng.integers(0, len(x) - 50)); b = a + int(rng.integers(20, min(max(2000, LMAX // 3), len(x) - a)))
            x = x.copy(); x[a:b] = synth.revcomp(x[a:b])
        elif r < 0.35 and len(x) > 100:
            a = int(rng.integers(0, len(x) - 50)); b = a + int(rng.integers(20, min(1500, len(x) - a)))
            x = np.concatenate([x, x[a:b]])
        elif r < 0.45:
            x = np.concatenate([x[: len(x) // 2], rng.integers(0, 4, int(rng.integers(1, max(800, LMAX // 4))), dtype=np.uint8), x[len(x) // 2:]])
        elif r < 0.5:
            x = synth.revcomp(x)
        elif r < 0.53:
            x = x[: int(rng.integers(0, 30))]
        gs.append(np.ascontiguousarray(x))
    w = int(rng.choice([5, 7, 9, 11, 13, 15, 17, 21]))
    pat = O.get_seed(w, int(rng.integers(0, 3)))
    what = 'setup'
    try:
        ctx.set_genomes(gs)
        for mode in (0, 1, 2):
            for ext in (True, False):
                what = 'seed mode %d ext %s' % (mode, ext)
                ln, st = ctx.seed_mums(pat, mode=mode, extend=ext)
                eln, est = O.find_matches(gs, pat, mode=mode, extend=ext)
                assert np.array_equal(ln, eln) and np.array_equal(st, est), what
        what = 'seed masked'
        mask = (1 << N) - 1
        ln, st = ctx.seed_mums(pat, mode=0, mask=mask)
        eln, est = O.find_matches(gs, pat, mode=0, mask=mask)
        assert np.array_equal(ln, eln) and np.array_equal(st, est)
        if min(len(g) for g in gs) >= 40:
            kw = dict(seed_weight=int(rng.choice([0, 7, 9, 11])), mode=int(rng.integers(0, 2)), recursive=int(rng.integers(0, 2)),
                      collinear=int(rng.random() < 0.2), add_unaligned=int(rng.integers(0, 2)), extend_lcbs=int(rng.random() < 0.5))
            for kv in os.environ.get('FUZZ_KW', '').split():
                k_, v_ = kv.split('='); kw[k_] = int(v_)
            what = 'align %s' % kw
            r = ctx.align(_lib.default_params(**kw))
            e = O.align(gs, O.default_params(**kw))['aln']
            assert all(np.array_equal(r[k], e[k]) for k in KEYS), what
            if N >= 3 and it % 3 == 0:
                what = 'progressive'
                r = ctx.progressive_align(_lib.default_params())
                e = O.progressive_align(gs)['aln']
                assert all(np.array_equal(r[k], e[k]) for k in ('left', 'right', 'reverse', 'col_off', 'cols', 'dp_score')), what
    except Exception as ex:
        print('FAIL it=%d seed0=%d N=%d L=%d div=%.2f w=%d lens=%s at %s: %r' % (it, seed0, N, L, div, w, [len(g) for g in gs], what, ex), flush=True)
        sys.exit(1)
    it += 1
    if it % 50 == 0:
        print('ok', it, flush=True)
print('FUZZ OK: %d cases' % it)
