"""time of the homology pass (mauve_apply_homology) on a bench configuration: hom_time.py <cfg> [scale]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mauvealigner_amd import _lib, synth
cfg = sys.argv[1]; scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
gs = synth.make_config(cfg, scale)
ctx = _lib.Context(0); ctx.set_genomes(gs)
p = _lib.default_params(seed_weight=15) if cfg in ('C2', 'C3') else _lib.default_params()
for i in range(3):
    sz = ctx.progressive_align(p, fetch=False) if cfg == 'C4' else ctx.align(p, fetch=False)
    t = time.perf_counter(); r = ctx.apply_homology(fetch=False); dt = time.perf_counter() - t
    t2 = time.perf_counter(); b = ctx.backbone(); dt2 = time.perf_counter() - t2
    print('%s: %d intervals, %d columns: homology pass %.2f ms (moved %d, columns now %d), backbone %.2f ms' % (cfg, sz['n_iv'], sz['n_cols'], dt * 1e3, r['n_moved'], r['n_cols'], dt2 * 1e3), flush=True)
