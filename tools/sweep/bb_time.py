"""Time of mauve_backbone on the alignment a config leaves in the context.  usage: bb_time.py <config> [scale]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mauvealigner_amd import _lib, synth
cfg = sys.argv[1]; scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
gs = synth.make_config(cfg, scale)
ctx = _lib.Context(0); ctx.set_genomes(gs)
p = _lib.default_params(seed_weight=15) if cfg in ('C2', 'C3') else _lib.default_params()
r = ctx.progressive_align(p, fetch=False) if cfg == 'C4' else ctx.align(p, fetch=False)
ctx.backbone()
for i in range(3):
    t = time.perf_counter(); b = ctx.backbone(island_gap=20); dt = time.perf_counter() - t
    print('%s: %d columns, %d intervals -> %d segments, %d islands in %.3f ms' % (cfg, r['n_cols'], r['n_iv'], len(b['seg_iv']), len(b['islands']), dt * 1e3), flush=True)
