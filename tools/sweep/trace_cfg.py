import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mauvealigner_amd import _lib, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
gs = synth.make_config(cfg, scale)
ctx = _lib.Context(0); ctx.set_genomes(gs)
p = _lib.default_params(seed_weight=15) if cfg in ('C2', 'C3') else _lib.default_params()
for kv in sys.argv[3:]:
    k, v = kv.split('=')
    setattr(p, k, int(v))
tot = sum(len(g) for g in gs)
for i in range(3):
    print('--- pass', i, file=sys.stderr)
    t = time.perf_counter()
    r = ctx.progressive_align(p, fetch=False) if cfg == 'C4' else ctx.align(p, fetch=False)
    dt = time.perf_counter() - t
    print('%s align ms %.3f = %.0f Mbp/s' % (cfg, dt * 1e3, tot / 1e6 / dt), ctx.stage_times(), {k: v for k, v in r.items() if k.startswith('n_')}, file=sys.stderr)
