"""time of a pass with the seed family (S3b) at a bench configuration: family_time.py <cfg>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mauvealigner_amd import _lib, synth
cfg = sys.argv[1]
gs = synth.make_config(cfg, 1.0)
ctx = _lib.Context(0); ctx.set_genomes(gs)
p = _lib.default_params(seed_weight=15, seed_family=1)
for i in range(3):
    t = time.perf_counter(); r = ctx.align(p, fetch=False); dt = time.perf_counter() - t
    print('%s seed family: %.2f ms' % (cfg, dt * 1e3), ctx.stage_times(), {k: v for k, v in r.items() if k in ('n_mums', 'n_anchor', 'n_lcb')}, flush=True)
