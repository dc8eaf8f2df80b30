"""The largest DP intervals of a config one at a time: kernel time of each alone, and (MAUVE_TRACE) what dp_core says about it.  dp_top.py <cfg> [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mauvealigner_amd import _lib, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
gs = synth.make_config(cfg, 1.0)
ctx = _lib.Context(0); ctx.set_genomes(gs)
p = _lib.default_params(seed_weight=15) if cfg in ("C2", "C3") else _lib.default_params()
n_dp, cost, cap = ctx.align_begin(p)
order = np.argsort(-cost, kind="stable")
ctx.profile(True)
for t in range(k):
    idx = np.sort(order[t:t + 1])
    for rep in range(2):
        ctx.profile_reset(); ctx.align_dp(idx, cap); ms = ctx.profile_get()["dp_step"]["ms"]
    print("interval %d: cost %d cap %d kernel %.3f ms" % (t, cost[order[t]], cap[order[t]], ms), flush=True)
