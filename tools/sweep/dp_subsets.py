"""Where does dp_step's time go at C3?  Kernel time (HIP events) of subsets of the C3 DP intervals."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mauvealigner_amd import _lib, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
gs = synth.make_config(cfg, 1.0)
ctx = _lib.Context(0); ctx.set_genomes(gs)
p = _lib.default_params(seed_weight=15)
n_dp, cost, cap = ctx.align_begin(p)
order = np.argsort(-cost, kind="stable")
print("n_dp", n_dp, "cells", int(cost.sum()), "top costs", cost[order[:12]].tolist(), "cap top", cap[order[:12]].tolist())
q = np.quantile(cost, [0.5, 0.9, 0.99, 0.999]); print("quantiles", q.tolist())
def run(name, idx):
    idx = np.sort(idx)
    ctx.profile(True)
    for rep in range(3):
        ctx.profile_reset()
        t = time.perf_counter(); ctx.align_dp(idx, cap); dt = time.perf_counter() - t
        k = ctx.profile_get()["dp_step"]
    print("%-28s n=%6d cells=%10d kernel %.3f ms  wall %.3f ms  %.1f GCUPS" % (name, len(idx), int(cost[idx].sum()), k["ms"], dt * 1e3, cost[idx].sum() / k["ms"] / 1e6))
run("all", order)
for k in (1, 10, 100, 1000, 5000):
    run("all but top %d" % k, order[k:])
run("top 100 only", order[:100])
run("top 1000 only", order[:1000])
run("bottom half", order[n_dp // 2:])
run("top 1", order[:1])
run("top 8", order[:8])
