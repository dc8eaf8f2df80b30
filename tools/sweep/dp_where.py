"""Where does the DP time go at a config?  One traced pass (MAUVE_TRACE: class counts, top single-wave step estimates, stage times of
dp_core / the device front) and, for the mauveAligner path, the kernel time of subsets of the DP intervals by size."""
import os, sys, time
os.environ["MAUVE_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mauvealigner_amd import _lib, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
gs = synth.make_config(cfg, scale)
ctx = _lib.Context(0); ctx.set_genomes(gs)
prog = cfg == "C4"
p = _lib.default_progressive_params() if prog else (_lib.default_params(seed_weight=15) if cfg in ("C2", "C3") else _lib.default_params())
for i in range(2):
    print("--- pass", i, file=sys.stderr)
    t = time.perf_counter()
    r = ctx.progressive_align(p, fetch=False) if prog else ctx.align(p, fetch=False)
    print("%s ms %.3f" % (cfg, (time.perf_counter() - t) * 1e3), ctx.stage_times(), {k: v for k, v in r.items() if k.startswith("n_")}, file=sys.stderr)
if not prog:
    n_dp, cost, cap = ctx.align_begin(p)
    order = np.argsort(-cost, kind="stable")
    print("n_dp", n_dp, "cells", int(cost.sum()), "top costs", cost[order[:16]].tolist(), "cap top", cap[order[:16]].tolist())
    print("quantiles 50/90/99/99.9/99.99", np.quantile(cost, [0.5, 0.9, 0.99, 0.999, 0.9999]).tolist())
    cs = np.cumsum(cost[order]); tot = cs[-1]
    for k in (1, 10, 100, 1000, 10000):
        if k <= n_dp:
            print("top %d hold %.1f %% of the cells" % (k, 100.0 * cs[k - 1] / tot))
    def run(name, idx):
        idx = np.sort(idx)
        ctx.profile(True)
        for rep in range(2):
            ctx.profile_reset()
            t = time.perf_counter(); ctx.align_dp(idx, cap); dt = time.perf_counter() - t
            k = ctx.profile_get()["dp_step"]
        print("%-28s n=%7d cells=%11d kernel %.3f ms  wall %.3f ms  %.1f GCUPS" % (name, len(idx), int(cost[idx].sum()), k["ms"], dt * 1e3, cost[idx].sum() / k["ms"] / 1e6))
    os.environ.pop("MAUVE_TRACE", None)
    run("all", order)
    for k in (1, 10, 100, 1000, 10000):
        if k < n_dp:
            run("all but top %d" % k, order[k:])
    run("top 1", order[:1]); run("top 10", order[:10]); run("top 100", order[:100]); run("top 1000", order[:1000])
