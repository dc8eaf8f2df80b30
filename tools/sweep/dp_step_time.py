"""per-step time of the one-wave DP sweep: T(m = 128) - T(m = 64) over n + 64 steps (the walk differs by 64 ops only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mauvealigner_amd import _lib
rng = np.random.default_rng(3)
ctx = _lib.Context(0); ctx.profile(True)
def t(m, n):
    a = rng.integers(0, 4, m, dtype=np.uint8); b = rng.integers(0, 4, n, dtype=np.uint8)
    best = 1e9
    for rep in range(5):
        ctx.profile_reset(); ctx.dp_batch([[a, b]]); best = min(best, ctx.profile_get()["dp_step"]["ms"])
    return best
for n in (2000,):
    t64, t128, t192 = t(64, n), t(128, n), t(192, n)
    print("n=%d: T64 %.3f T128 %.3f T192 %.3f ms -> per step %.1f ns (2nd stripe), %.1f ns (3rd)" % (n, t64, t128, t192, (t128 - t64) * 1e6 / (n + 64), (t192 - t128) * 1e6 / (n + 64)))
