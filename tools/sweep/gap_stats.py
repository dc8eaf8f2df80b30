"""distribution of the DP intervals of a config (from the oracle's anchors): shapes the DP kernel classes are sized for"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mauvealigner_amd import synth
from oracle import pyoracle as O
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
gs = synth.make_config(cfg, scale)
N = len(gs)
kw = dict(seed_weight=15) if cfg in ("C2", "C3") else {}
r = O.align(gs, O.default_params(**kw))["aln"]
al, ast, alcb = r["anchor_length"], r["anchor_start"], r["anchor_lcb"]
same = alcb[:-1] == alcb[1:]
a0, a1 = ast[:-1], ast[1:]
fwd = a0 > 0
lo = np.where(fwd, a0 + al[:-1, None], -a1 + al[1:, None])
hi = np.where(fwd, a1 - 1, -a0 - 1)
ln = np.maximum(hi - lo + 1, 0)
ln = ln[same]
tot = ln.sum(1); mx = ln.max(1); nonempty = (ln > 0).sum(1)
dp = (tot > 0) & (nonempty >= 2) & (mx <= 10000)
ln = ln[dp]; mx = mx[dp]
print("%s x%.2f: %d anchors, %d dp intervals" % (cfg, scale, len(al), len(ln)))
# cells and profile bound
cells = np.zeros(len(ln), np.int64); m = np.zeros(len(ln), np.int64)
for g in range(N):
    n = ln[:, g]
    first = (m == 0)
    cells += np.where(first, 0, m * n)
    m = np.where(first, n, m + n)
for q in (50, 75, 90, 95, 99, 99.9, 100):
    print("  pct %5.1f: longest %5d  sum %6d  cells %9d" % (q, np.percentile(mx, q), np.percentile(ln.sum(1), q), np.percentile(cells, q)))
order = np.argsort(-cells)
cs = np.cumsum(cells[order])
for frac in (0.001, 0.01, 0.05, 0.1, 0.25, 0.5):
    k = max(1, int(len(order) * frac))
    print("  top %5.1f%% of intervals (%6d) hold %5.1f%% of the cells; smallest of them: longest %d" % (frac * 100, k, 100.0 * cs[k - 1] / cs[-1], mx[order[k - 1]]))
for lim in (8, 16, 32, 64, 128, 256):
    sel = mx <= lim
    print("  longest <= %3d: %6d intervals (%.1f%%), %5.1f%% of cells" % (lim, sel.sum(), 100.0 * sel.mean(), 100.0 * cells[sel].sum() / cells.sum()))
print("  largest 10 (lens):", [ln[i].tolist() for i in order[:10]])
