"""GPU idle gaps inside one pass of the hot path: from a rocprofv3 kernel trace, the time between the end of a kernel and the start of the next one
(host round trips, launch latency), the largest of them with the kernels on either side.
usage (GPU box):  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python3 tools/sweep/gaps.py C3 ; python3 tools/sweep/gaps.py --read <dir>"""
import csv, glob, os, sys, time
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    def short(n):
        n = n.replace("(anonymous namespace)::", "").replace("devscan::", "").replace("void ", "")
        return n.split("(")[0][:44]
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows]
    mc = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*memory_copy_trace.csv"), recursive=True))
    if mc:
        for r in csv.DictReader(open(mc[-1])):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", "")[-14:]))
    ev.sort()
    # the last pass: from the last seed_extract_all launch that is followed by an as_fill
    starts = [i for i, e in enumerate(ev) if "seed_extract_all" in e[2]]
    fills = [i for i, e in enumerate(ev) if "as_fill" in e[2]]
    i1 = fills[-1]; i0 = max(s for s in starts if s < i1 and ev[s][1] - ev[s][0] > 50000)
    seg = ev[i0:i1 + 1]
    span = seg[-1][1] - seg[0][0]; busy = 0; gaps = []; cur_end = seg[0][0]
    for s, e, n in seg:
        if s > cur_end: gaps.append((s - cur_end, prev, n))
        busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e); prev = n
    print("pass: %.3f ms from the main extract to as_fill, %d events, busy %.3f ms, idle %.3f ms in %d gaps" % (span / 1e6, len(seg), busy / 1e6, (span - busy) / 1e6, len(gaps)))
    for g, a, b in sorted(gaps, reverse=True)[:45]:
        print("  %8.1f us   %-42s -> %s" % (g / 1e3, a, b))
    if len(sys.argv) > 3 and sys.argv[3] == "--timeline":     # every event of the pass in order: start (us from the first), duration, gap in front of it
        cur_end = seg[0][0]
        for s, e, n in seg:
            print("  t=%8.1f  dur %7.1f  gap %6.1f  %s" % ((s - seg[0][0]) / 1e3, (e - s) / 1e3, max(0, s - cur_end) / 1e3, n))
            cur_end = max(cur_end, e)
    small = [g for g, a, b in gaps if g < 5000]
    print("gaps < 5 us: %d, %.3f ms;  5-15 us: %d, %.3f ms;  >= 15 us: %d, %.3f ms" % (len(small), sum(small) / 1e6, len([g for g, _, _ in gaps if 5000 <= g < 15000]),
          sum(g for g, _, _ in gaps if 5000 <= g < 15000) / 1e6, len([g for g, _, _ in gaps if g >= 15000]), sum(g for g, _, _ in gaps if g >= 15000) / 1e6))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mauvealigner_amd import _lib, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
gs = synth.make_config(cfg, 1.0)
ctx = _lib.Context(0); ctx.set_genomes(gs)
p = _lib.default_params(seed_weight=15) if cfg in ("C2", "C3") else _lib.default_params()
for i in range(4):
    t = time.perf_counter(); r = ctx.align(p, fetch=False)
    print("%s ms %.3f" % (cfg, (time.perf_counter() - t) * 1e3), ctx.stage_times())
