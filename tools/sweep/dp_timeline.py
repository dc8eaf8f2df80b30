"""Timeline of the DP launches of one pass (do the workgroup launch on the second stream and the one-wave launch overlap?).
usage (GPU box):  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/sweep/dp_timeline.py C5 ; then
                  python3 tools/sweep/dp_timeline.py --read <dir>"""
import csv, glob, os, sys, time
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(f)) if "dp_step" in r["Kernel_Name"] or "dp_gather" in r["Kernel_Name"]]
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    for r in rows[-12:]:
        print("%-40s start %10.3f us  end %10.3f us  dur %9.3f us  grid %s wg %s" % (r["Kernel_Name"][:40], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                                                   (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size"), r.get("Workgroup_Size")))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mauvealigner_amd import _lib, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
gs = synth.make_config(cfg, 1.0)
ctx = _lib.Context(0); ctx.set_genomes(gs)
prog = cfg == "C4"
p = _lib.default_progressive_params() if prog else (_lib.default_params(seed_weight=15) if cfg in ("C2", "C3") else _lib.default_params())
for i in range(3):
    t = time.perf_counter()
    r = ctx.progressive_align(p, fetch=False) if prog else ctx.align(p, fetch=False)
    print("%s ms %.3f" % (cfg, (time.perf_counter() - t) * 1e3), ctx.stage_times())
