#!/bin/bash
# per-kernel times of a few passes of one config (GPU box): tools/sweep/kstats.sh <out-file> <cfg> [scale]
# rocprofv3 kernel trace of tools/sweep/trace_cfg.py; prints calls, total and average per kernel, largest first
out=$1; cfg=${2:-C3}; scale=${3:-1.0}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && unset MAUVE_TRACE
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $root/tools/sweep/trace_cfg.py $cfg $scale > /tmp/ks.log 2>&1 || { tail -5 /tmp/ks.log; exit 1; }
f=$(find /tmp/ks -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $root/$out <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("devscan::", "").replace("void ", "")
    return n.split("(")[0][:60]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print("%-62s calls %5d  total %9.1f us  avg %8.1f us" % (short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
P
grep "align ms" /tmp/ks.log >> $root/$out
