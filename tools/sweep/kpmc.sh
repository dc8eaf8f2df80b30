#!/bin/bash
# hardware counters of chosen kernels over a few passes of one config (GPU box): tools/sweep/kpmc.sh <out-file> <cfg> <kernel-substring> [counter ...]
# one rocprofv3 --pmc pass per group of counters (kernel trace only beside it); prints the per-launch average of every counter for the matching kernels
out=$1; cfg=$2; pat=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && unset MAUVE_TRACE
: > $root/$out
for grp in "$@"; do
  rm -rf /tmp/kp
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/kp -- python3 $root/tools/sweep/trace_cfg.py $cfg > /tmp/kp.log 2>&1 || { tail -5 /tmp/kp.log >> $root/$out; continue; }
  f=$(find /tmp/kp -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$pat" >> $root/$out <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if sys.argv[2] in n:
        acc[n.split("(")[0][:50]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k, cs in acc.items():
    for c, vs in cs.items():
        vs.sort(); big = max(v for _, v in vs)
        print("%-52s %-26s launches %3d  mean %.4g  max %.4g" % (k, c, len(vs), sum(v for _, v in vs) / len(vs), big))
P
done
