#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py's default workload (C3), and the PMC
# passes (FETCH_SIZE, WRITE_SIZE, LDS counters in separate runs, as MI355X_MICROARCH.md prescribes).
# Outputs under gpurun_out/prof_$TAG/.  usage: tools/profile_gpu.sh <tag> [pmc|nopmc] [C2|C3|C4|C5]   (default config: C3)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
CFG=${3:-C3}
OUT=$R/gpurun_out/prof_$TAG
if [ "$CFG" != "C3" ]; then OUT=$R/gpurun_out/prof_${TAG}_$CFG; fi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --config $CFG"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err
echo "trace done" >> $OUT/progress.log
if [ "${2:-}" = "pmc" ]; then
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
  echo "fetch done" >> $OUT/progress.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.json 2> $OUT/pmc_write.err
  echo "write done" >> $OUT/progress.log
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_lds -- python3 $R/bench.py $ARGS > $OUT/pmc_lds.json 2> $OUT/pmc_lds.err
  echo "lds done" >> $OUT/progress.log
fi
find $OUT -name "*stats*.csv" | head
