#!/bin/bash
# Runs on the GPU box (via gpurun): plain bench, rocprofv3 kernel trace + stats, and two PMC passes
# (FETCH_SIZE, WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes).  Outputs under gpurun_out/prof/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R && python3 bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err
echo "trace done" >> $OUT/progress.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "fetch done" >> $OUT/progress.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "write done" >> $OUT/progress.log
find $OUT -name "*.csv" | head -20
cat $OUT/bench.json
