#!/bin/bash
# the randomised GPU-vs-oracle sweeps of the round; usage: sweeps.sh <out dir> <seed base> <seconds per sweep>; stops at the first failure / GPU fault
D=$1; S=$2; T=$3; mkdir -p $D
step() { name=$1; shift; ( "$@" ) > $D/$name.log 2>&1; rc=$?; echo "$name rc=$rc: $(tail -n 1 $D/$name.log | cut -c1-300)" >> $D/summary.txt
         if [ $rc -ne 0 ] || grep -q "Memory access fault" $D/$name.log; then # (the sweeps with MAUVE_CANON_DEVICE_MIN keep a tiny pass's records in device memory; these take the default: records written straight to page-locked memory)
step fuzz4h env FUZZ_IT_FILE=$D/fuzz4h.it python tools/sweep/fuzz4.py $((S+8)) $T
cat $D/summary.txt; exit 1; fi; }
step fuzz1  env MAUVE_CANON_DEVICE_MIN=1 FUZZ_IT_FILE=$D/fuzz1.it python tools/sweep/fuzz.py $((S+1)) $T
step fuzz4  env MAUVE_CANON_DEVICE_MIN=1 FUZZ_IT_FILE=$D/fuzz4.it python tools/sweep/fuzz4.py $((S+2)) $T
step fuzz5  env MAUVE_CANON_DEVICE_MIN=1 FUZZ_IT_FILE=$D/fuzz5.it python tools/sweep/fuzz5.py $((S+3)) $T
step fuzz4w env MAUVE_DP_CLASS=wild MAUVE_CH_CL_MAX=3 MAUVE_CANON_DEVICE_MIN=1 FUZZ_IT_FILE=$D/fuzz4w.it python tools/sweep/fuzz4.py $((S+4)) $T
step fuzz5b env FUZZ_IT_FILE=$D/fuzz5b.it python tools/sweep/fuzz5.py $((S+5)) $T
# round 4: every interval with a dimension beyond one band through the wide sweep (dp_step_wide), both of its shapes
step fuzz4x env MAUVE_DP_WIDE_MIN=1 MAUVE_CANON_DEVICE_MIN=1 FUZZ_IT_FILE=$D/fuzz4x.it python tools/sweep/fuzz4.py $((S+6)) $T
step fuzz1x env MAUVE_DP_WIDE_MIN=1 MAUVE_DP_WIDE_R2=1 FUZZ_IT_FILE=$D/fuzz1x.it python tools/sweep/fuzz.py $((S+7)) $T
# (the sweeps with MAUVE_CANON_DEVICE_MIN keep a tiny pass's records in device memory; these take the default: records written straight to page-locked memory)
step fuzz4h env FUZZ_IT_FILE=$D/fuzz4h.it python tools/sweep/fuzz4.py $((S+8)) $T
cat $D/summary.txt
