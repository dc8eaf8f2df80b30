#!/bin/bash
# replay of one fuzz5 case with variants: locate2.sh <seed> <it> <outdir>
D=$3; mkdir -p $D
export MAUVE_CANON_DEVICE_MIN=1
for v in base "MAUVE_GIVEN_ON_HOST=1" "MAUVE_HOST_EXTEND=1" "MAUVE_NO_KEEP_MUMS=1" "MAUVE_NO_TINY=1" "FUZZ_KW=extend_lcbs=0" "FUZZ_KW=recursive=0" "MAUVE_HOST_GAP_CHAIN=1"; do
  n=$(echo $v | tr -c 'A-Za-z0-9\n' '_')
  if [ "$v" = base ]; then env FUZZ_IT=$2 python tools/sweep/fuzz5.py $1 100 > $D/$n.log 2>&1; else env "$v" FUZZ_IT=$2 python tools/sweep/fuzz5.py $1 100 > $D/$n.log 2>&1; fi
  echo "$n rc=$? $(grep -c FAIL $D/$n.log)" >> $D/summary.txt
  if grep -q "Memory access fault" $D/$n.log; then echo FAULT >> $D/summary.txt; break; fi
done
cat $D/summary.txt
