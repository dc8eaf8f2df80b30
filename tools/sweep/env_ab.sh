#!/bin/bash
for e in "NONE=1" "MAUVE_SCHEDULE=spin" "MAUVE_SCHEDULE=yield" "MAUVE_SCHEDULE=block"; do
  echo "== $e"
  env $e timeout -k 10 120 python tools/sweep/trace_cfg.py C3 1.0 2>&1 | grep -a "C3 align" | tail -2 | cut -c1-40
done
