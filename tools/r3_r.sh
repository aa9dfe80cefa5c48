#!/bin/bash
set -o pipefail
for e in "-" "DD3D_BWW_SK=0" "DD3D_SO_EARLY=0" "DD3D_BWW_SK=0 DD3D_SO_EARLY=0"; do
  [ "$e" = "-" ] && e=""
  env $e timeout -k 10 300 python tools/bench_cfg5.py --steps 2 --warmup 1 --classes 11 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"loss": [0-9.e+]*' | tr '\n' ' ' || exit 1
  echo " [$e]"
done
