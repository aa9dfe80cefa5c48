#!/bin/bash
set -o pipefail
for a in "--emulate-shard 0/8" "--emulate-shard 2/8" "--emulate-shard 0/4"; do
for e in "-" "DD3D_PIPELINE=1" "DD3D_PIPELINE=1 DD3D_BWW_AFTER=0"; do
  [ "$e" = "-" ] && e=""
  r=$(env $e timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline $a 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' ) || exit 1
  echo "$r  [$a] [$e]"
done; done
bash tools/abenv.sh "DD3D_PIPELINE=1 DD3D_BWW_AFTER=0" "DD3D_PIPELINE=1 DD3D_BWW_AFTER=0 DD3D_CLSW_LATE=0" || exit 1
