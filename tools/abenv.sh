#!/bin/bash
# usage: tools/abenv.sh "ENV1=a ENV2=b" "ENV1=c" ...   -- the default bench step (50 steps) once per environment setting ("-" = none)
set -o pipefail
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  r=$(env $e timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' ) || exit 1
  echo "$r  [$e]"
done
