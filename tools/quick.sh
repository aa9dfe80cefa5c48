#!/bin/bash
# quick perf check: full step and the 2-class share of an 8-rank job (ranks 0-1 own two classes)
set -o pipefail
for a in "" "--emulate-shard 0/8" ; do timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline $a 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"loss": [0-9.]*' | tr '\n' ' ' || exit 1; echo " [$a]"; done
