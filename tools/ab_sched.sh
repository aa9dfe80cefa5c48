#!/bin/bash
# A/B of two library builds on the full step: usage tools/ab_sched.sh <other.so>   (the other build is loaded through DD3D_LIB)
set -o pipefail
run() { for a in "" "--config cfg3" "--config cfg4 --steps 5 --warmup 2"; do timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline $a 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['achieved'],1))" || return 1; done; }
echo "== new"; run || exit 1
echo "== base"; DD3D_LIB="$1" run || exit 1
