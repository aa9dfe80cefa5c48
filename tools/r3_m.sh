#!/bin/bash
set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/tests.log 2>&1; rc=$?; tail -5 gpurun_out/tests.log; [ $rc -eq 0 ] || exit $rc
bash tools/abenv.sh - DD3D_BWW_NWG=768 DD3D_BWW_NWG=384 || exit 1
for a in "--emulate-shard 0/8" "--emulate-shard 2/8" "--config cfg3" "--metric-name ours"; do timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline $a 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' ' || exit 1; echo " [$a]"; done
