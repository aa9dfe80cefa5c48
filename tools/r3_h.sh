#!/bin/bash
set -o pipefail
for a in "" "--defer-loss"; do timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline $a 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' || exit 1; echo "[$a]"; done
DD3D_CLSW_LATE=0 DD3D_BWW_LATE=1 DD3D_W0_CHAIN=1 timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/w0chain.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/w0chain.log | cut -c1-400
