#!/bin/bash
# round 3, call A: GPU suite on the tree with close() / per-distiller counters, then A/B of the kernel gradient's dispatch order
set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/tests.log 2>&1; rc=$?; tail -5 gpurun_out/tests.log; [ $rc -eq 0 ] || exit $rc
for o in 1 0; do
  echo "== DD3D_BWW_ORDER=$o"
  DD3D_BWW_ORDER=$o timeout -k 10 100 python tools/kbench.py conv128_bww 2>&1 | tail -3 || exit 1
  DD3D_BWW_ORDER=$o timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"loss": [0-9.]*' | tr '\n' ' ' || exit 1
  echo
done
