#!/bin/bash
# round 3, call B: stream-K kernel gradient -- op tests, then isolated timing and the step for several grid sizes
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_ops.py -m gpu -x -q --timeout 180 --timeout-method thread -k "bwd_weight or conv_forward_backward or cloud_window" > gpurun_out/tests_bww.log 2>&1; rc=$?; tail -5 gpurun_out/tests_bww.log; [ $rc -eq 0 ] || exit $rc
for n in 512 768 256; do
  echo "== DD3D_BWW_NWG=$n"
  DD3D_BWW_NWG=$n timeout -k 10 100 python tools/kbench.py conv128_bww 2>&1 | grep "conv128_bww " || exit 1
  DD3D_BWW_NWG=$n timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"loss": [0-9.]*' | tr '\n' ' ' || exit 1
  echo
done
echo "== row-chunk kernel (DD3D_BWW_SK=0)"
DD3D_BWW_SK=0 timeout -k 10 100 python tools/kbench.py conv128_bww 2>&1 | grep "conv128_bww " || exit 1
DD3D_BWW_SK=0 timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"loss": [0-9.]*' | tr '\n' ' ' || exit 1
