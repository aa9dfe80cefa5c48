#!/bin/bash
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_ops.py -m gpu -x -q --timeout 180 --timeout-method thread -k "sparse_linear or classifier" > gpurun_out/tests_sl.log 2>&1; rc=$?; tail -5 gpurun_out/tests_sl.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u -m pytest tests/test_gpu_fused.py -m gpu -x -q --timeout 180 --timeout-method thread -k "dropin or oracle" > gpurun_out/tests_fused.log 2>&1; rc=$?; tail -3 gpurun_out/tests_fused.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"loss": [0-9.]*' | tr '\n' ' ' || exit 1
echo
bash tools/prof.sh prof_r03_b --steps 20 --warmup 3 || exit 1
python tools/stats.py gpurun_out/prof_r03_b 43 12
python tools/steptl.py gpurun_out/prof_r03_b 14 > gpurun_out/prof_r03_b/step.txt
