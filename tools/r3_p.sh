#!/bin/bash
set -o pipefail
timeout -k 10 1000 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_fused.py -m gpu -x -q -s --timeout 600 --timeout-method thread -k "true_divide or 3d_point or cfg4_full" > gpurun_out/tests_3d.log 2>&1; rc=$?; grep -E "differs|replayed|passed|failed|Error|assert" gpurun_out/tests_3d.log | tail -15 | cut -c1-300; exit $rc
