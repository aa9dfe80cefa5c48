#!/bin/bash
set -o pipefail
timeout -k 10 900 python -u -m pytest tests/test_gpu_fused.py tests/test_gpu_step.py -m gpu -x -q -s --timeout 300 --timeout-method thread -k "cfg2 or cfg3 or table2" > gpurun_out/tests_parity.log 2>&1; rc=$?; grep -E "replayed|discontinuity|Table|modmse|ours|passed|failed|Error" gpurun_out/tests_parity.log | tail -30; exit $rc
