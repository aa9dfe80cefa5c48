#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_fused.py -m gpu -x -q --timeout 120 --timeout-method thread -k "pipelined" > gpurun_out/tests_pipe.log 2>&1; rc=$?; tail -15 gpurun_out/tests_pipe.log | cut -c1-300; [ $rc -eq 0 ] || exit $rc
bash tools/abenv.sh - DD3D_PIPELINE=1 "DD3D_PIPELINE=1 DD3D_PIPE_ORDER=real" || exit 1
