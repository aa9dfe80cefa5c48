#!/bin/bash
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_fused.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/tests_fused.log 2>&1; rc=$?; tail -5 gpurun_out/tests_fused.log; [ $rc -eq 0 ] || exit $rc
bash tools/abenv.sh - DD3D_SO_EARLY=0 DD3D_CLSW_LATE=0 "DD3D_CLSW_LATE=0 DD3D_SO_EARLY=0" || exit 1
