#!/bin/bash
set -o pipefail
timeout -k 10 1000 python -u -m pytest tests/test_gpu_fused.py -m gpu -x -q -s --timeout 600 --timeout-method thread -k "3d_point or cfg4_full" > gpurun_out/tests_3d.log 2>&1; rc=$?; tail -15 gpurun_out/tests_3d.log | cut -c1-300; exit $rc
