#!/bin/bash
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_fused.py -m gpu -x -q --timeout 180 --timeout-method thread -k "graph or cfg2 or deferred" > gpurun_out/tests_fused.log 2>&1; rc=$?; tail -5 gpurun_out/tests_fused.log; [ $rc -eq 0 ] || exit $rc
bash tools/abenv.sh - DD3D_CLSW_LATE=0 DD3D_SO_STREAM=side || exit 1
bash tools/prof.sh prof_r03_c --steps 20 --warmup 3 || exit 1
python tools/steptl.py gpurun_out/prof_r03_c 14 > gpurun_out/prof_r03_c/step.txt
