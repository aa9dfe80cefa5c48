#!/bin/bash
set -o pipefail
timeout -k 10 900 python bench.py --config cfg5 --steps 3 --warmup 1 > gpurun_out/bench_cfg5.log 2>&1; rc=$?; tail -1 gpurun_out/bench_cfg5.log | cut -c1-1500; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --config cfg4 --steps 5 --warmup 2 2>&1 | tail -1 | cut -c1-600 || exit 1
