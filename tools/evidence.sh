#!/bin/bash
# The round's evidence set: default bench line (with cpu_baseline), rocprofv3 kernel stats of the same command, HBM-traffic PMC passes.
set -o pipefail
timeout -k 10 500 python bench.py > gpurun_out/bench_default.log 2>&1; rc=$?; tail -1 gpurun_out/bench_default.log | cut -c1-2000; [ $rc -eq 0 ] || exit $rc
bash tools/prof.sh prof_r01_graph --steps 20 --warmup 3 || exit 1
bash tools/pmc.sh pmc_rows || exit 1
python tools/pmc_to_json.py gpurun_out/pmc_rows gpurun_out/pmc_rows/summary.json
