#!/bin/bash
# The round's evidence set: default bench line (with cpu_baseline), rocprofv3 kernel stats of the same command, HBM-traffic PMC passes
# (roofline kernel + per-kernel table), MFMA-utilisation pass of the roofline kernel, record runs of the other configurations.
set -o pipefail
timeout -k 10 500 python bench.py > gpurun_out/bench_default.log 2>&1; rc=$?; tail -1 gpurun_out/bench_default.log | cut -c1-2000; [ $rc -eq 0 ] || exit $rc
bash tools/prof.sh prof_r02_graph --steps 20 --warmup 3 || exit 1
bash tools/pmc.sh pmc_rows || exit 1
python tools/pmc_to_json.py gpurun_out/pmc_rows gpurun_out/pmc_rows/summary.json > /dev/null || exit 1
python tools/pmc_table.py gpurun_out/pmc_rows gpurun_out/pmc_rows/table.md > /dev/null || exit 1
bash tools/pmc_mfma.sh pmc_mfma || exit 1
python tools/pmc_mfma_parse.py gpurun_out/pmc_mfma > gpurun_out/pmc_mfma/util.txt || exit 1
cat gpurun_out/pmc_mfma/util.txt
for cfg in "--config cfg3" "--config cfg4 --steps 5 --warmup 2" "--metric-name ours"; do
  timeout -k 10 400 python bench.py --no-cpu-baseline $cfg 2>&1 | tail -1 > "gpurun_out/bench_$(echo $cfg | tr -d ' -').json" || exit 1
done
ls gpurun_out/bench_*.json
