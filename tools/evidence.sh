#!/bin/bash
# The round's evidence set: default bench line (with cpu_baseline), rocprofv3 kernel stats of the same command, HBM-traffic PMC passes
# (roofline kernel + per-kernel table), MFMA-utilisation pass of the roofline kernel, record runs of the other configurations.
# Every long step appends to a file under gpurun_out/ (a silent run is taken to be hung).
set -o pipefail
R=${ROUND:-r03}
bash tools/prof.sh prof_${R}_graph --steps 20 --warmup 3 || exit 1
python tools/steptl.py gpurun_out/prof_${R}_graph 14 > gpurun_out/prof_${R}_graph/step.txt
bash tools/pmc.sh pmc_rows || exit 1
python tools/pmc_to_json.py gpurun_out/pmc_rows gpurun_out/pmc_rows/summary.json > /dev/null || exit 1
python tools/pmc_table.py gpurun_out/pmc_rows gpurun_out/pmc_rows/table.md > /dev/null || exit 1
bash tools/pmc_mfma.sh pmc_mfma || exit 1
python tools/pmc_mfma_parse.py gpurun_out/pmc_mfma > gpurun_out/pmc_mfma/util.txt || exit 1
cat gpurun_out/pmc_mfma/util.txt
bash tools/pmc_kbench.sh pmc_bww_sk_mfma conv128_bww k_conv128_bww > gpurun_out/pmc_bww_mfma.txt 2>&1 || exit 1
tail -2 gpurun_out/pmc_bww_mfma.txt
for cfg in "--config cfg3" "--config cfg4 --steps 5 --warmup 2" "--metric-name ours" "--emulate-shard 0/8" "--emulate-shard 2/8" "--emulate-shard 0/4" "--emulate-shard 0/2"; do
  timeout -k 10 400 python bench.py --no-cpu-baseline $cfg 2>&1 | tail -1 > "gpurun_out/bench_$(echo $cfg | tr -d ' /-').json" || exit 1
  echo "done $cfg"
done
timeout -k 10 900 python bench.py --config cfg5 --steps 3 --warmup 1 2>&1 | tail -1 > gpurun_out/bench_configcfg5.json || exit 1
ls gpurun_out/bench_*.json
