#!/bin/bash
# Copies the outputs of tools/evidence.sh (+ the default bench line) from gpurun_out/ into profiles/ under the round's names.
R=${ROUND:-r03}
set -e
f=$(ls -t gpurun_out/prof_${R}_graph/vm/*_kernel_stats.csv | head -1)
cp "$f" profiles/${R}_graph_step_kernel_stats.csv
cp gpurun_out/prof_${R}_graph/step.txt profiles/${R}_graph_step_timeline.txt
cp gpurun_out/pmc_rows/table.md profiles/${R}_hbm_per_kernel.md
cp gpurun_out/pmc_rows/summary.json profiles/${R}_pmc_conv128_rows.json
cp gpurun_out/pmc_mfma/util.txt profiles/${R}_pmc_conv128_mfma_util.txt
tail -2 gpurun_out/pmc_bww_mfma.txt > profiles/${R}_pmc_conv128_bww_mfma_util.txt
for p in "configcfg3:cfg3" "configcfg4steps5warmup2:cfg4" "configcfg5:cfg5" "metricnameours:cfg2_ours" "emulateshard02:shard_0of2" "emulateshard04:shard_0of4" "emulateshard08:shard_0of8" "emulateshard28:shard_2of8"; do
  cp gpurun_out/bench_${p%%:*}.json profiles/${R}_bench_${p##*:}.json
done
[ -s gpurun_out/bench_default.json ] && cp gpurun_out/bench_default.json profiles/${R}_bench_cfg2_default.json
echo copied
