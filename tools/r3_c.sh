#!/bin/bash
# round 3, call C: kernel trace of the default bench command with the stream-K kernel gradient; timeline of one replayed step
set -o pipefail
bash tools/prof.sh prof_r03_a --steps 20 --warmup 3 || exit 1
python tools/stats.py gpurun_out/prof_r03_a 23 30 > gpurun_out/prof_r03_a/stats.txt
python tools/steptl.py gpurun_out/prof_r03_a 14 > gpurun_out/prof_r03_a/step.txt
cat gpurun_out/prof_r03_a/stats.txt
