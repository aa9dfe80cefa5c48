#!/bin/bash
set -o pipefail
bash tools/abenv.sh - || exit 1
bash tools/prof.sh prof_r03_d --steps 20 --warmup 3 || exit 1
python tools/steptl.py gpurun_out/prof_r03_d 14 > gpurun_out/prof_r03_d/step.txt
