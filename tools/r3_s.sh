#!/bin/bash
set -o pipefail
echo "== HEAD"; timeout -k 10 600 python tools/bench_cfg5.py --steps 3 --warmup 1 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"loss": [0-9.e+]*' | tr '\n' ' ' || exit 1
echo; echo "== r02 tree"; (cd _r02 && timeout -k 10 600 python tools/bench_cfg5.py --steps 3 --warmup 1 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"loss": [0-9.e+]*' | tr '\n' ' ') || exit 1
echo; echo "== r02 tree, 11 classes"; (cd _r02 && timeout -k 10 600 python tools/bench_cfg5.py --steps 2 --warmup 1 --classes 11 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"loss": [0-9.e+]*' | tr '\n' ' ') || exit 1
echo
