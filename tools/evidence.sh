set -o pipefail
timeout -k 10 500 python bench.py > gpurun_out/bench_default.log 2>&1; rc=$?; tail -1 gpurun_out/bench_default.log | cut -c1-2000; [ $rc -eq 0 ] || exit $rc
bash tools/prof.sh prof_r01_graph --steps 20 --warmup 3
