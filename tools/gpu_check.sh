set -o pipefail
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/tests.log 2>&1; rc=$?; tail -5 gpurun_out/tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_graph.log 2>&1; rc=$?; tail -1 gpurun_out/bench_graph.log | cut -c1-1300; exit $rc
