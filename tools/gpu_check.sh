set -o pipefail
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/tests.log 2>&1; rc=$?; tail -3 gpurun_out/tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_graph.log 2>&1; rc=$?; tail -1 gpurun_out/bench_graph.log | grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*\|"loss": [0-9.]*'; [ $rc -eq 0 ] || exit $rc
timeout -k 10 100 python tools/kbench.py > gpurun_out/kbench.log 2>&1; cat gpurun_out/kbench.log
