set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_fused.py -x -q --timeout 120 --timeout-method thread > gpurun_out/t12.log 2>&1; rc=$?; tail -5 gpurun_out/t12.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_graph.log 2>&1; rc=$?; tail -2 gpurun_out/bench_graph.log | cut -c1-1200; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --path fused > gpurun_out/bench_fused.log 2>&1; rc=$?; tail -1 gpurun_out/bench_fused.log | cut -c1-600; exit $rc
