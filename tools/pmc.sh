#!/bin/bash
# usage: tools/pmc.sh <tag>  -- HBM traffic counters of bench.py, one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit together)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf $out && mkdir -p $out && cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 3 "$@" > $out/$c.log 2>&1 || { echo "pass $c failed"; tail -5 $out/$c.log; exit 1; }
  ls $out/$c/*/ | head
done
