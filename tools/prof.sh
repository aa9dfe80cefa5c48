#!/bin/bash
# usage: tools/prof.sh <tag> [bench args]  -- rocprofv3 kernel trace + stats of bench.py into gpurun_out/<tag>
tag=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$tag && cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag/run.log 2>&1
echo exit=$?
tail -1 $GRAFT_REPO_ROOT/gpurun_out/$tag/run.log | cut -c1-300
