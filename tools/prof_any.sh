#!/bin/bash
# usage: tools/prof_any.sh <tag> <script.py> [args]  -- rocprofv3 kernel trace + stats of a python tool into gpurun_out/<tag>
tag=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$tag && cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag/run.log 2>&1
echo exit=$?
