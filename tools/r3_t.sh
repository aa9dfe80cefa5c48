#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
bash tools/abenv.sh - "DD3D_BWW_SK=0" "DD3D_LIB=$R/tools/_ab/lib_chunk1024.so DD3D_BWW_SK=0" "DD3D_LIB=$R/tools/_ab/lib_chunk1536.so DD3D_BWW_SK=0" "DD3D_LIB=$R/tools/_ab/lib_chunk2048.so DD3D_BWW_SK=0" || exit 1
for c in 1024 2048; do DD3D_LIB=$R/tools/_ab/lib_chunk$c.so DD3D_BWW_SK=0 timeout -k 10 100 python tools/kbench.py conv128_bww 2>&1 | grep "conv128_bww " || exit 1; done
