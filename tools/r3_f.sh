#!/bin/bash
set -o pipefail
timeout -k 10 100 python tools/kbench.py sl_bwd_w 2>&1 | tail -1 || exit 1
bash tools/abenv.sh - DD3D_CLSW_LATE=0 DD3D_BWW_SK=0 "DD3D_BWW_SK=0 DD3D_CLSW_LATE=0" - || exit 1
