#!/bin/bash
# A/B of two builds of the library on tools/convbench.py: usage tools/ab.sh <other.so>   (run from the repo root on the GPU box)
set -o pipefail
cp dd3d_amd/libdd3d_hip.so /tmp/lib_new.so
echo "== new"; timeout -k 10 120 python tools/convbench.py || exit 1
cp "$1" dd3d_amd/libdd3d_hip.so
echo "== base"; timeout -k 10 120 python tools/convbench.py || exit 1
cp /tmp/lib_new.so dd3d_amd/libdd3d_hip.so
