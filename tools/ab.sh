#!/bin/bash
# A/B of two builds of the library on tools/convbench.py: usage tools/ab.sh <other.so>   (run from the repo root on the GPU box)
# The comparison build is loaded through DD3D_LIB (dd3d_amd/_lib.py); the product library is never touched.
set -o pipefail
echo "== new"; timeout -k 10 120 python tools/convbench.py || exit 1
echo "== base"; DD3D_LIB="$1" timeout -k 10 120 python tools/convbench.py || exit 1
