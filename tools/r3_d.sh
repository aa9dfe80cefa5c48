#!/bin/bash
set -o pipefail
echo "== stream-K"; bash tools/pmc_kb_traffic.sh pmc_bww_sk conv128_bww "bww" || exit 1
echo "== row-chunk"; DD3D_BWW_SK=0 bash tools/pmc_kb_traffic.sh pmc_bww_rc conv128_bww "k_" || exit 1
echo "== stream-K mfma"; bash tools/pmc_kbench.sh pmc_bww_sk_mfma conv128_bww k_conv128_bww || exit 1
