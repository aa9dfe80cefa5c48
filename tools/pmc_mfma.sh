#!/bin/bash
# MFMA utilisation + stall breakdown of the 128-channel convolution (tools/convbench.py), one PMC pass (8 SQ + 1 GRBM counters)
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_mfma}
rm -rf $out && mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/convbench.py > $out/run.log 2>&1
echo rc=$?; tail -2 $out/run.log
