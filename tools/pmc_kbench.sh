#!/bin/bash
# usage: tools/pmc_kbench.sh <tag> <kbench filter> <kernel substring>  -- MFMA utilisation + stall breakdown of one kernel of tools/kbench.py (one PMC pass)
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_kb}
rm -rf $out && mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/kbench.py "$2" > $out/run.log 2>&1
echo rc=$?; tail -2 $out/run.log
python3 - "$out" "$3" <<'PY'
import csv,glob,collections,sys
d,sub=sys.argv[1],sys.argv[2]
f=glob.glob(d+'/*/*counter_collection.csv')[0]
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for r in csv.DictReader(open(f)):
    if sub in r['Kernel_Name']:
        k=r['Kernel_Name'][:60]+' grid '+r['Grid_Size']
        a=agg[k][r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k in agg:
    c={n:v[1]/v[0] for n,v in agg[k].items()}
    gui=c['GRBM_GUI_ACTIVE']; wc=c['SQ_WAVE_CYCLES']
    print(k, 'cycles/XCD %.0fk'%(gui/8e3), ' MfmaUtil %.1f%%'%(100*c['SQ_VALU_MFMA_BUSY_CYCLES']/(gui/8*1024)), ' exec GFLOP %.1f'%(c['SQ_INSTS_VALU_MFMA_MOPS_F32']*512/1e9),
      ' wave-cycles: wait_any %.0f%% issue-stall %.0f%% active %.0f%% (lds-issue %.0f%%)  mean waves/SIMD %.2f'%(100*c['SQ_WAIT_ANY']/wc,100*c['SQ_WAIT_INST_ANY']/wc,100*c['SQ_ACTIVE_INST_ANY']/wc,100*c['SQ_WAIT_INST_LDS']/wc, wc*4/(gui/8*1024)))
PY
