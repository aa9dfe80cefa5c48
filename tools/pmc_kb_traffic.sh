#!/bin/bash
# usage: tools/pmc_kb_traffic.sh <tag> <kbench filter> <kernel substring>  -- HBM-side traffic (FETCH_SIZE x2, WRITE_SIZE; separate passes) of the
# kernels of one tools/kbench.py entry, per launch
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_kbt}
rm -rf $out && mkdir -p $out && cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -- python3 $GRAFT_REPO_ROOT/tools/kbench.py "$2" > $out/$c.log 2>&1 || { echo "pass $c failed"; tail -5 $out/$c.log; exit 1; }
done
python3 - "$out" "$3" <<'PY'
import csv,glob,collections,sys
d,sub=sys.argv[1],sys.argv[2]
for c,mul in (("FETCH_SIZE",2048.0),("WRITE_SIZE",1024.0)):
    f=glob.glob(d+'/'+c+'/*/*counter_collection.csv')[0]
    agg=collections.defaultdict(lambda:[0,0.0,0.0])
    for r in csv.DictReader(open(f)):
        if sub in r['Kernel_Name'] and r['Counter_Name']==c:
            a=agg[r['Kernel_Name'].replace('(anonymous namespace)::','')[:50]]; a[0]+=1; a[1]+=float(r['Counter_Value'])*mul; a[2]+=float(r['End_Timestamp'])-float(r['Start_Timestamp'])
    for k,(n,v,t) in agg.items():
        print(f"{c:11s} {k:52s} launches {n:4d}  {v/n/1e6:8.1f} MB/launch  {t/n/1e3:7.1f} us")
PY
