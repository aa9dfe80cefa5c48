"""One replayed step's kernel timeline from a rocprofv3 kernel trace.  usage: python tools/steptl.py <dir> [step index]"""
import csv, glob, os, re, sys
f = max(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'), key=os.path.getmtime)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in csv.DictReader(open(f)))
idx = [i for i, r in enumerate(rows) if 'k_syn_head' in r[3] or 'k_assemble_batch' in r[3]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
a, b = idx[k], idx[k + 1]
t0 = rows[a][0]
print("step wall us", (rows[b][0] - t0) / 1e3)
for s, e, q, n in rows[a:b]:
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)[:40]
    print(f"{(s-t0)/1e3:8.1f} -> {(e-t0)/1e3:8.1f}  dur {(e-s)/1e3:6.1f}  q{q} {n}")
