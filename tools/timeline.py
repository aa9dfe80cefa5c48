"""Timeline summary of a rocprofv3 kernel trace: per-queue busy time, union busy time and idle gaps
over the steady-state tail of the run.  usage: tools_timeline.py <dir> [tail_fraction=0.5] [steps_in_tail]"""
import csv, glob, sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t1 - (t1 - t0) * frac
rows = [r for r in rows if r[0] >= lo]
win = (max(r[1] for r in rows) - rows[0][0]) / 1e3
print(f"window {win/1e3:.2f} ms, {len(rows)} dispatches")
perq = defaultdict(float)
for s, e, q, _ in rows:
    perq[q] += (e - s) / 1e3
for q, v in sorted(perq.items()):
    print(f"queue {q}: busy {v/1e3:.2f} ms = {100*v/win:.1f}% of window")
# union
busy, cur_s, cur_e = 0.0, rows[0][0], rows[0][1]
gaps = []
for s, e, q, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"union busy {busy/1e6:.2f} ms = {100*busy/1e3/win:.1f}% of window; idle {win/1e3-busy/1e6:.2f} ms in {len(gaps)} gaps")
bykernel = defaultdict(lambda: [0, 0.0])
for g, n in gaps:
    bykernel[n][0] += 1
    bykernel[n][1] += g / 1e3
print("idle time attributed to the kernel that ended the gap:")
for n, (c, t) in sorted(bykernel.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"  {t/1e3:7.2f} ms in {c:5d} gaps (avg {t/c:6.1f} us) before {n[:70]}")
